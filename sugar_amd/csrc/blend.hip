// blend.hip -- per-tile front-to-back alpha compositing (forward) and its backward for gfx950.
//
// Replaces renderCUDA forward  DGR/cuda_rasterizer/forward.cu:261-374
//      and renderCUDA backward DGR/cuda_rasterizer/backward.cu:399-557.
//
// One 256-thread workgroup (4 wave64) per 16x16 tile.  Forward: wave w owns the 8x8 pixel block (w & 1, w >> 1), lane l its
// pixel (l & 7, l >> 3); backward: wave w owns the 16x4 strip of rows 4w.., lane l pixel (l & 15, l >> 4) (its phase B walks
// rows of 16 pixels).
// The tile's depth-sorted Gaussian list is staged through LDS in batches: each thread gathers ONE 48-byte
// GeomRec (three dwordx4 loads from one or two cache lines) and the whole workgroup then walks the batch
// with uniform-address (broadcast, conflict-free) LDS reads.
//
// Backward: see the comment above k_blend_bwd -- per-pair terms are transposed through a wave-private LDS panel and
// summed in registers, so a (tile, Gaussian) pair costs at most one 48-byte record of global atomics instead of the
// reference's up to 256 x 9 (backward.cu:523-554).  The walk starts at the tile's deepest contributor (tile_maxc,
// recorded by the forward) instead of at the end of the tile's list.
#include "sgr_common.h"

int g_sgr_blend_variant = 3;  // development switch (sgr_set_blend_variant): bit 0 = wave-per-block forward, bit 1 = wave-per-block backward

#ifdef SGR_COUNT
__device__ unsigned long long g_sgr_count[8];
extern "C" void sgr_debug_counts(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sgr_count), sizeof(g_sgr_count)); unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_sgr_count), z, sizeof(z)); }
#endif

namespace {

#define BATCH 256
#define LOG2E 1.4426950408889634f

// Which of the tile's four 16x4 pixel strips (one per wave) can a Gaussian contribute to?  A pair contributes only if
// power <= 0 and opacity * exp(power) >= 1/255 (forward.cu:336-345), i.e. the pixel lies inside the ellipse
//     q(d) = cx dx^2 + 2 cy dx dy + cz dy^2 <= tau^2 = 2 ln(255 opacity)        (d = pixel - centre).
// The test is exact for the strip's rectangle: q is convex, so unless the centre lies in the rectangle its minimum over
// the rectangle is attained on one of the four edges, where q is a 1-D parabola,
//     q(dx, .) = (det/cz) dx^2 + cz (dy - dy*)^2,  dy* = -cy dx / cz      (and symmetrically for a horizontal edge),
// minimised by clamping dy* to the edge.  tau^2 is inflated by 0.2 % + 1e-3 against rounding (at the threshold that is a
// 1 % margin in alpha, the float evaluation of the pair differs from the exact one by ~1e-6).  Skipping an entry for a
// strip is therefore exact: no pixel of that strip would have passed the reference's tests.  Elongated, diagonal
// splats miss most of the strips of their bounding box.  Degenerate conics fall back to "all strips".
__device__ __forceinline__ uint32_t strip_hit_mask(float gxc, float gyc, float cx, float cy, float cz, float op, float x0,
                                                   float y0)
{
    if (op < 1.0f / 255.0f) return 0u;  // alpha <= opacity < 1/255 everywhere
    const float det = cx * cz - cy * cy;
    if (!(det > 0.f) || !(cx > 0.f) || !(cz > 0.f)) return 0xFu;
    const float tau2 = 2.0f * __logf(255.0f * op) * 1.002f + 1e-3f;
    const float icx = __builtin_amdgcn_rcpf(cx), icz = __builtin_amdgcn_rcpf(cz);
    const float det_cz = det * icz, det_cx = det * icx;   // curvature left along an edge after minimising across it
    const float kyx = -cy * icz, kxy = -cy * icx;         // dy* = kyx dx,  dx* = kxy dy
    const float dxl = x0 - gxc, dxr = dxl + 15.0f;
    const bool in_x = dxl <= 0.f && dxr >= 0.f;
    const float sL = kyx * dxl, sR = kyx * dxr;           // optimal dy on the left / right edge
    const float bL = det_cz * dxl * dxl, bR = det_cz * dxr * dxr;
    uint32_t m = 0u;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const float dyl = y0 + 4.0f * w - gyc, dyh = dyl + 3.0f;
        float tL = fminf(fmaxf(sL, dyl), dyh) - sL, tR = fminf(fmaxf(sR, dyl), dyh) - sR;
        float q = fminf(bL + cz * tL * tL, bR + cz * tR * tR);
        const float sT = kxy * dyl, sB = kxy * dyh;       // optimal dx on the top / bottom edge
        const float tT = fminf(fmaxf(sT, dxl), dxr) - sT, tB = fminf(fmaxf(sB, dxl), dxr) - sB;
        q = fminf(q, fminf(det_cx * dyl * dyl + cx * tT * tT, det_cx * dyh * dyh + cx * tB * tB));
        const bool inside = in_x && dyl <= 0.f && dyh >= 0.f;
        if (inside || q <= tau2) m |= 1u << w;
    }
    return m;
}

// The same test for the forward's wave shape: four 8 x 8 pixel blocks in a 2 x 2 arrangement (bit w: block (w & 1, w >> 1)).
// A square block is touched by ~10 % fewer splats than a 16 x 4 strip of the same area.
__device__ __forceinline__ uint32_t block_hit_mask(float gxc, float gyc, float cx, float cy, float cz, float op, float x0,
                                                   float y0)
{
    if (op < 1.0f / 255.0f) return 0u;
    const float det = cx * cz - cy * cy;
    if (!(det > 0.f) || !(cx > 0.f) || !(cz > 0.f)) return 0xFu;
    const float tau2 = 2.0f * __logf(255.0f * op) * 1.002f + 1e-3f;
    const float icx = __builtin_amdgcn_rcpf(cx), icz = __builtin_amdgcn_rcpf(cz);
    const float det_cz = det * icz, det_cx = det * icx;
    const float kyx = -cy * icz, kxy = -cy * icx;
    uint32_t m = 0u;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const float dxl = x0 + 8.0f * (w & 1) - gxc, dxr = dxl + 7.0f;
        const float dyl = y0 + 8.0f * (w >> 1) - gyc, dyh = dyl + 7.0f;
        const float sL = kyx * dxl, sR = kyx * dxr;
        const float tL = fminf(fmaxf(sL, dyl), dyh) - sL, tR = fminf(fmaxf(sR, dyl), dyh) - sR;
        float q = fminf(det_cz * dxl * dxl + cz * tL * tL, det_cz * dxr * dxr + cz * tR * tR);
        const float sT = kxy * dyl, sB = kxy * dyh;
        const float tT = fminf(fmaxf(sT, dxl), dxr) - sT, tB = fminf(fmaxf(sB, dxl), dxr) - sB;
        q = fminf(q, fminf(det_cx * dyl * dyl + cx * tT * tT, det_cx * dyh * dyh + cx * tB * tB));
        const bool inside = dxl <= 0.f && dxr >= 0.f && dyl <= 0.f && dyh >= 0.f;
        if (inside || q <= tau2) m |= 1u << w;
    }
    return m;
}

// x summed over the four 16-lane rows of the wave (lanes l, l+16, l+32, l+48), result in every lane: two gfx950 lane-swap
// VALU ops instead of two ds_bpermute round trips through the LDS pipeline.
__device__ __forceinline__ float sum_over_rows(float x)
{
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // {r0,r0,r2,r2} , {r1,r1,r3,r3}
    const uint32_t y = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(y, y, false, false);      // {lo,lo} , {hi,hi}
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

struct StageFwd {
    float4 a[BATCH];      // x, y, -0.5*conic.x*log2e, -conic.y*log2e
    float4 b[BATCH];      // -0.5*conic.z*log2e, opacity, r, g
    float c[BATCH];       // b
    uint8_t list[4][BATCH];  // per wave (its 8x8 block): staged indices of the entries that can touch it, in list order
    uint32_t cnt[4][4];   // [staging wave][strip] hit counts
};

// Forward blend.  The workgroup stages 256 list entries at a time (one 48-B record gather per thread), computes each
// entry's block mask and compacts, per wave, the indices of the entries that can touch it.  Every wave then walks ONLY
// its own compacted list, in list order, with a branch-free body (the skip tests of forward.cu:336-351 become lane
// predicates), and stops as soon as its own 64 pixels are done.  Conic terms are pre-scaled by log2(e) at staging so a
// pair costs one v_exp_f32.
__global__ void __launch_bounds__(256) k_blend_fwd(int W, int H, int gx, const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ point_list,
                                                   const GeomRec* __restrict__ rec, const float* __restrict__ bg,
                                                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                   uint32_t* __restrict__ tile_maxc, uint32_t* __restrict__ tile_walked,
                                                   float* __restrict__ out_color, const uint32_t* __restrict__ guard_hdr,
                                                   uint32_t list_cap)
{
    // sync-free forward (sgr_forward_ex with a binning capacity): the host has not seen R.  If the lists did not fit the
    // capacity, or the level-1 binning overflowed, the ranges are meaningless: touch nothing (the caller reads the header,
    // discards this forward and repeats it)
    if (guard_hdr && (guard_hdr[SGR_HDR_R] > list_cap || guard_hdr[4 + SGR_B2_HDR_OVERFLOW])) return;
    __shared__ StageFwd st;
    __shared__ int s_done[4];
    __shared__ uint32_t s_maxc[4];
    __shared__ uint32_t s_walk[4];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int x0 = tx * SGR_TILE_X, y0 = ty * SGR_TILE_Y;
    // wave w owns the 8 x 8 pixel block (w & 1, w >> 1) of the tile, lane l the pixel (l & 7, l >> 3) of it
    const int px = x0 + 8 * (wave & 1) + (lane & 7), py = y0 + 8 * (wave >> 1) + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile], r1 = tile_start[tile + 1];
    const int total = (int)(r1 - r0);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0;
    uint32_t wave_walked = (uint32_t)total;  // list position at which this wave's last pixel finished (whole list: never)
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;

    for (int base = 0; base < total; base += BATCH) {
        // workgroup vote: stop when every pixel is done (forward.cu:309-311)
        const bool wave_done = __ballot(!done) == 0ull;
        if (lane == 0) s_done[wave] = wave_done;
        __syncthreads();
        if (s_done[0] & s_done[1] & s_done[2] & s_done[3]) break;
        const int nb = min(BATCH, total - base);
        uint32_t hit = 0u;
        if (tid < nb) {
            const uint32_t id = point_list[r0 + base + tid];
            const float4* rp = reinterpret_cast<const float4*>(rec + id);
            const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
            hit = block_hit_mask(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, (float)x0, (float)y0);
            st.a[tid] = make_float4(v0.x, v0.y, -0.5f * LOG2E * v0.z, -LOG2E * v0.w);
            st.b[tid] = make_float4(-0.5f * LOG2E * v1.x, v1.y, v2.x, v2.y);
            st.c[tid] = v2.z;
        }
        unsigned long long bal[4];
#pragma unroll
        for (int w = 0; w < 4; w++) bal[w] = __ballot((hit >> w) & 1u);
        if (lane == 0) {
#pragma unroll
            for (int w = 0; w < 4; w++) st.cnt[wave][w] = (uint32_t)__popcll(bal[w]);
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if ((hit >> w) & 1u) {
                uint32_t pos = (uint32_t)__popcll(bal[w] & lt_mask);
                for (int c = 0; c < wave; c++) pos += st.cnt[c][w];
                st.list[w][pos] = (uint8_t)tid;
            }
        }
        __syncthreads();
        if (wave_done) continue;  // this wave's pixels are finished; it only helps staging
        const int n_mine = __builtin_amdgcn_readfirstlane(
            (int)(st.cnt[0][wave] + st.cnt[1][wave] + st.cnt[2][wave] + st.cnt[3][wave]));
        for (int k = 0; k < n_mine; k++) {
            const int j = __builtin_amdgcn_readfirstlane((int)st.list[wave][k]);  // wave-uniform: keep it scalar
            const float4 a = st.a[j];
            const float4 b = st.b[j];
            const float cb = st.c[j];
            const uint32_t pos = (uint32_t)(base + j + 1);
            const float dx = a.x - pixfx, dy = a.y - pixfy;
            const float power2 = dx * (a.z * dx + a.w * dy) + b.x * dy * dy;  // log2(e) * power
            const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power2));
            const bool ok = !done && !(power2 > 0.0f) && !(alpha < 1.0f / 255.0f);
            const float test_T = T * (1.0f - alpha);
            const bool stop = ok && (test_T < 0.0001f);
            const bool upd = ok && !stop;
            const float w = upd ? alpha * T : 0.0f;
            C0 += b.z * w; C1 += b.w * w; C2 += cb * w;
            T = upd ? test_T : T;
            last_contributor = upd ? pos : last_contributor;
            done = done || stop;
#ifdef SGR_COUNT
            {   // lanes that pass the tests per wave iteration, and how many 8x2 row pairs / 4x4 sub-blocks hold one
                const unsigned long long okm = __ballot(ok);
                if (lane == 0) {
                    atomicAdd(&g_sgr_count[0], 1ull);                               // (entry, wave) iterations
                    atomicAdd(&g_sgr_count[1], (unsigned long long)__popcll(okm));  // passing lanes
                    int rows = ((okm & 0xFFFFull) != 0) + ((okm & 0xFFFF0000ull) != 0) + ((okm & 0xFFFF00000000ull) != 0) + ((okm >> 48) != 0);
                    atomicAdd(&g_sgr_count[2], (unsigned long long)rows);
                    // lane = 8 * y + x: sub-block (x >> 2, y >> 2)
                    const unsigned long long m00 = 0x0F0F0F0Full, m10 = 0xF0F0F0F0ull;
                    const int blocks = ((okm & m00) != 0) + ((okm & m10) != 0) + ((okm & (m00 << 32)) != 0) + ((okm & (m10 << 32)) != 0);
                    atomicAdd(&g_sgr_count[3], (unsigned long long)blocks);
                }
            }
#endif
            if (__ballot(!done) == 0ull) { wave_walked = pos; break; }
        }
    }
    if (inside) {
        const size_t pix_id = (size_t)W * py + px;
        const size_t HW = (size_t)H * W;
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
        out_color[pix_id] = C0 + T * bg[0];
        out_color[HW + pix_id] = C1 + T * bg[1];
        out_color[2 * HW + pix_id] = C2 + T * bg[2];
    }
    // deepest contributor of the tile, consumed by the backward
    uint32_t mc = inside ? last_contributor : 0u;
    for (int o = 32; o > 0; o >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, o));
    // furthest list position any pixel examined (R_f of the roofline accounting): the entry that
    // finished the wave's last pixel, or the whole list for a wave with a pixel that never saturated
    uint32_t wk = __ballot(inside) ? wave_walked : 0u;
    if (lane == 0) { s_maxc[wave] = mc; s_walk[wave] = wk; }
    __syncthreads();
    if (tid == 0) {
        tile_maxc[tile] = max(max(s_maxc[0], s_maxc[1]), max(s_maxc[2], s_maxc[3]));
        tile_walked[tile] = max(max(s_walk[0], s_walk[1]), max(s_walk[2], s_walk[3]));
    }
}

// ---------------------------------------------------------------------------------------------
// Forward blend, one WAVE per 8 x 8 pixel block (variant 1).
//
// k_blend_fwd above runs four waves per tile in lock step: three workgroup barriers per 256-entry batch, and a wave whose
// 64 pixels are saturated keeps staging for the others until the whole tile is done.  Here every 8 x 8 block is its own
// 64-thread workgroup: the wave gathers 64 list entries (one 48-byte record per lane), culls them against ITS block with the
// exact ellipse-vs-rectangle test, compacts the survivors into LDS (wave-private: LDS operations of a wave execute in
// order, no barrier at all) and walks them; the gather of the next 64 entries is in flight during the walk.  The four blocks
// of a tile re-read the same records (L2 hits: they are placed on the same XCD) and each computes one block test per entry
// instead of four -- ~4 wave-instructions per entry in total, against ~30 per walked (entry, block) pair.
__device__ __forceinline__ bool block_hit(float gxc, float gyc, float cx, float cy, float cz, float op, float bx0, float by0)
{
    if (op < 1.0f / 255.0f) return false;
    const float det = cx * cz - cy * cy;
    if (!(det > 0.f) || !(cx > 0.f) || !(cz > 0.f)) return true;
    const float tau2 = 2.0f * __logf(255.0f * op) * 1.002f + 1e-3f;
    const float icx = __builtin_amdgcn_rcpf(cx), icz = __builtin_amdgcn_rcpf(cz);
    const float det_cz = det * icz, det_cx = det * icx;
    const float kyx = -cy * icz, kxy = -cy * icx;
    const float dxl = bx0 - gxc, dxr = dxl + 7.0f;
    const float dyl = by0 - gyc, dyh = dyl + 7.0f;
    const float sL = kyx * dxl, sR = kyx * dxr;
    const float tL = fminf(fmaxf(sL, dyl), dyh) - sL, tR = fminf(fmaxf(sR, dyl), dyh) - sR;
    float q = fminf(det_cz * dxl * dxl + cz * tL * tL, det_cz * dxr * dxr + cz * tR * tR);
    const float sT = kxy * dyl, sB = kxy * dyh;
    const float tT = fminf(fmaxf(sT, dxl), dxr) - sT, tB = fminf(fmaxf(sB, dxl), dxr) - sB;
    q = fminf(q, fminf(det_cx * dyl * dyl + cx * tT * tT, det_cx * dyh * dyh + cx * tB * tB));
    const bool inside = dxl <= 0.f && dxr >= 0.f && dyl <= 0.f && dyh >= 0.f;
    return inside || q <= tau2;
}

// The walk over a compacted batch, hand-scheduled (the compiler's version of this loop carries ~30 VALU and ~23 SALU
// instructions per entry; this one 22 and 9).  Lanes whose pixel is finished are simply removed from EXEC for the whole walk,
// the reference's three skip tests (forward.cu:336-351) narrow EXEC further inside an iteration, and the state updates are
// plain moves under that mask -- no v_cndmask, no per-lane bookkeeping.  The 10 dwords of entry k+1 are fetched from LDS
// (uniform address: broadcast) while entry k is evaluated (two register sets, A = v[40:49], B = v[50:59]).
//   in : exec-independent; live = lanes still accumulating, n = entries to walk (> 0), addr = LDS byte address of entry 0
//   out: live updated, n = entries not yet started when every lane had finished (0: the batch was walked to its end)
// gfx950 hazards observed by hand (the compiler does not look inside): one independent instruction between v_exp_f32 and the
// use of its result; EXEC is only ever written by SALU instructions before VALU instructions depend on it.
#define SGR_FWD_BODY(X, Y, A, B, CZ, OP, R, G, BL, POS)                                             \
    "v_sub_f32 v60, " X ", %[px]\n"                                                                 \
    "v_sub_f32 v61, " Y ", %[py]\n"                                                                 \
    "v_mul_f32 v62, " B ", v61\n"                                                                   \
    "v_fmac_f32 v62, " A ", v60\n"                                                                  \
    "v_mul_f32 v63, " CZ ", v61\n"                                                                  \
    "v_mul_f32 v63, v63, v61\n"                                                                     \
    "v_fmac_f32 v63, v60, v62\n"        /* log2(e) * power */                                       \
    "v_exp_f32 v62, v63\n"                                                                          \
    "s_mov_b64 %[live], exec\n"                                                                     \
    "v_cmp_nlt_f32 vcc, 0, v63\n"       /* !(power > 0) */                                          \
    "v_mul_f32 v62, " OP ", v62\n"                                                                  \
    "v_min_f32 v62, 0x3f7d70a4, v62\n"  /* alpha = min(0.99, opacity * G) */                        \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v62\n" /* !(alpha < 1/255) */                                   \
    "v_sub_f32 v60, 1.0, v62\n"                                                                     \
    "v_mul_f32 v60, %[T], v60\n"        /* test_T */                                                \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_gt_f32 vcc, 0x38d1b717, v60\n" /* test_T < 0.0001: this lane is finished */              \
    "v_mul_f32 v61, v62, %[T]\n"                                                                    \
    "s_andn2_b64 %[live], %[live], vcc\n"                                                           \
    "s_andn2_b64 exec, exec, vcc\n"                                                                 \
    "v_mov_b32 %[T], v60\n"                                                                         \
    "v_fmac_f32 %[C0], " R ", v61\n"                                                                \
    "v_fmac_f32 %[C1], " G ", v61\n"                                                                \
    "v_fmac_f32 %[C2], " BL ", v61\n"                                                               \
    "v_mov_b32 %[last], " POS "\n"                                                                  \
    "s_mov_b64 exec, %[live]\n"

#define SGR_FWD_ENTRY_BYTES 48


__device__ __forceinline__ void fwd_walk(uint32_t addr, int& n, unsigned long long& live, float pixfx, float pixfy, float& T,
                                         float& C0, float& C1, float& C2, uint32_t& last)
{
    unsigned long long full;
    asm volatile(
        "s_mov_b64 %[full], exec\n"
        "s_and_b64 exec, exec, %[live]\n"
        "s_waitcnt lgkmcnt(0)\n"  /* scalar loads return out of order: none may be pending while LDS reads are counted */
        "ds_read_b128 v[40:43], %[addr]\n"
        "ds_read_b128 v[44:47], %[addr] offset:16\n"
        "ds_read_b64 v[48:49], %[addr] offset:32\n"
        "1:\n"
        "ds_read_b128 v[50:53], %[addr] offset:48\n"
        "ds_read_b128 v[54:57], %[addr] offset:64\n"
        "ds_read_b64 v[58:59], %[addr] offset:80\n"
        "s_waitcnt lgkmcnt(3)\n"
        SGR_FWD_BODY("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49")
        "s_cbranch_execz 3f\n"
        "s_add_i32 %[n], %[n], -1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 3f\n"
        "ds_read_b128 v[40:43], %[addr] offset:96\n"
        "ds_read_b128 v[44:47], %[addr] offset:112\n"
        "ds_read_b64 v[48:49], %[addr] offset:128\n"
        "v_add_u32 %[addr], 96, %[addr]\n"
        "s_waitcnt lgkmcnt(3)\n"
        SGR_FWD_BODY("v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59")
        "s_cbranch_execz 3f\n"
        "s_add_i32 %[n], %[n], -1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc0 1b\n"
        "3:\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, %[full]\n"
        : [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [last] "+v"(last), [addr] "+v"(addr), [n] "+s"(n),
          [live] "+s"(live), [full] "=&s"(full)
        : [px] "v"(pixfx), [py] "v"(pixfy)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56",
          "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc", "scc", "memory");
}

__global__ void __launch_bounds__(64) k_blend_fwd_w(int W, int H, int gx, int T_tiles, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ point_list, const GeomRec* __restrict__ rec,
                                                    const float* __restrict__ bg, float* __restrict__ final_T,
                                                    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_maxc,
                                                    uint32_t* __restrict__ tile_walked, float* __restrict__ out_color,
                                                    const uint32_t* __restrict__ guard_hdr, uint32_t list_cap)
{
    if (guard_hdr && (guard_hdr[SGR_HDR_R] > list_cap || guard_hdr[4 + SGR_B2_HDR_OVERFLOW])) return;
    // entry k: {x, y, -0.5*conic.x*log2e, -conic.y*log2e | -0.5*conic.z*log2e, opacity, r, g | b, bitcast(1-based list position), -, -}
    // (one spare entry: the walk's look-ahead reads one entry past the last)
    __shared__ __attribute__((aligned(16))) float s_e[65 * (SGR_FWD_ENTRY_BYTES / 4)];
    // workgroup b runs on XCD b % 8: the four blocks of a tile share an XCD (and its L2)
    const int wg = blockIdx.x;
    const int sub = (wg >> 3) & 3;
    const int tile = ((wg >> 5) << 3) + (wg & 7);
    if (tile >= T_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int bx0 = tx * SGR_TILE_X + 8 * (sub & 1), by0 = ty * SGR_TILE_Y + 8 * (sub >> 1);
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile];
    const int total = (int)(tile_start[tile + 1] - r0);

    float T = 1.0f;
    uint32_t last_contributor = 0;
    uint32_t walked = (uint32_t)total;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    unsigned long long live = __ballot(inside);  // lanes still accumulating

    // The ids of the next batch are fetched while this one is walked; the records are gathered at the top of the iteration
    // (compiler-scheduled loads: other resident waves cover the latency.  Loads issued from inline asm into registers that
    // stay in flight across the loop edge are NOT safe: the compiler may copy or reuse the destination registers before
    // the data lands.)
    // (loads are unconditional with clamped indices: a load under a lane predicate makes the compiler's wait counts
    // path-dependent and it then drains everything at the first use)
    uint32_t id_next = 0u;
    if (total > 0) id_next = point_list[r0 + (uint32_t)min(lane, total - 1)];  // (uniform branch)
    const uint32_t lds0 = (uint32_t)(uintptr_t)s_e;  // LDS byte address of the staging area
    for (int base = 0; base < total && live != 0ull; base += 64) {
        const float4* rp = reinterpret_cast<const float4*>(rec + id_next);
        const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
        id_next = point_list[r0 + (uint32_t)min(base + 64 + lane, total - 1)];
        const bool hit = (base + lane < total) && block_hit(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, (float)bx0, (float)by0);
        const unsigned long long m = __ballot(hit);
        int n = __popcll(m);
        if (hit) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            float4* e = reinterpret_cast<float4*>(s_e + pos * (SGR_FWD_ENTRY_BYTES / 4));
            e[0] = make_float4(v0.x, v0.y, -0.5f * LOG2E * v0.z, -LOG2E * v0.w);
            e[1] = make_float4(-0.5f * LOG2E * v1.x, v1.y, v2.x, v2.y);
            *reinterpret_cast<float2*>(e + 2) = make_float2(v2.z, __uint_as_float((uint32_t)(base + lane + 1)));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n > 0) {
            const int n0 = n;
            fwd_walk(lds0, n, live, pixfx, pixfy, T, C0, C1, C2, last_contributor);
            if (live == 0ull) {  // every pixel finished at compacted entry n0 - n: its list position is the furthest examined
                walked = __float_as_uint(s_e[(n0 - n) * (SGR_FWD_ENTRY_BYTES / 4) + 9]);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (inside) {
        const size_t pix_id = (size_t)W * py + px;
        const size_t HW = (size_t)H * W;
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
        out_color[pix_id] = C0 + T * bg[0];
        out_color[HW + pix_id] = C1 + T * bg[1];
        out_color[2 * HW + pix_id] = C2 + T * bg[2];
    }
    // deepest contributor / furthest examined position of the TILE (zeroed by the launcher): maximum over its four blocks
    uint32_t mc = inside ? last_contributor : 0u;
    for (int o = 32; o > 0; o >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, o));
    if (lane == 0) {
        atomicMax(&tile_maxc[tile], mc);
        atomicMax(&tile_walked[tile], __ballot(inside) ? walked : 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// Backward blend.
//
// The reference adds nine floats with global atomics for every contributing (pixel, Gaussian) pair
// (backward.cu:523-554).  Here a workgroup walks its tile's list back to front in batches of 64 Gaussians, and every
// wave alternates two lane mappings over sub-batches of 16 Gaussians:
//
//   phase A  lane = pixel (16x4 strip of the wave).  The per-pixel recurrence of backward.cu:486-534 (T /= 1-alpha,
//            accum_rec, dL_dalpha) runs sequentially over the 16 Gaussians; for each pair the lane stores just
//            Z = G * dL_dalpha and Wt = alpha * T into a wave-private LDS panel zw[g][pixel].
//   phase B  lane = (Gaussian g, pixel row q).  Each lane streams the 16 pixels of its row out of the panel and
//            accumulates, in registers, the colour sums  sum Wt*dL_dpix  and the moments  sum Z, sum Z*dx, sum Z*dx^2
//            (dy is constant along a row, so the y-moments factor out).  No cross-lane traffic at all in the loop;
//            one 4-lane shuffle reduction per sub-batch, then LDS float atomics into a per-workgroup table.
//
// All gradient terms of a pair are linear in {Wt*g_c, Z, Z*dx, Z*dy, Z*dx^2, Z*dx*dy, Z*dy^2} with per-Gaussian
// coefficients (backward.cu:538-554), so only those nine sums leave the workgroup: at most one 48-byte record of global
// float atomics per (tile, Gaussian), into acc[P][12].  The coefficients are applied once per Gaussian by the fused
// backward-preprocess kernel.  The panel's row stride (65 float2) makes both the phase-A writes and the phase-B reads
// bank-conflict free.
#define BWD_BATCH 64
#define BWD_SUB 16
#define ZW_STRIDE 65

struct __attribute__((aligned(16))) BwdShared {
    float4 a[BWD_BATCH];                  // x, y, conic.x, conic.y
    float4 b[BWD_BATCH];                  // conic.z, opacity, r, g
    float c[BWD_BATCH];                   // b
    uint32_t id[BWD_BATCH];
    uint32_t hit[BWD_BATCH];              // strip_hit_mask: which waves' strips the entry can touch
    float part[BWD_BATCH][12];            // per-workgroup sums of the current batch
    float2 zw[4][BWD_SUB * ZW_STRIDE];    // wave-private (Z, Wt) panels; before the walk starts the same memory carries
                                          // dL_dpix of the tile once (gpix view below): 39 KB total -> 4 workgroups per CU
};

__global__ void __launch_bounds__(256) k_blend_bwd(int W, int H, int gx, const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ point_list,
                                                   const GeomRec* __restrict__ rec, const float* __restrict__ bg,
                                                   const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                                                   const uint32_t* __restrict__ tile_maxc, const float* __restrict__ dL_dpix,
                                                   float* __restrict__ acc)
{
    __shared__ BwdShared sh;
    const int tile = blockIdx.x;
    const int total = (int)tile_maxc[tile];  // entries at list positions > tile_maxc contribute to no pixel
    if (total == 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int x0 = tx * SGR_TILE_X, y0 = ty * SGR_TILE_Y;
    const int px = x0 + (tid & 15), py = y0 + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile];

    const size_t pix_id = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_Ts[pix_id] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) { g0 = dL_dpix[pix_id]; g1 = dL_dpix[HW + pix_id]; g2 = dL_dpix[2 * HW + pix_id]; }
    const float neg_Tfinal_bg = -T_final * (bg[0] * g0 + bg[1] * g1 + bg[2] * g2);
    // accum_rec and last_color of backward.cu:514-516 only ever meet dL_dpixel in a dot product: the recurrence is carried
    // for the scalars  accum_rec . dL_dpixel  and  last_color . dL_dpixel  (6 instead of 12 instructions per pair)
    float acc_g = 0.f, lc_g = 0.f;
    float last_alpha = 0.f;

    // phase-B lane role and the dL_dpix of its 16-pixel row, kept in registers for the whole kernel
    const int bg_g = lane & 15, bq = lane >> 4;
    float* gpix = reinterpret_cast<float*>(&sh.zw[0][0]);  // [3][256], one-time exchange through the panel memory
    gpix[tid] = g0; gpix[256 + tid] = g1; gpix[512 + tid] = g2;
    for (int i = tid; i < BWD_BATCH * 12; i += 256) (&sh.part[0][0])[i] = 0.f;
    __syncthreads();
    float rg0[16], rg1[16], rg2[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int p = wave * 64 + bq * 16 + i;
        rg0[i] = gpix[p]; rg1[i] = gpix[256 + p]; rg2[i] = gpix[512 + p];
    }
    __syncthreads();  // the panels are written from here on
    const float rowy = (float)(y0 + wave * 4 + bq);
    float2* zw = sh.zw[wave];

    for (int base = 0; base < total; base += BWD_BATCH) {
        const int nb = min(BWD_BATCH, total - base);
        if (tid < BWD_BATCH) {
            uint32_t hit = 0u;
            if (tid < nb) {
                const uint32_t id = point_list[r0 + (uint32_t)(total - 1 - base - tid)];
                const float4* rp = reinterpret_cast<const float4*>(rec + id);
                const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
                // conic pre-scaled by log2(e) (and the -1/2 folded in): a pair costs one v_exp_f32 and no extra multiplies
                sh.a[tid] = make_float4(v0.x, v0.y, -0.5f * LOG2E * v0.z, -LOG2E * v0.w);
                sh.b[tid] = make_float4(-0.5f * LOG2E * v1.x, v1.y, v2.x, v2.y);
                sh.c[tid] = v2.z; sh.id[tid] = id;
                hit = strip_hit_mask(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, (float)x0, (float)y0);
            }
            sh.hit[tid] = hit;
        }
        __syncthreads();
        // entries of this batch that can touch this wave's strip, walked in list order in groups of <= 16 panel rows
        unsigned long long bits = __ballot((sh.hit[lane] >> wave) & 1u);
        while (bits) {
            // ---------------- phase A: lane = pixel
            int rows = 0;   // wave-uniform: panel rows in use
            int myj = 0;    // phase-B role: batch index of the Gaussian in panel row bg_g
            while (bits && rows < BWD_SUB) {
                const int j = __builtin_ctzll(bits);
                bits &= bits - 1;
                const int pos = total - base - j;  // 1-based list position of this entry
                const float4 a = sh.a[j];
                const float4 b = sh.b[j];
                const float dx = a.x - pixfx, dy = a.y - pixfy;
                const float power = dx * (a.z * dx + a.w * dy) + b.x * dy * dy;  // log2(e) * power of backward.cu:489
                const float G = __builtin_amdgcn_exp2f(power);
                const float alpha = fminf(0.99f, b.y * G);
                const bool active = (pos <= last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__ballot(active) == 0ull) continue;  // wave-uniform
                float Z = 0.f, Wt = 0.f;
                if (active) {
                    // 1/(1-alpha) with the hardware reciprocal (1 ulp): the reference's two IEEE divisions
                    // (backward.cu:503,534) cost ~24 instructions per pair; 1-alpha >= 0.01, so no range issue
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * inv_1ma;
                    Wt = alpha * T;
                    const float c0 = b.z, c1 = b.w, c2 = sh.c[j];
                    // accum_rec = last_alpha * last_color + (1 - last_alpha) * accum_rec  (backward.cu:514-516)
                    acc_g += last_alpha * (lc_g - acc_g);
                    lc_g = c0 * g0 + c1 * g1 + c2 * g2;
                    float dL_dalpha = lc_g - acc_g;
                    last_alpha = alpha;
                    dL_dalpha = dL_dalpha * T + neg_Tfinal_bg * inv_1ma;  // backward.cu:523-529
                    Z = G * dL_dalpha;
                }
                zw[rows * ZW_STRIDE + lane] = make_float2(Z, Wt);
                if (bg_g == rows) myj = j;
                rows++;
            }
            if (rows == 0) continue;  // wave-uniform (bits is now 0)
            // the panel is exchanged between lanes of ONE wave: LDS operations of a wave execute in order, so a
            // wave-scope fence (compiler ordering) is all that is needed, no workgroup barrier
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---------------- phase B: lane = (panel row bg_g, pixel row bq)
            if (bg_g < rows) {
                const float4 a = sh.a[myj];
                const float dyr = a.y - rowy;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f;
                const float2* row = zw + bg_g * ZW_STRIDE + bq * 16;
                const float xb = a.x - (float)x0;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float2 v = row[i];
                    s0 += v.x; s1 += v.x * (float)i; s2 += v.x * (float)(i * i);  // moments about the tile's left edge
                    k0 += v.y * rg0[i]; k1 += v.y * rg1[i]; k2 += v.y * rg2[i];
                }
                // sum Z (xb - i) and sum Z (xb - i)^2 from the raw moments
                const float sx = xb * s0 - s1;
                const float sxx = xb * (xb * s0 - 2.f * s1) + s2;
                float o[9] = {k0, k1, k2, s0, sx, dyr * s0, sxx, dyr * sx, dyr * dyr * s0};
#pragma unroll
                for (int v = 0; v < 9; v++) o[v] = sum_over_rows(o[v]);
                if (bq == 0) {
                    float* dst = sh.part[myj];
#pragma unroll
                    for (int v = 0; v < 9; v++) atomicAdd(&dst[v], o[v]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // flush this batch: 4 lanes per Gaussian, 3 values each; untouched Gaussians cost nothing
        {
            const int g = tid >> 2, cpart = tid & 3;
            if (g < nb && cpart < 3) {
                float* src = &sh.part[g][cpart * 3];
                float* dst = acc + (size_t)sh.id[g] * SGR_ACC_STRIDE + cpart * 3;
#pragma unroll
                for (int v = 0; v < 3; v++) {
                    const float val = src[v];
                    if (val != 0.f) { atomicAdd(&dst[v], val); src[v] = 0.f; }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Backward blend, one WAVE per 8 x 8 pixel block (variant bit 1).
//
// Same two-phase scheme as k_blend_bwd (phase A lane = pixel, phase B lane = (Gaussian, two pixel rows)), restructured like
// k_blend_fwd_w: no workgroup barriers, the gather of the next 64 list entries in flight, exact block culling with
// compaction -- and the compacted entries go through a small LDS queue, so phase B always sees full groups of 16 Gaussians
// (a batch of 64 list entries leaves ~19 survivors: without the queue every second phase B would run a quarter full).
// A (block, Gaussian) pair is met exactly once, so its nine sums go straight to global memory (nine float atomics by the
// sixteen q == 0 lanes); the moments are taken about the block origin and shifted to the Gaussian's centre once per pair.
// Phase A is hand-scheduled like the forward walk: ~30 VALU per entry, the reference's tests as EXEC masks.
#define BW_QCAP 80          // queue entries: at most 15 left over + 64 new
#define BW_SUB 16
#define BW_ZW_STRIDE 65     // float2 units: conflict-free for the phase-A writes and the phase-B reads
#define BW_ENTRY_DW 12      // x, y, A, B | C, opacity, r, g | b, list position (1-based), id, -

#define SGR_BWD_BODY(X, Y, A, B, CZ, OP, R, G_, BL, POS)                                          \
    "v_sub_f32 v120, " X ", %[px]\n"                                                              \
    "v_sub_f32 v121, " Y ", %[py]\n"                                                              \
    "v_mul_f32 v122, " B ", v121\n"                                                               \
    "v_fmac_f32 v122, " A ", v120\n"                                                              \
    "v_mul_f32 v123, " CZ ", v121\n"                                                              \
    "v_mul_f32 v123, v123, v121\n"                                                                \
    "v_fmac_f32 v123, v120, v122\n"      /* log2(e) * power */                                    \
    "v_exp_f32 v124, v123\n"             /* G */                                                  \
    "v_mov_b32 v126, 0\n"                                                                         \
    "v_mov_b32 v127, 0\n"                                                                         \
    "v_cmp_nlt_f32 %[m0], 0, v123\n"     /* !(power > 0) */                                       \
    "v_cmp_le_u32 %[m1], " POS ", %[lastc]\n" /* at or before this pixel's last contributor */    \
    "v_mul_f32 v125, " OP ", v124\n"                                                              \
    "v_min_f32 v125, 0x3f7d70a4, v125\n" /* alpha */                                              \
    "s_and_b64 %[m0], %[m0], %[m1]\n"                                                             \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v125\n" /* !(alpha < 1/255) */                                \
    "s_and_b64 %[m0], %[m0], %[inside]\n"                                                         \
    "s_and_b64 exec, %[m0], vcc\n"                                                                \
    "v_sub_f32 v120, 1.0, v125\n"                                                                 \
    "v_rcp_f32 v120, v120\n"             /* 1 / (1 - alpha) */                                    \
    "v_sub_f32 v121, %[lc], %[acc]\n"                                                             \
    "v_fmac_f32 %[acc], %[la], v121\n"   /* accum_rec . g  (backward.cu:514-516) */               \
    "v_mul_f32 %[T], %[T], v120\n"                                                                \
    "v_mul_f32 %[lc], " R ", %[g0]\n"                                                             \
    "v_fmac_f32 %[lc], " G_ ", %[g1]\n"                                                           \
    "v_fmac_f32 %[lc], " BL ", %[g2]\n"  /* last_color . g */                                     \
    "v_mul_f32 v127, v125, %[T]\n"       /* Wt = alpha * T */                                     \
    "v_sub_f32 v122, %[lc], %[acc]\n"                                                             \
    "v_mul_f32 v122, v122, %[T]\n"                                                                \
    "v_fmac_f32 v122, %[ntb], v120\n"    /* dL_dalpha (backward.cu:523-529) */                    \
    "v_mov_b32 %[la], v125\n"                                                                     \
    "v_mul_f32 v126, v124, v122\n"       /* Z = G * dL_dalpha */                                  \
    "s_mov_b64 exec, %[full]\n"                                                                   \
    "ds_write_b64 %[waddr], v[126:127]\n"                                                         \
    "v_add_u32 %[waddr], 520, %[waddr]\n"

// rows (1..16) queue entries starting at LDS address e_addr -> panel rows 0..rows-1 at w_addr (+ 8 * lane already added)
__device__ __forceinline__ void bwd_phase_a(uint32_t e_addr, uint32_t w_addr, int rows, unsigned long long inside_mask, float pixfx,
                                            float pixfy, float g0, float g1, float g2, float ntb, uint32_t lastc, float& T,
                                            float& acc_g, float& lc_g, float& last_alpha)
{
    unsigned long long full, m0, m1;
    asm volatile(
        "s_mov_b64 %[full], exec\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b128 v[100:103], %[eaddr]\n"
        "ds_read_b128 v[104:107], %[eaddr] offset:16\n"
        "ds_read_b64 v[108:109], %[eaddr] offset:32\n"
        "1:\n"
        "ds_read_b128 v[110:113], %[eaddr] offset:48\n"
        "ds_read_b128 v[114:117], %[eaddr] offset:64\n"
        "ds_read_b64 v[118:119], %[eaddr] offset:80\n"
        "s_waitcnt lgkmcnt(3)\n"
        SGR_BWD_BODY("v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109")
        "s_add_i32 %[n], %[n], -1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 3f\n"
        "ds_read_b128 v[100:103], %[eaddr] offset:96\n"
        "ds_read_b128 v[104:107], %[eaddr] offset:112\n"
        "ds_read_b64 v[108:109], %[eaddr] offset:128\n"
        "v_add_u32 %[eaddr], 96, %[eaddr]\n"
        "s_waitcnt lgkmcnt(4)\n"  /* the panel write of the previous entry may still be counted */
        SGR_BWD_BODY("v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119")
        "s_add_i32 %[n], %[n], -1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc0 1b\n"
        "3:\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [T] "+v"(T), [acc] "+v"(acc_g), [lc] "+v"(lc_g), [la] "+v"(last_alpha), [eaddr] "+v"(e_addr), [waddr] "+v"(w_addr),
          [n] "+s"(rows), [full] "=&s"(full), [m0] "=&s"(m0), [m1] "=&s"(m1)
        : [px] "v"(pixfx), [py] "v"(pixfy), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [ntb] "v"(ntb), [lastc] "v"(lastc),
          [inside] "s"(inside_mask)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
          "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "vcc", "scc",
          "memory");
}

#ifdef SGR_BWD_REF_A
// C++ restatement of bwd_phase_a (debug aid: -DSGR_BWD_REF_A)
__device__ __forceinline__ void bwd_phase_a_ref(const float* q, float2* zw, int lane, int rows, bool inside, float pixfx, float pixfy,
                                                float g0, float g1, float g2, float ntb, uint32_t lastc, float& T, float& acc_g,
                                                float& lc_g, float& last_alpha)
{
    for (int r = 0; r < rows; r++) {
        const float* e = q + r * BW_ENTRY_DW;
        const float dx = e[0] - pixfx, dy = e[1] - pixfy;
        const float power = dx * (e[2] * dx + e[3] * dy) + e[4] * dy * dy;
        const float G = __builtin_amdgcn_exp2f(power);
        const float alpha = fminf(0.99f, e[5] * G);
        const bool active = inside && (__float_as_uint(e[9]) <= lastc) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        float Z = 0.f, Wt = 0.f;
        if (active) {
            const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
            acc_g += last_alpha * (lc_g - acc_g);
            T = T * inv;
            lc_g = e[6] * g0 + e[7] * g1 + e[8] * g2;
            Wt = alpha * T;
            const float d = (lc_g - acc_g) * T + ntb * inv;
            last_alpha = alpha;
            Z = G * d;
        }
        zw[r * BW_ZW_STRIDE + lane] = make_float2(Z, Wt);
    }
}
#endif

__global__ void __launch_bounds__(64) k_blend_bwd_w(int W, int H, int gx, int T_tiles, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ point_list, const GeomRec* __restrict__ rec,
                                                    const float* __restrict__ bg, const float* __restrict__ final_Ts,
                                                    const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_maxc,
                                                    const float* __restrict__ dL_dpix, float* __restrict__ acc)
{
    __shared__ __attribute__((aligned(16))) float s_q[(BW_QCAP + 1) * BW_ENTRY_DW];  // (+1: phase A's look-ahead)
    __shared__ float2 s_zw[BW_SUB * BW_ZW_STRIDE];
    const int wg = blockIdx.x;
    const int sub = (wg >> 3) & 3;
    const int tile = ((wg >> 5) << 3) + (wg & 7);
    if (tile >= T_tiles) return;
    const int total = (int)tile_maxc[tile];  // entries at list positions > tile_maxc contribute to no pixel of the tile
    if (total == 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int bx0 = tx * SGR_TILE_X + 8 * (sub & 1), by0 = ty * SGR_TILE_Y + 8 * (sub >> 1);
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const unsigned long long inside_mask = __ballot(inside);
    if (inside_mask == 0ull) return;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile];
    const size_t pix_id = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_Ts[pix_id] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) { g0 = dL_dpix[pix_id]; g1 = dL_dpix[HW + pix_id]; g2 = dL_dpix[2 * HW + pix_id]; }
    const float ntb = -T_final * (bg[0] * g0 + bg[1] * g1 + bg[2] * g2);
    float acc_g = 0.f, lc_g = 0.f, last_alpha = 0.f;

    // phase-B role: panel row bg_ (Gaussian), pixels 16 bq .. 16 bq + 15 of the block (rows 2 bq and 2 bq + 1)
    const int bg_ = lane & 15, bq = lane >> 4;
    float rg0[16], rg1[16], rg2[16];
    {
        float* gp = reinterpret_cast<float*>(s_zw);
        gp[lane] = g0; gp[64 + lane] = g1; gp[128 + lane] = g2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; i++) { rg0[i] = gp[16 * bq + i]; rg1[i] = gp[64 + 16 * bq + i]; rg2[i] = gp[128 + 16 * bq + i]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t q_lds = (uint32_t)(uintptr_t)s_q;
    const uint32_t zw_lds = (uint32_t)(uintptr_t)s_zw + 8u * (uint32_t)lane;

    // back to front: lane l of batch `base` holds list entry total - 1 - base - l.  Software pipeline over the batches: the
    // gather of batch b (and the ids of batch b + 1) is ISSUED, then the groups already waiting in the queue are processed
    // while those loads travel, and only then are the loaded records culled and appended to the queue.  (The loads are
    // plain compiler-scheduled ones, issued and consumed inside the same iteration, unconditional with clamped indices.)
    int qn = 0;  // entries waiting in the queue (they sit at its front)
    // process the queued entries in groups of 16 (all of them if `last_batch`), then move the rest to the front
    auto drain = [&](const bool last_batch) {
        int qs = 0;
        while (qn - qs >= BW_SUB || (last_batch && qn > qs)) {
            const int rows = min(BW_SUB, qn - qs);
#ifdef SGR_BWD_REF_A
            bwd_phase_a_ref(s_q + qs * BW_ENTRY_DW, s_zw, lane, rows, inside, pixfx, pixfy, g0, g1, g2, ntb, last_contributor, T, acc_g,
                            lc_g, last_alpha);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#else
            bwd_phase_a(q_lds + (uint32_t)qs * (BW_ENTRY_DW * 4), zw_lds, rows, inside_mask, pixfx, pixfy, g0, g1, g2, ntb,
                        last_contributor, T, acc_g, lc_g, last_alpha);
#endif
            // ---------------- phase B: lane = (panel row bg_, pixel rows 2 bq and 2 bq + 1)
            if (bg_ < rows) {
                const float* e = s_q + (qs + bg_) * BW_ENTRY_DW;
                const float2* row = s_zw + bg_ * BW_ZW_STRIDE + 16 * bq;
                float s0 = 0.f, sx = 0.f, sxx = 0.f, t0 = 0.f, tx1 = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) {
#ifdef SGR_BWD_HALF
                    if (i == 8) __builtin_amdgcn_sched_barrier(0);  // two groups of eight loads: fewer registers in flight
#endif
                    const float2 v = row[i];
                    const float xi = (float)(i & 7);
                    s0 += v.x; sx += v.x * xi; sxx += v.x * (xi * xi);
                    if (i >= 8) { t0 += v.x; tx1 += v.x * xi; }
                    k0 += v.y * rg0[i]; k1 += v.y * rg1[i]; k2 += v.y * rg2[i];
                }
                // raw moments about the block origin: y = 2 bq + (i >> 3)
                const float y0 = (float)(2 * bq);
                float o[9] = {k0, k1, k2, s0, sx, y0 * s0 + t0, sxx, y0 * sx + tx1, y0 * y0 * s0 + (2.f * y0 + 1.f) * t0};
#pragma unroll
                for (int v = 0; v < 9; v++) o[v] = sum_over_rows(o[v]);
                {
                    // d = centre - pixel = (xb - x, yb - y): shift the raw moments to the Gaussian's centre (every lane: the
                    // four lanes of a Gaussian hold the same sums after the reduction)
                    const float xb = e[0] - (float)bx0, yb = e[1] - (float)by0;
                    const float S0 = o[3], Sx = o[4], Sy = o[5], Sxx = o[6], Sxy = o[7], Syy = o[8];
                    const float dxs = xb * S0 - Sx, dys = yb * S0 - Sy;
                    const float dxx = xb * (xb * S0 - 2.f * Sx) + Sxx;
                    const float dyy = yb * (yb * S0 - 2.f * Sy) + Syy;
                    const float dxy = xb * dys - yb * Sx + Sxy;  // xb yb S0 - xb Sy - yb Sx + Sxy
                    // the nine sums of a pair are handed to nine neighbouring lanes through LDS (the panel is free now), so
                    // that one atomic instruction carries whole 36-byte records (four Gaussians at a time) and the memory
                    // system sees ONE request per (block, Gaussian) pair instead of nine
                    float* tb = reinterpret_cast<float*>(s_zw) + bg_ * 16;
                    if (bq == 0) { tb[0] = o[0]; tb[1] = o[1]; tb[2] = o[2]; tb[3] = S0; }
                    else if (bq == 1) { tb[4] = dxs; tb[5] = dys; tb[6] = dxx; tb[7] = dxy; }
                    else if (bq == 2) { tb[8] = dyy; tb[9] = e[10]; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                const float* tbl = reinterpret_cast<const float*>(s_zw);
                const int c = lane & 15;
#pragma unroll
                for (int pass = 0; pass < 4; pass++) {
                    const int g = 4 * pass + (lane >> 4);
                    if (g < rows && c < 9) {
                        const float val = tbl[g * 16 + c];
#ifndef SGR_BWD_NO_ATOMICS
                        if (val != 0.f) atomicAdd(acc + (size_t)__float_as_uint(tbl[g * 16 + 9]) * SGR_ACC_STRIDE + c, val);
#else
                        if (val == 12345.f) atomicAdd(acc + (size_t)__float_as_uint(tbl[g * 16 + 9]) * SGR_ACC_STRIDE + c, val);
#endif
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            qs += rows;
        }
        // the (fewer than 16) entries left over move to the front of the queue
        const int rem = qn - qs;
        if (qs > 0 && rem > 0) {
            float4 t0v, t1v, t2v;
            if (lane < rem) {
                const float4* src = reinterpret_cast<const float4*>(s_q + (qs + lane) * BW_ENTRY_DW);
                t0v = src[0]; t1v = src[1]; t2v = src[2];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < rem) {
                float4* dstq = reinterpret_cast<float4*>(s_q + lane * BW_ENTRY_DW);
                dstq[0] = t0v; dstq[1] = t1v; dstq[2] = t2v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        qn = rem;
    };
    uint32_t id_next = point_list[r0 + (uint32_t)max(total - 1 - lane, 0)];
    for (int base = 0; base < total; base += 64) {
        const uint32_t id_cur = id_next;
        const float4* rp = reinterpret_cast<const float4*>(rec + id_cur);
        const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
        id_next = point_list[r0 + (uint32_t)max(total - 1 - base - 64 - lane, 0)];
        drain(false);
        const bool hit = (base + lane < total) && block_hit(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, (float)bx0, (float)by0);
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const uint32_t pos = (uint32_t)qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            float4* e = reinterpret_cast<float4*>(s_q + pos * BW_ENTRY_DW);
            e[0] = make_float4(v0.x, v0.y, -0.5f * LOG2E * v0.z, -LOG2E * v0.w);
            e[1] = make_float4(-0.5f * LOG2E * v1.x, v1.y, v2.x, v2.y);
            e[2] = make_float4(v2.z, __uint_as_float((uint32_t)(total - base - lane)), __uint_as_float(id_cur), 0.f);
        }
        qn += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    drain(true);
}

}  // namespace

void sgr_launch_blend_fwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const GeomRec* rec, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc,
                          uint32_t* tile_walked, float* out_color, const uint32_t* guard_hdr, uint32_t list_cap, hipStream_t s)
{
    if (g_sgr_blend_variant & 1) {
        const int T = gx * gy;
        // tile_maxc and tile_walked are adjacent (each padded to 256 bytes): one memset
        (void)hipMemsetAsync(tile_maxc, 0, (size_t)((char*)tile_walked - (char*)tile_maxc) + (size_t)T * 4, s);
        hipLaunchKernelGGL(k_blend_fwd_w, dim3(32 * ((T + 7) / 8)), dim3(64), 0, s, W, H, gx, T, tile_start, point_list, rec, bg,
                           final_T, n_contrib, tile_maxc, tile_walked, out_color, guard_hdr, list_cap);
        return;
    }
    hipLaunchKernelGGL(k_blend_fwd, dim3(gx * gy), dim3(256), 0, s, W, H, gx, tile_start, point_list, rec, bg, final_T,
                       n_contrib, tile_maxc, tile_walked, out_color, guard_hdr, list_cap);
}

void sgr_launch_blend_bwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const GeomRec* rec, const float* bg, const float* final_T, const uint32_t* n_contrib,
                          const uint32_t* tile_maxc, const float* dL_dpix, float* acc, hipStream_t s)
{
    if (g_sgr_blend_variant & 2) {
        const int T = gx * gy;
        hipLaunchKernelGGL(k_blend_bwd_w, dim3(32 * ((T + 7) / 8)), dim3(64), 0, s, W, H, gx, T, tile_start, point_list, rec, bg,
                           final_T, n_contrib, tile_maxc, dL_dpix, acc);
        return;
    }
    hipLaunchKernelGGL(k_blend_bwd, dim3(gx * gy), dim3(256), 0, s, W, H, gx, tile_start, point_list, rec, bg, final_T,
                       n_contrib, tile_maxc, dL_dpix, acc);
}
