// blend.hip -- per-tile front-to-back alpha compositing (forward) and its backward for gfx950.
//
// Replaces renderCUDA forward  DGR/cuda_rasterizer/forward.cu:261-374
//      and renderCUDA backward DGR/cuda_rasterizer/backward.cu:399-557.
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; lane l of wave w owns pixel (x = l & 15, y = 4w + (l >> 4)).
// The tile's depth-sorted Gaussian list is staged through LDS in batches: each thread gathers ONE 48-byte
// GeomRec (three dwordx4 loads from one or two cache lines) and the whole workgroup then walks the batch
// with uniform-address (broadcast, conflict-free) LDS reads.
//
// Backward (v1): the per-(pixel, Gaussian) gradient terms are reduced across the 64 lanes of a wave with a
// butterfly before touching memory, so a (tile, Gaussian) pair costs 4 x 9 global atomics instead of the
// reference's up to 256 x 9 (backward.cu:523-554).  The walk starts at the tile's deepest contributor
// (tile_maxc, recorded by the forward) instead of at the end of the tile's list.
#include "sgr_common.h"

namespace {

#define BATCH 256

struct StageFwd {
    float4 a[BATCH];  // x, y, conic.x, conic.y
    float4 b[BATCH];  // conic.z, opacity, r, g
    float c[BATCH];   // b
};

__global__ void __launch_bounds__(256) k_blend_fwd(int W, int H, int gx, const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ point_list,
                                                   const GeomRec* __restrict__ rec, const float* __restrict__ bg,
                                                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                   uint32_t* __restrict__ tile_maxc, uint32_t* __restrict__ tile_walked,
                                                   float* __restrict__ out_color)
{
    __shared__ StageFwd st;
    __shared__ int s_done[4];
    __shared__ uint32_t s_maxc[4];
    __shared__ uint32_t s_walk[4];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int px = tx * SGR_TILE_X + (tid & 15), py = ty * SGR_TILE_Y + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile], r1 = tile_start[tile + 1];
    const int total = (int)(r1 - r0);

    bool done = !inside;
    float T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;

    for (int base = 0; base < total; base += BATCH) {
        // workgroup vote: stop when every pixel is done (forward.cu:309-311)
        const unsigned long long m = __ballot(done);
        if ((tid & 63) == 0) s_done[wave] = (m == ~0ull);
        __syncthreads();
        if (s_done[0] & s_done[1] & s_done[2] & s_done[3]) break;
        const int nb = min(BATCH, total - base);
        if (tid < nb) {
            const uint32_t id = point_list[r0 + base + tid];
            const float4* rp = reinterpret_cast<const float4*>(rec + id);
            const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
            st.a[tid] = v0; st.b[tid] = v1; st.c[tid] = v2.x;
        }
        __syncthreads();
        for (int j = 0; !done && j < nb; j++) {
            contributor++;
            const float4 a = st.a[j];
            const float4 b = st.b[j];
            const float dx = a.x - pixfx, dy = a.y - pixfy;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, b.y * __expf(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) { done = true; continue; }
            const float w = alpha * T;
            C0 += b.z * w; C1 += b.w * w; C2 += st.c[j] * w;
            T = test_T;
            last_contributor = contributor;
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
    // furthest list position any pixel looked at (R_f of the roofline accounting)
    uint32_t wk = inside ? contributor : 0u;
    for (int o = 32; o > 0; o >>= 1) wk = max(wk, (uint32_t)__shfl_xor((int)wk, o));
    if ((tid & 63) == 0) { s_maxc[wave] = mc; s_walk[wave] = wk; }
    __syncthreads();
    if (tid == 0) {
        tile_maxc[tile] = max(max(s_maxc[0], s_maxc[1]), max(s_maxc[2], s_maxc[3]));
        tile_walked[tile] = max(max(s_walk[0], s_walk[1]), max(s_walk[2], s_walk[3]));
    }
}

struct StageBwd {
    float4 a[BATCH];
    float4 b[BATCH];
    float c[BATCH];
    uint32_t id[BATCH];
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256) k_blend_bwd(int W, int H, int gx, const uint32_t* __restrict__ tile_start,
                                                   const uint32_t* __restrict__ point_list,
                                                   const GeomRec* __restrict__ rec, const float* __restrict__ bg,
                                                   const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                                                   const uint32_t* __restrict__ tile_maxc, const float* __restrict__ dL_dpix,
                                                   float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                                                   float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor)
{
    __shared__ StageBwd st;
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int px = tx * SGR_TILE_X + (tid & 15), py = ty * SGR_TILE_Y + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile];
    const int total = (int)tile_maxc[tile];  // entries at list positions > tile_maxc contribute to no pixel
    if (total == 0) return;

    const size_t pix_id = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_Ts[pix_id] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) { g0 = dL_dpix[pix_id]; g1 = dL_dpix[HW + pix_id]; g2 = dL_dpix[2 * HW + pix_id]; }
    const float bg_dot_dpixel = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;  // accum_rec
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;     // last_color
    float last_alpha = 0.f;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // walk list positions total, total-1, ..., 1 (1-based, as n_contrib counts them)
    for (int base = 0; base < total; base += BATCH) {
        __syncthreads();
        const int nb = min(BATCH, total - base);
        if (tid < nb) {
            const uint32_t id = point_list[r0 + (uint32_t)(total - 1 - base - tid)];
            const float4* rp = reinterpret_cast<const float4*>(rec + id);
            const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
            st.a[tid] = v0; st.b[tid] = v1; st.c[tid] = v2.x; st.id[tid] = id;
        }
        __syncthreads();
        for (int j = 0; j < nb; j++) {
            const int pos = total - base - j;  // 1-based list position of this entry
            const float4 a = st.a[j];
            const float4 b = st.b[j];
            const float dx = a.x - pixfx, dy = a.y - pixfy;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, b.y * G);
            const bool active = (pos <= last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            if (__ballot(active) == 0ull) continue;  // wave-uniform
            float dcol0 = 0.f, dcol1 = 0.f, dcol2 = 0.f, dmx = 0.f, dmy = 0.f, dcx = 0.f, dcy = 0.f, dcz = 0.f, dop = 0.f;
            if (active) {
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                const float c0 = b.z, c1 = b.w, c2 = st.c[j];
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = c0;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = c1;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = c2;
                float dL_dalpha = (c0 - acc0) * g0 + (c1 - acc1) * g1 + (c2 - acc2) * g2;
                dcol0 = dchannel_dcolor * g0; dcol1 = dchannel_dcolor * g1; dcol2 = dchannel_dcolor * g2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                dmx = dL_dG * dG_ddelx * ddelx_dx;
                dmy = dL_dG * dG_ddely * ddely_dy;
                dcx = -0.5f * gdx * dx * dL_dG;
                dcy = -0.5f * gdx * dy * dL_dG;
                dcz = -0.5f * gdy * dy * dL_dG;
                dop = G * dL_dalpha;
            }
            dcol0 = wave_sum(dcol0); dcol1 = wave_sum(dcol1); dcol2 = wave_sum(dcol2);
            dmx = wave_sum(dmx); dmy = wave_sum(dmy);
            dcx = wave_sum(dcx); dcy = wave_sum(dcy); dcz = wave_sum(dcz);
            dop = wave_sum(dop);
            if (lane == 0) {
                const size_t id = st.id[j];
                atomicAdd(&dL_dcolor[3 * id + 0], dcol0);
                atomicAdd(&dL_dcolor[3 * id + 1], dcol1);
                atomicAdd(&dL_dcolor[3 * id + 2], dcol2);
                atomicAdd(&dL_dmean2D[3 * id + 0], dmx);
                atomicAdd(&dL_dmean2D[3 * id + 1], dmy);
                atomicAdd(&dL_dconic[4 * id + 0], dcx);
                atomicAdd(&dL_dconic[4 * id + 1], dcy);
                atomicAdd(&dL_dconic[4 * id + 3], dcz);
                atomicAdd(&dL_dopacity[id], dop);
            }
        }
    }
}

}  // namespace

void sgr_launch_blend_fwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const GeomRec* rec, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc,
                          uint32_t* tile_walked, float* out_color, hipStream_t s)
{
    hipLaunchKernelGGL(k_blend_fwd, dim3(gx * gy), dim3(256), 0, s, W, H, gx, tile_start, point_list, rec, bg, final_T,
                       n_contrib, tile_maxc, tile_walked, out_color);
}

void sgr_launch_blend_bwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const GeomRec* rec, const float* bg, const float* final_T, const uint32_t* n_contrib,
                          const uint32_t* tile_maxc, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                          float* dL_dopacity, float* dL_dcolor, hipStream_t s)
{
    hipLaunchKernelGGL(k_blend_bwd, dim3(gx * gy), dim3(256), 0, s, W, H, gx, tile_start, point_list, rec, bg, final_T,
                       n_contrib, tile_maxc, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
}
