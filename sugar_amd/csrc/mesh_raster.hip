// mesh_raster.hip -- nearest-face triangle z-buffer for gfx950: the `fragments.zbuf` / `fragments.pix_to_face` the level-set
// sampler of SuGaR's coarse-mesh extraction reads from `pytorch3d.renderer.MeshRasterizer`
//   /root/reference/sugar_scene/sugar_model.py:1880-1893 (the rasterizer it builds: blur_radius 0, faces_per_pixel 10),
//   :1912-1928 (splat mesh -> fragments -> depth), :1966 (front Gaussian = pix_to_face // n_triangles_per_gaussian),
//   /root/reference/sugar_extractors/coarse_mesh.py:26 (use_gaussian_depth_for_surface_levels = False), :216-225, :271-287.
//
// Replaces pytorch3d 0.7.4's `_C.rasterize_meshes` (csrc/rasterize_meshes/rasterize_meshes.cu) for ONE mesh at blur_radius = 0.
// pytorch3d is absent from this image and from /root/reference: the per-pixel rule below restates its published
// CheckPixelInsideFace / geometry_utils.cuh formulas (see oracle/mesh_rasterizer.c, which cites them function by function and
// is PARITY-UNPINNED against pytorch3d itself).  This translation unit is compiled with -ffp-contract=off: every float
// operation of the rule is an individually rounded IEEE operation in the oracle's order, so pix_to_face, zbuf, the
// barycentric coordinates and the distances agree with the oracle BIT FOR BIT (tests/test_gpu_mesh_raster.py).
//
// Design (not pytorch3d's coarse-to-fine bins): the faces go through the SAME depth sort and two-level ordered tile binning as
// the Gaussians (binning.hip, binning2.hip) -- sort key = a lower bound of the depth any pixel of the face can have -- so every
// 16 x 16 tile's list comes out front to back.  One wave per 8 x 8 pixel block then walks the list (lane = pixel): a batch of
// 64 faces is gathered (one 48-byte record per lane), culled against the block's NDC rectangle, compacted into LDS and
// evaluated with broadcast LDS reads; a pixel keeps its K nearest covering faces as a sorted register array of
// (depth bits, face) 64-bit keys.  Because the list is ordered by the depth bound, a block STOPS as soon as all its pixels
// hold K faces nearer than the next face's bound -- the z-buffer of a 1M-Gaussian splat mesh touches the front few per cent of
// the (tile, face) instances.  The result does not depend on the order (the K smallest (z, face) pairs, ascending).
//
// Differences from pytorch3d that the oracle shares or documents:
//   * entries with exactly equal z at one pixel are ordered by face index (pytorch3d: by the history of its unsorted queue);
//   * blur_radius must be 0 (what SuGaR passes); clip_barycentric_coords must be false (pytorch3d's default for blur 0);
//   * `clipped_faces_neighbor_idx` (the two halves of a face split by the near-plane clip) only acts at blur 0 if a pixel
//     centre is STRICTLY inside both halves, which takes opposite roundings of the shared edge's two edge functions: not handled;
//   * a face with a non-finite coordinate is skipped (in pytorch3d its barycentrics are NaN and no pixel is inside);
//   * no max_faces_per_bin: pytorch3d drops the faces of an overflowing bin with a warning, nothing is dropped here.
#include "sgr_common.h"

#include <string>

int sgr_fail(int code, const char* msg);  // capi.hip: sets sgr_last_error() of the calling thread

namespace {

#define MR_EPS 1e-8f          // kEpsilon, pytorch3d/csrc/utils/geometry_utils.cuh
#define MR_EMPTY64 0xFFFFFFFFFFFFFFFFull
#define MR_ENTRY_DW 24        // dwords per staged face

// one face, 48 bytes: the nine floats of face_verts, the sort key's value (lower bound of the depth of any covered pixel),
// and BarycentricCoordsForward's `area`
struct MeshRec { float x0, y0, z0, x1, y1, z1, x2, y2, z2, bound, area, pad; };
static_assert(sizeof(MeshRec) == 48, "MeshRec must be 48 bytes");

// rasterization_utils.cuh: PixToNonSquareNdc (identical float operations to oracle/mesh_rasterizer.c)
__device__ __forceinline__ float pix_to_ndc(int i, int S1, int S2)
{
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S1;
}

// pixel index (possibly fractional, possibly far outside the image) whose centre has NDC coordinate x: the inverse of the
// function above, used only for conservative tile rectangles
__device__ __forceinline__ float ndc_to_pix(float x, int S1, int S2)
{
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    const float offset = range / 2.0f;
    return ((x + offset) * (float)S1 - offset) / range;
}

// Per face: validity (CheckPixelInsideFace's per-face tests), the 48-byte record, the depth-sort key and the packed tile
// rectangle -- the contract of binning.hip's depth sort (keys, rect_by_id, per-workgroup key range, zeroed counters).
__global__ void __launch_bounds__(256) k_mesh_setup(int F, const float* __restrict__ fv, int W, int H, int gx, int gy, int persp,
                                                    int cull_back, MeshRec* __restrict__ rec, uint32_t* __restrict__ keys,
                                                    uint2* __restrict__ rect_by_id, uint2* __restrict__ key_minmax,
                                                    uint32_t* __restrict__ sort_counters, int n_counters)
{
    __shared__ float s_v[256 * 9];
    __shared__ uint32_t s_mm[2][4];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * 256;
    if (blockIdx.x == 0)
        for (int i = tid; i < n_counters; i += 256) sort_counters[i] = 0u;
    // the 256 faces of the workgroup are 2304 consecutive floats: coalesced loads, then a stride-9 read per lane (no bank conflicts)
    const size_t g0 = (size_t)base * 9, gl = (size_t)F * 9 - 1;
    float t[9];
#pragma unroll
    for (int j = 0; j < 9; j++) { const size_t g = g0 + (size_t)(j * 256 + tid); t[j] = fv[g < gl ? g : gl]; }
#pragma unroll
    for (int j = 0; j < 9; j++) s_v[j * 256 + tid] = t[j];
    __syncthreads();
    const int f = base + tid;
    float v[9];
#pragma unroll
    for (int j = 0; j < 9; j++) v[j] = s_v[tid * 9 + j];
    const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
    bool ok = f < F;
#pragma unroll
    for (int j = 0; j < 9; j++) ok = ok && __builtin_isfinite(v[j]);
    const float zmin = fminf(fminf(z0, z1), z2);
    if (zmin < MR_EPS) ok = false;  // CheckPointOutsideBoundingBox: z_invalid
    const float face_area = (x0 - x1) * (y2 - y1) - (y0 - y1) * (x2 - x1);  // EdgeFunctionForward(v0, v1, v2)
    if (face_area <= MR_EPS && face_area >= -1.0f * MR_EPS) ok = false;
    if (cull_back && face_area < 0.0f) ok = false;
    const float xmin = fminf(fminf(x0, x1), x2), xmax = fmaxf(fmaxf(x0, x1), x2);
    const float ymin = fminf(fminf(y0, y1), y2), ymax = fmaxf(fmaxf(y0, y1), y2);
    const float area = ((x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)) + MR_EPS;  // BarycentricCoordsForward: EdgeFunctionForward(v2, v0, v1) + kEpsilon

    // tile rectangle, conservative by a pixel on every side (coverage itself is decided per pixel).  Output column c looks along
    // NDC x of pixel index W-1-c (the kernel's "reverse ordering of X and Y axes")
    uint2 rect = make_uint2(0u, 0u);
    if (ok) {
        const float lo_x = fmaxf(floorf(ndc_to_pix(xmin, W, H)) - 1.0f, 0.0f), hi_x = fminf(ceilf(ndc_to_pix(xmax, W, H)) + 1.0f, (float)(W - 1));
        const float lo_y = fmaxf(floorf(ndc_to_pix(ymin, H, W)) - 1.0f, 0.0f), hi_y = fminf(ceilf(ndc_to_pix(ymax, H, W)) + 1.0f, (float)(H - 1));
        if (lo_x <= hi_x && lo_y <= hi_y) {
            const int c0 = W - 1 - (int)hi_x, c1 = W - 1 - (int)lo_x, r0 = H - 1 - (int)hi_y, r1 = H - 1 - (int)lo_y;
            const int tx0 = c0 / SGR_TILE_X, tx1 = c1 / SGR_TILE_X + 1, ty0 = r0 / SGR_TILE_Y, ty1 = r1 / SGR_TILE_Y + 1;
            rect.x = (uint32_t)tx0 | ((uint32_t)ty0 << 16);
            rect.y = (uint32_t)(tx1 - tx0) | ((uint32_t)(ty1 - ty0) << 16);
        } else {
            ok = false;
        }
    }
    // Lower bound of pz over the pixels the face can cover.  With perspective-correct barycentrics b_i = top_i / sum(top) the
    // depth is a convex combination of the vertex depths (b_i > 0 for a covered pixel), so pz >= zmin up to a few ulps -- unless
    // sum(top) < kEpsilon engages the clamp of BarycentricPerspectiveCorrectionForward, which needs depths below ~1e-4 or edge
    // functions that do not add up to the area (slivers: rounding of order 4e-7 * diagonal^2 against the area).  Faces outside
    // that regime get the smallest bound: they sort to the front of every list and are never skipped.
    const float d2 = (xmax - xmin) * (xmax - xmin) + (ymax - ymin) * (ymax - ymin);
    const bool regular = zmin >= 1e-3f && fabsf(area) >= 1e-6f && fabsf(area) >= 1e-4f * d2;
    const float bound = regular ? zmin * (persp ? (1.0f - 4e-6f) : 0.96f) : 1e-30f;
    const uint32_t key = ok ? __float_as_uint(bound) : 0xFFFFFFFFu;
    if (f < F) {
        float4* dst = reinterpret_cast<float4*>(rec + f);
        dst[0] = make_float4(x0, y0, z0, x1);
        dst[1] = make_float4(y1, z1, x2, y2);
        dst[2] = make_float4(z2, bound, area, 0.0f);
        keys[f] = key;
        rect_by_id[f] = ok ? rect : make_uint2(0u, 0u);
    }
    uint32_t kmin = key, kmax = key == 0xFFFFFFFFu ? 0u : key;
    for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o));
    }
    if ((tid & 63) == 0) { s_mm[0][tid >> 6] = kmin; s_mm[1][tid >> 6] = kmax; }
    __syncthreads();
    if (tid == 0)
        key_minmax[blockIdx.x] = make_uint2(min(min(s_mm[0][0], s_mm[0][1]), min(s_mm[0][2], s_mm[0][3])),
                                            max(max(s_mm[1][0], s_mm[1][1]), max(s_mm[1][2], s_mm[1][3])));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}

// One wave per 8 x 8 pixel block (workgroup b runs on XCD b % 8: the four blocks of a tile share an L2, as in blend.hip).
// KQ = capacity of the per-pixel queue (>= K; the first K entries are written).
template <int KQ>
__global__ void __launch_bounds__(64) k_mesh_fine(int W, int H, int gx, int T_tiles, const uint32_t* __restrict__ tile_start,
                                                  const uint32_t* __restrict__ point_list, const MeshRec* __restrict__ rec, int K,
                                                  int persp, long long face_base, long long* __restrict__ pix_to_face,
                                                  float* __restrict__ zbuf)
{
    __shared__ __attribute__((aligned(16))) float s_e[64 * MR_ENTRY_DW];
    const int wg = blockIdx.x;
    const int sub = (wg >> 3) & 3;
    const int tile = ((wg >> 5) << 3) + (wg & 7);
    if (tile >= T_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int bx0 = tx * SGR_TILE_X + 8 * (sub & 1), by0 = ty * SGR_TILE_Y + 8 * (sub >> 1);
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool active = px < W && py < H;
    if (__ballot(active) == 0ull) return;
    // NDC of this lane's pixel centre and of the block's rectangle (pixel index W-1-px grows to the left: reversed axes)
    const float pxf = pix_to_ndc(W - 1 - px, W, H), pyf = pix_to_ndc(H - 1 - py, H, W);
    const float bxmin = pix_to_ndc(W - 1 - (bx0 + 7), W, H), bxmax = pix_to_ndc(W - 1 - bx0, W, H);
    const float bymin = pix_to_ndc(H - 1 - (by0 + 7), H, W), bymax = pix_to_ndc(H - 1 - by0, H, W);

    unsigned long long q[KQ];
#pragma unroll
    for (int i = 0; i < KQ; i++) q[i] = MR_EMPTY64;
    // depth bits of the farthest kept face, maximum over the block's pixels; 0xFFFFFFFF while any pixel has a free slot
    uint32_t zmax_bits = 0xFFFFFFFFu;

    const uint32_t r0 = tile_start[tile];
    const int total = (int)(tile_start[tile + 1] - r0);
    uint32_t id_next = 0u;
    if (total > 0) id_next = point_list[r0 + (uint32_t)min(lane, total - 1)];
    for (int base = 0; base < total; base += 64) {
        const float4* rp = reinterpret_cast<const float4*>(rec + id_next);
        const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];  // {x0,y0,z0,x1} {y1,z1,x2,y2} {z2,bound,area,-}
        const uint32_t id = id_next;
        id_next = point_list[r0 + (uint32_t)min(base + 64 + lane, total - 1)];
        const uint32_t kb = __float_as_uint(v2.y);
        // the list is ordered by the bound: once the first face of a batch cannot enter any pixel's queue, none behind it can
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)kb) > zmax_bits) break;
        const float x0 = v0.x, y0 = v0.y, z0 = v0.z, x1 = v0.w, y1 = v1.x, z1 = v1.y, x2 = v1.z, y2 = v1.w, z2 = v2.x;
        const float xmin = fminf(fminf(x0, x1), x2), xmax = fmaxf(fmaxf(x0, x1), x2);
        const float ymin = fminf(fminf(y0, y1), y2), ymax = fmaxf(fmaxf(y0, y1), y2);
        // same comparisons as the per-pixel bounding-box test, on the block's outermost pixel centres
        const bool hit = (base + lane < total) && !(bxmin > xmax || bxmax < xmin || bymin > ymax || bymax < ymin) && !(kb > zmax_bits);
        const unsigned long long m = __ballot(hit);
        const int n = __popcll(m);
        if (hit) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            float4* e = reinterpret_cast<float4*>(s_e + pos * MR_ENTRY_DW);
            e[0] = make_float4(x0, y0, x1, y1);
            e[1] = make_float4(x2, y2, y2 - y1, x2 - x1);
            e[2] = make_float4(y0 - y2, x0 - x2, y1 - y0, x1 - x0);
            e[3] = make_float4(z0, z1, z2, v2.z);
            e[4] = make_float4(xmin, xmax, ymin, ymax);
            e[5] = make_float4(__uint_as_float(id), 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < n; k++) {
            const float4* e = reinterpret_cast<const float4*>(s_e + k * MR_ENTRY_DW);
            const float4 a = e[0], b = e[1], c = e[2], z = e[3], bb = e[4];
            const bool in_box = active && !(pxf > bb.y || pxf < bb.x || pyf > bb.w || pyf < bb.z);
            // EdgeFunctionForward(p, v1, v2), (p, v2, v0), (p, v0, v1)
            const float e0 = (pxf - a.z) * b.z - (pyf - a.w) * b.w;
            const float e1 = (pxf - b.x) * c.x - (pyf - b.y) * c.y;
            const float e2 = (pxf - a.x) * c.z - (pyf - a.y) * c.w;
            // w_i = e_i / area > 0 needs e_i and area of one sign: a necessary condition, the exact rule follows for the lanes that pass
            const bool pre = in_box && (z.w >= 0.0f ? (e0 > 0.0f && e1 > 0.0f && e2 > 0.0f) : (e0 < 0.0f && e1 < 0.0f && e2 < 0.0f));
            if (__ballot(pre) == 0ull) continue;
            const float w0 = e0 / z.w, w1 = e1 / z.w, w2 = e2 / z.w;  // BarycentricCoordsForward
            float b0 = w0, b1 = w1, b2 = w2;
            if (persp) {  // BarycentricPerspectiveCorrectionForward
                const float t0 = w0 * z.y * z.z;
                const float t1 = z.x * w1 * z.z;
                const float t2 = z.x * z.y * w2;
                const float den = fmaxf(t0 + t1 + t2, MR_EPS);
                b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
            }
            const float pz = b0 * z.x + b1 * z.y + b2 * z.z;
            const bool ins = pre && !(pz < 0.0f) && b0 > 0.0f && b1 > 0.0f && b2 > 0.0f;
            unsigned long long nk = ins ? (((unsigned long long)__float_as_uint(pz) << 32) | (unsigned long long)__float_as_uint(e[5].x))
                                        : MR_EMPTY64;
            if (__ballot(nk < q[KQ - 1]) == 0ull) continue;
#pragma unroll
            for (int i = 0; i < KQ; i++) {  // sorted insertion: ascending (depth bits, face)
                const unsigned long long lo = nk < q[i] ? nk : q[i];
                const unsigned long long hi = nk < q[i] ? q[i] : nk;
                q[i] = lo;
                nk = hi;
            }
        }
        zmax_bits = wave_max_u32(active ? (uint32_t)(q[KQ - 1] >> 32) : 0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (active) {
        const size_t o = ((size_t)py * W + px) * (size_t)K;
#pragma unroll
        for (int k = 0; k < KQ; k++) {
            if (k < K) {
                const uint32_t code = (uint32_t)q[k];
                const bool have = q[k] != MR_EMPTY64;
                pix_to_face[o + k] = have ? (long long)code + face_base : -1ll;
                zbuf[o + k] = have ? __uint_as_float((uint32_t)(q[k] >> 32)) : -1.0f;
            }
        }
    }
}

// geometry_utils.cuh: PointLineDistanceForward (squared distance to the segment)
__device__ __forceinline__ float point_line_distance(float px, float py, float ax, float ay, float bx, float by)
{
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    if (l2 <= MR_EPS) return (px - bx) * (px - bx) + (py - by) * (py - by);
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = qx - px, dy = qy - py;
    return dx * dx + dy * dy;
}

// The interpolation weights and the image-plane distance of the kept faces, recomputed with the rule's own operations
// (bit-identical to what decided them): one thread per (pixel, slot).
__global__ void __launch_bounds__(256) k_mesh_attrs(long long n_slots, int W, int H, int K, int persp, long long face_base,
                                                    const float* __restrict__ fv, const long long* __restrict__ pix_to_face,
                                                    float* __restrict__ bary, float* __restrict__ dists)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    const long long f = pix_to_face[i];
    float b0 = -1.0f, b1 = -1.0f, b2 = -1.0f, sd = -1.0f;
    if (f >= 0) {
        const long long pix = i / K;
        const int py = (int)(pix / W), px = (int)(pix - (long long)py * W);
        const float pxf = pix_to_ndc(W - 1 - px, W, H), pyf = pix_to_ndc(H - 1 - py, H, W);
        const float* v = fv + 9 * (f - face_base);
        const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
        const float area = ((x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)) + MR_EPS;
        const float w0 = ((pxf - x1) * (y2 - y1) - (pyf - y1) * (x2 - x1)) / area;
        const float w1 = ((pxf - x2) * (y0 - y2) - (pyf - y2) * (x0 - x2)) / area;
        const float w2 = ((pxf - x0) * (y1 - y0) - (pyf - y0) * (x1 - x0)) / area;
        b0 = w0; b1 = w1; b2 = w2;
        if (persp) {
            const float t0 = w0 * z1 * z2;
            const float t1 = z0 * w1 * z2;
            const float t2 = z0 * z1 * w2;
            const float den = fmaxf(t0 + t1 + t2, MR_EPS);
            b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
        }
        const float e01 = point_line_distance(pxf, pyf, x0, y0, x1, y1);
        const float e02 = point_line_distance(pxf, pyf, x0, y0, x2, y2);
        const float e12 = point_line_distance(pxf, pyf, x1, y1, x2, y2);
        sd = -fminf(fminf(e01, e02), e12);  // inside the face: signed_dist = -dist
    }
    if (bary) { bary[3 * i] = b0; bary[3 * i + 1] = b1; bary[3 * i + 2] = b2; }
    if (dists) dists[i] = sd;
}

__global__ void __launch_bounds__(256) k_mesh_fill_empty(long long n_slots, long long* __restrict__ pix_to_face, float* __restrict__ zbuf,
                                                         float* __restrict__ bary, float* __restrict__ dists)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    pix_to_face[i] = -1ll;
    zbuf[i] = -1.0f;
    if (bary) { bary[3 * i] = -1.0f; bary[3 * i + 1] = -1.0f; bary[3 * i + 2] = -1.0f; }
    if (dists) dists[i] = -1.0f;
}

// The splat mesh of SuGaR's level-set sampler, straight from the Gaussian buffers: per Gaussian the four corners of its
// "diamond" (or "square") in the plane of its two larger axes (SuGaR.triangle_vertices, sugar_scene/sugar_model.py:481-514),
// pushed along the viewing ray onto the plane through the Gaussian's centre that faces the camera (SuGaR.splat_mesh,
// :695-716, mode 'perspective'), projected like pytorch3d's MeshRasterizer.transform does (NDC x, y; view-space z), and emitted
// as the two faces [0,2,1], [0,3,2] (:254-256) in the face_verts layout sgr_rasterize_meshes consumes.  The reference builds
// the same numbers with ~40 tensor operations over 4 P vertices (argsort, gathers, a quaternion -> matrix -> quaternion round
// trip, two camera transforms and their inverse): 3.6 ms at 1M Gaussians against 0.03 ms here.  Differences are float rounding
// of that round trip (1e-6 relative); the tests bound them.
//   scaling[P,3] activated, quaternions[P,4] UNIT, real part first; prim[12] = the four canonical corners (x = 0);
//   w2v / proj: pytorch3d row-vector 4x4 matrices (get_world_to_view_transform / get_projection_transform .get_matrix()).
struct SplatArgs { const float* points; const float* scaling; const float* quats; const float* prim; const float* w2v; const float* proj; float tri_scale; };
__global__ void __launch_bounds__(256) k_splat_face_verts(int P, SplatArgs a, float* __restrict__ fv)
{
    __shared__ float s_m[44];
    if (threadIdx.x < 16) s_m[threadIdx.x] = a.w2v[threadIdx.x];
    else if (threadIdx.x < 32) s_m[threadIdx.x] = a.proj[threadIdx.x - 16];
    else if (threadIdx.x < 44) s_m[threadIdx.x] = a.prim[threadIdx.x - 32];
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
    const float* V = s_m; const float* Pm = s_m + 16; const float* prim = s_m + 32;
    const float px = a.points[3 * (size_t)g], py = a.points[3 * (size_t)g + 1], pz = a.points[3 * (size_t)g + 2];
    const float s[3] = {a.scaling[3 * (size_t)g], a.scaling[3 * (size_t)g + 1], a.scaling[3 * (size_t)g + 2]};
    const float4 q = *reinterpret_cast<const float4*>(a.quats + 4 * (size_t)g);
    // smallest axis first, the two others in cyclic order (:494-496; ties: the lowest index)
    int ia = 0;
    if (s[1] < s[ia]) ia = 1;
    if (s[2] < s[ia]) ia = 2;
    const int ib = (ia + 1) % 3, ic = (ia + 2) % 3;
    const float r = q.x, i = q.y, j = q.z, k = q.w;  // pytorch3d quaternion_to_matrix (unit quaternion: two_s = 2)
    const float R[3][3] = {{1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r)},
                           {2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r)},
                           {2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)}};
    const float sa = a.tri_scale * s[ia], sb = a.tri_scale * s[ib], sc = a.tri_scale * s[ic];
    // centre in view space (row vectors: x_view = [x 1] @ w2v), the projection direction and the centre's projection on it
    const float cx = px * V[0] + py * V[4] + pz * V[8] + V[12], cy = px * V[1] + py * V[5] + pz * V[9] + V[13],
                cz = px * V[2] + py * V[6] + pz * V[10] + V[14];
    const float cn = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);  // F.normalize
    const float dx = cx / cn, dy = cy / cn, dz = cz / cn;
    const float cproj = cx * dx + cy * dy + cz * dz;
    float vx[4], vy[4], vz[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float l0 = prim[3 * c] * sa, l1 = prim[3 * c + 1] * sb, l2 = prim[3 * c + 2] * sc;  // corner in the sorted local frame
        const float wx = px + (R[0][ia] * l0 + R[0][ib] * l1 + R[0][ic] * l2), wy = py + (R[1][ia] * l0 + R[1][ib] * l1 + R[1][ic] * l2),
                    wz = pz + (R[2][ia] * l0 + R[2][ib] * l1 + R[2][ic] * l2);
        const float ex = wx * V[0] + wy * V[4] + wz * V[8] + V[12], ey = wx * V[1] + wy * V[5] + wz * V[9] + V[13],
                    ez = wx * V[2] + wy * V[6] + wz * V[10] + V[14];
        const float vproj = ex * dx + ey * dy + ez * dz;
        const float f = cproj / vproj;  // (:712: the corner slides along ITS ray onto the plane through the centre)
        const float sx = f * ex, sy = f * ey, sz = f * ez;
        // pytorch3d projection (row vectors): ndc_h = [x y z 1] @ proj, NDC = ndc_h.xy / ndc_h.w; z keeps the view-space depth
        const float hx = sx * Pm[0] + sy * Pm[4] + sz * Pm[8] + Pm[12], hy = sx * Pm[1] + sy * Pm[5] + sz * Pm[9] + Pm[13],
                    hw = sx * Pm[3] + sy * Pm[7] + sz * Pm[11] + Pm[15];
        vx[c] = hx / hw; vy[c] = hy / hw; vz[c] = sz;
    }
    float* o = fv + 18 * (size_t)g;
    const int tri[6] = {0, 2, 1, 0, 3, 2};
#pragma unroll
    for (int t = 0; t < 6; t++) { o[3 * t] = vx[tri[t]]; o[3 * t + 1] = vy[tri[t]]; o[3 * t + 2] = vz[tri[t]]; }
}

struct MeshLayout { size_t rec, sort, img, bin2, total; };
MeshLayout mesh_layout(int F, int W, int H)
{
    MeshLayout L;
    const ImgLayout IL = sgr_img_layout(W, H);
    const Bin2Layout B2 = sgr_bin2_layout(F, IL.gx, IL.gy);
    size_t off = 0;
    L.rec = off;  off = sgr_align(off + (size_t)(F > 0 ? F : 1) * sizeof(MeshRec));
    L.sort = off; off = sgr_align(off + sgr_sort_scratch_bytes(F));
    L.img = off;  off = sgr_align(off + IL.total);
    L.bin2 = off; off = sgr_align(off + B2.total);
    L.total = off;
    return L;
}

struct PinnedHdr { uint32_t* p = nullptr; };
thread_local PinnedHdr g_hdr;

#define MR_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return sgr_fail(SGR_E_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

template <int KQ>
void launch_fine(int W, int H, const ImgLayout& IL, const uint32_t* tile_start, const uint32_t* point_list, const MeshRec* rec, int K,
                 int persp, long long face_base, long long* p2f, float* zbuf, hipStream_t s)
{
    const int groups = (IL.T + 7) / 8;  // 32 workgroups (8 tiles x 4 blocks) per group: see the index arithmetic of the kernel
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mesh_fine<KQ>), dim3(groups * 32), dim3(64), 0, s, W, H, IL.gx, IL.T, tile_start, point_list, rec, K,
                       persp, face_base, p2f, zbuf);
}

}  // namespace

extern "C" {

size_t sgr_rasterize_meshes_scratch_bytes(int64_t F, int width, int height)
{
    if (F < 0 || F > 0x7FFFFF00ll || width <= 0 || height <= 0) return 0;
    return mesh_layout((int)F, width, height).total;
}

int64_t sgr_rasterize_meshes(const float* face_verts, int64_t F, int64_t face_index_base, int width, int height, float blur_radius,
                             int faces_per_pixel, int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                             char* scratch, size_t scratch_bytes, sgr_alloc_fn list_alloc, void* list_user, int64_t* pix_to_face,
                             float* zbuf, float* bary_coords, float* dists, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (width <= 0 || height <= 0 || F < 0) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: width, height must be positive, F >= 0");
    if (F > 0x7FFFFF00ll) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: more than 2^31 faces");
    if (blur_radius != 0.0f) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: only blur_radius == 0 (hard rasterization) is implemented");
    if (clip_barycentric_coords) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: clip_barycentric_coords is not implemented (pytorch3d's default for blur_radius 0 is False)");
    if (faces_per_pixel < 1 || faces_per_pixel > 16) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: faces_per_pixel must be in 1..16");
    if (!pix_to_face || !zbuf) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: null output");
    const int K = faces_per_pixel;
    const long long n_slots = (long long)width * height * K;
    const ImgLayout IL = sgr_img_layout(width, height);
    if (IL.gx > 65535 || IL.gy > 65535) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: image too large");
    if (F == 0) {
        hipLaunchKernelGGL(k_mesh_fill_empty, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, n_slots, (long long*)pix_to_face, zbuf,
                           bary_coords, dists);
        MR_HIP(hipGetLastError());
        return 0;
    }
    if (!face_verts || !scratch || !list_alloc) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: null input / scratch / allocator");
    const int Fi = (int)F;
    const MeshLayout ML = mesh_layout(Fi, width, height);
    if (scratch_bytes < ML.total) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: scratch smaller than sgr_rasterize_meshes_scratch_bytes()");
    const Bin2Layout B2 = sgr_bin2_layout(Fi, IL.gx, IL.gy);
    MeshRec* rec = reinterpret_cast<MeshRec*>(scratch + ML.rec);
    char* sort_scratch = scratch + ML.sort;
    char* img = scratch + ML.img;
    char* bin2 = scratch + ML.bin2;
    uint32_t* tile_start = reinterpret_cast<uint32_t*>(img + IL.tile_start);
    uint32_t* tile_cursor = reinterpret_cast<uint32_t*>(img + IL.tile_cursor);
    uint32_t* tile_maxc = reinterpret_cast<uint32_t*>(img + IL.tile_maxc);
    uint32_t* tile_walked = reinterpret_cast<uint32_t*>(img + IL.tile_walked);
    uint32_t* header = reinterpret_cast<uint32_t*>(img + IL.header);
    uint32_t* blk_hist = reinterpret_cast<uint32_t*>(img + IL.blk_hist);
    uint2* rects = reinterpret_cast<uint2*>(sort_scratch + sgr_sort_rects_offset(Fi));
    uint2* rect_by_id = reinterpret_cast<uint2*>(sort_scratch + sgr_sort_rect_by_id_offset(Fi));

    hipLaunchKernelGGL(k_mesh_setup, dim3((Fi + 255) / 256), dim3(256), 0, s, Fi, face_verts, width, height, IL.gx, IL.gy,
                       perspective_correct ? 1 : 0, cull_backfaces ? 1 : 0, rec, reinterpret_cast<uint32_t*>(sort_scratch), rect_by_id,
                       reinterpret_cast<uint2*>(sort_scratch + sgr_sort_minmax_offset(Fi)),
                       reinterpret_cast<uint32_t*>(sort_scratch + sgr_sort_counters_offset(Fi)), sgr_sort_counter_words());
    const uint32_t* order = nullptr;
    sgr_launch_gaussian_sort(Fi, sort_scratch, &order, rect_by_id, rects, s);
    sgr_launch_bin2_count(Fi, IL.gx, IL.gy, B2, bin2, header + 4, rects, order, tile_cursor, 0u, s);
    sgr_launch_tile_scan(IL.T, tile_cursor, tile_start, header, tile_maxc, tile_walked, 0, nullptr, nullptr, s);
    MR_HIP(hipGetLastError());
    if (!g_hdr.p) MR_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_hdr.p), 64, hipHostMallocDefault));
    MR_HIP(hipMemcpyAsync(g_hdr.p, header, 32, hipMemcpyDeviceToHost, s));
    MR_HIP(hipStreamSynchronize(s));  // the one host round trip: the instance list is sized by the count
    bool two_level = true;
    const bool legacy_ok = IL.n_blocks > 0;
    const int per_block = legacy_ok ? (((Fi + IL.n_blocks - 1) / IL.n_blocks + 63) / 64) * 64 : 0;
    if (g_hdr.p[4 + SGR_B2_HDR_OVERFLOW]) {
        // more (face, super-tile) pairs than the level-1 list holds (many screen-filling faces): single-level binning
        if (!legacy_ok) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: level-1 binning overflow on an image too large for the single-level fallback");
        two_level = false;
        sgr_launch_bin_count(Fi, IL.gx, IL.gy, IL.n_blocks, per_block, order, rects, blk_hist, s);
        sgr_launch_hist_scan(IL.T, IL.n_blocks, blk_hist, tile_cursor, s);
        sgr_launch_tile_scan(IL.T, tile_cursor, tile_start, header, tile_maxc, tile_walked, 1, nullptr, nullptr, s);
        MR_HIP(hipMemcpyAsync(g_hdr.p, header, 32, hipMemcpyDeviceToHost, s));
        MR_HIP(hipStreamSynchronize(s));
    }
    const int64_t R = (int64_t)g_hdr.p[SGR_HDR_R];
    // (k_tile_scan saturates the count at 2^32 - 1: the offsets of such a list have wrapped, as in sgr_forward_ex)
    if (R >= 0xFFFFFFFFll) return sgr_fail(SGR_E_INVALID, "rasterize_meshes: more than 2^32 - 2 (tile, face) instances in one view");
    const uint32_t n_chunks = g_hdr.p[4 + SGR_B2_HDR_CHUNKS];
    char* list = list_alloc(list_user, (size_t)(R > 0 ? R : 1) * 4 + 256);
    if (!list) return sgr_fail(SGR_E_ALLOC, "rasterize_meshes: instance list allocation failed");
    uint32_t* point_list = reinterpret_cast<uint32_t*>(list);
    if (R > 0) {
        if (two_level)
            sgr_launch_bin2_write(IL.gx, IL.gy, B2, bin2, header + 4, n_chunks, rects, order, tile_start, point_list, 0xFFFFFFFFu, nullptr, s);
        else
            sgr_launch_bin_scatter(Fi, IL.gx, IL.gy, IL.n_blocks, per_block, order, rects, tile_start, blk_hist, point_list, s);
    }
    const int persp = perspective_correct ? 1 : 0;
    long long* p2f = reinterpret_cast<long long*>(pix_to_face);
    if (K == 1) launch_fine<1>(width, height, IL, tile_start, point_list, rec, K, persp, face_index_base, p2f, zbuf, s);
    else if (K == 2) launch_fine<2>(width, height, IL, tile_start, point_list, rec, K, persp, face_index_base, p2f, zbuf, s);
    else if (K <= 4) launch_fine<4>(width, height, IL, tile_start, point_list, rec, K, persp, face_index_base, p2f, zbuf, s);
    else if (K <= 8) launch_fine<8>(width, height, IL, tile_start, point_list, rec, K, persp, face_index_base, p2f, zbuf, s);
    else if (K <= 10) launch_fine<10>(width, height, IL, tile_start, point_list, rec, K, persp, face_index_base, p2f, zbuf, s);
    else launch_fine<16>(width, height, IL, tile_start, point_list, rec, K, persp, face_index_base, p2f, zbuf, s);
    if (bary_coords || dists)
        hipLaunchKernelGGL(k_mesh_attrs, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, n_slots, width, height, K, persp,
                           (long long)face_index_base, face_verts, (const long long*)p2f, bary_coords, dists);
    MR_HIP(hipGetLastError());
    return R;
}

int sgr_splat_mesh_face_verts(int P, const float* points, const float* scaling, const float* quaternions, const float* primitive_verts,
                              float triangle_scale, const float* world_to_view, const float* projection, float* face_verts, void* stream)
{
    if (P <= 0) return 0;
    if (!points || !scaling || !quaternions || !primitive_verts || !world_to_view || !projection || !face_verts)
        return sgr_fail(SGR_E_INVALID, "splat_mesh_face_verts: null pointer");
    SplatArgs a = {points, scaling, quaternions, primitive_verts, world_to_view, projection, triangle_scale};
    hipLaunchKernelGGL(k_splat_face_verts, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, a, face_verts);
    return hipGetLastError() == hipSuccess ? 0 : sgr_fail(SGR_E_HIP, "splat_mesh_face_verts: launch failed");
}

}  // extern "C"
