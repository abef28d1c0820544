// knn.hip -- exact k-nearest-neighbour kernels for gfx950: the LDS-tiled exhaustive search (below: the parity anchor and
// `method="brute"`) and the uniform-grid search built on the same arithmetic (k_grid_*, k_knn_ball: further down).
//
//   sgr_dist2 : distCUDA2 of simple-knn (simple-knn/simple_knn.cu:147-221): mean of the three smallest squared
//               distances to the OTHER points.  The reference gets there with a Morton sort + box pruning; the
//               value it defines is exact, so any exact search reproduces it.
//   sgr_knn   : pytorch3d.ops.knn_points semantics as used by SuGaR (sugar_scene/sugar_model.py:49,235,1028,1342).
//
// One lane per query point; the reference set streams through LDS in 1024-point tiles (12 KB), every lane reads
// the same address (broadcast, conflict-free).  The running K-best list lives in registers, sorted ascending;
// a candidate is rejected with a single compare against the current worst.  Squared distances are evaluated as
// (dx*dx + dy*dy) + dz*dz with individually rounded ops (-ffp-contract=off), identical to the oracle.
#include "../../include/sugar_raster.h"
#include "sgr_common.h"
#include <cstdlib>
#include <string>

namespace {

#define KNN_TILE 1024

// `qlist` / `qcount` (grid search fallback): the queries are qlist[0 .. *qcount), taken only when there are more than
// `min_count` of them (fewer go to k_knn_far, one workgroup per query)
template <int K, bool EXCLUDE_SELF>
__global__ void __launch_bounds__(256) k_knn(int N, const float* __restrict__ query, int M, const float* __restrict__ ref,
                                             float* __restrict__ out_d, int64_t* __restrict__ out_i,
                                             float* __restrict__ out_mean, const int* __restrict__ qlist = nullptr,
                                             const unsigned int* __restrict__ qcount = nullptr, unsigned int min_count = 0)
{
    __shared__ float s_ref[KNN_TILE * 3];
    int q = blockIdx.x * 256 + threadIdx.x;
    bool live = q < N;
    if (qlist) {
        const unsigned int n = min(*qcount, (unsigned int)N);
        if (n <= min_count || (unsigned int)blockIdx.x * 256u >= n) return;  // (uniform per workgroup: before any barrier)
        live = (unsigned int)q < n;
        q = live ? qlist[q] : 0;
    }
    float qx = 0, qy = 0, qz = 0;
    if (live) { qx = query[3 * (size_t)q]; qy = query[3 * (size_t)q + 1]; qz = query[3 * (size_t)q + 2]; }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = 3.402823466e+38f; bi[k] = 0x7FFFFFFF; }
    for (int base = 0; base < M; base += KNN_TILE) {
        const int nt = min(KNN_TILE, M - base);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * 3; i += 256) s_ref[i] = ref[3 * (size_t)base + i];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < nt; j++) {
            const int gi = base + j;
            if (EXCLUDE_SELF && gi == q) continue;
            const float dx = s_ref[3 * j] - qx, dy = s_ref[3 * j + 1] - qy, dz = s_ref[3 * j + 2] - qz;
            float d = dx * dx + dy * dy + dz * dz;
            if (!(d < bd[K - 1])) continue;
            int id = gi;
#pragma unroll
            for (int k = 0; k < K; k++) {  // insertion ordered on (distance, index): the lower index wins ties, and a
                                           // displaced entry keeps its place relative to equal-distance successors
                if (d < bd[k] || (d == bd[k] && id < bi[k])) {
                    const float td = bd[k]; const int ti = bi[k];
                    bd[k] = d; bi[k] = id; d = td; id = ti;
                }
            }
        }
    }
    if (!live) return;
    if (out_mean) {
        out_mean[q] = (bd[0] + bd[1] + bd[2]) / 3.0f;  // simple_knn.cu:182
    } else {
#pragma unroll
        for (int k = 0; k < K; k++) { out_d[(size_t)q * K + k] = bd[k]; out_i[(size_t)q * K + k] = (bi[k] == 0x7FFFFFFF) ? -1 : (int64_t)bi[k]; }
    }
}

thread_local std::string g_knn_err;

template <int K>
void launch_knn(int N, const float* query, int M, const float* ref, float* d, int64_t* i, hipStream_t s)
{
    hipLaunchKernelGGL((k_knn<K, false>), dim3((N + 255) / 256), dim3(256), 0, s, N, query, M, ref, d, i, (float*)nullptr);
}

}  // namespace

extern "C" {

int sgr_dist2(int P, const float* points, float* meanDists, void* stream)
{
    if (P <= 0) return 0;
    if (!points || !meanDists) return SGR_E_INVALID;
    hipLaunchKernelGGL((k_knn<3, true>), dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, points, P, points,
                       (float*)nullptr, (int64_t*)nullptr, meanDists);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

int sgr_knn(int N, const float* query, int M, const float* ref, int K, float* dists, int64_t* idx, void* stream)
{
    if (N <= 0) return 0;
    if (!query || !ref || !dists || !idx || M <= 0) return SGR_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 1: launch_knn<1>(N, query, M, ref, dists, idx, s); break;
        case 2: launch_knn<2>(N, query, M, ref, dists, idx, s); break;
        case 3: launch_knn<3>(N, query, M, ref, dists, idx, s); break;
        case 4: launch_knn<4>(N, query, M, ref, dists, idx, s); break;
        case 8: launch_knn<8>(N, query, M, ref, dists, idx, s); break;
        case 16: launch_knn<16>(N, query, M, ref, dists, idx, s); break;
        case 32: launch_knn<32>(N, query, M, ref, dists, idx, s); break;
        default: return SGR_E_INVALID;  // supported K: 1,2,3,4,8,16,32
    }
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

}  // extern "C"

// =====================================================================================================================
// Exact k-NN on a uniform grid (the default for large sets).
//
// The reference's neighbour rebuild is pytorch3d's exhaustive knn_points (O(N*M); 0.37 s at 1M points even with the LDS-tiled
// kernel above) and simple-knn's Morton-box search.  Here the reference set is counting-sorted into a uniform grid sized for
// ~6 points per cell; a query walks Chebyshev rings of cells around its own cell and stops after ring r as soon as its K-th
// best squared distance is <= (r*h)^2 -- every unvisited point is at least r*h away -- so the result is EXACT.  Distances use
// the same individually rounded arithmetic as the exhaustive kernel and ties are resolved on (distance, index), so both
// kernels return identical values and indices.  Everything (bounding box, cell size) stays on the device: no host sync.
// =====================================================================================================================
namespace {

// order-preserving uint encodings of the bbox; far_count: queries handed to the exhaustive kernels, ball_count: to the ball scan
struct GridHdr { unsigned int minb[3], maxb[3], far_count, ball_count, occupied, pad;
};  // occupied: non-empty cells (k_grid_count)
// Rings a lane walks on its own before it hands its query to k_knn_ball (with K candidates in hand) -- measured, 124k queries
// against 1M points, 0 / 10 / 30 % of them outside the cloud: cap 1: 1.06 / 1.39 / 2.02 ms, 2: 1.12 / 1.85 / 2.46, 3: 1.13 / 2.57 /
// 3.15, 4: 1.13 / 3.49 / 4.15 (round 3, exhaustive fallback after 4 rings: 14.3 ms at 10 %).  Self queries (the neighbour rebuild,
// distCUDA2) do not depend on it: 2.3 / 0.8 ms at 1M.
#define GRID_MAX_RING 1
// Occupancy of the grid at a quarter of its resolution, for the ball cover of k_knn_ball: word (z / 4) * 32 + (y / 4), bit x / 4 = some cell
// of that 4 x 4 x 4 group holds a point (G <= 128: at most 32 x 32 words of 32 bits).  A far query's ball is mostly empty space: its
// cell rows are tested against these 4 KB (in LDS) before the two dependent look-ups of their point range.
#define GRID_COARSE_WORDS 1024
#define BALL_SAMPLES 2048u     // reference points a query without a bound is compared with in k_knn_ball (32 per lane; 32 KB of LDS)
#define FAR_SINGLE_MAX 16384u  // up to this many far queries: one workgroup each; more: the tiled exhaustive kernel

__device__ __forceinline__ unsigned int f2ord(float f)
{
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void k_grid_init(GridHdr* h)
{
    if (threadIdx.x < 3) { h->minb[threadIdx.x] = 0xFFFFFFFFu; h->maxb[threadIdx.x] = 0u; }
    if (threadIdx.x == 3) { h->far_count = 0u; h->ball_count = 0u; h->occupied = 0u; h->pad = 0u; }
}

// Far queries, Q per workgroup (grid-stride over the fallback list): the 256 threads split the reference set -- every point is
// loaded once for the Q queries --, keep their own K best per query in registers, and a query's K best are then drawn from the
// 256 sorted lists in K rounds of a block-wide arg-min on (distance, index).  Same arithmetic and order as k_knn: same result.
template <int K, bool EXCLUDE_SELF, int Q>
__global__ void __launch_bounds__(256) k_knn_far(const float* __restrict__ query, int M, const float4* __restrict__ sorted,
                                                 float* __restrict__ out_d, int64_t* __restrict__ out_i, float* __restrict__ out_mean,
                                                 const int* __restrict__ qlist, const unsigned int* __restrict__ qcount, unsigned int cap)
{
    __shared__ float s_d[4];
    __shared__ int s_i[4], s_t[4];
    const unsigned int n = min(*qcount, cap);
    if (n > FAR_SINGLE_MAX) return;  // (k_knn takes them)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (unsigned int w = blockIdx.x * Q; w < n; w += gridDim.x * Q) {
        int qid[Q];
        float qx[Q], qy[Q], qz[Q];
        float bd[Q][K];
        int bi[Q][K];
#pragma unroll
        for (int u = 0; u < Q; u++) {
            qid[u] = qlist[min(w + u, n - 1)];  // (a short last group repeats its last query)
            qx[u] = query[3 * (size_t)qid[u]]; qy[u] = query[3 * (size_t)qid[u] + 1]; qz[u] = query[3 * (size_t)qid[u] + 2];
#pragma unroll
            for (int k = 0; k < K; k++) { bd[u][k] = 3.402823466e+38f; bi[u][k] = 0x7FFFFFFF; }
        }
        for (int j = tid; j < M; j += 256) {  // (the cell-sorted copy: one 16-byte load per point; any visiting order gives the same lists)
            const float4 p = sorted[j];
            const int pid = __float_as_int(p.w);
#pragma unroll
            for (int u = 0; u < Q; u++) {
                if (EXCLUDE_SELF && pid == qid[u]) continue;
                const float dx = p.x - qx[u], dy = p.y - qy[u], dz = p.z - qz[u];
                float d = dx * dx + dy * dy + dz * dz;
                int id = pid;
                if (!(d < bd[u][K - 1] || (d == bd[u][K - 1] && id < bi[u][K - 1]))) continue;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (d < bd[u][k] || (d == bd[u][k] && id < bi[u][k])) {
                        const float td = bd[u][k]; const int ti = bi[u][k];
                        bd[u][k] = d; bi[u][k] = id; d = td; id = ti;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < Q; u++) {
            if (w + u >= n) break;  // (uniform)
            const int q = qid[u];
            float sum3 = 0.f;
            for (int k = 0; k < K; k++) {
                // block-wide arg-min of the list heads on (distance, index)
                float d = bd[u][0]; int id = bi[u][0]; int who = tid;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float od = __shfl_xor(d, o); const int oi = __shfl_xor(id, o); const int ow = __shfl_xor(who, o);
                    if (od < d || (od == d && oi < id)) { d = od; id = oi; who = ow; }
                }
                __syncthreads();
                if (lane == 0) { s_d[wave] = d; s_i[wave] = id; s_t[wave] = who; }
                __syncthreads();
                d = s_d[0]; id = s_i[0]; who = s_t[0];
#pragma unroll
                for (int v = 1; v < 4; v++)
                    if (s_d[v] < d || (s_d[v] == d && s_i[v] < id)) { d = s_d[v]; id = s_i[v]; who = s_t[v]; }
                if (tid == who) {  // pop the winner's head
#pragma unroll
                    for (int j = 0; j + 1 < K; j++) { bd[u][j] = bd[u][j + 1]; bi[u][j] = bi[u][j + 1]; }
                    bd[u][K - 1] = 3.402823466e+38f; bi[u][K - 1] = 0x7FFFFFFF;
                }
                if (tid == 0) {
                    if (out_mean) { if (k < 3) sum3 += d; }
                    else { out_d[(size_t)q * K + k] = d; out_i[(size_t)q * K + k] = (id == 0x7FFFFFFF) ? -1 : (int64_t)id; }
                }
            }
            if (tid == 0 && out_mean) out_mean[q] = sum3 / 3.0f;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_grid_bbox(int M, const float* __restrict__ pts, GridHdr* h)
{
    __shared__ unsigned int s_min[3], s_max[3];
    if (threadIdx.x < 3) { s_min[threadIdx.x] = 0xFFFFFFFFu; s_max[threadIdx.x] = 0u; }
    __syncthreads();
    unsigned int mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    // four points per thread and round, all twelve loads requested before the first is used (a plain loop waits for each), clamped
    // indices (a repeated point does not change a minimum); one LDS atomic per wave and bound instead of one per thread: 38 -> ~10 us at 1M
    const int stride = gridDim.x * 256;
    for (int base = blockIdx.x * 256 + threadIdx.x; base < M; base += 4 * stride) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t i = (size_t)min(base + u * stride, M - 1);
#pragma unroll
            for (int c = 0; c < 3; c++) v[u][c] = pts[3 * i + c];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < 3; c++) { const unsigned int o = f2ord(v[u][c]); mn[c] = min(mn[c], o); mx[c] = max(mx[c], o); }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = min(mn[c], (unsigned int)__shfl_xor((int)mn[c], o));
            mx[c] = max(mx[c], (unsigned int)__shfl_xor((int)mx[c], o));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_min[c], mn[c]); atomicMax(&s_max[c], mx[c]); }
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&h->minb[threadIdx.x], s_min[threadIdx.x]); atomicMax(&h->maxb[threadIdx.x], s_max[threadIdx.x]); }
}

struct GridGeom { float ox, oy, oz, h, inv_h; int G; };

__device__ __forceinline__ GridGeom grid_geom(const GridHdr* h, int G)
{
    GridGeom g;
    g.ox = ord2f(h->minb[0]); g.oy = ord2f(h->minb[1]); g.oz = ord2f(h->minb[2]);
    const float ex = ord2f(h->maxb[0]) - g.ox, ey = ord2f(h->maxb[1]) - g.oy, ez = ord2f(h->maxb[2]) - g.oz;
    const float ext = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-30f));
    g.h = ext / (float)G * 1.0001f;  // cubic cells; the slack keeps the max point inside the last cell
    g.inv_h = 1.0f / g.h;
    g.G = G;
    return g;
}
__device__ __forceinline__ int cell_coord(float v, float o, const GridGeom& g)
{
    const int c = (int)((v - o) * g.inv_h);
    return min(max(c, 0), g.G - 1);
}

// lane-run detection inside a wave: `head` = this lane's cell differs from the previous lane's (or it is lane 0); len = lanes from this
// head up to the next head (or the end of the live lanes).  Dead lanes carry the key 0xFFFFFFFF and are never heads.
__device__ __forceinline__ bool grid_run_head(unsigned int c, bool live, unsigned int& len)
{
    const int lane = threadIdx.x & 63;
    const unsigned int prev = (unsigned int)__shfl_up((int)c, 1);
    const bool head = live && (lane == 0 || prev != c);
    const unsigned long long hm = __ballot(head), lm = __ballot(live);
    // lanes after this one that start a run, or are dead: the nearest ends this run
    const unsigned long long stop = (hm | ~lm) & ~((2ull << lane) - 1ull);
    const int end = stop ? (int)__builtin_ctzll(stop) : 64;
    len = (unsigned int)(end - lane);
    return head;
}

__global__ void __launch_bounds__(256) k_grid_count(int M, const float* __restrict__ pts, GridHdr* __restrict__ hdr, int G,
                                                    unsigned int* __restrict__ cell_count, unsigned int* __restrict__ cell_of)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < M;
    const GridGeom g = grid_geom(hdr, G);
    unsigned int c = 0xFFFFFFFFu;
    if (live) {
        const int cx = cell_coord(pts[3 * (size_t)i], g.ox, g), cy = cell_coord(pts[3 * (size_t)i + 1], g.oy, g),
                  cz = cell_coord(pts[3 * (size_t)i + 2], g.oz, g);
        c = ((unsigned int)cz * G + cy) * G + cx;
        cell_of[i] = c;
    }
    // Runs of neighbouring lanes in the same cell (points of a mesh or of a scan come in spatial order) take ONE atomic: the run's
    // first lane adds the run's length.  The occupied-cell count is NOT kept here any more: one atomic per first touch of a cell (then
    // per wave) on ONE address serialised the whole kernel -- 0.13 ms at 1M points; k_grid_blocksum counts the non-empty cells it
    // reads anyway.
    unsigned int len;
    if (grid_run_head(c, live, len)) atomicAdd(&cell_count[c], len);
}

// exclusive scan of the cell counts (G^3 <= 2M cells), two launches: sums of 4096-cell blocks, then every block adds up the sums
// before it (at most 512) and scans its own cells -- 16 consecutive cells per thread, 64 contiguous bytes per lane.  (Until round 4
// one 1024-thread workgroup walked the whole array with a stride of cells/1024 between neighbouring threads: 0.27 ms at 176k cells.)
#define GRID_SCAN_BLOCK 4096
__device__ __forceinline__ unsigned int grid_scan_load16(int n, const unsigned int* __restrict__ cnt, int base, unsigned int v[16])
{
    unsigned int sum = 0;
    if (base + 16 <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(cnt + base);
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = (base + i < n) ? cnt[base + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) sum += v[i];
    return sum;
}

// sum over the 256 threads of a workgroup, returned to every thread; `mine_excl` = sum over the threads before this one
__device__ __forceinline__ unsigned int grid_block_scan(unsigned int x, unsigned int* s_wave, unsigned int& mine_excl)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int y = (unsigned int)__shfl_up((int)incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const unsigned int t = s_wave[w];
        if (w < wave) before += t;
        total += t;
    }
    mine_excl = before + incl - x;
    __syncthreads();
    return total;
}

__global__ void __launch_bounds__(256) k_grid_blocksum(int n, const unsigned int* __restrict__ cnt, unsigned int* __restrict__ part,
                                                       GridHdr* __restrict__ hdr)
{
    __shared__ unsigned int s_wave[4];
    unsigned int v[16], excl;
    const unsigned int sum = grid_scan_load16(n, cnt, (int)blockIdx.x * GRID_SCAN_BLOCK + (int)threadIdx.x * 16, v);
    // non-empty cells (the grid's occupancy, read by the host's asynchronous probe): one atomic per wave of this small grid
    unsigned int occ = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) occ += v[j] != 0u ? 1u : 0u;
    __shared__ unsigned int s_occ[4];
    for (int o = 32; o > 0; o >>= 1) occ += (unsigned int)__shfl_xor((int)occ, o);
    if ((threadIdx.x & 63) == 0) s_occ[threadIdx.x >> 6] = occ;
    const unsigned int total = grid_block_scan(sum, s_wave, excl);   // (has the workgroup barriers that publish s_occ)
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = total;
        const unsigned int o4 = s_occ[0] + s_occ[1] + s_occ[2] + s_occ[3];
        if (o4) atomicAdd(&hdr->occupied, o4);   // one atomic per workgroup: the word is one address
    }
}

__global__ void __launch_bounds__(256) k_grid_scan(int n, const unsigned int* __restrict__ cnt, const unsigned int* __restrict__ part,
                                                   unsigned int* __restrict__ start)
{
    __shared__ unsigned int s_wave[4];
    unsigned int before = 0, excl;
    for (int j = (int)threadIdx.x; j < (int)blockIdx.x; j += 256) before += part[j];
    const unsigned int offset = grid_block_scan(before, s_wave, excl);
    unsigned int v[16];
    const int base = (int)blockIdx.x * GRID_SCAN_BLOCK + (int)threadIdx.x * 16;
    const unsigned int sum = grid_scan_load16(n, cnt, base, v);
    const unsigned int total = grid_block_scan(sum, s_wave, excl);
    unsigned int run = offset + excl;
    if (base + 16 <= n) {
        unsigned int o[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { o[i] = run; run += v[i]; }
        uint4* q = reinterpret_cast<uint4*>(start + base);
        q[0] = make_uint4(o[0], o[1], o[2], o[3]); q[1] = make_uint4(o[4], o[5], o[6], o[7]);
        q[2] = make_uint4(o[8], o[9], o[10], o[11]); q[3] = make_uint4(o[12], o[13], o[14], o[15]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) { if (base + i < n) start[base + i] = run; run += v[i]; }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) start[n] = offset + total;
}

__global__ void __launch_bounds__(256) k_grid_scatter(int M, const float* __restrict__ pts, const unsigned int* __restrict__ cell_of,
                                                      const unsigned int* __restrict__ cell_start, unsigned int* __restrict__ cursor,
                                                      float4* __restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < M;
    const unsigned int c = live ? cell_of[i] : 0xFFFFFFFFu;
    // (as in k_grid_count: one returning atomic per run of lanes in the same cell; the run's lanes take consecutive slots)
    unsigned int len;
    const bool head = grid_run_head(c, live, len);
    unsigned int base = 0u;
    if (head) base = cell_start[c] + atomicAdd(&cursor[c], len);
    const int lane = threadIdx.x & 63;
    const unsigned long long hm = __ballot(head);
    if (!live) return;
    const int my_head = 63 - (int)__builtin_clzll(hm & ((2ull << lane) - 1ull));   // the nearest head at or before this lane
    const unsigned int pos = (unsigned int)__shfl((int)base, my_head) + (unsigned int)(lane - my_head);
    sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}

template <int K, bool EXCLUDE_SELF, bool SELF_QUERY>
__global__ void __launch_bounds__(128) k_grid_query(int N, const float* __restrict__ query, const GridHdr* __restrict__ hdr, int G,
                                                    const unsigned int* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                    float* __restrict__ out_d, int64_t* __restrict__ out_i,
                                                    float* __restrict__ out_mean, unsigned int* __restrict__ far_count,
                                                    int* __restrict__ far_list, unsigned int far_cap, int max_ring,
                                                    unsigned int* __restrict__ ball_count, int* __restrict__ ball_list,
                                                    float* __restrict__ ball_u2)
{
    const int t = blockIdx.x * 128 + threadIdx.x;
    if (t >= N) return;
    // SELF_QUERY: lane t takes the t-th point of the cell-sorted reference set, so neighbouring lanes search the same cells
    int q = t;
    float qx, qy, qz;
    if (SELF_QUERY) {
        const float4 p = sorted[t];
        qx = p.x; qy = p.y; qz = p.z; q = __float_as_int(p.w);
    } else {
        qx = query[3 * (size_t)t]; qy = query[3 * (size_t)t + 1]; qz = query[3 * (size_t)t + 2];
    }
    const GridGeom g = grid_geom(hdr, G);
    const int cx = cell_coord(qx, g.ox, g), cy = cell_coord(qy, g.oy, g), cz = cell_coord(qz, g.oz, g);
    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = 3.402823466e+38f; bi[k] = 0x7FFFFFFF; }
    auto consider = [&](const float4 p) {
        int id = __float_as_int(p.w);
        if (EXCLUDE_SELF && id == q) return;
        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
        float d = dx * dx + dy * dy + dz * dz;
        if (!(d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1]))) return;
#pragma unroll
        for (int k = 0; k < K; k++) {  // ordered on (distance, index): independent of the visiting order
            if (d < bd[k] || (d == bd[k] && id < bi[k])) {
                const float td = bd[k]; const int ti = bi[k];
                bd[k] = d; bi[k] = id; d = td; id = ti;
            }
        }
    };
    {   // Rings 0 and 1 together, as the nine cell rows (z, y) of the 3 x 3 x 3 box: a row's cells are consecutive in the cell-sorted
        // array, so a row is ONE range -- 18 independent look-ups issued together instead of 27 cells x two dependent ones (round 5:
        // this walk was 1.0 ms of the level-set sampler's 124k queries, most of it waiting on look-ups of empty cells).  The K
        // best are ordered on (distance, index): visiting ring 1 before knowing whether ring 0 sufficed changes nothing.
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G - 1);
        unsigned int rb[9], re[9];
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const int z = cz + j / 3 - 1, y = cy + j % 3 - 1;
            const bool in = z >= 0 && z < G && y >= 0 && y < G;
            const unsigned int c0 = in ? ((unsigned int)z * G + y) * G + x0 : 0u;
            rb[j] = cell_start[c0];
            re[j] = in ? cell_start[c0 + (unsigned int)(x1 - x0) + 1u] : rb[j];
        }
#pragma unroll
        for (int j = 0; j < 9; j++)
            for (unsigned int s = rb[j]; s < re[j]; s++) consider(sorted[s]);
    }
    for (int r = 1; r < G; r++) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, G - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, G - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, G - 1);
        if (r >= 2)
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                const bool shell_row = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                // inside the shell only the two end cells of the row are new; on a shell face the whole row is
                for (int x = x0; x <= x1; x += (shell_row ? 1 : max(1, x1 - x0))) {
                    if (!shell_row && x != cx - r && x != cx + r) continue;
                    const unsigned int c = ((unsigned int)z * G + y) * G + x;
                    const unsigned int b = cell_start[c], e = cell_start[c + 1];
                    for (unsigned int s = b; s < e; s++) consider(sorted[s]);
                }
            }
        if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == G - 1 && y1 == G - 1 && z1 == G - 1) break;  // whole grid searched
        // Lower bound of the distance to every point NOT yet visited.  Such a point lies beyond one of the faces of the visited box
        // that are inside the grid (beyond a face on the grid's boundary there are no points), so along that axis it is at least the
        // query's distance to the face away, and along the two other axes at least the query's distance to the GRID (zero for a
        // query inside it): bound^2 = min over inner faces of face^2 + sum of the other axes' grid distances^2.  For a query inside
        // the cloud this is the old r * h (plus its offset in its own cell); for one OUTSIDE -- a pixel of the level-set sampler
        // unprojected in front of the cloud -- the second term carries the distance to the cloud, and a handful of rings suffice
        // where r * h alone needed (distance / h) rings: 10 % of the sampler's 124k queries used to fall through to the exhaustive
        // kernels for 13 of its 14.3 ms.  Planes and cell assignment are float computations: the margin keeps the bound valid.
        const float qa[3] = {qx, qy, qz}, oa[3] = {g.ox, g.oy, g.oz};
        const int ca[3] = {cx, cy, cz};
        float gd2[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float dlo = oa[a] - qa[a], dhi = qa[a] - (oa[a] + (float)G * g.h);
            const float dd = fmaxf(fmaxf(dlo, dhi), 0.0f) * (1.0f - 1e-5f);
            gd2[a] = dd * dd;
        }
        float reach2 = 3.402823466e+38f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float others = gd2[(a + 1) % 3] + gd2[(a + 2) % 3];
            const float margin = 4e-6f * (fabsf(qa[a]) + fabsf(oa[a]) + (float)G * g.h);
            if (ca[a] - r > 0) {
                const float fd = fmaxf(qa[a] - (oa[a] + (float)(ca[a] - r) * g.h) - margin, 0.0f);
                reach2 = fminf(reach2, fd * fd + others);
            }
            if (ca[a] + r < G - 1) {
                const float fd = fmaxf((oa[a] + (float)(ca[a] + r + 1) * g.h) - qa[a] - margin, 0.0f);
                reach2 = fminf(reach2, fd * fd + others);
            }
        }
        if (bd[K - 1] < reach2) break;
        // A query far from the data (an outlier; a pixel of the level-set sampler unprojected in front of the cloud) keeps ONE lane
        // walking hundreds of cells, two dependent loads each, while the other lanes of its wave idle: 10 % such queries cost 13
        // of the sampler's 14.3 ms (and 6.6 ms still with the tighter bound above).  After `max_ring` rings the lane stops: once
        // it holds K candidates its K-th best distance U bounds the answer -- the K nearest lie in the ball (q, U) -- and the query
        // goes to k_knn_ball, where a whole wave scans exactly the cells that ball touches.  Same distances, same (distance, index)
        // order everywhere: same result.  (Only when the hand-over list is full does a lane walk on; the exhaustive kernels behind
        // this one take what the grid scratch's far list holds -- nothing since round 5.)
        if (r >= max_ring && far_list) {
            // (round 5) a lane WITHOUT K candidates hands its query over too, with "no bound yet": k_knn_ball then takes the bound
            // from a spread sample of the reference set.  It used to walk on, alone, for up to 12 rings (~15 000 cells, two dependent
            // loads each) and then went to the exhaustive kernels: the level-set sampler's pixels over a SURFACE-bound cloud
            // (BASELINE config 4: flat Gaussians on a mesh, most grid cells empty, queries 3 - 30 cells off the surface) cost 55 ms
            // per 124k queries that way.
            const unsigned int pos = atomicAdd(ball_count, 1u);
            if (pos < far_cap) { ball_list[pos] = q; ball_u2[pos] = bd[K - 1]; return; }
        }
    }
    if (out_mean) {
        out_mean[q] = (bd[0] + bd[1] + bd[2]) / 3.0f;
    } else {
#pragma unroll
        for (int k = 0; k < K; k++) {
            out_d[(size_t)q * K + k] = bd[k];
            out_i[(size_t)q * K + k] = (bi[k] == 0x7FFFFFFF) ? -1 : (int64_t)bi[k];
        }
    }
}

// The points of up to 64 cell rows -- lane r holds row r as the contiguous range [b, e) of the cell-sorted array -- visited by the
// WHOLE wave: the ranges are concatenated (exclusive scan of their lengths) and lane l takes elements l, l + 64, ... of the
// concatenation (a 6-step search over the scanned offsets through ds_bpermute finds the row an element belongs to).  Every lane
// works whatever the rows' lengths are; a step is VISIT_WIDTH batches of 64 independent loads, all in flight before any is used.
// (round 5: before, a lane walked its own row serially -- one dependent load per point, and a surface cloud puts the points of a
// ball into ~15 of the 64 lanes: 355k cycles per query by the cycle counter of a development build (profiles/r05_knn_ball_counters.txt), almost all
// of it waiting.)  `f(point, valid)` is called by ALL lanes (it may ballot); call visit_rows wave-uniformly.
#ifndef VISIT_WIDTH
#define VISIT_WIDTH 2
#endif
template <class F>
__device__ __forceinline__ void visit_rows(const float4* __restrict__ sorted, unsigned int b, unsigned int e, int lane, F&& f)
{
    const unsigned int cnt = e > b ? e - b : 0u;
    unsigned int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    const unsigned int total = __shfl(inc, 63), excl = inc - cnt;
    for (unsigned int base = 0; base < total; base += 64u * VISIT_WIDTH) {
        unsigned int sl[VISIT_WIDTH], at[VISIT_WIDTH];
        int r[VISIT_WIDTH];   // the last row whose offset is <= sl (empty rows share their successor's offset and are stepped over)
#pragma unroll
        for (int u = 0; u < VISIT_WIDTH; u++) { sl[u] = base + 64u * u + (unsigned int)lane; r[u] = 0; }
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
#pragma unroll
            for (int u = 0; u < VISIT_WIDTH; u++) { const unsigned int o = __shfl(excl, r[u] + step); if (o <= sl[u]) r[u] += step; }
        }
#pragma unroll
        for (int u = 0; u < VISIT_WIDTH; u++) at[u] = __shfl(b, r[u]) + (sl[u] - __shfl(excl, r[u]));
        const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 p[VISIT_WIDTH];
#pragma unroll
        for (int u = 0; u < VISIT_WIDTH; u++) p[u] = sl[u] < total ? sorted[at[u]] : none;
#pragma unroll
        for (int u = 0; u < VISIT_WIDTH; u++)
            if (base + 64u * u < total) f(p[u], sl[u] < total);   // (uniform)
    }
}

// the k-th smallest (k = 0: the smallest) of the 64 lanes' values, ties by lane: every lane counts the lanes ordered before it --
// 64 independent v_readlane + compare pairs instead of k + 1 dependent six-step butterflies
__device__ __forceinline__ float wave_kth_smallest(float v, int lane, int k)
{
    int rank = 0;
#pragma unroll 8
    for (int i = 0; i < 64; i++) {   // (eight at a time: all 64 at once spill scalar registers)
        const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
        rank += (o < v || (o == v && i < lane)) ? 1 : 0;
    }
    const unsigned long long m = __ballot(rank == k);
    return __shfl(v, (int)__builtin_ctzll(m | (1ull << 63)));
}

// The kernel is bound by dependent look-ups (cell range -> points, and the cross-lane steps between them): what it needs is waves
// to switch to.  52 KB of LDS per workgroup let three of them share a CU; six waves per SIMD need <= 80 VGPRs (no spills at that,
// scripts/kernel_meta.py).  Same box, 124k queries over config 4's cloud: 1.80 ms at two workgroups per CU / 104 VGPRs, 1.60 at three.
#ifndef BALL_OCCUPANCY
#define BALL_OCCUPANCY __attribute__((amdgpu_waves_per_eu(6, 6)))
#endif
#ifndef BALL_GRID_MAX
#define BALL_GRID_MAX 768u   // three workgroups per CU
#endif
#define BALL_WAVES 8           // waves of a k_knn_ball workgroup: they share the staged samples and the occupancy mask
#define BALL_ROW_LIST 128      // per-wave list of cell rows waiting to be scanned (a flush takes 64)
#define BALL_CAND 128          // per-wave list of candidates (distance, index) within the bound; reduced to the K best when it fills
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// sorted index of sample j of BALL_SAMPLES spread evenly over the M points of the cell-sorted array (every point once for a small set)
__device__ __forceinline__ unsigned int ball_sample_at(unsigned int j, unsigned int M)
{
    return M <= BALL_SAMPLES ? j : (unsigned int)(((float)j + 0.5f) * ((float)M / (float)BALL_SAMPLES));
}
// The samples as one dense array (32 KB): k_knn_ball stages it in LDS once per workgroup.  (round 5: read in place, the strided
// 16-byte loads of a query each pulled their own cache line through L2 -- 0.5 MB per query, more than everything else it read.)
__global__ void __launch_bounds__(256) k_grid_samples(int G, const unsigned int* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                      float4* __restrict__ samples)
{
    const unsigned int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= BALL_SAMPLES) return;
    const unsigned int M = cell_start[(unsigned int)G * G * G], at = ball_sample_at(j, M);
    samples[j] = at < M ? sorted[at] : make_float4(3.0e+38f, 3.0e+38f, 3.0e+38f, __int_as_float(0x7FFFFFFF));  // (distance +inf to anything)
}

// Orders the C (<= BALL_CAND) candidates of a wave's list on (distance, index) and keeps the first min(C, K) at the head of the
// list: every candidate counts the candidates ordered before it (indices are distinct, so the counts are a permutation) and moves
// to that place.  Broadcast LDS reads and compares only -- no dependent cross-lane steps.  Returns the new count.
template <int K>
__device__ __forceinline__ unsigned int ball_select(uint2* __restrict__ cand, unsigned int C, int lane)
{
    const uint2 none = make_uint2(__float_as_uint(3.402823466e+38f), 0x7FFFFFFFu);
    const bool h0 = (unsigned int)lane < C, h1 = (unsigned int)lane + 64u < C;
    const uint2 e0 = h0 ? cand[lane] : none, e1 = h1 ? cand[lane + 64] : none;
    const float d0 = __uint_as_float(e0.x), d1 = __uint_as_float(e1.x);
    const int i0 = (int)e0.y, i1 = (int)e1.y;
    int r0 = 0, r1 = 0;
    for (unsigned int j = 0; j < C; j++) {
        const uint2 o = cand[j];
        const float od = __uint_as_float(o.x);
        const int oi = (int)o.y;
        r0 += (od < d0 || (od == d0 && oi < i0)) ? 1 : 0;
        r1 += (od < d1 || (od == d1 && oi < i1)) ? 1 : 0;
    }
    wave_lds_fence();
    if (h0 && r0 < K) cand[r0] = e0;
    if (h1 && r1 < K) cand[r1] = e1;
    wave_lds_fence();
    return C < (unsigned int)K ? C : (unsigned int)K;
}

// Queries the ring walk handed over, with an upper bound U^2 on their K-th nearest distance or without one: ONE WAVE per query.
// The K nearest lie in the ball (q, U); cell rows (fixed z, y) farther than U in the (y, z) plane are skipped, the others contribute
// the x-interval sqrt(U^2 - dy^2 - dz^2) around the query, and the cells of a row are consecutive in the cell-sorted array: a row
// is one contiguous range of points.  The wave visits the rows' points together (visit_rows); a point within the bound is appended
// to a candidate list in LDS, which is cut to its K best on (distance, index) -- tightening the bound -- when it fills and at the
// end: the answer, in order.  The cover is taken with a margin (float rounding of planes and cell assignment); a larger cover
// only costs time.  (round 5, third form of this kernel: per-lane sorted lists in registers and a K-round cross-lane merge before;
// 2 x K sorted (distance, index) registers per lane, an insertion network per visited batch and 18 K dependent ds_bpermute at the end.)
template <int K, bool EXCLUDE_SELF>
__global__ void __launch_bounds__(64 * BALL_WAVES) BALL_OCCUPANCY k_knn_ball(const float* __restrict__ query, const GridHdr* __restrict__ hdr, int G,
                                                  const unsigned int* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                  float* __restrict__ out_d, int64_t* __restrict__ out_i, float* __restrict__ out_mean,
                                                  const int* __restrict__ qlist, const float* __restrict__ qu2,
                                                  const unsigned int* __restrict__ qcount, unsigned int cap,
                                                  const unsigned int* __restrict__ coarse, const float4* __restrict__ samples)
{
    static_assert(K <= 64 && K <= BALL_CAND - 64, "the candidate list takes a batch of 64 on top of the K kept");
    __shared__ unsigned int s_occ[GRID_COARSE_WORDS];
    __shared__ float4 s_samp[BALL_SAMPLES];
    __shared__ uint2 s_rows[BALL_WAVES][BALL_ROW_LIST];
    __shared__ uint2 s_cand[BALL_WAVES][BALL_CAND];
    const unsigned int n = min(*qcount, cap);
    if (blockIdx.x * (unsigned int)BALL_WAVES >= n) return;   // (no query for this workgroup: uniform over the block)
    for (int i = threadIdx.x; i < GRID_COARSE_WORDS; i += 64 * BALL_WAVES) s_occ[i] = coarse[i];
    for (int i = threadIdx.x; i < (int)BALL_SAMPLES; i += 64 * BALL_WAVES) s_samp[i] = samples[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint2* const cand = s_cand[wv];
    const GridGeom g = grid_geom(hdr, G);
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (unsigned int w = blockIdx.x * BALL_WAVES + wv; w < n; w += gridDim.x * BALL_WAVES) {
        const int q = qlist[w];
        const float qx = query[3 * (size_t)q], qy = query[3 * (size_t)q + 1], qz = query[3 * (size_t)q + 2];
        const float ext = (float)G * g.h;
        float u2 = qu2[w];
        if (!(u2 < 3.0e+38f)) {
            // No bound from the ring walk.  (1) BALL_SAMPLES points spread evenly over the cell-sorted array (i.e. over the cloud):
            // every lane keeps the nearest of its share; the K-th smallest of the 64 lane minima is the distance of K DISTINCT
            // reference points, hence an upper bound of the K-th nearest distance -- but a loose one (the samples are ~0.08 of the
            // cloud's extent apart: the ball it gives cuts thousands of points out of a surface).  (2) A descent from the nearest
            // sample: look at the 27 cells around the current best point, move to the nearest point found there, repeat until it
            // stays in its cell; the K-th smallest of the lanes' nearest points in the LAST neighbourhood (distinct points again) is
            // a bound of the size of the answer itself when the descent ends near the global minimum (a far query's distance field
            // over a bumpy surface has local minima: the bound of one is looser, the answer is the same).  The smaller of the two
            // is used.  With fewer than K points in reach the bound stays infinite and the cover below is the whole grid: exact.
            float best = 3.402823466e+38f, bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll 8
            for (unsigned int j = 0; j < BALL_SAMPLES / 64u; j++) {
                const float4 p = s_samp[j * 64u + (unsigned int)lane];
                const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                const float d = (EXCLUDE_SELF && __float_as_int(p.w) == q) ? 3.402823466e+38f : dx * dx + dy * dy + dz * dz;
                if (d < best) { best = d; bx = p.x; by = p.y; bz = p.z; }
            }
            float nearest = best;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nearest = fminf(nearest, __shfl_xor(nearest, o));
            u2 = wave_kth_smallest(best, lane, K - 1);
            if (nearest < 3.0e+38f) {
                const int first = (int)__builtin_ctzll(__ballot(best == nearest) | (1ull << 63));
                float cpx = __shfl(bx, first), cpy = __shfl(by, first), cpz = __shfl(bz, first), cur_d = nearest;
                float lbest = 3.402823466e+38f;
                for (int it = 0; it < 16; it++) {
                    const int ccx = cell_coord(cpx, g.ox, g), ccy = cell_coord(cpy, g.oy, g), ccz = cell_coord(cpz, g.oz, g);
                    // lanes 0..8: the nine rows (z, y) of the neighbourhood; a row's three cells are one contiguous range of points
                    unsigned int rb = 0u, re = 0u;
                    if (lane < 9) {
                        const int z = ccz + lane / 3 - 1, y = ccy + lane % 3 - 1;
                        if (z >= 0 && z < G && y >= 0 && y < G) {
                            const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, G - 1);
                            const unsigned int c0 = ((unsigned int)z * G + y) * G + x0;
                            rb = cell_start[c0]; re = cell_start[c0 + (unsigned int)(x1 - x0) + 1u];
                        }
                    }
                    lbest = 3.402823466e+38f;
                    float lx = 0.f, ly = 0.f, lz = 0.f;
                    visit_rows(sorted, rb, re, lane, [&](const float4 p, bool valid) {
                        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                        const float d = dx * dx + dy * dy + dz * dz;
                        if (valid && !(EXCLUDE_SELF && __float_as_int(p.w) == q) && d < lbest) { lbest = d; lx = p.x; ly = p.y; lz = p.z; }
                    });
                    float nb = lbest;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) nb = fminf(nb, __shfl_xor(nb, o));
                    if (!(nb < cur_d)) break;   // the neighbourhood holds nothing nearer than the point it was built around
                    const int win = (int)__builtin_ctzll(__ballot(lbest == nb) | (1ull << 63));
                    cpx = __shfl(lx, win); cpy = __shfl(ly, win); cpz = __shfl(lz, win);
                    cur_d = nb;
                }
                // (lbest: every lane's nearest point of the last neighbourhood visited -- no second pass over it)
                u2 = fminf(u2, wave_kth_smallest(lbest, lane, K - 1));
            }
        }
        const bool unbounded = !(u2 < 3.0e+38f);
        const float U = unbounded ? 4.0f * (ext + fabsf(qx - g.ox) + fabsf(qy - g.oy) + fabsf(qz - g.oz))
                                  : sqrtf(u2) * (1.0f + 1e-5f) + 1e-5f * (ext + fabsf(qx) + fabsf(qy) + fabsf(qz));
        float U2 = U * U;    // (of the cover, fixed) ...
        float keep2 = U2;    // ... and of the candidates kept: shrinks to the K-th best so far whenever the list is cut
        // cell rows the ball can touch.  The points live inside the grid, so along y (z) the ball only reaches as far as what the
        // query's distance to the grid along the two other axes leaves of U: for a query outside the cloud that is a small cap
        const float gx = fmaxf(fmaxf(g.ox - qx, qx - (g.ox + ext)), 0.0f) * (1.0f - 1e-5f);
        const float gy = fmaxf(fmaxf(g.oy - qy, qy - (g.oy + ext)), 0.0f) * (1.0f - 1e-5f);
        const float gz = fmaxf(fmaxf(g.oz - qz, qz - (g.oz + ext)), 0.0f) * (1.0f - 1e-5f);
        const float Uz = sqrtf(fmaxf(U2 - gx * gx - gy * gy, 0.0f)), Uy = sqrtf(fmaxf(U2 - gx * gx - gz * gz, 0.0f));
        const int z0 = cell_coord(qz - Uz, g.oz, g), z1 = cell_coord(qz + Uz, g.oz, g);
        const int y0 = cell_coord(qy - Uy, g.oy, g), y1 = cell_coord(qy + Uy, g.oy, g);
        const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
        // The rows are enumerated 64 at a time without integer division (a slice z takes 8 / 16 / 32 / 64 lanes by how many rows y
        // it has, several slices per step).  Enumeration and scanning are decoupled: a row that the ball reaches and whose
        // 4 x 4 x 4 groups hold anything -- LDS and arithmetic only -- is appended to a per-wave list, and the list is scanned 64
        // rows at a time.  A far query's ball has thousands of rows of which a tenth pass: before, each step of 64 rows paid the
        // latency of the cell_start look-up and of the points for the few that passed.
        const int sh = ny <= 8 ? 3 : (ny <= 16 ? 4 : (ny <= 32 ? 5 : 6));
        const int ly = lane & ((1 << sh) - 1), lz = lane >> sh, zstep = 64 >> sh;
        unsigned int pending = 0u, C = 0u;
        int zb = 0, yb = 0;
        for (;;) {
            const bool more = zb < nz;
            if (more) {
                const int z = z0 + zb + lz, y = y0 + yb + ly;
                bool pass = false;
                unsigned int c0 = 0u, ncell = 0u;
                if (z <= z1 && y <= y1) {
                    // distance of the query to the slab of cell row (z, y) along each axis (zero inside the slab)
                    const float zl = g.oz + (float)z * g.h, yl = g.oy + (float)y * g.h;
                    const float dz = fmaxf(fmaxf(zl - qz, qz - (zl + g.h)), 0.0f), dy = fmaxf(fmaxf(yl - qy, qy - (yl + g.h)), 0.0f);
                    const float rem = U2 - dz * dz - dy * dy;
                    const float xr = sqrtf(fmaxf(rem, 0.0f));
                    // (a row whose x-interval misses the grid is dropped: the clamp of cell_coord would scan its end cell)
                    if (rem >= 0.0f && !(qx - xr > g.ox + ext || qx + xr < g.ox)) {
                        const int x0 = cell_coord(qx - xr, g.ox, g), x1 = cell_coord(qx + xr, g.ox, g);
                        // nothing in any 4 x 4 x 4 group the row's x-interval passes through: no look-up
                        const unsigned int occ = s_occ[(z >> 2) * 32 + (y >> 2)];
                        const int xa = x0 >> 2, xb = x1 >> 2;
                        const unsigned int upto = xb >= 31 ? 0xFFFFFFFFu : ((1u << (xb + 1)) - 1u);
                        pass = (occ & upto & ~((1u << xa) - 1u)) != 0u;
                        c0 = ((unsigned int)z * G + y) * G + x0;
                        ncell = (unsigned int)(x1 - x0) + 1u;
                    }
                }
                const unsigned long long pm = __ballot(pass);
                if (pass) s_rows[wv][pending + (unsigned int)__popcll(pm & lt)] = make_uint2(c0, ncell);
                pending += (unsigned int)__popcll(pm);
                yb += 1 << sh;
                if (yb >= ny) { yb = 0; zb += zstep; }
                wave_lds_fence();
            }
            if (pending >= 64u || (!more && pending > 0u)) {
                const unsigned int take = pending < 64u ? pending : 64u;
                unsigned int b = 0u, e = 0u;
                if ((unsigned int)lane < take) {
                    const uint2 r = s_rows[wv][lane];
                    b = cell_start[r.x]; e = cell_start[r.x + r.y];
                }
                pending -= take;
                uint2 keep = make_uint2(0u, 0u);
                if ((unsigned int)lane < pending) keep = s_rows[wv][64 + lane];
                wave_lds_fence();
                if ((unsigned int)lane < pending) s_rows[wv][lane] = keep;
                wave_lds_fence();
                visit_rows(sorted, b, e, lane, [&](const float4 p, bool valid) {
                    const int id = __float_as_int(p.w);
                    const float dx = p.x - qx, ddy = p.y - qy, ddz = p.z - qz;
                    const float d = dx * dx + ddy * ddy + ddz * ddz;
                    // keep2 bounds the K-th nearest distance from above (U^2 with its margin, then the K-th best met so far): a
                    // point beyond it is not among the answer -- the cells of the cover hold ~8 times the points of the ball
                    const bool in = valid && !(EXCLUDE_SELF && id == q) && !(d > keep2);
                    const unsigned long long im = __ballot(in);
                    if (im == 0ull) return;
                    if (in) cand[C + (unsigned int)__popcll(im & lt)] = make_uint2(__float_as_uint(d), (unsigned int)id);
                    C += (unsigned int)__popcll(im);
                    wave_lds_fence();
                    if (C > (unsigned int)(BALL_CAND - 64)) {
                        C = ball_select<K>(cand, C, lane);
                        if (C == (unsigned int)K) keep2 = fminf(keep2, __uint_as_float(cand[K - 1].x));
                    }
                });
            } else if (!more) break;
        }
        // the answer: the list's K best in order (fewer than K points in the whole set: the rest reads "none")
        C = ball_select<K>(cand, C, lane);
        if (out_mean) {
            if (lane == 0) {
                float sum3 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) sum3 += (unsigned int)k < C ? __uint_as_float(cand[k].x) : 3.402823466e+38f;
                out_mean[q] = sum3 / 3.0f;
            }
        } else if (lane < K) {
            const bool has = (unsigned int)lane < C;
            const uint2 c = has ? cand[lane] : make_uint2(0u, 0u);
            out_d[(size_t)q * K + lane] = has ? __uint_as_float(c.x) : 3.402823466e+38f;
            out_i[(size_t)q * K + lane] = has ? (int64_t)(int)c.y : (int64_t)-1;
        }
        wave_lds_fence();   // (the list is reused by the wave's next query)
    }
}

__global__ void __launch_bounds__(256) k_ball_all(int N, int* __restrict__ ball_list, float* __restrict__ ball_u2, unsigned int* __restrict__ ball_count)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) *ball_count = (unsigned int)N;
    if (t < N) { ball_list[t] = t; ball_u2[t] = 3.402823466e+38f; }
}

// query sets up to this size skip the lane-per-query ring walk (SGR_KNN_DIRECT_MAX overrides; 0: never)
int grid_direct_max()
{
    static const int v = [] { const char* e = getenv("SGR_KNN_DIRECT_MAX"); return e ? atoi(e) : 262144; }();
    return v;
}

int grid_max_ring()
{
    static const int v = [] { const char* e = getenv("SGR_KNN_MAX_RING"); const int x = e ? atoi(e) : 0; return x > 0 ? x : GRID_MAX_RING; }();
    return v;
}

int grid_res(int M)
{
    int G = (int)ceil(cbrt((double)M / 6.0));
    if (G < 1) G = 1;
    if (G > 128) G = 128;
    return G;
}
// The resolution above gives ~6 points per cell when the points FILL their bounding box.  The clouds this library meets in
// practice lie on surfaces (Gaussians of a reconstructed scene; BASELINE config 4 binds them to a mesh): then only ~3 G^2 of the
// G^3 cells hold anything, each ~M / (3 G^2) points -- 110 at 1M -- and a query's first ring of 27 cells scans thousands of
// points (3.6 ms per 124k queries, self query 3.9 ms against 2.3 for a volume).  The finest resolution worth having for such a
// set, ~6 points per OCCUPIED cell of a surface: the scratch is sized for it, and build_grid picks between the two after
// counting how many cells the coarse grid actually fills (one 4-byte read-back: the only host round trip of a query, and only
// one query in GRID_PROBE_REUSE + 1 of the same size pays it: build_grid).
int grid_res_max(int M)
{
    int G = (int)ceil(sqrt((double)M / 18.0));
    const int G0 = grid_res(M);
    if (G < G0) G = G0;
    if (G > 128) G = 128;
    return G;
}
#define GRID_PROBE_MIN_POINTS 50000  // below this a query costs microseconds either way: no probe, no round trip

struct GridScratch { GridHdr* hdr; unsigned int* coarse; unsigned int *cell_count, *cursor, *cell_start, *cell_of, *scan_part; float4* sorted; int* far_list;
                     int* ball_list; float* ball_u2; float4* samples; unsigned int far_cap; size_t total; };
GridScratch carve_grid(char* base, int M)
{
    const int G = grid_res_max(M);
    const size_t cells = (size_t)G * G * G;
    GridScratch s;
    size_t off = 0;
    s.hdr = reinterpret_cast<GridHdr*>(base + off); off = sgr_align(off + sizeof(GridHdr));
    s.coarse = reinterpret_cast<unsigned int*>(base + off); off = sgr_align(off + GRID_COARSE_WORDS * 4);
    s.cell_count = reinterpret_cast<unsigned int*>(base + off); off = sgr_align(off + cells * 4);
    s.cursor = reinterpret_cast<unsigned int*>(base + off); off = sgr_align(off + cells * 4);
    s.cell_start = reinterpret_cast<unsigned int*>(base + off); off = sgr_align(off + (cells + 1) * 4);
    s.cell_of = reinterpret_cast<unsigned int*>(base + off); off = sgr_align(off + (size_t)M * 4);
    s.scan_part = reinterpret_cast<unsigned int*>(base + off); off = sgr_align(off + ((cells + GRID_SCAN_BLOCK - 1) / GRID_SCAN_BLOCK) * 4);
    s.sorted = reinterpret_cast<float4*>(base + off); off = sgr_align(off + (size_t)M * 16);
    s.far_cap = (unsigned int)(M > 65536 ? M : 65536);  // queries the exhaustive fallback can take (the rest walk on)
    s.far_list = reinterpret_cast<int*>(base + off); off = sgr_align(off + (size_t)s.far_cap * 4);
    s.ball_list = reinterpret_cast<int*>(base + off); off = sgr_align(off + (size_t)s.far_cap * 4);
    s.ball_u2 = reinterpret_cast<float*>(base + off); off = sgr_align(off + (size_t)s.far_cap * 4);
    s.samples = reinterpret_cast<float4*>(base + off); off = sgr_align(off + (size_t)BALL_SAMPLES * 16);
    s.total = off;
    return s;
}

__global__ void __launch_bounds__(256) k_grid_coarse(int G, const unsigned int* __restrict__ cell_start, unsigned int* __restrict__ coarse)
{
    const int Gc = (G + 3) >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;       // (fine row (z, y), coarse x)
    if (t >= G * G * Gc) return;
    const int xc = t % Gc, row = t / Gc, y = row % G, z = row / G;
    const unsigned int base = ((unsigned int)z * G + y) * G;
    const int xa = 4 * xc, xb = min(4 * xc + 4, G);
    if (cell_start[base + xb] > cell_start[base + xa]) atomicOr(&coarse[(z >> 2) * 32 + (y >> 2)], 1u << xc);
}

#define GRID_PROBE_REUSE 63   // sets of the same size that reuse a resolution before the occupancy is looked at again
// The occupancy probe is ASYNCHRONOUS (round 6; it used to synchronise the caller's stream on every 64th call): a call copies the
// occupied-cell count of the grid it built into pinned memory behind its kernels and records an event; a LATER call for a set of the
// same size reads the value once the event has completed (hipEventQuery, never a wait) and picks its resolution from it.  The
// resolution is a matter of speed only -- any G gives the same neighbours -- so a set's first query simply runs on the default grid.
struct ProbeSlot {
    unsigned int* p = nullptr;
    hipEvent_t ev = nullptr;
    int last_M = 0, last_G = 0, reused = 0;
    bool pending = false;
    int pending_M = 0, pending_G = 0;
    ~ProbeSlot() { /* leaked on purpose: the HIP runtime may be gone at thread exit */ }
};
thread_local ProbeSlot g_probe;

int count_cells(int M, const float* ref, const GridScratch& gs, int G, hipStream_t s)
{
    // cell_count and cursor are adjacent: one memset (sized for the finest grid the scratch was carved for)
    if (hipMemsetAsync(gs.cell_count, 0, reinterpret_cast<char*>(gs.cell_start) - reinterpret_cast<char*>(gs.cell_count), s) != hipSuccess)
        return SGR_E_HIP;
    hipLaunchKernelGGL(k_grid_count, dim3((M + 255) / 256), dim3(256), 0, s, M, ref, gs.hdr, G, gs.cell_count, gs.cell_of);
    return 0;
}

// returns the resolution the grid was built with (> 0), or a negative error code
int build_grid(int M, const float* ref, const GridScratch& gs, hipStream_t s)
{
    int G = grid_res(M);
    hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(64), 0, s, gs.hdr);
    hipLaunchKernelGGL(k_grid_bbox, dim3(min((M + 255) / 256, 1024)), dim3(256), 0, s, M, ref, gs.hdr);
    const int Gmax = grid_res_max(M);
    // The resolution is a matter of speed only (any G gives the same answer), and a caller queries the same kind of set again and
    // again (the level-set sampler: all Gaussians, once per view): the choice made for a set of M points is reused for the next
    // GRID_PROBE_REUSE sets of that size -- no second counting pass (0.13 ms at 1M) and no host round trip -- and probed again then.
    const bool probe = M >= GRID_PROBE_MIN_POINTS && Gmax > G;
    if (probe) {
        ProbeSlot& pr = g_probe;
        if (pr.pending && hipEventQuery(pr.ev) == hipSuccess) {
            pr.pending = false;
            if (pr.pending_M == M) {
                // how full was that grid?  ~6 points per occupied cell: the set fills its box, keep it.  Many more: the set is a
                // surface (or a few clusters): refine until an occupied cell holds ~6 again -- occupied cells of a surface grow with G^2
                const double per_cell = (double)M / (double)(pr.p[0] ? pr.p[0] : 1u);
                int Gn = pr.pending_G;
                if (per_cell > 24.0 || (per_cell < 2.0 && Gn > G)) {
                    Gn = (int)ceil((double)pr.pending_G * sqrt(per_cell / 6.0));
                    if (Gn > Gmax) Gn = Gmax;
                    if (Gn < G) Gn = G;
                }
                pr.last_M = M; pr.last_G = Gn; pr.reused = 0;
            }
        } else if (pr.pending) {
            (void)hipGetLastError();  // (hipErrorNotReady is not an error)
        }
        if (pr.last_M == M && pr.last_G > 0) G = pr.last_G;
    }
    {
        const int rc = count_cells(M, ref, gs, G, s);
        if (rc < 0) return rc;
    }
    const size_t cells = (size_t)G * G * G;
    const int scan_blocks = (int)((cells + GRID_SCAN_BLOCK - 1) / GRID_SCAN_BLOCK);
    hipLaunchKernelGGL(k_grid_blocksum, dim3(scan_blocks), dim3(256), 0, s, (int)cells, gs.cell_count, gs.scan_part, gs.hdr);
    if (probe) {
        ProbeSlot& pr = g_probe;
        const bool due = !(pr.last_M == M && pr.last_G > 0) || ++pr.reused > GRID_PROBE_REUSE;
        if (due && !pr.pending) {
            if (!pr.p && hipHostMalloc(reinterpret_cast<void**>(&pr.p), 64, hipHostMallocDefault) != hipSuccess) return SGR_E_HIP;
            if (!pr.ev && hipEventCreateWithFlags(&pr.ev, hipEventDisableTiming) != hipSuccess) return SGR_E_HIP;
            if (hipMemcpyAsync(pr.p, &gs.hdr->occupied, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return SGR_E_HIP;
            if (hipEventRecord(pr.ev, s) != hipSuccess) return SGR_E_HIP;
            pr.pending = true; pr.pending_M = M; pr.pending_G = G;
        }
    }
    hipLaunchKernelGGL(k_grid_scan, dim3(scan_blocks), dim3(256), 0, s, (int)cells, gs.cell_count, gs.scan_part, gs.cell_start);
    hipLaunchKernelGGL(k_grid_scatter, dim3((M + 255) / 256), dim3(256), 0, s, M, ref, gs.cell_of, gs.cell_start, gs.cursor, gs.sorted);
    if (hipMemsetAsync(gs.coarse, 0, GRID_COARSE_WORDS * 4, s) != hipSuccess) return SGR_E_HIP;
    const int coarse_threads = G * G * ((G + 3) >> 2);
    hipLaunchKernelGGL(k_grid_coarse, dim3((coarse_threads + 255) / 256), dim3(256), 0, s, G, gs.cell_start, gs.coarse);
    hipLaunchKernelGGL(k_grid_samples, dim3(BALL_SAMPLES / 256), dim3(256), 0, s, G, gs.cell_start, gs.sorted, gs.samples);
    return G;
}

// the two exhaustive fallbacks of a grid query (each looks at the far count and takes its own regime)
template <int K, bool EXCLUDE_SELF>
void launch_far(int N, const float* query, int M, const float* ref, const GridScratch& gs, int G, float* d, int64_t* i, float* mean, hipStream_t s)
{
    const unsigned int cap = gs.far_cap < (unsigned int)N ? gs.far_cap : (unsigned int)N;
    // one wave per query, BALL_WAVES per workgroup, grid-stride over the (device-side) count: two workgroups per CU hold their 76 KB of LDS
    const unsigned int ball_groups = (cap + BALL_WAVES - 1) / BALL_WAVES;
    hipLaunchKernelGGL((k_knn_ball<K, EXCLUDE_SELF>), dim3(ball_groups < BALL_GRID_MAX ? ball_groups : BALL_GRID_MAX), dim3(64 * BALL_WAVES), 0, s, query,
                       gs.hdr, G, gs.cell_start, gs.sorted, d, i, mean, gs.ball_list, gs.ball_u2, &gs.hdr->ball_count, cap, gs.coarse,
                       gs.samples);
    constexpr int Q = K <= 4 ? 8 : (K <= 16 ? 4 : 2);  // queries per workgroup: Q * K (distance, index) pairs per thread in registers
    const unsigned int groups = (cap + Q - 1) / Q;
    hipLaunchKernelGGL((k_knn_far<K, EXCLUDE_SELF, Q>), dim3(groups < 4096u ? groups : 4096u), dim3(256), 0, s, query, M, gs.sorted, d, i,
                       mean, gs.far_list, &gs.hdr->far_count, cap);
    hipLaunchKernelGGL((k_knn<K, EXCLUDE_SELF>), dim3((cap + 255) / 256), dim3(256), 0, s, (int)cap, query, M, ref, d, i, mean,
                       (const int*)gs.far_list, (const unsigned int*)&gs.hdr->far_count, FAR_SINGLE_MAX);
}

template <int K>
void launch_grid_query(bool self, int N, const float* query, int M, const float* ref, const GridScratch& gs, int G, float* d, int64_t* i,
                       hipStream_t s)
{
    if (self)
        hipLaunchKernelGGL((k_grid_query<K, false, true>), dim3((N + 127) / 128), dim3(128), 0, s, N, query, gs.hdr, G, gs.cell_start,
                           gs.sorted, d, i, (float*)nullptr, &gs.hdr->far_count, gs.far_list, gs.far_cap, grid_max_ring(), &gs.hdr->ball_count,
                           gs.ball_list, gs.ball_u2);
    else if ((unsigned int)N <= gs.far_cap && N <= grid_direct_max() && G > grid_res(M))
        // A query set too small to fill the device one LANE per query (the level-set sampler's 124k pixels are 7 waves per CU, each
        // as slow as its slowest lane's serial walk) goes to the wave-per-query kernel directly, every query without a bound --
        // when the reference set is a SURFACE (build_grid refined the grid: G > grid_res(M)).  Same box, 124k queries: config 4's
        // cloud 1.50 -> 1.09 ms on the surface, 3.26 -> 2.77 at distance 1; over a VOLUME (the metric scene, 0 / 10 / 30 % of the
        // queries outside it) the ring walk finishes most queries itself and stays ahead: 0.74 / 0.64 / 0.69 against 1.02 / 0.93 / 1.00.
        hipLaunchKernelGGL(k_ball_all, dim3((N + 255) / 256), dim3(256), 0, s, N, gs.ball_list, gs.ball_u2, &gs.hdr->ball_count);
    else
        hipLaunchKernelGGL((k_grid_query<K, false, false>), dim3((N + 127) / 128), dim3(128), 0, s, N, query, gs.hdr, G, gs.cell_start,
                           gs.sorted, d, i, (float*)nullptr, &gs.hdr->far_count, gs.far_list, gs.far_cap, grid_max_ring(), &gs.hdr->ball_count,
                           gs.ball_list, gs.ball_u2);
    launch_far<K, false>(N, query, M, ref, gs, G, d, i, nullptr, s);
}

}  // namespace

extern "C" {

size_t sgr_knn_grid_scratch_bytes(int M) { return carve_grid(nullptr, M > 0 ? M : 1).total; }

int sgr_knn_grid(int N, const float* query, int M, const float* ref, int K, float* dists, int64_t* idx, char* scratch, void* stream)
{
    if (N <= 0) return 0;
    if (!query || !ref || !dists || !idx || !scratch || M <= 0) return SGR_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const GridScratch gs = carve_grid(scratch, M);
    const int G = build_grid(M, ref, gs, s);
    if (G < 0) return G;
    const bool self = (query == ref && N == M);
    switch (K) {
        case 1: launch_grid_query<1>(self, N, query, M, ref, gs, G, dists, idx, s); break;
        case 2: launch_grid_query<2>(self, N, query, M, ref, gs, G, dists, idx, s); break;
        case 3: launch_grid_query<3>(self, N, query, M, ref, gs, G, dists, idx, s); break;
        case 4: launch_grid_query<4>(self, N, query, M, ref, gs, G, dists, idx, s); break;
        case 8: launch_grid_query<8>(self, N, query, M, ref, gs, G, dists, idx, s); break;
        case 16: launch_grid_query<16>(self, N, query, M, ref, gs, G, dists, idx, s); break;
        case 32: launch_grid_query<32>(self, N, query, M, ref, gs, G, dists, idx, s); break;
        default: return SGR_E_INVALID;
    }
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

int sgr_dist2_grid(int P, const float* points, float* meanDists, char* scratch, void* stream)
{
    if (P <= 0) return 0;
    if (!points || !meanDists || !scratch) return SGR_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const GridScratch gs = carve_grid(scratch, P);
    const int G = build_grid(P, points, gs, s);
    if (G < 0) return G;
    hipLaunchKernelGGL((k_grid_query<3, true, true>), dim3((P + 127) / 128), dim3(128), 0, s, P, points, gs.hdr, G,
                       gs.cell_start, gs.sorted, (float*)nullptr, (int64_t*)nullptr, meanDists, &gs.hdr->far_count, gs.far_list,
                       gs.far_cap, grid_max_ring(), &gs.hdr->ball_count, gs.ball_list, gs.ball_u2);
    launch_far<3, true>(P, points, P, points, gs, G, nullptr, nullptr, meanDists, s);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

}  // extern "C"
