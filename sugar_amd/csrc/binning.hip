// binning.hip -- per-tile binning for gfx950: tile scan, bucket scatter, per-tile depth sort.
//
// Replaces the reference's global pipeline
//   cub::DeviceScan::InclusiveSum over P Gaussians      DGR/cuda_rasterizer/rasterizer_impl.cu:277
//   duplicateWithKeys  (64-bit tile|depth keys)          :70-111
//   cub::DeviceRadixSort::SortPairs over R instances     :303-308   (6 passes x 24 B x R at 1080p)
//   cudaMemset + identifyTileRanges                      :310-317, :116-138
// with a multi-workgroup counting sort on the tile id followed by an independent in-LDS sort of each tile's bucket.
// No global atomics: workgroup b of a persistent grid owns a contiguous slice of Gaussians and a private LDS histogram.
//   (preprocess): blk_hist[b][t] = instances of tile t among workgroup b's Gaussians          (LDS atomics)
//   hist_scan  : column-wise exclusive scan over b (in place) + per-tile totals
//   tile_scan  : exclusive scan of the totals -> tile_start[T+1]; ranges fall out for free; R and the largest count
//   scatter    : workgroup b re-walks its slice; slot = tile_start[t] + blk_hist[b][t] + (LDS atomic rank); stores the
//                64-bit key (depth_bits << 32 | gaussian_id).  A workgroup's instances of a tile are contiguous.
//   tile_sort  : one workgroup per tile sorts its bucket in LDS (bitonic network on u64 keys) and writes the ids.
// HBM traffic per instance: 8 B scatter + 8 B read + 4 B write (+ 16 B per (workgroup, tile) of histogram traffic),
// versus ~144 B per instance for the 6-pass radix sort of 12-byte pairs.
//
// Order contract: within a tile the reference's stable sort orders by depth bits, ties by ascending
// Gaussian index (the emission order of duplicateWithKeys).  Sorting the composite key
// (depth_bits << 32 | id) reproduces that order exactly and makes the result independent of the order
// in which the scatter's atomics were served.
#include "sgr_device.h"

namespace {

// ---- tile scan: one 1024-thread workgroup, T <= a few 10^4 -----------------------------------
__global__ void __launch_bounds__(1024) k_tile_scan(int T, const uint32_t* __restrict__ tile_count,
                                                    uint32_t* __restrict__ tile_start, uint32_t* __restrict__ header)
{
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_max[16];
    const int tid = threadIdx.x;
    const int per = (T + 1023) / 1024;
    const int b = tid * per, e = min(T, b + per);
    uint32_t sum = 0, mx = 0;
    for (int i = b; i < e; i++) { uint32_t c = tile_count[i]; sum += c; mx = max(mx, c); }
    s_part[tid] = sum;
    // wave max
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if ((tid & 63) == 0) s_max[tid >> 6] = mx;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int o = 1; o < 1024; o <<= 1) {
        uint32_t v = (tid >= o) ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;  // exclusive prefix of this thread's chunk
    for (int i = b; i < e; i++) { tile_start[i] = run; run += tile_count[i]; }
    if (tid == 1023) {
        tile_start[T] = s_part[1023];
        header[SGR_HDR_R] = s_part[1023];
        header[SGR_HDR_R_HI] = 0;
        uint32_t m = 0;
        for (int w = 0; w < 16; w++) m = max(m, s_max[w]);
        header[SGR_HDR_MAXCOUNT] = m;
    }
}

// ---- column scan of the per-workgroup histograms --------------------------------------------
// One lane per tile walks the n_blocks rows (coalesced across tiles): blk_hist[b][t] becomes the exclusive prefix over b,
// tile_count[t] the total.  Loads are issued UNROLL at a time so each lane keeps several rows in flight.
__global__ void __launch_bounds__(256) k_hist_scan(int T, int n_blocks, uint32_t* __restrict__ blk_hist,
                                                   uint32_t* __restrict__ tile_count)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    constexpr int UNROLL = 16;
    uint32_t run = 0;
    int b = 0;
    for (; b + UNROLL <= n_blocks; b += UNROLL) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = blk_hist[(size_t)(b + u) * T + t];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { blk_hist[(size_t)(b + u) * T + t] = run; run += v[u]; }
    }
    for (; b < n_blocks; b++) { const uint32_t v = blk_hist[(size_t)b * T + t]; blk_hist[(size_t)b * T + t] = run; run += v; }
    tile_count[t] = run;
}

// ---- scatter ---------------------------------------------------------------------------------
// LDS variant: persistent grid matching the preprocess kernel's slices.  s_base[t] starts at
// tile_start[t] + (exclusive prefix of this workgroup's predecessors) and is bumped with returning LDS atomics.
__global__ void __launch_bounds__(256) k_scatter_lds(int P, int gx, int gy, int per_block, const GeomRec* __restrict__ rec,
                                                     const uint32_t* __restrict__ tile_start,
                                                     const uint32_t* __restrict__ blk_hist, uint64_t* __restrict__ keys)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_base[];
    const int T = gx * gy;
    const uint32_t* mine = blk_hist + (size_t)blockIdx.x * T;
    for (int i = threadIdx.x; i < T; i += 256) s_base[i] = tile_start[i] + mine[i];
    __syncthreads();
    const int begin = blockIdx.x * per_block;
    const int end = min(P, begin + per_block);
    for (int base = begin; base < end; base += 256) {
        const int idx = base + threadIdx.x;
        if (idx >= end) continue;
        const float4* rp = reinterpret_cast<const float4*>(rec + idx);
        const float4 r2 = rp[2];
        const int radius = __float_as_int(r2.z);
        if (!(radius > 0)) continue;
        const float4 r0 = rp[0];
        int minx, miny, maxx, maxy;
        sgr_get_rect(r0.x, r0.y, radius, gx, gy, minx, miny, maxx, maxy);
        const uint64_t key = ((uint64_t)__float_as_uint(r2.y) << 32) | (uint32_t)idx;
        for (int y = miny; y < maxy; y++)
            for (int x = minx; x < maxx; x++) {
                const uint32_t slot = atomicAdd(&s_base[y * gx + x], 1u);
                keys[slot] = key;
            }
    }
}

// Fallback (tile histogram too large for LDS): one lane per Gaussian, returning global atomics on a cursor array.
__global__ void __launch_bounds__(256) k_scatter_atomic(int P, int gx, int gy, const GeomRec* __restrict__ rec,
                                                        const uint32_t* __restrict__ tile_start,
                                                        uint32_t* __restrict__ tile_cursor, uint64_t* __restrict__ keys)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float4* rp = reinterpret_cast<const float4*>(rec + idx);
    const float4 r0 = rp[0];
    const float4 r2 = rp[2];
    const int radius = __float_as_int(r2.z);
    if (!(radius > 0)) return;
    int minx, miny, maxx, maxy;
    sgr_get_rect(r0.x, r0.y, radius, gx, gy, minx, miny, maxx, maxy);
    const uint64_t key = ((uint64_t)__float_as_uint(r2.y) << 32) | (uint32_t)idx;
    for (int y = miny; y < maxy; y++)
        for (int x = minx; x < maxx; x++) {
            const int t = y * gx + x;
            const uint32_t slot = tile_start[t] + atomicAdd(&tile_cursor[t], 1u);
            keys[slot] = key;
        }
}

// ---- per-tile sort ---------------------------------------------------------------------------
// Bitonic sorting network for an ARBITRARY count n (no physical padding): the "flip + disperse" form in
// which every comparator points the same way (min to the lower index).  Positions >= n behave as +inf;
// since a comparator (i < l) only swaps when s[i] > s[l], a virtual +inf at l never moves, so pairs
// with l >= n are simply skipped.  Works for LDS and global pointers alike.
template <int NT, typename Ptr>
__device__ __forceinline__ void bitonic_sort_n(Ptr s, uint32_t n, int tid)
{
    uint32_t N = 1, logN = 0;
    while (N < n) { N <<= 1; logN++; }
    for (uint32_t lk = 1; lk <= logN; lk++) {
        const uint32_t k = 1u << lk, half = k >> 1;
        for (uint32_t t = tid; t < (N >> 1); t += NT) {  // flip
            const uint32_t blk = t >> (lk - 1), off = t & (half - 1);
            const uint32_t i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < n) {
                const uint64_t a = s[i], b = s[l];
                if (a > b) { s[i] = b; s[l] = a; }
            }
        }
        __syncthreads();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {  // disperse
            for (uint32_t t = tid; t < (N >> 1); t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t l = i | j;
                if (l < n) {
                    const uint64_t a = s[i], b = s[l];
                    if (a > b) { s[i] = b; s[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Tiles with lo < count <= hi are sorted in LDS by this launch; other tiles exit immediately.
template <int NT>
__global__ void __launch_bounds__(NT) k_tile_sort_lds(const uint32_t* __restrict__ tile_start, uint32_t lo, uint32_t hi,
                                                      const uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
    const int tile = blockIdx.x;
    const uint32_t b = tile_start[tile], e = tile_start[tile + 1];
    const uint32_t n = e - b;
    if (n <= lo || n > hi) return;
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < n; i += NT) s[i] = keys[b + i];
    __syncthreads();
    bitonic_sort_n<NT>(s, n, tid);
    for (uint32_t i = tid; i < n; i += NT) point_list[b + i] = (uint32_t)s[i];
}

// Fallback for tiles too large for LDS: the same network directly on the bucket in global memory
// (one 1024-thread workgroup per tile; slow, only for degenerate scenes).
__global__ void __launch_bounds__(1024) k_tile_sort_global(const uint32_t* __restrict__ tile_start, uint32_t lo,
                                                           uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list)
{
    const int tile = blockIdx.x;
    const uint32_t b = tile_start[tile], e = tile_start[tile + 1];
    const uint32_t n = e - b;
    if (n <= lo) return;
    const int tid = threadIdx.x;
    volatile uint64_t* s = keys + b;
    bitonic_sort_n<1024>(s, n, tid);
    for (uint32_t i = tid; i < n; i += 1024) point_list[b + i] = (uint32_t)s[i];
}

}  // namespace

void sgr_launch_tile_scan(int T, const uint32_t* tile_count, uint32_t* tile_start, uint32_t* header, hipStream_t s)
{
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, T, tile_count, tile_start, header);
}

void sgr_launch_hist_scan(int T, int n_blocks, uint32_t* blk_hist, uint32_t* tile_count, hipStream_t s)
{
    hipLaunchKernelGGL(k_hist_scan, dim3((T + 255) / 256), dim3(256), 0, s, T, n_blocks, blk_hist, tile_count);
}

void sgr_launch_scatter(int P, int gx, int gy, const GeomRec* rec, const uint32_t* tile_start, uint32_t* tile_cursor,
                        const uint32_t* blk_hist, int n_blocks, int per_block, uint64_t* keys, hipStream_t s)
{
    if (P <= 0) return;
    if (blk_hist) {
        const size_t lds = (size_t)gx * gy * 4;
        static size_t configured = 0;
        if (lds > configured) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
            configured = lds;
        }
        hipLaunchKernelGGL(k_scatter_lds, dim3(n_blocks), dim3(256), lds, s, P, gx, gy, per_block, rec, tile_start, blk_hist, keys);
    } else {
        hipLaunchKernelGGL(k_scatter_atomic, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, gy, rec, tile_start, tile_cursor, keys);
    }
}

// Size classes of the per-tile sort: LDS per workgroup bounds how many tiles a CU sorts concurrently.
//   count <= 1024  :   8 KB, 128 threads      count <= 4096 : 32 KB, 256 threads
//   count <= 16384 : 128 KB, 1024 threads     larger        : in place in global memory (degenerate scenes)
void sgr_launch_tile_sort(int T, uint32_t max_count, const uint32_t* tile_start, uint64_t* keys, uint32_t* point_list,
                          hipStream_t s)
{
    if (T <= 0 || max_count == 0) return;
    hipLaunchKernelGGL(k_tile_sort_lds<128>, dim3(T), dim3(128), 1024 * 8, s, tile_start, 0u, 1024u, keys, point_list);
    if (max_count > 1024u)
        hipLaunchKernelGGL(k_tile_sort_lds<256>, dim3(T), dim3(256), 4096 * 8, s, tile_start, 1024u, 4096u, keys, point_list);
    if (max_count > 4096u) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_sort_lds<1024>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            attr_set = true;
        }
        hipLaunchKernelGGL(k_tile_sort_lds<1024>, dim3(T), dim3(1024), 16384 * 8, s, tile_start, 4096u, 16384u, keys, point_list);
    }
    if (max_count > 16384u)
        hipLaunchKernelGGL(k_tile_sort_global, dim3(T), dim3(1024), 0, s, tile_start, 16384u, keys, point_list);
}
