// binning.hip -- per-tile binning for gfx950: tile scan, bucket scatter, per-tile depth sort.
//
// Replaces the reference's global pipeline
//   cub::DeviceScan::InclusiveSum over P Gaussians      DGR/cuda_rasterizer/rasterizer_impl.cu:277
//   duplicateWithKeys  (64-bit tile|depth keys)          :70-111
//   cub::DeviceRadixSort::SortPairs over R instances     :303-308   (6 passes x 24 B x R at 1080p)
//   cudaMemset + identifyTileRanges                      :310-317, :116-138
// with a counting sort on the tile id (the counts come from the preprocess kernel's per-tile atomics)
// followed by an independent in-LDS sort of each tile's bucket:
//   tile_scan  : exclusive scan of tile_count[T] -> tile_start[T+1]; ranges fall out for free
//   scatter    : every (Gaussian, tile) instance claims a slot in its tile's bucket and stores the
//                64-bit key (depth_bits << 32 | gaussian_id)
//   tile_sort  : one workgroup per tile sorts its bucket in LDS (bitonic, u64 keys) and writes the ids.
// HBM traffic per instance: 8 B scatter + 8 B read + 4 B write, versus ~144 B for the radix sort.
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

// ---- scatter: one lane per Gaussian (v1: serial loop over the Gaussian's tile rectangle) ------
__global__ void __launch_bounds__(256) k_scatter(int P, int gx, int gy, const GeomRec* __restrict__ rec,
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

void sgr_launch_scatter(int P, int gx, int gy, const GeomRec* rec, const uint32_t* tile_start, uint32_t* tile_cursor,
                        uint64_t* keys, hipStream_t s)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_scatter, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, gy, rec, tile_start, tile_cursor, keys);
}

#define SGR_SORT_SMALL 2048u   // <= 16 KB of LDS per workgroup: many tiles per CU
#define SGR_SORT_LARGE 16384u  // <= 128 KB of LDS: one tile per CU

void sgr_launch_tile_sort(int T, uint32_t max_count, const uint32_t* tile_start, uint64_t* keys, uint32_t* point_list,
                          hipStream_t s)
{
    if (T <= 0 || max_count == 0) return;
    hipLaunchKernelGGL(k_tile_sort_lds<256>, dim3(T), dim3(256), SGR_SORT_SMALL * 8, s, tile_start, 0u, SGR_SORT_SMALL,
                       keys, point_list);
    if (max_count > SGR_SORT_SMALL) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_sort_lds<1024>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SGR_SORT_LARGE * 8);
            attr_set = true;
        }
        hipLaunchKernelGGL(k_tile_sort_lds<1024>, dim3(T), dim3(1024), SGR_SORT_LARGE * 8, s, tile_start, SGR_SORT_SMALL,
                           SGR_SORT_LARGE, keys, point_list);
    }
    if (max_count > SGR_SORT_LARGE)
        hipLaunchKernelGGL(k_tile_sort_global, dim3(T), dim3(1024), 0, s, tile_start, SGR_SORT_LARGE, keys, point_list);
}
