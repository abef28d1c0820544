// pick.hip -- the pixel subset and the row compaction of the level-set sampling pass WITHOUT host round trips (round 6).
//
// sugar_scene/sugar_model.py:1929-1957 (compute_level_surface_points_from_camera_fast): of the pixels that have a depth, a uniformly
// random subset of n_surface_points is kept (`torch.randperm(n_valid)[:n]`, drawn on the CPU there).  The sizes of everything behind
// it then depend on how many pixels were valid -- as tensor code that is a `nonzero`, a `min(n, n_valid)` and a slice: three waits
// for the GPU per view.  Here the subset is chosen on the device into a buffer of FIXED size (n_surface_points rows) with the count
// beside it, and nothing waits:
//
//   key(i)   = a bijective 32-bit hash of (pixel index ^ seed): distinct pixels have distinct keys, so "the k smallest keys among the
//              valid pixels" is a well-defined, uniformly distributed k-subset (every pixel is equally likely to be in it; the order of
//              the output is raster order, which no consumer of the sampler depends on)
//   select   = three-level radix select of the k-th smallest key (11 + 11 + 10 key bits): a 2048-bin histogram per level -- level 1 over
//              all valid pixels through workgroup-private LDS histograms, levels 2 and 3 over the few keys under the prefix found so
//              far -- and after each one workgroup finds the bin the k-th key falls into
//   compact  = ordered compaction of the pixels with key <= threshold (block counts, one-workgroup scan, write)
//
// and, behind the level-set kernel, sgr_compact_level_rows gathers every level's valid rows to the front of fixed-size outputs and
// leaves the per-level counts on the device.
#include "../../include/sugar_raster.h"
#include "sgr_common.h"

namespace {

#define PICK_BINS 2048     // bins per level of the radix select: key bits 31..21, 20..10, 9..0 (the last level uses 1024 of them)
#define PICK_BLOCK 1024   // pixels per workgroup of the compaction passes

// selection state (device): n_valid; k_eff = min(k, n_valid); prefix = the key bits fixed so far (levels above the current one);
// need = keys still to take inside the current prefix; thresh = the k-th smallest key (final); none = "select nothing"
struct PickState { uint32_t n_valid, k_eff, prefix, need, thresh, none, pad0, pad1; };

__device__ __forceinline__ uint32_t pick_key(uint32_t i, uint32_t seed)
{
    uint32_t h = i ^ (seed * 0x9E3779B9u);   // (murmur3's finaliser: a bijection of the 32-bit integers)
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ bool pick_valid(float d) { return !(d < 0.f); }   // sugar_model.py:1929: no_proj_mask = depth < 0

// level 1: the 11 upper key bits of every valid pixel, in a workgroup-private LDS histogram first (a 65 536-bin histogram in global
// memory -- the first version -- was 727 000 global atomics per view, 75 us at the ~10 atomics per ns this part sustains)
__global__ void __launch_bounds__(1024) k_pick_hist1(int n, const float* __restrict__ depth, uint32_t seed, uint32_t* __restrict__ hist,
                                                     PickState* __restrict__ st)
{
    __shared__ uint32_t s_h[PICK_BINS];
    for (int b = threadIdx.x; b < PICK_BINS; b += 1024) s_h[b] = 0u;
    __syncthreads();
    uint32_t cnt = 0;
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) {
        if (pick_valid(depth[i])) { atomicAdd(&s_h[pick_key((uint32_t)i, seed) >> 21], 1u); cnt++; }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&st->n_valid, cnt);
    __syncthreads();
    for (int b = threadIdx.x; b < PICK_BINS; b += 1024) { const uint32_t c = s_h[b]; if (c) atomicAdd(&hist[b], c); }
}

// levels 2 and 3: only the keys under the prefix found so far (a few hundred, then a handful): straight to the global histogram
__global__ void __launch_bounds__(256) k_pick_hist23(int n, const float* __restrict__ depth, uint32_t seed, uint32_t* __restrict__ hist,
                                                     const PickState* __restrict__ st, int level)
{
    if (st->none) return;
    const uint32_t prefix = st->prefix;
    const int sh_hi = level == 2 ? 21 : 10, sh_lo = level == 2 ? 10 : 0;
    const uint32_t mask = level == 2 ? 0x7FFu : 0x3FFu;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!pick_valid(depth[i])) continue;
        const uint32_t k = pick_key((uint32_t)i, seed);
        if ((k >> sh_hi) == (prefix >> sh_hi)) atomicAdd(&hist[(k >> sh_lo) & mask], 1u);
    }
}

// one workgroup: the smallest bin b with (bins 0..b summed) >= need; the prefix takes the bin's bits, need becomes the rank inside it
__global__ void __launch_bounds__(1024) k_pick_select(const uint32_t* __restrict__ hist, PickState* __restrict__ st, int level, uint32_t k)
{
    __shared__ uint32_t s_wave[16];
    const int tid = threadIdx.x;
    uint32_t need;
    if (level == 1) {
        need = min(k, st->n_valid);
        if (tid == 0) { st->k_eff = need; st->none = need == 0u ? 1u : 0u; }
    } else {
        need = st->need;
        if (st->none) return;
    }
    if (need == 0u) return;
    constexpr int PER = PICK_BINS / 1024;
    uint32_t sum = 0;
    for (int j = 0; j < PER; j++) sum += hist[tid * PER + j];
    uint32_t incl = sum;
    const int lane = tid & 63, wave = tid >> 6;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += y; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    incl += before;
    const uint32_t excl = incl - sum;
    if (excl < need && need <= incl) {   // the crossing lies in this thread's bins (exactly one thread)
        uint32_t run = excl;
        for (int j = 0; j < PER; j++) {
            const uint32_t c = hist[tid * PER + j];
            if (run + c >= need) {
                const uint32_t bin = (uint32_t)(tid * PER + j);
                const uint32_t p = level == 1 ? (bin << 21) : (level == 2 ? (st->prefix | (bin << 10)) : (st->prefix | bin));
                st->prefix = p;
                st->need = need - run;
                if (level == 3) st->thresh = p;   // keys are distinct: exactly one pixel carries it
                break;
            }
            run += c;
        }
    }
}

__device__ __forceinline__ bool pick_taken(int i, int n, const float* depth, uint32_t seed, const PickState& s)
{
    return i < n && !s.none && pick_valid(depth[i]) && pick_key((uint32_t)i, seed) <= s.thresh;
}

__global__ void __launch_bounds__(PICK_BLOCK) k_pick_count(int n, const float* __restrict__ depth, uint32_t seed, const PickState* __restrict__ st,
                                                           uint32_t* __restrict__ blk_count)
{
    __shared__ uint32_t s_w[PICK_BLOCK / 64];
    const PickState s = *st;
    const int i = blockIdx.x * PICK_BLOCK + threadIdx.x;
    const unsigned long long m = __ballot(pick_taken(i, n, depth, seed, s));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int w = 0; w < PICK_BLOCK / 64; w++) c += s_w[w];
        blk_count[blockIdx.x] = c;
    }
}

// exclusive scan of the block counts in place, one workgroup
__global__ void __launch_bounds__(1024) k_pick_scan(int n_blocks, uint32_t* __restrict__ blk_count)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < n_blocks ? blk_count[i] : 0u;
        uint32_t incl = v;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += y; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave; w++) before += s_wave[w];
        if (i < n_blocks) blk_count[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(PICK_BLOCK) k_pick_write(int n, const float* __restrict__ depth, uint32_t seed, const PickState* __restrict__ st,
                                                           const uint32_t* __restrict__ blk_start, int k, int64_t* __restrict__ picked,
                                                           uint32_t* __restrict__ count_out, uint32_t* __restrict__ n_valid_out)
{
    __shared__ uint32_t s_w[PICK_BLOCK / 64];
    const PickState s = *st;
    const int i = blockIdx.x * PICK_BLOCK + threadIdx.x;
    const bool take = pick_taken(i, n, depth, seed, s);
    const unsigned long long m = __ballot(take);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = blk_start[blockIdx.x];
    for (int w = 0; w < wave; w++) off += s_w[w];
    if (take) {
        const uint32_t slot = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (slot < (uint32_t)k) picked[slot] = (int64_t)i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_out) *count_out = s.k_eff;
        if (n_valid_out) *n_valid_out = s.n_valid;
    }
}

// rows behind the count repeat the first picked pixel (a valid pixel: everything behind the pick stays finite); with no valid pixel
// at all they are pixel 0
__global__ void __launch_bounds__(256) k_pick_pad(int k, const PickState* __restrict__ st, int64_t* __restrict__ picked)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = st->k_eff;
    if (j < k && (uint32_t)j >= c) picked[j] = c ? picked[0] : 0;
}

// per level: the rows with valid[l][n] != 0 and n < *n_rows, in order, to the front.  Two launches over (row blocks) x (levels):
// block counts, then every workgroup sums the counts of the blocks before it (at most a few hundred words) and writes its rows.
// (First version: one workgroup per level walking all rows -- 0.2 ms for 124 000 rows, a tenth of the whole sampling pass.)
#define ROWS_BLOCK 1024
__device__ __forceinline__ bool row_taken(int i, uint32_t lim, const uint8_t* v) { return (uint32_t)i < lim && v[i] != 0; }

__global__ void __launch_bounds__(ROWS_BLOCK) k_rows_count(int N, const uint8_t* __restrict__ valid, const uint32_t* __restrict__ n_rows,
                                                           uint32_t* __restrict__ blk_count)
{
    __shared__ uint32_t s_w[ROWS_BLOCK / 64];
    const int l = blockIdx.y, i = blockIdx.x * ROWS_BLOCK + threadIdx.x;
    const uint32_t lim = n_rows ? min((uint32_t)N, *n_rows) : (uint32_t)N;
    const unsigned long long m = __ballot(row_taken(i, lim, valid + (size_t)l * N));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int w = 0; w < ROWS_BLOCK / 64; w++) c += s_w[w];
        blk_count[l * gridDim.x + blockIdx.x] = c;
    }
}

__global__ void __launch_bounds__(ROWS_BLOCK) k_rows_write(int N, const uint8_t* __restrict__ valid, const uint32_t* __restrict__ n_rows,
                                                           const uint32_t* __restrict__ blk_count, const float* __restrict__ pts,
                                                           const float* __restrict__ nrm, const int64_t* __restrict__ tag_a,
                                                           const int64_t* __restrict__ tag_b, int64_t* __restrict__ rows_out,
                                                           float* __restrict__ pts_out, float* __restrict__ nrm_out,
                                                           int64_t* __restrict__ tag_a_out, int64_t* __restrict__ tag_b_out,
                                                           uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s_w[ROWS_BLOCK / 64];
    __shared__ uint32_t s_before;
    const int l = blockIdx.y, i = blockIdx.x * ROWS_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lim = n_rows ? min((uint32_t)N, *n_rows) : (uint32_t)N;
    const bool take = row_taken(i, lim, valid + (size_t)l * N);
    const unsigned long long m = __ballot(take);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    if (wave == 0) {   // rows of this level in the blocks before this one (and, in the last block, the level's total)
        uint32_t c = 0;
        for (int b = lane; b < (int)blockIdx.x; b += 64) c += blk_count[l * gridDim.x + b];
        for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
        if (lane == 0) s_before = c;
    }
    __syncthreads();
    uint32_t off = s_before;
    for (int w = 0; w < wave; w++) off += s_w[w];
    if (take) {
        const size_t o = (size_t)l * N + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const size_t src = (size_t)l * N + i;
        rows_out[o] = i;
        pts_out[3 * o] = pts[3 * src]; pts_out[3 * o + 1] = pts[3 * src + 1]; pts_out[3 * o + 2] = pts[3 * src + 2];
        if (nrm_out) { nrm_out[3 * o] = nrm[3 * src]; nrm_out[3 * o + 1] = nrm[3 * src + 1]; nrm_out[3 * o + 2] = nrm[3 * src + 2]; }
        if (tag_a_out) tag_a_out[o] = tag_a[i];
        if (tag_b_out) tag_b_out[o] = tag_b[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        uint32_t c = s_before;
        for (int w = 0; w < ROWS_BLOCK / 64; w++) c += s_w[w];
        counts[l] = c;
    }
}

}  // namespace

extern "C" {

size_t sgr_pick_pixels_scratch_bytes(int n_pix)
{
    const size_t blocks = ((size_t)(n_pix > 0 ? n_pix : 1) + PICK_BLOCK - 1) / PICK_BLOCK;
    return sgr_align(3 * (size_t)PICK_BINS * 4 + sizeof(PickState)) + sgr_align(blocks * 4);
}

size_t sgr_compact_level_rows_scratch_bytes(int N, int L)
{
    return sgr_align((size_t)((N > 0 ? N : 1) + ROWS_BLOCK - 1) / ROWS_BLOCK * (size_t)(L > 0 ? L : 1) * 4);
}

int sgr_pick_pixels(int n_pix, const float* depth, int k, uint32_t seed, int64_t* picked, uint32_t* count, uint32_t* n_valid, char* scratch,
                    void* stream)
{
    if (n_pix <= 0 || k <= 0 || !depth || !picked || !scratch) return SGR_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    uint32_t* hist1 = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* hist2 = hist1 + PICK_BINS;
    uint32_t* hist3 = hist2 + PICK_BINS;
    PickState* st = reinterpret_cast<PickState*>(hist3 + PICK_BINS);
    uint32_t* blk = reinterpret_cast<uint32_t*>(scratch + sgr_align(3 * (size_t)PICK_BINS * 4 + sizeof(PickState)));
    if (hipMemsetAsync(scratch, 0, 3 * (size_t)PICK_BINS * 4 + sizeof(PickState), s) != hipSuccess) return SGR_E_HIP;
    const int grid = (n_pix + 255) / 256 < 2048 ? (n_pix + 255) / 256 : 2048;
    const int grid1 = (n_pix + 32767) / 32768 < 256 ? (n_pix + 32767) / 32768 : 256;   // ~32 pixels per thread: few flushes of the LDS histogram
    const int blocks = (n_pix + PICK_BLOCK - 1) / PICK_BLOCK;
    hipLaunchKernelGGL(k_pick_hist1, dim3(grid1), dim3(1024), 0, s, n_pix, depth, seed, hist1, st);
    hipLaunchKernelGGL(k_pick_select, dim3(1), dim3(1024), 0, s, hist1, st, 1, (uint32_t)k);
    hipLaunchKernelGGL(k_pick_hist23, dim3(grid), dim3(256), 0, s, n_pix, depth, seed, hist2, st, 2);
    hipLaunchKernelGGL(k_pick_select, dim3(1), dim3(1024), 0, s, hist2, st, 2, (uint32_t)k);
    hipLaunchKernelGGL(k_pick_hist23, dim3(grid), dim3(256), 0, s, n_pix, depth, seed, hist3, st, 3);
    hipLaunchKernelGGL(k_pick_select, dim3(1), dim3(1024), 0, s, hist3, st, 3, (uint32_t)k);
    hipLaunchKernelGGL(k_pick_count, dim3(blocks), dim3(PICK_BLOCK), 0, s, n_pix, depth, seed, st, blk);
    hipLaunchKernelGGL(k_pick_scan, dim3(1), dim3(1024), 0, s, blocks, blk);
    hipLaunchKernelGGL(k_pick_write, dim3(blocks), dim3(PICK_BLOCK), 0, s, n_pix, depth, seed, st, blk, k, picked, count, n_valid);
    hipLaunchKernelGGL(k_pick_pad, dim3((k + 255) / 256), dim3(256), 0, s, k, st, picked);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

int sgr_compact_level_rows(int N, int L, const uint8_t* valid, const uint32_t* n_rows, const float* points, const float* normals,
                           const int64_t* tag_a, const int64_t* tag_b, int64_t* rows_out, float* points_out, float* normals_out,
                           int64_t* tag_a_out, int64_t* tag_b_out, uint32_t* counts, char* scratch, void* stream)
{
    if (N <= 0 || L <= 0 || !valid || !points || !rows_out || !points_out || !counts) return SGR_E_INVALID;
    if ((normals_out && !normals) || (tag_a_out && !tag_a) || (tag_b_out && !tag_b)) return SGR_E_INVALID;
    if (!scratch) return SGR_E_INVALID;
    const int nb = (N + ROWS_BLOCK - 1) / ROWS_BLOCK;
    uint32_t* blk = reinterpret_cast<uint32_t*>(scratch);
    hipLaunchKernelGGL(k_rows_count, dim3(nb, L), dim3(ROWS_BLOCK), 0, (hipStream_t)stream, N, valid, n_rows, blk);
    hipLaunchKernelGGL(k_rows_write, dim3(nb, L), dim3(ROWS_BLOCK), 0, (hipStream_t)stream, N, valid, n_rows, blk, points, normals, tag_a, tag_b,
                       rows_out, points_out, normals_out, tag_a_out, tag_b_out, counts);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

}  // extern "C"
