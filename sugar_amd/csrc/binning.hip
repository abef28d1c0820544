// binning.hip -- depth-ordered tile binning for gfx950.
//
// Replaces the reference's instance-level pipeline
//   cub::DeviceScan::InclusiveSum over P Gaussians       DGR/cuda_rasterizer/rasterizer_impl.cu:277
//   duplicateWithKeys  (64-bit tile|depth key per instance) :70-111
//   cub::DeviceRadixSort::SortPairs over R instances      :303-308   (6 passes x 24 B x R; R ~ 19 P at 1080p)
//   cudaMemset + identifyTileRanges                       :310-317, :116-138
// by sorting the P GAUSSIANS by depth once and then binning them into tiles IN THAT ORDER with a stable counting sort,
// so every tile's list comes out depth-sorted and no per-instance sort exists at all:
//
// (This file holds the depth sort and the SINGLE-LEVEL binning, which is the fallback path; the default two-level binning
//  lives in binning2.hip.)
//   gsort      : stable LSD radix sort (8-bit passes) of (depth_bits, gaussian_id) over the P Gaussians.  Culled
//                Gaussians carry key 0xFFFFFFFF and sink to the end.  Ties keep ascending id.  The digits are taken from
//                key - min(key) (culled: max + 1 - min), and when that range fits 24 bits -- depths within two or three
//                binades, the rule -- the fourth pass is an identity and its kernel returns at once.  One kernel per pass
//                (chunk tickets + look-back through published per-digit counts), one histogram kernel for all passes.
//                The last pass also carries each Gaussian's packed tile rectangle (8 bytes, written by the preprocess
//                kernel) into sorted order: the walks below stream it.
//   bin_count  : workgroup b owns a contiguous slice of the sorted order and a private histogram over all T tiles in LDS;
//                one lane per Gaussian (counting needs no order).
//   hist_scan  : column-wise exclusive scan over the slices of blk_hist[slice][t] (in place) + per-tile totals
//   tile_scan  : exclusive scan of the totals -> tile_start[T+1] (the ranges), R, largest tile count
//   bin_scatter: ONE wave per slice takes the slice's Gaussians in order and spreads each rectangle over its 64 lanes;
//                slot = tile_start[t] + blk_hist[slice][t] + (running LDS counter, returning atomic) and the Gaussian id
//                goes straight into point_list.  Slices are ordered, a wave processes its slice sequentially, and one
//                Gaussian never hits a tile twice, so the per-tile order is exactly the sorted order.
//
// Order contract: the reference's stable radix sort orders a tile's list by depth bits, ties by ascending Gaussian index
// (emission order of duplicateWithKeys).  (depth_bits, id) ascending is the same order -- verified bit-exactly against the
// oracle and against the reference's own code (tests/test_gpu_parity.py, tests/test_gpu_reference.py).
//
// HBM traffic: ~100 B per Gaussian for the sort, 4 B per instance for point_list, 16 B per (workgroup, tile) of histogram
// traffic -- versus ~144 B per INSTANCE for the reference's 6-pass radix sort of 12-byte pairs.  Global atomics only in the
// sort: 1024 digit counters and one ticket per workgroup and pass.
#include "sgr_device.h"
#include <cstdlib>

namespace {

#define WAVE_FENCE()                                          \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// Global stable radix sort of (u32 key, u32 value) pairs, 8 bits per pass.
// ---------------------------------------------------------------------------------------------------------------------
#define LDS_ORDER()                          \
    do {                                     \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_wave_barrier();     \
    } while (0)

#ifndef RS_ITEMS
#define RS_ITEMS 4096  // keys per workgroup chunk (256 threads x 16 keys)
#endif

// sort parameters, reduced from the preprocess kernel's per-workgroup key ranges by the first histogram kernel
struct RsParams { uint32_t kmin, kmax1 /* max + 1: the stand-in of a culled key */, skip3 /* (kmax1 - kmin) < 2^24 */, pad; };

__device__ __forceinline__ uint32_t rs_digit(uint32_t key, uint32_t kmin, uint32_t kmax1, int shift)
{
    return (((key == 0xFFFFFFFFu ? kmax1 : key) - kmin) >> shift) & 255u;
}

// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += y;
    }
    return v;
}

// Chained scan ("decoupled look-back") state of a pass: one word per (chunk, digit), zero = nothing published yet,
// otherwise the count in the low 30 bits and one of two flags: A = this chunk's own count, P = the inclusive sum over this
// chunk and all chunks before it.  Value and flag share a word, so relaxed agent-scope loads and stores are enough.
#define RS_FLAG_A 0x40000000u
#define RS_FLAG_P 0x80000000u
#define RS_VALUE 0x3FFFFFFFu
#define RS_COUNTER_WORDS (4 * 256 + 4)
#ifndef RS_LOOKBACK
#define RS_LOOKBACK 8       // published words a digit's thread requests per round trip of the look-back
#endif  // global digit histograms of the four passes + one chunk ticket per pass

__device__ __forceinline__ uint32_t rs_peek(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rs_post(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Before the passes: reduce the preprocess kernel's per-workgroup key ranges (every workgroup redundantly: 8 bytes per 256
// Gaussians; workgroup 0 publishes the sort parameters), histogram all four digits of every key into the global counters
// (they do not depend on the order of the keys, so one read of the keys serves every pass) and clear the look-back state.
// `counters` is zeroed by the preprocess kernel.
__global__ void __launch_bounds__(256) k_rs_prepare(int n, const uint32_t* __restrict__ keys, int n_chunks, uint32_t* __restrict__ status,
                                                    const uint2* __restrict__ minmax, int n_minmax, RsParams* __restrict__ params,
                                                    uint32_t* __restrict__ counters, int allow_skip)
{
    __shared__ uint32_t s_h[4][256];
    __shared__ uint32_t s_mm[2][4];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int base = 0; base < n_minmax; base += 256 * 16) {  // 16 loads in flight per thread (a plain loop waits for each)
        uint2 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = minmax[min(base + j * 256 + (int)threadIdx.x, n_minmax - 1)];
#pragma unroll
        for (int j = 0; j < 16; j++) { lo = min(lo, v[j].x); hi = max(hi, v[j].y); }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) { s_mm[0][threadIdx.x >> 6] = lo; s_mm[1][threadIdx.x >> 6] = hi; }
    s_h[0][threadIdx.x] = 0; s_h[1][threadIdx.x] = 0; s_h[2][threadIdx.x] = 0; s_h[3][threadIdx.x] = 0;
    __syncthreads();
    lo = min(min(s_mm[0][0], s_mm[0][1]), min(s_mm[0][2], s_mm[0][3]));
    hi = max(max(s_mm[1][0], s_mm[1][1]), max(s_mm[1][2], s_mm[1][3]));
    if (lo > hi) { lo = 0u; hi = 0u; }  // nothing visible
    const uint32_t kmin = lo, kmax1 = hi + 1u;  // (hi < 0x7F800000: a positive float)
    const bool skip3 = allow_skip && ((kmax1 - kmin) >> 24) == 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        RsParams pr; pr.kmin = kmin; pr.kmax1 = kmax1; pr.skip3 = skip3 ? 1u : 0u; pr.pad = 0u;
        *params = pr;
    }
#pragma unroll
    for (int p = 0; p < 4; p++) status[((size_t)p * n_chunks + blockIdx.x) * 256 + threadIdx.x] = 0u;
    const int base = blockIdx.x * RS_ITEMS;
    uint32_t kk[RS_ITEMS / 256];  // (all of the chunk's keys requested before the first is used)
#pragma unroll
    for (int i = 0; i < RS_ITEMS / 256; i++) kk[i] = keys[min(base + i * 256 + (int)threadIdx.x, n - 1)];
#pragma unroll
    for (int i = 0; i < RS_ITEMS / 256; i++) {
        const int k = base + i * 256 + threadIdx.x;
        if (k < n) {
            const uint32_t key = kk[i];
            const uint32_t nk = (key == 0xFFFFFFFFu ? kmax1 : key) - kmin;
            atomicAdd(&s_h[0][nk & 255u], 1u);
            atomicAdd(&s_h[1][(nk >> 8) & 255u], 1u);
            atomicAdd(&s_h[2][(nk >> 16) & 255u], 1u);
            if (!skip3) atomicAdd(&s_h[3][nk >> 24], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t c = s_h[p][threadIdx.x];
        if (c) atomicAdd(&counters[p * 256 + threadIdx.x], c);
    }
}

// One pass = ONE kernel, one 256-thread workgroup per chunk of 4096 keys (chunk = a ticket, so a workgroup only ever waits
// for workgroups that started before it).  Wave w ranks keys [1024 w, 1024 w + 1024) of the chunk 64 at a time IN ORDER
// (rank among equal digits of a group from eight ballots, running per-wave digit counters in LDS) and the four waves'
// counts are combined into the chunk's stable order.  Thread d then owns digit d: it publishes the chunk's count, takes the
// digit's global start from the pass histogram, adds the counts of the earlier chunks by looking back through their
// published words, eight at a time, until it meets an inclusive sum (separate histogram and scan kernels cost two more
// launches and two more trips through HBM per pass: 12 dependent launches for a 1M-key sort that moves 50 MB), and publishes
// its own inclusive sum.  The pairs are permuted into the chunk's order through LDS, and consecutive threads store
// consecutive slots of every digit's run: coalesced runs instead of one transaction per key.  Stable by construction.
#ifndef RS_WAVES
#define RS_WAVES 16  // waves per chunk: a chunk's keys are ranked 64 at a time by each wave, a serial chain per wave
#endif
#define RS_THREADS (64 * RS_WAVES)
__global__ void __launch_bounds__(RS_THREADS) k_rs_pass(int n, const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                 uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int shift,
                                                 int n_chunks, uint32_t* __restrict__ status, uint32_t* __restrict__ counters,
                                                 const uint2* __restrict__ aux_by_val, uint2* __restrict__ aux_out,
                                                 const RsParams* __restrict__ params, int pass,
                                                 uint32_t* __restrict__ keys_out_skip, uint32_t* __restrict__ vals_out_skip)
{
    // pass 2 is the last one when the key range fits 24 bits: it then writes where pass 3 would have (and carries the tile
    // rectangles); pass 3 returns at once
    const RsParams pr = *params;
    if (pass == 3 && pr.skip3) return;
    if (pass == 2) {
        if (pr.skip3) { keys_out = keys_out_skip; vals_out = vals_out_skip; }
        else aux_out = nullptr;
    }
    const uint32_t kmin = pr.kmin, kmax1 = pr.kmax1;
    __shared__ uint32_t s_cnt[RS_WAVES][256];   // per-wave digit counts, then per-wave first slot (chunk-local)
    __shared__ uint32_t s_loc[256];      // chunk-local first slot of every digit
    __shared__ uint32_t s_base[256];     // global first slot of every digit's run for this chunk
    __shared__ uint32_t s_scan[8];
    __shared__ uint32_t s_k[RS_ITEMS], s_v[RS_ITEMS];
    __shared__ int s_chunk;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_chunk = (int)atomicAdd(&counters[4 * 256 + pass], 1u);
    for (int i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&s_cnt[0][0])[i] = 0u;
    __syncthreads();
    const int chunk = s_chunk;
    const int base = chunk * RS_ITEMS;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    constexpr int G = RS_ITEMS / RS_THREADS;  // groups of 64 keys per wave
    uint32_t rk[G], rv[G];
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int k = base + wave * (RS_ITEMS / RS_WAVES) + i * 64 + lane;
        rk[i] = (k < n) ? keys_in[k] : 0xFFFFFFFFu;
        rv[i] = (k < n) ? (vals_in ? vals_in[k] : (uint32_t)k) : 0u;
    }
    uint32_t lrank[G];  // rank of the key among the keys of ITS WAVE with the same digit
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int k = base + wave * (RS_ITEMS / RS_WAVES) + i * 64 + lane;
        const bool live = k < n;
        const uint32_t digit = rs_digit(rk[i], kmin, kmax1, shift);
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t prev = s_cnt[wave][digit];
        lrank[i] = prev + (uint32_t)__popcll(same & below);
        WAVE_FENCE();
        if (live && (same & below) == 0ull) s_cnt[wave][digit] = prev + (uint32_t)__popcll(same);
        WAVE_FENCE();
    }
    __syncthreads();
    // threads 0..255: thread d owns digit d (the first four waves; the others only keep the barriers company)
    const bool owner = tid < 256;
    uint32_t mine = 0, tot = 0, i_mine = 0, i_tot = 0;
    uint32_t* st = status + (size_t)pass * n_chunks * 256 + (owner ? tid : 0);  // this pass, digit d: word of chunk k at st[k * 256]
    if (owner) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) mine += s_cnt[w][tid];
        rs_post(st + (size_t)chunk * 256, mine | (chunk == 0 ? RS_FLAG_P : RS_FLAG_A));
        tot = counters[pass * 256 + tid];
        // two exclusive scans over the 256 digits: the chunk's counts (chunk-local starts) and the global totals (run
        // starts): shuffle scans inside the four waves, then the earlier waves' totals (one barrier instead of 32)
        i_mine = wave_incl_scan(mine); i_tot = wave_incl_scan(tot);
        if (lane == 63) { s_scan[wave] = i_mine; s_scan[4 + wave] = i_tot; }
    }
    __syncthreads();
    if (owner) {
        uint32_t loc = i_mine - mine, gbase = i_tot - tot;
#pragma unroll
        for (int w = 0; w < 3; w++)
            if (w < wave) { loc += s_scan[w]; gbase += s_scan[4 + w]; }
        // look back: keys of this digit in the chunks before this one
        uint32_t prefix = 0;
        int k = chunk - 1;
        bool done = k < 0;
        while (!done) {
            uint32_t v[RS_LOOKBACK];
#pragma unroll
            for (int j = 0; j < RS_LOOKBACK; j++) v[j] = (k - j >= 0) ? rs_peek(st + (size_t)(k - j) * 256) : RS_FLAG_P;
            bool stalled = false;
            int adv = 0;
#pragma unroll
            for (int j = 0; j < RS_LOOKBACK; j++) {
                if (!done && !stalled) {
                    if (v[j] == 0u) stalled = true;  // not published yet: ask again from here
                    else { prefix += v[j] & RS_VALUE; adv++; done = (v[j] & RS_FLAG_P) != 0u; }
                }
            }
            k -= adv;
        }
        if (chunk != 0) rs_post(st + (size_t)chunk * 256, (prefix + mine) | RS_FLAG_P);
        s_loc[tid] = loc;
        s_base[tid] = gbase + prefix;
        uint32_t run = loc;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int k = base + wave * (RS_ITEMS / RS_WAVES) + i * 64 + lane;
        if (k < n) {
            const uint32_t digit = rs_digit(rk[i], kmin, kmax1, shift);
            const uint32_t pos = s_cnt[wave][digit] + lrank[i];
            s_k[pos] = rk[i]; s_v[pos] = rv[i];
        }
    }
    __syncthreads();
    const int live_n = min(RS_ITEMS, n - base);
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int j = i * RS_THREADS + tid;
        if (j < live_n) {
            const uint32_t key = s_k[j], val = s_v[j];
            const uint32_t digit = rs_digit(key, kmin, kmax1, shift);
            const uint32_t dst = s_base[digit] + ((uint32_t)j - s_loc[digit]);
            keys_out[dst] = key;
            vals_out[dst] = val;
            if (aux_out) aux_out[dst] = aux_by_val[val];  // last pass: the tile rectangles in sorted order
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ordered binning (count and scatter share one walk).
// ---------------------------------------------------------------------------------------------------------------------
// Tile rectangles in sorted order, packed {minx | miny << 16, w | h << 16} (w == 0: culled), so the ordered walks stream
// 8 contiguous bytes per Gaussian instead of chasing order[] -> rec[] through two dependent random loads.
// Counting needs no order: workgroup b histograms the tiles of its slice of the depth order with one lane per Gaussian
// (LDS atomics), 16 waves per CU.
__global__ void __launch_bounds__(256) k_bin_count(int P, int gx, int gy, int per_slice, const uint2* __restrict__ rects,
                                                   uint32_t* __restrict__ blk_hist)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];
    const int T = gx * gy;
    const int tid = threadIdx.x;
    for (int t = tid; t < T; t += 256) s_cnt[t] = 0u;
    __syncthreads();
    const int begin = blockIdx.x * per_slice;
    const int end = min(P, begin + per_slice);
    for (int s = begin + tid; s < end; s += 256) {
        const uint2 r = rects[s];
        const int w = (int)(r.y & 0xFFFFu), h = (int)(r.y >> 16);
        const int minx = (int)(r.x & 0xFFFFu), miny = (int)(r.x >> 16);
        for (int y = 0; y < h; y++) {
            uint32_t* rowp = s_cnt + (miny + y) * gx + minx;
            for (int x = 0; x < w; x++) atomicAdd(&rowp[x], 1u);
        }
    }
    __syncthreads();
    uint32_t* row = blk_hist + (size_t)blockIdx.x * T;
    for (int t = tid; t < T; t += 256) row[t] = s_cnt[t];
}

// The scatter must hand out a tile's slots in depth order: ONE wave per slice takes the slice's Gaussians in order and
// spreads each rectangle over its 64 lanes (returning LDS atomics on distinct tiles); four Gaussians are in flight per
// step (LDS operations of one wave execute in program order, so the order of the atomics is the order of the Gaussians).
__global__ void __launch_bounds__(64) k_bin_scatter(int P, int gx, int gy, int per_slice, const uint32_t* __restrict__ order,
                                                    const uint2* __restrict__ rects, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ blk_hist, uint32_t* __restrict__ point_list)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];
    const int T = gx * gy;
    const int lane = threadIdx.x;
    const uint32_t* row = blk_hist + (size_t)blockIdx.x * T;
    for (int t = lane; t < T; t += 64) s_cnt[t] = tile_start[t] + row[t];
    WAVE_FENCE();
    const int begin = blockIdx.x * per_slice;
    const int end = min(P, begin + per_slice);
    for (int base = begin; base < end; base += 64) {
        const int s = base + lane;
        int id = 0, t0 = 0, w = 1, n = 0;
        float inv_w = 1.0f;
        if (s < end) {
            const uint2 r = rects[s];
            const int rw = (int)(r.y & 0xFFFFu);
            if (rw > 0) {
                w = rw;
                n = rw * (int)(r.y >> 16);
                t0 = (int)(r.x >> 16) * gx + (int)(r.x & 0xFFFFu);
                id = (int)order[s];
                inv_w = __builtin_amdgcn_rcpf((float)rw);  // k / w below is exact for k < 2^20 with a 1-ulp reciprocal
            }
        }
        unsigned long long todo = __ballot(n > 0);
        while (todo) {
            int jn[4], jw[4], jt[4];
            float jinv[4];
            uint32_t jid[4];
            bool big = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (todo) {
                    const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                    todo &= todo - 1;
                    jn[u] = __builtin_amdgcn_readlane(n, j);
                    jw[u] = __builtin_amdgcn_readlane(w, j);
                    jt[u] = __builtin_amdgcn_readlane(t0, j);
                    jinv[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_w), j));
                    jid[u] = (uint32_t)__builtin_amdgcn_readlane(id, j);
                    big = big || jn[u] > 64;
                } else {
                    jn[u] = 0; jw[u] = 1; jt[u] = 0; jid[u] = 0u; jinv[u] = 1.0f;
                }
            }
            if (!big) {
                uint32_t slot[4];
                int tt[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int ty = (int)(((float)lane + 0.5f) * jinv[u]);
                    tt[u] = jt[u] + ty * gx + (lane - ty * jw[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (lane < jn[u]) slot[u] = atomicAdd(&s_cnt[tt[u]], 1u);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (lane < jn[u]) point_list[slot[u]] = jid[u];
            } else {
                // a rectangle larger than 64 tiles: strictly one Gaussian at a time
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    for (int k = lane; k < jn[u]; k += 64) {
                        const int ty = (int)(((float)k + 0.5f) * jinv[u]);
                        const int t = jt[u] + ty * gx + (k - ty * jw[u]);
                        const uint32_t sl = atomicAdd(&s_cnt[t], 1u);
                        point_list[sl] = jid[u];
                    }
                    WAVE_FENCE();
                }
            }
            WAVE_FENCE();
        }
    }
}

// ---- column scan of the per-workgroup histograms --------------------------------------------
// One lane per tile walks the n_blocks rows (coalesced across tiles): blk_hist[b][t] becomes the exclusive prefix over b,
// tile_count[t] the total.  Loads are issued UNROLL at a time so each lane keeps several rows in flight.
__global__ void __launch_bounds__(64) k_hist_scan(int T, int n_blocks, uint32_t* __restrict__ blk_hist,
                                                  uint32_t* __restrict__ tile_count)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= T) return;
    constexpr int UNROLL = 32;
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

// ---- tile scan: one 1024-thread workgroup, T <= a few 10^5 -----------------------------------
// One workgroup: tile ranges (exclusive scan of the tile counts), R, the largest count; also clears the two per-tile maxima
// the blend kernels combine into.  Everything a lane touches in HBM is next to what its neighbours touch: the tiles are
// taken 8192 at a time as 8 rows of 1024 (thread t: tiles t, t + 1024, ...), each 64-tile segment is scanned inside its wave
// with shuffles, the 128 segment totals by wave 0, and the loads of a round go out as one batch.  (With thread t owning 8
// CONSECUTIVE tiles the one CU this runs on spent 17 us mostly on 32-byte-strided stores.)
__global__ void __launch_bounds__(1024) k_tile_scan(int T, const uint32_t* __restrict__ tile_count,
                                                    uint32_t* __restrict__ tile_start, uint32_t* __restrict__ header,
                                                    uint32_t* __restrict__ tile_maxc, uint32_t* __restrict__ tile_walked,
                                                    int clear_b2_words, uint32_t* __restrict__ host_a, uint32_t* __restrict__ host_b,
                                                    uint32_t* __restrict__ repair_flag)
{
    __shared__ uint32_t s_seg[128];
    __shared__ uint32_t s_max[16];
    __shared__ uint32_t s_total;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t carry = 0, mx = 0;
    unsigned long long exact = 0ull;  // this thread's share of the total in 64 bits: the 32-bit scan wraps at 2^32 instances
    for (int base = 0; base < T; base += 8192) {
        uint32_t c[8], incl[8];
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = tile_count[min(base + j * 1024 + tid, T - 1)];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (base + j * 1024 + tid >= T) c[j] = 0u;
            exact += c[j];
            mx = max(mx, c[j]);
            incl[j] = wave_incl_scan(c[j]);
            if (lane == 63) s_seg[j * 16 + wave] = incl[j];
        }
        __syncthreads();
        if (wave == 0) {  // exclusive scan of the 128 segment totals, two per lane
            const uint32_t a0 = s_seg[2 * lane], a1 = s_seg[2 * lane + 1];
            const uint32_t p = wave_incl_scan(a0 + a1);
            s_seg[2 * lane] = p - a0 - a1;
            s_seg[2 * lane + 1] = p - a1;
            if (lane == 63) s_total = p;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = base + j * 1024 + tid;
            if (i < T) {
                tile_start[i] = carry + s_seg[j * 16 + wave] + incl[j] - c[j]; tile_maxc[i] = 0u; tile_walked[i] = 0u;
                if (repair_flag) repair_flag[i] = 0u;
            }
        }
        carry += s_total;
        __syncthreads();  // s_seg / s_total are rewritten by the next round
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    for (int o = 32; o > 0; o >>= 1) exact += __shfl_xor(exact, o);
    __shared__ unsigned long long s_exact[16];
    if (lane == 0) { s_max[wave] = mx; s_exact[wave] = exact; }
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        unsigned long long total = 0ull;
        for (int w = 0; w < 16; w++) { m = max(m, s_max[w]); total += s_exact[w]; }
        // 2^32 - 1 or more instances: the word SATURATES (the wrapped sum could pass for a small, valid count); no capacity is
        // ever that large (sgr_forward_ex), so such a forward is invalid for every reader of the header
        if (total >= 0xFFFFFFFFull) carry = 0xFFFFFFFFu;
        tile_start[T] = carry;
        header[SGR_HDR_R] = carry;
        header[SGR_HDR_R_HI] = (uint32_t)(total >> 32);
        header[SGR_HDR_MAXCOUNT] = m;
        header[SGR_HDR_HINT_MISS] = 0;
        if (clear_b2_words) { header[4] = 0; header[5] = 0; header[6] = 0; }  // single-level path: k_sup_scan did not run
        header[7] = 0;
        header[SGR_HDR_DEEP] = 0;
        // the header for the host, written straight into its pinned memory (a 32-byte copy command of its own cost the stream
        // 8 us, twice per forward); words 4-6 were left by k_sup_scan, an earlier kernel on this stream
        const uint32_t w4 = clear_b2_words ? 0u : header[4], w5 = clear_b2_words ? 0u : header[5], w6 = clear_b2_words ? 0u : header[6];
        for (int k = 0; k < 2; k++) {
            uint32_t* h = k ? host_b : host_a;
            if (!h) continue;
            h[0] = carry; h[1] = m; h[2] = (uint32_t)(total >> 32); h[3] = 0; h[4] = w4; h[5] = w5; h[6] = w6; h[7] = 0;
        }
    }
}

}  // namespace

// sort scratch: [ keys_a | keys_b | vals_a | vals_b | look-back state | counters | rects (sorted) | rects (by id) | keys_c |
//                 vals_c | key ranges per preprocess workgroup | parameters ]
static size_t sort_state_bytes(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    const size_t chunks = (n + RS_ITEMS - 1) / RS_ITEMS;
    return sgr_align(chunks * 256 * 4 * 4) + sgr_align(RS_COUNTER_WORDS * 4);
}

static size_t sort_base_bytes(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    return sgr_align(n * 4) * 4 + sort_state_bytes(P) + 2 * sgr_align(n * 8);  // ... + packed rectangles (sorted, by id)
}

size_t sgr_sort_minmax_offset(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    return sort_base_bytes(P) + 2 * sgr_align(n * 4);
}

size_t sgr_sort_scratch_bytes(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    return sgr_sort_minmax_offset(P) + sgr_align((n + 255) / 256 * 8) + 256;
}

size_t sgr_sort_rect_by_id_offset(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    return sgr_sort_rects_offset(P) + sgr_align(n * 8);
}

size_t sgr_sort_rects_offset(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    return sgr_align(n * 4) * 4 + sort_state_bytes(P);
}

// the RS_COUNTER_WORDS words the preprocess kernel zeroes for the sort that follows it
size_t sgr_sort_counters_offset(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    const size_t chunks = (n + RS_ITEMS - 1) / RS_ITEMS;
    return sgr_align(n * 4) * 4 + sgr_align(chunks * 256 * 4 * 4);
}
int sgr_sort_counter_words() { return RS_COUNTER_WORDS; }

// keys_a (the first array of sort_scratch) must hold the keys and the key ranges theirs, both written by the preprocess
// kernel, which also zeroes the counters; on return *order_out points at the sorted Gaussian ids (inside sort_scratch).
// Buffers: pass 0 a -> b, pass 1 b -> c, pass 2 c -> b (or -> a when it is the last one), pass 3 b -> a.
void sgr_launch_gaussian_sort(int P, char* sort_scratch, const uint32_t** order_out, const uint2* rect_by_id, uint2* rects_sorted,
                              hipStream_t s, int part)
{
    const size_t n = (size_t)P;
    const size_t arr = sgr_align(n * 4);
    uint32_t* keys_a = reinterpret_cast<uint32_t*>(sort_scratch);
    uint32_t* keys_b = reinterpret_cast<uint32_t*>(sort_scratch + arr);
    uint32_t* vals_a = reinterpret_cast<uint32_t*>(sort_scratch + 2 * arr);
    uint32_t* vals_b = reinterpret_cast<uint32_t*>(sort_scratch + 3 * arr);
    uint32_t* status = reinterpret_cast<uint32_t*>(sort_scratch + 4 * arr);
    const int chunks = (P + RS_ITEMS - 1) / RS_ITEMS;
    uint32_t* counters = reinterpret_cast<uint32_t*>(sort_scratch + sgr_sort_counters_offset(P));
    uint32_t* keys_c = reinterpret_cast<uint32_t*>(sort_scratch + sort_base_bytes(P));
    uint32_t* vals_c = reinterpret_cast<uint32_t*>(sort_scratch + sort_base_bytes(P) + arr);
    const uint2* minmax = reinterpret_cast<const uint2*>(sort_scratch + sgr_sort_minmax_offset(P));
    const int n_minmax = (P + 255) / 256;
    RsParams* params = reinterpret_cast<RsParams*>(sort_scratch + sgr_sort_minmax_offset(P) + sgr_align((size_t)n_minmax * 8));
    const uint32_t* kin[4] = {keys_a, keys_b, keys_c, keys_b};
    const uint32_t* vin[4] = {nullptr, vals_b, vals_c, vals_b};
    uint32_t* kout[4] = {keys_b, keys_c, keys_b, keys_a};
    uint32_t* vout[4] = {vals_b, vals_c, vals_b, vals_a};
    static const int allow_skip = getenv("SGR_SORT_FOUR_PASSES") ? 0 : 1;  // (development: always run the fourth pass)
    if (part != 2)
        hipLaunchKernelGGL(k_rs_prepare, dim3(chunks), dim3(256), 0, s, P, keys_a, chunks, status, minmax, n_minmax, params, counters, allow_skip);
    for (int pass = (part == 2 ? 2 : 0); pass < (part == 1 ? 2 : 4); pass++)
        hipLaunchKernelGGL(k_rs_pass, dim3(chunks), dim3(RS_THREADS), 0, s, P, kin[pass], vin[pass], kout[pass], vout[pass], 8 * pass, chunks,
                           status, counters, rect_by_id, pass >= 2 ? rects_sorted : (uint2*)nullptr, params, pass, keys_a, vals_a);
    *order_out = vals_a;
}

static void set_lds_limit(const void* fn, size_t bytes, size_t& configured)
{
    if (bytes > configured) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        configured = bytes;
    }
}

void sgr_launch_bin_count(int P, int gx, int gy, int n_slices, int per_slice, const uint32_t* order, const uint2* rects,
                          uint32_t* blk_hist, hipStream_t s)
{
    (void)order;
    static size_t configured = 0;
    const size_t bytes = (size_t)gx * gy * 4;
    set_lds_limit(reinterpret_cast<const void*>(&k_bin_count), bytes, configured);
    hipLaunchKernelGGL(k_bin_count, dim3(n_slices), dim3(256), bytes, s, P, gx, gy, per_slice, rects, blk_hist);
}

void sgr_launch_bin_scatter(int P, int gx, int gy, int n_slices, int per_slice, const uint32_t* order, const uint2* rects,
                            const uint32_t* tile_start, uint32_t* blk_hist, uint32_t* point_list, hipStream_t s)
{
    static size_t configured = 0;
    const size_t bytes = (size_t)gx * gy * 4;
    set_lds_limit(reinterpret_cast<const void*>(&k_bin_scatter), bytes, configured);
    hipLaunchKernelGGL(k_bin_scatter, dim3(n_slices), dim3(64), bytes, s, P, gx, gy, per_slice, order, rects, tile_start, blk_hist,
                       point_list);
}

void sgr_launch_hist_scan(int T, int n_blocks, uint32_t* blk_hist, uint32_t* tile_count, hipStream_t s)
{
    hipLaunchKernelGGL(k_hist_scan, dim3((T + 63) / 64), dim3(64), 0, s, T, n_blocks, blk_hist, tile_count);
}

void sgr_launch_tile_scan(int T, const uint32_t* tile_count, uint32_t* tile_start, uint32_t* header, uint32_t* tile_maxc,
                          uint32_t* tile_walked, int clear_b2_words, uint32_t* host_a, uint32_t* host_b, hipStream_t s, uint32_t* repair_flag)
{
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, T, tile_count, tile_start, header, tile_maxc, tile_walked,
                       clear_b2_words, host_a, host_b, repair_flag);
}
