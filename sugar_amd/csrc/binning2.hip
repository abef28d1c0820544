// binning2.hip -- two-level depth-ordered tile binning for gfx950 (the default path; binning.hip's single-level scatter
// stays as the fallback for scenes whose Gaussians cover more super-tiles than the level-1 list can hold).
//
// Same contract as binning.hip (DGR/cuda_rasterizer/rasterizer_impl.cu:70-138, 277-317 replaced): every tile's list holds
// the Gaussians whose rectangle covers the tile, in (depth bits, id) order -- bit-identical lists and ranges.
//
// Why two levels.  The single-level scatter keeps one open write stream per (slice, tile): 1024 waves x 8160 tiles at
// 1080p.  No cache can hold that many partially written lines, so the 74 MB of list entries cost ~570 MB of HBM writes
// (rocprofv3 WRITE_SIZE) and the kernel is latency-bound with one wave per SIMD.  Here a Gaussian is first appended, in
// depth order, to the lists of the 8x8-tile SUPER-TILES its rectangle touches (135 bins at 1080p, ~2 entries per
// Gaussian: the ordered scatter is cheap and runs 2048 waves), and each super-tile list is then cut into chunks of 512
// entries.  A workgroup per chunk walks its entries in order with lane = entry, one ballot per tile of the super-tile, and
// appends the ids to the tiles' lists: sequential write streams, full lines, no atomics.
//
//   sup_count   : slice b of the depth order histograms its super-tiles in LDS (one lane per Gaussian)
//   sup_hist_scan: exclusive scan over slices, per-super-tile totals
//   sup_scan    : super-tile list starts, chunk table (chunk_base[s] = first chunk of super-tile s), capacity check
//   sup_scatter : one wave per slice appends the sorted positions of its Gaussians to the super-tile lists, in order
//   tile_pass<0>: per chunk, per tile of the super-tile: number of covering entries  -> cnt2[chunk][64]
//   tile_scan2  : per super-tile, exclusive scan of cnt2 over its chunks (in place) -> per-tile totals
//   tile_scan   : (binning.hip) ranges, R
//   tile_pass<1>: per chunk, the same walk again, now storing ids at tile_start[t] + cnt2[chunk][t] + running count
#include "sgr_device.h"

namespace {

#define LDS_ORDER()                          \
    do {                                     \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_wave_barrier();     \
    } while (0)

#define SGR_B2_MAXN 16  // super-tiles per Gaussian handled by the lane-parallel level-1 path

#define WAVE_FENCE()                                          \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)

// rectangle of a packed tile rectangle in super-tile units; returns false for a culled Gaussian
__device__ __forceinline__ bool sup_rect(uint2 r, int& sx0, int& sy0, int& w1, int& h1)
{
    const int w = (int)(r.y & 0xFFFFu), h = (int)(r.y >> 16);
    if (w == 0) return false;
    const int minx = (int)(r.x & 0xFFFFu), miny = (int)(r.x >> 16);
    sx0 = minx >> SGR_SUP_SHIFT; sy0 = miny >> SGR_SUP_SHIFT;
    w1 = ((minx + w - 1) >> SGR_SUP_SHIFT) - sx0 + 1;
    h1 = ((miny + h - 1) >> SGR_SUP_SHIFT) - sy0 + 1;
    return true;
}

__global__ void __launch_bounds__(256) k_sup_count(int P, int sgx, int T1, int per_slice, const uint2* __restrict__ rects,
                                                   uint32_t* __restrict__ hist1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];
    const int tid = threadIdx.x;
    for (int t = tid; t < T1; t += 256) s_cnt[t] = 0u;
    __syncthreads();
    const int begin = blockIdx.x * per_slice;
    const int end = min(P, begin + per_slice);
    for (int s = begin + tid; s < end; s += 256) {
        int sx0, sy0, w1, h1;
        if (!sup_rect(rects[s], sx0, sy0, w1, h1)) continue;
        for (int y = 0; y < h1; y++)
            for (int x = 0; x < w1; x++) atomicAdd(&s_cnt[(sy0 + y) * sgx + sx0 + x], 1u);
    }
    __syncthreads();
    uint32_t* row = hist1 + (size_t)blockIdx.x * T1;
    for (int t = tid; t < T1; t += 256) row[t] = s_cnt[t];
}

// one 1024-thread workgroup: list starts, chunk table, totals
__global__ void __launch_bounds__(1024) k_sup_scan(int T1, uint32_t cap1, uint32_t chunk_cap, const uint32_t* __restrict__ sup_count,
                                                   uint32_t* __restrict__ sup_start, uint32_t* __restrict__ chunk_base,
                                                   uint4* __restrict__ chunk_info, uint32_t* __restrict__ hdr)
{
    __shared__ uint32_t s_a[1024], s_b[1024];
    const int tid = threadIdx.x;
    const int per = (T1 + 1023) / 1024;
    const int b = tid * per, e = min(T1, b + per);
    uint32_t sa = 0, sb = 0;
    for (int i = b; i < e; i++) { const uint32_t c = sup_count[i]; sa += c; sb += (c + SGR_B2_CHUNK - 1) / SGR_B2_CHUNK; }
    s_a[tid] = sa; s_b[tid] = sb;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t va = (tid >= o) ? s_a[tid - o] : 0, vb = (tid >= o) ? s_b[tid - o] : 0;
        __syncthreads();
        s_a[tid] += va; s_b[tid] += vb;
        __syncthreads();
    }
    uint32_t ra = s_a[tid] - sa, rb = s_b[tid] - sb;
    for (int i = b; i < e; i++) {
        const uint32_t c = sup_count[i];
        sup_start[i] = ra; chunk_base[i] = rb;
        const uint32_t nc = (c + SGR_B2_CHUNK - 1) / SGR_B2_CHUNK;
        // chunk -> {super-tile, first level-1 entry, entries}: ONE 16-byte load per chunk in the tile passes (the chain
        // chunk -> super-tile -> list start / first chunk -> entry range was three dependent round trips per wave)
        for (uint32_t k = 0; k < nc && rb + k < chunk_cap; k++)
            chunk_info[rb + k] = make_uint4((uint32_t)i, ra + k * SGR_B2_CHUNK, min((uint32_t)SGR_B2_CHUNK, c - k * SGR_B2_CHUNK), 0u);
        ra += c; rb += nc;
    }
    if (tid == 1023) {
        const uint32_t R1 = s_a[1023], nch = s_b[1023];
        sup_start[T1] = R1; chunk_base[T1] = nch;
        hdr[SGR_B2_HDR_R1] = R1;
        hdr[SGR_B2_HDR_CHUNKS] = nch;
        hdr[SGR_B2_HDR_OVERFLOW] = (R1 > cap1 || nch > chunk_cap) ? 1u : 0u;
    }
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)  // set bits of m in lanes below this one
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Ordered append to the super-tile lists: ONE wave per slice walks its Gaussians in depth order, 64 per step with one
// lane per Gaussian.  The (Gaussian, super-tile) pairs of a step are laid out in LDS ordered by Gaussian (a rectangle
// touches one to four super-tiles unless the splat is wider than 128 px), then taken 64 at a time: pairs with the same
// super-tile find each other with one ballot per key bit, the first of them reserves the group's slots with ONE returning
// LDS atomic and every pair stores at base + its rank.  (A step with a splat over more than SGR_B2_MAXN super-tiles is
// taken one Gaussian at a time.)  Equal keys inside a batch are ranked by pair index = depth order,
// batches are sequential: a super-tile's list comes out in depth order.  The entry is the Gaussian's position in the
// depth order (rects[] and order[] are indexed by it).
__global__ void __launch_bounds__(64) k_sup_scatter(int P, int sgx, int T1, int key_bits, int per_slice,
                                                    const uint2* __restrict__ rects, const uint32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ sup_start,
                                                    const uint32_t* __restrict__ hist1, const uint32_t* __restrict__ hdr,
                                                    uint4* __restrict__ L1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];  // [T1] running slots, then [64 * SGR_B2_MAXN] pairs
    if (hdr[SGR_B2_HDR_OVERFLOW]) return;
    uint32_t* s_pair = s_cnt + T1;
    const int lane = threadIdx.x;
    const uint32_t* row = hist1 + (size_t)blockIdx.x * T1;
    for (int tb = 0; tb < T1; tb += 256) {  // (four loads of each table in flight: a plain loop waits for every pair)
        uint32_t a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int t = min(tb + 64 * j + lane, T1 - 1); a[j] = sup_start[t]; b[j] = row[t]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { const int t = tb + 64 * j + lane; if (t < T1) s_cnt[t] = a[j] + b[j]; }
    }
    WAVE_FENCE();
    const int begin = blockIdx.x * per_slice;
    const int end = min(P, begin + per_slice);
    // (the next step's rectangle and id travel while this step is scattered)
    uint2 r_next = rects[min(begin + lane, P - 1)];
    uint32_t id_next = order[min(begin + lane, P - 1)];
    for (int base = begin; base < end; base += 64) {
        const int s = base + lane;
        const uint2 r_cur = r_next;
        const uint32_t id_cur = id_next;
        r_next = rects[min(s + 64, P - 1)];
        id_next = order[min(s + 64, P - 1)];
        int t0 = 0, w = 1, n = 0;
        if (s < end) {
            int sx0, sy0, w1, h1;
            if (sup_rect(r_cur, sx0, sy0, w1, h1)) { w = w1; n = w1 * h1; t0 = sy0 * sgx + sx0; }
        }
        if (__ballot(n > SGR_B2_MAXN) == 0ull) {
            // exclusive prefix of n over the lanes: the pairs of this step, ordered by Gaussian
            int incl = n;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(incl, d);
                if (lane >= d) incl += y;
            }
            const uint32_t off = (uint32_t)(incl - n);
            const int m = __builtin_amdgcn_readlane(incl, 63);
            const float inv_w = 1.0f / (float)w;
            for (int p = 0; p < n; p++) {
                const int ty = (int)(((float)p + 0.5f) * inv_w);  // p / w, exact for these sizes
                s_pair[off + p] = ((uint32_t)(t0 + ty * sgx + (p - ty * w)) << 8) | (uint32_t)lane;
            }
            WAVE_FENCE();
            for (int q = 0; q < m; q += 64) {
                const bool on = q + lane < m;
                const uint32_t v = on ? s_pair[q + lane] : 0u;
                const uint32_t key = v >> 8;
                unsigned long long mm = __ballot(on);
                for (int b = 0; b < key_bits; b++) {
                    const unsigned long long bb = __ballot(on && ((key >> b) & 1u));
                    mm &= ((key >> b) & 1u) ? bb : ~bb;
                }
                const uint32_t rank = lanes_below(mm);
                uint32_t slot = 0;
                if (on && rank == 0) slot = atomicAdd(&s_cnt[key], (uint32_t)__popcll(mm));
                const int leader = on ? (int)__builtin_ctzll(mm) : lane;
                slot = (uint32_t)__shfl((int)slot, leader);
                // the entry is self-contained -- {packed rectangle, Gaussian id} of the lane that owns the pair -- so the tile
                // passes stream 16 bytes per entry instead of chasing position -> rectangle -> id through two gathers (4- and
                // 8-byte reads of 64-byte sectors: 330 MiB fetched for 40 MB of data, rocprofv3 FETCH_SIZE)
                const int owner = (int)(v & 0xFFu);
                const uint4 ent = make_uint4((uint32_t)__shfl((int)r_cur.x, owner), (uint32_t)__shfl((int)r_cur.y, owner),
                                             (uint32_t)__shfl((int)id_cur, owner), 0u);
                if (on) L1[slot + rank] = ent;
                asm volatile("" ::: "memory");  // keep the batches' atomics in program order (compiler only)
            }
            WAVE_FENCE();  // s_pair is rewritten by the next step
        } else {
            // a huge splat: strictly one Gaussian at a time for this step
            unsigned long long todo = __ballot(n > 0);
            while (todo) {
                const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                todo &= todo - 1;
                const int jn = __builtin_amdgcn_readlane(n, j), jw = __builtin_amdgcn_readlane(w, j);
                const int jt = __builtin_amdgcn_readlane(t0, j);
                const uint4 ent = make_uint4((uint32_t)__builtin_amdgcn_readlane((int)r_cur.x, j), (uint32_t)__builtin_amdgcn_readlane((int)r_cur.y, j),
                                             (uint32_t)__builtin_amdgcn_readlane((int)id_cur, j), 0u);
                for (int k = lane; k < jn; k += 64) {
                    const int ty = k / jw;
                    const uint32_t sl = atomicAdd(&s_cnt[jt + ty * sgx + (k - ty * jw)], 1u);
                    L1[sl] = ent;
                }
                asm volatile("" ::: "memory");
            }
        }
    }
}

// Per chunk of a super-tile list (<= 512 entries = 8 batches of one entry per lane): a lane turns the rectangle
// of its entry into the 64-bit mask of the super-tile's 8 x 8 tiles it covers (a dozen instructions for 64 entries).
// Then, per tile t, one ballot of bit t per batch gives the entries that cover it, in list order:
//   WRITE = false: the popcounts are the tile's count for this chunk (kept in lane t of one register);
//   WRITE = true : the covering lanes drop their ids into an LDS row at (running count + number of covering lanes below),
//                  i.e. the row is the tile's list segment of this chunk, and the wave then copies it out 64 ids per
//                  store instruction -- contiguous, coalesced, no atomics.  (One global store per (batch, tile) with ten
//                  active lanes kept the CU's address unit busy for 24 cycles each: 4x the instructions.)
__device__ __forceinline__ void tile_mask(uint2 r, int ox, int oy, uint32_t& lo, uint32_t& hi)
{
    const int minx = (int)(r.x & 0xFFFFu), miny = (int)(r.x >> 16);
    const int x0 = max(minx - ox, 0), x1 = min(minx + (int)(r.y & 0xFFFFu) - ox, SGR_SUP);
    const int y0 = max(miny - oy, 0), y1 = min(miny + (int)(r.y >> 16) - oy, SGR_SUP);
    const uint32_t xb = ((1u << (x1 - x0)) - 1u) << x0;  // columns, 8 bits
    const uint32_t yb = ((1u << (y1 - y0)) - 1u) << y0;  // rows, 8 bits
    const uint32_t rows4 = xb * 0x01010101u;             // the column pattern in four rows
    // row bit r -> byte r all ones: spread the four bits to the byte lsbs, then * 0xFF
    lo = rows4 & ((((yb & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu);
    hi = rows4 & ((((yb >> 4) * 0x00204081u) & 0x01010101u) * 0xFFu);
}

#define SGR_B2_BATCHES (SGR_B2_CHUNK / 64)

// SEVERAL waves per chunk (one workgroup), an equal share of the super-tile's 64 tiles each.  One wave walking all 64 tiles of a
// chunk is a serial chain of ~5000 instructions (64 tiles x 8 batches of ballot / rank / LDS store): ~40 us by itself, however
// few chunks there are -- the walk hint below skips four chunks in five at the metric workload and the pass did not get
// any shorter until the chain was cut up.  Every wave loads the chunk's 512 entries and turns them into tile masks (a
// dozen instructions per batch, repeated by every wave) and then keeps the half of the masks its tiles live in.
// How many waves: measured at the metric workload (3661 chunks) -- count pass 28 / 21 / 23 / 29 / 44 us with 1 / 2 / 4 / 8 / 16
// waves per chunk (its chain is short, the repeated loads cost more); write pass without a hint 52 / 48 / 45 / 46 / 62 us and
// with nearly every chunk skipped 40 / 24 / 17 / 14 / 18 us.
#define SGR_B2_COUNT_WAVES 2
#define SGR_B2_WRITE_WAVES 8
template <bool WRITE, int SGR_B2_WAVES>
__global__ void __launch_bounds__(64 * SGR_B2_WAVES) k_tile_pass(int gx, int gy, int sgx, int T1, const uint4* __restrict__ chunk_info,
                                                                 const uint32_t* __restrict__ hdr, const uint4* __restrict__ L1,
                                                                 uint32_t* __restrict__ cnt2,
                                                                 const uint32_t* __restrict__ tile_start,
                                                                 uint32_t* __restrict__ point_list, uint32_t list_cap,
                                                                 const uint32_t* __restrict__ tile_need, const uint32_t* __restrict__ gate)
{
    constexpr int SGR_B2_TILES_PER_WAVE = 64 / SGR_B2_WAVES;
    __shared__ uint32_t s_rows[WRITE ? SGR_B2_WAVES * SGR_B2_CHUNK : 1];
    if (hdr[SGR_B2_HDR_OVERFLOW]) return;
    if (gate && *gate == 0u) return;  // (the repair pass of the walk hint: no tile outran its hint)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* s_row = s_rows + (WRITE ? wave * SGR_B2_CHUNK : 0);
    const int n_chunks = (int)hdr[SGR_B2_HDR_CHUNKS];
    for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint4 ci = chunk_info[c];
        const int sup = (int)ci.x;
        const uint32_t e0 = ci.y;
        const int n = (int)ci.z;
        const int ox = (sup % sgx) * SGR_SUP, oy = (sup / sgx) * SGR_SUP;
        uint32_t base = 0;
        if (WRITE) {
            // lane t: first slot of tile t's segment for this chunk
            uint32_t before = 0, need = 0, total = 0;
            const int tx = ox + (lane & (SGR_SUP - 1)), ty = oy + (lane >> SGR_SUP_SHIFT);
            if (tx < gx && ty < gy) {
                const int t = ty * gx + tx;
                before = cnt2[(size_t)c * 64 + lane];  // entries of the tile in the earlier chunks of this super-tile
                const uint32_t t0 = tile_start[t];
                total = tile_start[t + 1] - t0;
                base = t0 + before;
                need = tile_need ? tile_need[t] : 0xFFFFFFFFu;
            }
            // walk hint: the lists are in depth order and so are the chunks -- if every tile of the super-tile already holds
            // the entries it is expected to walk (or all it has), this chunk and every later one is never read: nothing to
            // write.  (All four waves evaluate the same test.)
            if (__ballot(before < min(need, total)) == 0ull) continue;
        }
        const int half = (wave * SGR_B2_TILES_PER_WAVE) >> 5;           // this wave's tiles sit in the low or the high mask word
        const int t_first = (wave * SGR_B2_TILES_PER_WAVE) & 31;
        uint32_t msk[SGR_B2_BATCHES], id[SGR_B2_BATCHES];
#pragma unroll
        for (int b = 0; b < SGR_B2_BATCHES; b++) {
            msk[b] = 0u; id[b] = 0u;
            if (b * 64 + lane < n) {  // (requesting all eight up front at clamped indices made no difference here: measured)
                const uint4 e = L1[e0 + b * 64 + lane];
                uint32_t lo, hi;
                tile_mask(make_uint2(e.x, e.y), ox, oy, lo, hi);
                msk[b] = half ? hi : lo;
                id[b] = e.z;
            }
        }
        if (!WRITE) {
            uint32_t run = 0;  // lane 16 wave + tt: entries of this chunk covering that tile
#pragma unroll
            for (int b = 0; b < SGR_B2_BATCHES; b++) {
                if (b * 64 >= n) break;  // (uniform)
                uint32_t add = 0;
#pragma unroll
                for (int tt = 0; tt < SGR_B2_TILES_PER_WAVE; tt++) {
                    const unsigned long long M = __ballot((msk[b] & (1u << (t_first + tt))) != 0u);
                    add = (lane == wave * SGR_B2_TILES_PER_WAVE + tt) ? (uint32_t)__popcll(M) : add;
                }
                run += add;
            }
            if (lane / SGR_B2_TILES_PER_WAVE == wave) cnt2[(size_t)c * 64 + lane] = run;
        } else {
            for (int tt = 0; tt < SGR_B2_TILES_PER_WAVE; tt++) {
                const uint32_t sel = 1u << (t_first + tt);
                uint32_t cnt = 0;  // wave-uniform
#pragma unroll
                for (int b = 0; b < SGR_B2_BATCHES; b++) {
                    const bool bit = (msk[b] & sel) != 0u;
                    const unsigned long long M = __ballot(bit);
                    if (bit) s_row[cnt + lanes_below(M)] = id[b];
                    cnt += (uint32_t)__popcll(M);
                }
                if (cnt) {
                    // the LDS pipeline of a wave is in order: the row's writes land before these reads, and the reads
                    // before the next tile's writes; only the compiler has to keep that order (a fence would drain
                    // the stores)
                    LDS_ORDER();
                    const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)base, wave * SGR_B2_TILES_PER_WAVE + tt);
                    for (uint32_t j = lane; j < cnt; j += 64)
                        if (dst + j < list_cap) point_list[dst + j] = s_row[j];  // (sync-free mode: the list has the caller's capacity)
                    LDS_ORDER();
                }
            }
        }
    }
}

// per super-tile: exclusive scan of cnt2 over its chunks (lane = tile), totals to tile_count
__global__ void __launch_bounds__(64) k_tile_scan2(int gx, int gy, int sgx, const uint32_t* __restrict__ chunk_base,
                                                   const uint32_t* __restrict__ hdr, uint32_t* __restrict__ cnt2,
                                                   uint32_t* __restrict__ tile_count)
{
    if (hdr[SGR_B2_HDR_OVERFLOW]) return;
    const int sup = blockIdx.x, lane = threadIdx.x;
    const int c0 = (int)chunk_base[sup], c1 = (int)chunk_base[sup + 1];
    constexpr int UNROLL = 8;
    uint32_t run = 0;
    int c = c0;
    for (; c + UNROLL <= c1; c += UNROLL) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = cnt2[(size_t)(c + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { cnt2[(size_t)(c + u) * 64 + lane] = run; run += v[u]; }
    }
    for (; c < c1; c++) { const uint32_t v = cnt2[(size_t)c * 64 + lane]; cnt2[(size_t)c * 64 + lane] = run; run += v; }
    const int tx = (sup % sgx) * SGR_SUP + (lane & (SGR_SUP - 1));
    const int ty = (sup / sgx) * SGR_SUP + (lane >> SGR_SUP_SHIFT);
    if (tx < gx && ty < gy) tile_count[ty * gx + tx] = run;
}

// exclusive scan over the slices of hist1[slice][t] for one bin t per workgroup (the bins are few, the slices many:
// binning.hip's lane-per-bin column walk would leave the chip idle).  Thread i owns SGR_B2_SLICES / 256 consecutive slices;
// their loads go out as one batch and the scan is a shuffle scan per wave plus the four wave totals.
__global__ void __launch_bounds__(256) k_sup_hist_scan(int T1, uint32_t* __restrict__ hist1, uint32_t* __restrict__ sup_count)
{
    constexpr int PER = SGR_B2_SLICES / 256;
    __shared__ uint32_t s_w[4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    uint32_t v[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) v[j] = hist1[(size_t)(tid * PER + j) * T1 + t];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) sum += v[j];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) s_w[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t x = s_w[w]; if (w < (tid >> 6)) before += x; total += x; }
    uint32_t run = before + incl - sum;
#pragma unroll
    for (int j = 0; j < PER; j++) { hist1[(size_t)(tid * PER + j) * T1 + t] = run; run += v[j]; }
    if (tid == 0) sup_count[t] = total;
}

void set_lds_limit(const void* fn, size_t bytes, size_t& configured)
{
    if (bytes > 48 * 1024 && bytes > configured) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        configured = bytes;
    }
}

}  // namespace

Bin2Layout sgr_bin2_layout(int P, int gx, int gy)
{
    Bin2Layout L;
    L.sgx = (gx + SGR_SUP - 1) / SGR_SUP;
    L.sgy = (gy + SGR_SUP - 1) / SGR_SUP;
    L.T1 = L.sgx * L.sgy;
    const size_t cap = (size_t)(P > 0 ? P : 1) * 4 + 65536;  // level-1 entries: ~2 per Gaussian in practice
    L.cap1 = cap > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)cap;
    L.chunk_cap = L.cap1 / SGR_B2_CHUNK + (uint32_t)L.T1 + 1;
    L.per_slice = (((P + SGR_B2_SLICES - 1) / SGR_B2_SLICES + 63) / 64) * 64;
    if (L.per_slice < 64) L.per_slice = 64;
    size_t off = 0;
    L.hist1 = off;      off = sgr_align(off + (size_t)SGR_B2_SLICES * L.T1 * 4);
    L.sup_count = off;  off = sgr_align(off + (size_t)L.T1 * 4);
    L.sup_start = off;  off = sgr_align(off + (size_t)(L.T1 + 1) * 4);
    L.chunk_base = off; off = sgr_align(off + (size_t)(L.T1 + 1) * 4);
    L.hdr = off;        off = sgr_align(off + 64);
    L.L1 = off;         off = sgr_align(off + (size_t)L.cap1 * 16);  // {packed rectangle, id, -} per entry
    L.cnt2 = off;       off = sgr_align(off + (size_t)L.chunk_cap * 64 * 4);
    L.chunk_sup = off;  off = sgr_align(off + (size_t)L.chunk_cap * 16);  // chunk -> {super-tile, first entry, entries, -}
    L.total = off;
    return L;
}

// level 1 + the counting half of level 2: leaves tile_count[T] (consumed by sgr_launch_tile_scan) and the header
void sgr_launch_bin2_count(int P, int gx, int gy, const Bin2Layout& L, char* scratch, uint32_t* hdr, const uint2* rects,
                           const uint32_t* order, uint32_t* tile_count, uint32_t chunk_grid, hipStream_t s)
{
    uint32_t* hist1 = reinterpret_cast<uint32_t*>(scratch + L.hist1);
    uint32_t* sup_count = reinterpret_cast<uint32_t*>(scratch + L.sup_count);
    uint32_t* sup_start = reinterpret_cast<uint32_t*>(scratch + L.sup_start);
    uint32_t* chunk_base = reinterpret_cast<uint32_t*>(scratch + L.chunk_base);
    uint4* L1 = reinterpret_cast<uint4*>(scratch + L.L1);
    uint32_t* cnt2 = reinterpret_cast<uint32_t*>(scratch + L.cnt2);
    uint4* chunk_info = reinterpret_cast<uint4*>(scratch + L.chunk_sup);
    static size_t conf_a = 0, conf_b = 0;
    const size_t lds = (size_t)L.T1 * 4, lds_sc = lds + 64 * SGR_B2_MAXN * 4;
    int key_bits = 1;
    while ((1 << key_bits) < L.T1) key_bits++;
    set_lds_limit(reinterpret_cast<const void*>(&k_sup_count), lds, conf_a);
    set_lds_limit(reinterpret_cast<const void*>(&k_sup_scatter), lds_sc, conf_b);
    hipLaunchKernelGGL(k_sup_count, dim3(SGR_B2_SLICES), dim3(256), lds, s, P, L.sgx, L.T1, L.per_slice, rects, hist1);
    hipLaunchKernelGGL(k_sup_hist_scan, dim3(L.T1), dim3(256), 0, s, L.T1, hist1, sup_count);
    hipLaunchKernelGGL(k_sup_scan, dim3(1), dim3(1024), 0, s, L.T1, L.cap1, L.chunk_cap, sup_count, sup_start, chunk_base,
                       chunk_info, hdr);
    hipLaunchKernelGGL(k_sup_scatter, dim3(SGR_B2_SLICES), dim3(64), lds_sc, s, P, L.sgx, L.T1, key_bits, L.per_slice, rects, order,
                       sup_start, hist1, hdr, L1);
    uint32_t grid = L.chunk_cap < 8192u ? L.chunk_cap : 8192u;  // the chunk count lives on the device: grid-stride
    if (chunk_grid && chunk_grid < grid) grid = chunk_grid;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_pass<false, SGR_B2_COUNT_WAVES>), dim3(grid), dim3(64 * SGR_B2_COUNT_WAVES), 0, s, gx, gy, L.sgx, L.T1, chunk_info, hdr, L1,
                       cnt2, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    hipLaunchKernelGGL(k_tile_scan2, dim3(L.T1), dim3(64), 0, s, gx, gy, L.sgx, chunk_base, hdr, cnt2, tile_count);
}

void sgr_launch_bin2_write(int gx, int gy, const Bin2Layout& L, char* scratch, const uint32_t* hdr, uint32_t n_chunks, const uint2* rects,
                           const uint32_t* order, const uint32_t* tile_start, uint32_t* point_list, uint32_t list_cap,
                           const uint32_t* tile_need, hipStream_t s, const uint32_t* gate, uint32_t max_grid)
{
    if (n_chunks == 0) return;
    if (max_grid && n_chunks > max_grid) n_chunks = max_grid;  // (grid-stride over the device-side chunk count)
    const uint4* L1 = reinterpret_cast<const uint4*>(scratch + L.L1);
    uint32_t* cnt2 = reinterpret_cast<uint32_t*>(scratch + L.cnt2);
    const uint4* chunk_info = reinterpret_cast<const uint4*>(scratch + L.chunk_sup);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_pass<true, SGR_B2_WRITE_WAVES>), dim3(n_chunks), dim3(64 * SGR_B2_WRITE_WAVES), 0, s, gx, gy, L.sgx, L.T1, chunk_info, hdr, L1,
                       cnt2, tile_start, point_list, list_cap, tile_need, gate);
}
