// tile_order.h -- behind a blend forward, by ONE workgroup of NT threads: this view's launch order (tiles by the depth of their
// deepest contributor, deepest first: a counting sort over 1024 depth classes, ties in arbitrary order), the walk hint for the
// camera's next visit and the second header copy for the host (include/sugar_raster.h: sgr_forward_opts.tile_order_out,
// tile_need_out, header_host).  Device code shared by blend.hip (a kernel of its own: k_tile_order) and loss.hip (a spare
// workgroup of the loss forward kernel, which is what follows the blend in the train step: the 12 us of this serial,
// single-workgroup job leave the step's chain).
#pragma once
#include "sgr_common.h"

struct SgrTileOrderJob {
    int T;                       // tiles; 0 = no job
    uint32_t list_cap;           // capacity the forward ran with (SGR_FORWARD_INVALID)
    const uint32_t* tile_maxc;   // deepest contributor per tile (image scratch)
    const uint32_t* tile_walked; // entries walked per tile, or NULL (no hint wanted)
    const uint32_t* header;      // device header of the forward
    uint32_t* order;             // the backward's launch order (image scratch)
    uint32_t* order_copy;        // the caller's copy (tile_order_out), or NULL
    uint32_t* need_out;          // walk hint out, or NULL
    uint32_t* header_host;       // device address of the pinned header copy, or NULL
    float margin;
};

template <int NT>
__device__ __forceinline__ void sgr_tile_order_block(const SgrTileOrderJob& j)
{
    static_assert(NT == 256 || NT == 1024, "one or four classes per thread");
    constexpr int CPT = 1024 / NT;
    __shared__ uint32_t s_cls[1024];
    __shared__ uint32_t s_w[NT / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int T = j.T;
    // the second header copy for the host (word 3, the hint-miss flag, is final now that the blend kernel is done)
    if (j.header_host && tid < 8) j.header_host[8 + tid] = j.header[tid];
    if (SGR_FORWARD_INVALID(j.header, j.list_cap)) return;  // (an invalid forward leaves the previous order and hint in place)
    if (j.need_out)  // what the tile walked now, plus a margin, plus one batch
        for (int i = tid; i < T; i += NT) {
            const uint32_t w = j.tile_walked[i];
            j.need_out[i] = w + (uint32_t)((float)w * j.margin) + 64u;
        }
    const uint32_t mc = j.header[SGR_HDR_MAXCOUNT];
    const int shift = mc >= 1024u ? (32 - __builtin_clz(mc)) - 10 : 0;  // class = 1023 - (depth >> shift): class 0 = deepest
    for (int c = tid; c < 1024; c += NT) s_cls[c] = 0u;
    __syncthreads();
    for (int i = tid; i < T; i += NT) atomicAdd(&s_cls[1023u - min(j.tile_maxc[i] >> shift, 1023u)], 1u);
    __syncthreads();
    uint32_t loc[CPT], mine = 0;
#pragma unroll
    for (int k = 0; k < CPT; k++) { loc[k] = s_cls[tid * CPT + k]; mine += loc[k]; }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
    for (int k = 0; k < CPT; k++) { s_cls[tid * CPT + k] = run; run += loc[k]; }  // first slot of the class
    __syncthreads();
    for (int i = tid; i < T; i += NT) {
        const uint32_t slot = atomicAdd(&s_cls[1023u - min(j.tile_maxc[i] >> shift, 1023u)], 1u);
        j.order[slot] = (uint32_t)i;
        if (j.order_copy) j.order_copy[slot] = (uint32_t)i;  // (may alias the forward's tile_order: the blend is done)
    }
}
