/*
 * sugar_raster.h -- C ABI of the MI355X-native (gfx950) differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary for the one hot path of Anttwo/SuGaR: the tile rasterizer that the
 * reference reaches through
 *     DGR = gaussian_splatting/submodules/diff-gaussian-rasterization
 *     CudaRasterizer::Rasterizer::{forward,backward,markVisible}   DGR/cuda_rasterizer/rasterizer.h:24-84
 * bound to Python by DGR/rasterize_points.cu:35-217 and DGR/ext.cpp:15-19, plus
 *     SimpleKNN::knn / distCUDA2    simple-knn/simple_knn.h:15-19, simple-knn/spatial.cu:15-26.
 *
 * Plain pointers and sizes only (no torch types).  Every pointer is a DEVICE pointer unless said
 * otherwise; a null pointer means "input absent", exactly like the reference (DGR/cuda_rasterizer/
 * forward.cu:205,241).  `stream` is a hipStream_t passed as void*; all work is enqueued on it.
 * The library never frees caller memory; scratch comes from the three allocation callbacks (the
 * std::function<char*(size_t)> allocators of rasterizer.h:24-27 made C-callable).
 *
 * Return convention: functions return >= 0 on success and a negative SGR_E_* code on failure;
 * sgr_last_error() gives the message for the calling thread.
 */
#ifndef SUGAR_RASTER_H_INCLUDED
#define SUGAR_RASTER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGR_ABI_VERSION 4 /* 2: sgr_forward_ex takes an options struct, binning mode per call, trainer API; 3: sgr_forward_info.speculation, SGR_FLAG_SPECULATIVE;
                             4: sgr_train_view.flags, sgr_train_exchange ranges, exact alpha the default, sgr_compact_level_rows / sgr_pick_pixels */

#define SGR_E_INVALID (-1) /* bad argument (e.g. NUM_CHANNELS != 3 path, rasterizer_impl.cu:242-245) */
#define SGR_E_HIP (-2)     /* a HIP runtime call or kernel failed (CHECK_CUDA, auxiliary.h:166-173) */
#define SGR_E_ALLOC (-3)   /* an allocation callback returned NULL */

/* Scratch allocator: must return a device buffer of at least `bytes` bytes, 256-byte aligned, that
 * stays valid until the matching backward has run.  Replaces resizeFunctional, rasterize_points.cu:27-33. */
typedef char* (*sgr_alloc_fn)(void* user, size_t bytes);

int sgr_abi_version(void);
const char* sgr_last_error(void);

/* Rasterizer::forward, DGR/cuda_rasterizer/rasterizer.h:31-55 / rasterizer_impl.cu:198-336.
 *   P Gaussians, D active SH degree, M SH coefficients per Gaussian (0 if shs == NULL).
 *   means3D[P*3], shs[P*M*3], colors_precomp[P*3], opacities[P], scales[P*3], rotations[P*4],
 *   cov3D_precomp[P*6], viewmatrix[16], projmatrix[16], cam_pos[3], background[3]  (all float32).
 *   out_color[3*H*W] and radii[P] are written (radii may be NULL).
 * Returns num_rendered (Gaussian x tile instances), or a negative error code.
 * The layout of the three scratch buffers is private to this library (a15 in SURVEY.md section 8a). */
int64_t sgr_forward(sgr_alloc_fn geom_alloc, void* geom_user,
                    sgr_alloc_fn binning_alloc, void* binning_user,
                    sgr_alloc_fn img_alloc, void* img_user,
                    int P, int D, int M,
                    const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp,
                    const float* opacities, const float* scales, float scale_modifier,
                    const float* rotations, const float* cov3D_precomp,
                    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                    float tan_fovx, float tan_fovy, int prefiltered,
                    float* out_color, int* radii, int debug, void* stream);

/* sgr_forward with options (extension; every field may be zero / NULL: then exactly sgr_forward).
 *
 * binning_capacity > 0 selects the SYNC-FREE forward.  The reference copies num_rendered to the host in the middle of the
 *   forward to size the instance list (rasterizer_impl.cu:280-281): the GPU idles for the round trip.  Here the list is
 *   allocated for `binning_capacity` instances, nothing is waited for and the call returns binning_capacity (pass it to
 *   sgr_backward as R).  The device-side header (8 uint32 at sgr_img_header_offset() of the image scratch, SGR_HDR_*) tells
 *   whether the forward is VALID: word 0 = the real num_rendered; if it exceeds the capacity, or word 6 != 0 (level-1 binning
 *   overflow), the blend kernel has returned without touching its outputs and the caller must discard this forward and
 *   repeat it (binning_capacity = 0, or a larger one).  sgr_backward on an invalid forward is a no-op on the device (its
 *   kernels read the same header), so a caller may check late.
 * header_host (pinned HOST memory, 16 uint32) / header_event (a hipEvent_t): the header is copied to header_host[0..7]
 *   right behind the tile scan -- before the list write and the blend -- and once more to header_host[8..15] behind the
 *   blend (word 3, the hint-miss flag, is only final there); header_event, if given, is recorded behind the second copy.
 * tile_need (device, uint32 per tile, extension): WALK HINT.  The lists are depth ordered and a tile stops walking once all
 *   its pixels are saturated -- at the metric workload 88 % of the instances written are never read.  With tile_need[t] = the
 *   number of list entries tile t is expected to walk (e.g. its tile_walked of the previous visit of the same camera, plus a
 *   margin), the list-write pass skips every 512-entry chunk of a super-tile none of whose tiles needs it, and the blend
 *   walks at most tile_need[t] entries of tile t.  Ranges, num_rendered and the written prefix of every list are unchanged
 *   (bit-identical to rasterizer_impl.cu:70-138).  If a tile would have walked further, header word 3 (SGR_HDR_HINT_MISS) is
 *   set: the forward is then INVALID exactly like a capacity overflow (repeat it without the hint).
 * tile_need_out (device, uint32 per tile): receives the hint for the next visit: tile_walked * (1 + hint_margin) + 64
 *   (hint_margin <= 0: 0.25).
 * tile_order (device, uint32 per tile, extension): LAUNCH ORDER of the blend kernel.  A permutation of 0..tiles-1; the blend
 *   kernel hands its workgroups to the tiles in this order.  With the deepest tiles first the waves still running when the grid
 *   drains are the short ones (raster order: 118 us, the order of the camera's previous visit: 110 us at the metric workload).
 *   The image does not depend on it.  It must be a permutation (as written by tile_order_out); anything else leaves tiles
 *   unrendered.
 * tile_order_out (device, uint32 per tile): receives this view's tiles sorted by the depth of their deepest contributor,
 *   deepest first (a counting sort over 1024 depth classes; ties in arbitrary order) -- the order the backward of this view
 *   uses (pass SGR_BWD_TILE_ORDER_READY to sgr_backward_ex and it does not sort again) and a good tile_order for the next
 *   visit of the same camera.  May alias tile_order.
 * info (host, may be NULL): what the call did (binning path taken).
 * Header word 5 (SGR_HDR_CHUNKS): level-2 chunks of this view (see chunk_grid). */
#define SGR_HDR_R 0          /* header words (device): total instances */
#define SGR_HDR_MAXCOUNT 1   /* largest per-tile instance count */
#define SGR_HDR_R_HI 2
#define SGR_HDR_HINT_MISS 3  /* != 0: a tile needed more entries than its walk hint allowed */
#define SGR_HDR_CHUNKS 5      /* 512-entry chunks of the super-tile lists (two-level binning) */
#define SGR_HDR_L1_OVERFLOW 6 /* != 0: the level-1 (super-tile) list overflowed its capacity */
#define SGR_HDR_LAYOUT_CAP 8  /* the instance capacity the binning buffer of this forward was laid out for (written by the forward
                                 blend kernel, read by the backward blend kernel: the binning buffer may be cloned or moved between them) */
#define SGR_HDR_DEEP 9        /* tiles of this view whose blocks were blended by the eight-wave kernel for long lists (round 6) */
#define SGR_HDR_REPAIR 7      /* tiles that outran their walk hint and were rendered again inside the same forward (round 5): a hint
                                 that is too short costs those tiles a second pass, not the forward; SGR_HDR_HINT_MISS is only raised
                                 when more than 1024 tiles did */
typedef struct sgr_forward_info {
    int binning_mode;        /* 0: two-level binning, 1: single-level */
    int sync_free;           /* 1: the call did not wait for the device */
    int speculation;         /* SGR_FLAG_SPECULATIVE: 0 not asked for, 1 hit (the capacity held), 2 miss (list pass and blend ran twice) */
} sgr_forward_info;
typedef struct sgr_forward_opts {
    int64_t binning_capacity;
    int flags;               /* SGR_FLAG_* */
    uint32_t* header_host;
    void* header_event;
    const uint32_t* tile_need;
    uint32_t* tile_need_out;
    float hint_margin;
    uint32_t chunk_grid;     /* sync-free forward: workgroups to launch for the level-2 tile passes (their count lives on the
                                device; the passes are grid-stride loops, so any value is correct).  0: the capacity bound,
                                about twice what a typical view needs; a trainer passes header word 5 of the camera's
                                previous visit plus a margin */
    sgr_forward_info* info;
    const uint32_t* tile_order;
    uint32_t* tile_order_out;
} sgr_forward_opts;
int64_t sgr_forward_ex(sgr_alloc_fn geom_alloc, void* geom_user,
                       sgr_alloc_fn binning_alloc, void* binning_user,
                       sgr_alloc_fn img_alloc, void* img_user,
                       int P, int D, int M,
                       const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                       float tan_fovx, float tan_fovy, int prefiltered,
                       float* out_color, int* radii, int debug, void* stream, const sgr_forward_opts* opts);

/* flags of sgr_forward_ex.  SGR_FLAG_RAW_PARAMS: `scales`, `rotations` and `opacities` are the RAW parameters of the 3DGS model
 * (log scale, unnormalised quaternion, opacity logit) and the activations of gaussian_model.py:92-117 (exp, F.normalize,
 * sigmoid) are applied inside the preprocess kernel -- same arithmetic as sgr_activations_forward, without its round trip
 * through memory.  The matching backward is sgr_backward_phase with SGR_MODE_RAW_PARAMS: dL_dscale, dL_drot and dL_dopacity
 * are then the gradients w.r.t. the raw parameters (sgr_activations_backward folded in).  Ignored with cov3D_precomp. */
#define SGR_FLAG_RAW_PARAMS 1
/* SGR_FLAG_SINGLE_LEVEL_BINNING: take the single-level ordered scatter (binning.hip) instead of the default two-level
 * binning (super-tiles of 8x8 tiles, then tiles; falls back to the single level by itself when the level-1 list would
 * overflow).  Both produce bit-identical lists and ranges; the single level needs one LDS counter per tile (at most about
 * 38 000 tiles) and always takes the host round trip. */
#define SGR_FLAG_SINGLE_LEVEL_BINNING 2
/* SGR_FLAG_DEFER_POST (used by sgr_trainer_step): the forward stops behind the blend kernel -- launch order, walk hint and the
 * second header copy are NOT written and header_event is NOT recorded by the call; the caller has them done by the kernel that
 * follows (the loss forward carries that job in a spare workgroup) and records the event itself.  Needs header_host to be
 * device-mapped pinned memory; the call fails with SGR_E_INVALID otherwise. */
#define SGR_FLAG_DEFER_POST 4
/* SGR_FLAG_SPECULATIVE (with binning_capacity > 0, two-level binning): the reference's forward semantics -- the call returns the
 * TRUE instance count, as Rasterizer::forward does (rasterizer_impl.cu:280-281,335) -- without its idle GPU: every kernel of the
 * forward is enqueued sync-free with `binning_capacity` as the list capacity, and the ONE host wait (for the header the tile scan
 * writes) happens at the END of the call, while the list-write pass and the blend kernel are still queued or running.  When the
 * true count exceeds the capacity (or the level-1 list overflowed) the queued kernels were no-ops and the call runs them once more
 * with the true count (binning_alloc is then called a second time).  The instance list is laid out for the capacity; sgr_backward
 * finds that layout through the binning buffer's address, so the caller passes the returned count as usual.  Ignored with
 * SGR_FLAG_DEFER_POST.  sgr_forward_info.speculation reports hit / miss. */
#define SGR_FLAG_SPECULATIVE 8
/* Exact alpha (round 6; THE DEFAULT): the blend kernels evaluate alpha exactly as the reference does -- `power` rounded
 * operation by operation in the order of forward.cu:333 / backward.cu:492, G = expf(power) by the device library's own two-term
 * algorithm, test_T = T (1 - alpha).  alpha, T, final_T and n_contrib are then bit-identical to the reference's kernels (compiled
 * without FP contraction) and every gradient tensor agrees with them to the reference's own float-atomic noise (about 4e-6
 * norm-wise at BASELINE sizes).  The alternative, sgr_set_exact_alpha(0) or SGR_EXACT_ALPHA=0 in the environment ("fast alpha"):
 * conic pre-scaled by log2(e), two FMAs and one v_exp_f32 -- 12 fewer vector instructions per (list entry, 8x8 block), ~6 % of a
 * train step -- evaluates alpha just as accurately but with other roundings (~1e-7 relative apart), and the sums over a splat's
 * pixels cancel so heavily that this is 3e-5 .. 1.4e-4 norm-wise in dL/dscale and dL/drotation: around north_star's 1e-4 bar, over
 * it for some cameras (profiles/r06_grad_switch_table.txt, profiles/r06_fullsize_parity_fast_alpha.json).  The mode is process-wide;
 * SGR_FLAG_EXACT_ALPHA / SGR_BWD_EXACT_ALPHA force it on for one call.  Forward and backward of a view must run in the same mode. */
#define SGR_FLAG_EXACT_ALPHA 16
/* Long lists (round 6).  With a walk hint (sgr_forward_opts.tile_need) the 8x8 blocks of a tile whose hinted list is longer than
 * `entries` are blended by an eight-wave kernel -- alpha for eight 64-entry batches in parallel, then the cheap sequential
 * transmittance chain -- beside the one-wave kernel, on a stream of the library's own; bit-identical results.  Default 0 = off
 * (SGR_DEEP_MIN in the environment sets it): measured on BASELINE config 4 the kernel is bit-exact but not yet faster than the
 * one-wave walk (profiles/r06_blend_fwd_deep_lists_ab.txt).  Header word SGR_HDR_DEEP reports how many tiles took that path. */
#define SGR_FLAG_NO_DEEP 32   /* this call: one wave per block whatever the hint says (a caller that knows its lists are short saves the
                                 side stream's fork and join: sgr_trainer_step sets it unless the camera's last visit left a hint
                                 above the threshold -- host header word 9, second copy, carries the largest hint written) */
void sgr_set_deep_min(int entries);
int sgr_get_deep_min(void);
void sgr_set_exact_alpha(int on);
int sgr_get_exact_alpha(void);
#define SGR_MODE_RAW_PARAMS 4 /* or-ed into the `phase` argument of sgr_backward_phase (phases 0, 1, 2 as before) */
/* Compact SH mode only (dL_dsh == NULL): the backward skips the SH block altogether -- no read of shs, and dL_dmean3D
 * comes out WITHOUT the term through the view direction; sgr_sh_adam_from_views_ex forms that term. */
#define SGR_MODE_SH_DIR_ELSEWHERE 8

/* Rasterizer::backward, DGR/cuda_rasterizer/rasterizer.h:57-84 / rasterizer_impl.cu:340-434.
 *   R = the value sgr_forward returned; geom/binning/img = the buffers its callbacks handed out.
 *   dL_dpix[3*H*W] in; gradient outputs (float32):
 *     dL_dmean2D[P*3] (x,y used; NDC-scaled, backward.cu:460-461), dL_dconic[P*4] (slots 0,1,3 used),
 *     dL_dopacity[P], dL_dcolor[P*3], dL_dmean3D[P*3], dL_dcov3D[P*6], dL_dsh[P*M*3],
 *     dL_dscale[P*3], dL_drot[P*4].
 *   Every output row is fully written by this call (rows of culled Gaussians are set to zero), so the
 *   caller does NOT need to pre-zero them; pre-zeroed buffers (rasterize_points.cu:151-159) work too.
 *   Compact SH mode: with shs given and dL_dsh == NULL no SH gradient is written and dL_dcolor receives the colour
 *   gradients masked by the forward's clamp flags (the dL_dRGB of backward.cu:31-34); sgr_sh_grad_from_views rebuilds
 *   the SH gradient, summed over any number of views, from those 3 floats per Gaussian and view.
 *   dL_dmean2D, dL_dconic and (without cov3D_precomp) dL_dcov3D may be NULL: they are then not written (a training step
 *   that only wants the gradients of its parameters saves 52 bytes of writes per Gaussian). */
int sgr_backward(int P, int D, int M, int64_t R,
                 const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* img_buffer,
                 const float* dL_dpix,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 int debug, void* stream);

/* sgr_backward in two halves for the compact SH mode (shs given, dL_dsh == NULL), same arguments:
 *   phase 1: accumulator reset + blend backward, then the clamp-masked colour gradients into dL_dcolor[P*3] -- they are
 *            final here, so a view-sharded trainer can start their all-gather;
 *   phase 2: the backward preprocess (every other output; dL_dcolor is not touched again);
 *   phase 0: both, identical to sgr_backward. */
int sgr_backward_phase(int phase, int P, int D, int M, int64_t R,
                       const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                       char* geom_buffer, char* binning_buffer, char* img_buffer,
                       const float* dL_dpix,
                       float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                       int debug, void* stream);

/* sgr_backward_phase with options.  Densification statistics of the train loop (gaussian_splatting/train.py:111-123,
 * sugar_scene/sugar_densifier.py:156-164), fused into the backward preprocess: for every Gaussian with radius > 0
 *   max_radii2D[i] = max(max_radii2D[i], radius_i);  grad_accum[i] += |dL_dmean2D[i].xy|;  denom[i] += 1
 * (all float[P], updated in place; NULL = not wanted).  dL_dmean2D itself need not be written for that. */
typedef struct sgr_backward_opts {
    float* max_radii2D;
    float* grad_accum;
    float* denom;
    float* campos_row;   /* compact SH mode: receives cam_pos[3] next to the colour gradients (the last row of an all-gather send
                            buffer), written by the kernel that writes dL_dcolor; NULL = not wanted */
    int flags;           /* SGR_BWD_* */
} sgr_backward_opts;
#define SGR_BWD_EXACT_ALPHA 2      /* see SGR_FLAG_EXACT_ALPHA */
#define SGR_BWD_TILE_ORDER_READY 1 /* the forward of this view was given tile_order_out: its launch order is in the image scratch */
int sgr_backward_ex(int phase, int P, int D, int M, int64_t R,
                    const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp,
                    const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                    const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                    char* geom_buffer, char* binning_buffer, char* img_buffer,
                    const float* dL_dpix,
                    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                    int debug, void* stream, const sgr_backward_opts* opts);

/* SH gradient from per-view masked colour gradients (view-sharded training exchanges 12 B per Gaussian and view instead
 * of 12*M B):  dL_dsh[P*M*3] = sum_v basis(normalize(means3D - campos_all[v])) (x) dcolor_all[v][P*3]   (the per-view
 * computeColorFromSH backward, backward.cu:47-97, summed over the n_views views).  campos_all[n_views*3] and
 * dcolor_all[n_views*P*3] are device arrays; view_stride = rows between consecutive views in dcolor_all (0 = P: dense;
 * the all-gather buffer of the trainer carries one extra row per view). */
int sgr_sh_grad_from_views(int P, int n_views, int D, int M, const float* means3D, const float* campos_all,
                           const float* dcolor_all, int64_t view_stride, float* dL_dsh, void* stream);

/* The same sum consumed on the spot: one Adam step (the update of sgr_adam_step, learning rate lr_dc for the three DC
 * values of a Gaussian and lr_rest for the others, gaussian_model.py:157-158) on sh_params[P*M*3] with its moment buffers;
 * the 48-float SH gradient is never written to memory.  means3D must be the values the views were rendered with (call
 * this before the positions are updated). */
int sgr_sh_adam_from_views(int P, int n_views, int D, int M, const float* means3D, const float* campos_all,
                           const float* dcolor_all, int64_t view_stride, float* sh_params, float* exp_avg, float* exp_avg_sq,
                           float lr_dc,
                           float lr_rest, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* As above, and the SH coefficients' other reader moves in with them: with dmean_extra != NULL the kernel also forms the
 * view-direction part of dL/dmean3D (computeColorFromSH backward, backward.cu:99-139, + dnormvdv), summed over the views,
 * and writes it to dmean_extra[P*3] (every row).  The backward that produced dcolor_all must then have been run with
 * SGR_MODE_SH_DIR_ELSEWHERE (its dL_dmean3D lacks exactly that term and it does not read the SH tensor at all), and the
 * position gradient is completed by sgr_adam_step_ex.  Saves one pass over the SH tensor per step (192 B per Gaussian). */
int sgr_sh_adam_from_views_ex(int P, int n_views, int D, int M, const float* means3D, const float* campos_all,
                              const float* dcolor_all, int64_t view_stride, float* sh_params, float* exp_avg, float* exp_avg_sq,
                              float lr_dc, float lr_rest, float beta1, float beta2, float eps, int step, float grad_scale,
                              float* dmean_extra, void* stream);

/* Rasterizer::markVisible, DGR/cuda_rasterizer/rasterizer.h:24-29 / rasterizer_impl.cu:141-153.
 * present[P] is one byte per Gaussian (bool). */
int sgr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Scratch sizes (bytes) for a given problem.  The geometry and binning callbacks are asked for exactly these; the image
 * callback is asked for sgr_img_bytes(width, height) PLUS the scratch of the two-level binning (which also depends on P),
 * appended after the image state -- the offsets below stay valid. */
size_t sgr_geom_bytes(int P);
size_t sgr_img_bytes(int width, int height);
size_t sgr_binning_bytes(int64_t R, int width, int height); /* instance list + per-block survivor masks for the backward */

/* ---- introspection for parity tests (read-only views into the private scratch layout) ---------
 * Each returns a byte offset into the corresponding buffer.  The geometry record of Gaussian i is
 * 12 floats at geom + sgr_geom_rec_offset() + 48*i:
 *   {x, y, conic.x, conic.y, conic.z, opacity, depth, bitcast(radius), r, g, b, bitcast(clamped bits)} */
size_t sgr_geom_rec_offset(int P);
size_t sgr_img_final_T_offset(int width, int height);   /* float[W*H] */
size_t sgr_img_n_contrib_offset(int width, int height); /* uint32[W*H] */
size_t sgr_img_tile_start_offset(int width, int height);/* uint32[T+1]: tile t owns [start[t], start[t+1]) */
size_t sgr_img_tile_maxc_offset(int width, int height); /* uint32[T]: max n_contrib over the tile's pixels */
size_t sgr_img_tile_walked_offset(int width, int height); /* uint32[T]: furthest list position any pixel examined */
size_t sgr_img_header_offset(int width, int height);      /* 8 x uint32: see sgr_forward_ex */
size_t sgr_binning_point_list_offset(int64_t R);        /* uint32[R]: Gaussian ids, tile-major, depth order */

/* ---- optional per-stage timing (HIP events recorded on the caller's stream, process-wide) --
 * Stages: 0 preprocess, 1 ordered tile count + scans, 2 ordered scatter into the tile lists, 3 global depth sort of
 * the Gaussians, 4 blend forward (k_blend_fwd_w alone), 5 blend backward, 6 preprocess backward, 7 the two launches behind the blend that
 * repair a walk hint (empty when every hint held); and the rest of sgr_trainer_step, so that the stages cover every launch of the
 * native train step: 8 post-blend bookkeeping when it is a launch of its own (it rides in the loss kernel otherwise), 9 loss
 * forward, 10 loss backward, 11 reset of the backward's accumulator table, 12 SH-Adam from the colour gradients, 13 flat Adam,
 * 14 the masked-colour kernel of the two-phase (exchange) backward.  sgr_profile_read synchronises the recorded events, returns
 * the summed milliseconds and launch counts per stage since the last read, and clears the record. */
#define SGR_N_STAGES 15
void sgr_profile_enable(int stage_mask); /* bit s set: record events around stage s (0 disables; each event pair costs
                                            several microseconds of GPU pipeline, so time only what is being measured) */
int sgr_profile_read(double* ms_sum, int64_t* count, int n_stages);


/* ---- k-NN helpers sharing the Gaussian position buffer ------------------------------------------
 * sgr_dist2: simple_knn._C.distCUDA2 (simple-knn/spatial.cu:15-26 -> SimpleKNN::knn, simple-knn/
 *   simple_knn.cu:185-221): meanDists[i] = mean of the 3 smallest SQUARED distances from point i to the
 *   other points (simple_knn.cu:182).  points[P*3] -> meanDists[P].
 * sgr_knn: the exact K-nearest-neighbour query SuGaR obtains from pytorch3d.ops.knn_points
 *   (sugar_scene/sugar_model.py:49,235,1028,1342): for each of N query points the K (<= 32) nearest of
 *   M reference points; dists[N*K] squared distances ascending, idx[N*K] int64 reference indices
 *   (equal distances: lower index first).  When query and reference are the same buffer the point itself
 *   comes first with distance 0. */
int sgr_dist2(int P, const float* points, float* meanDists, void* stream);
int sgr_knn(int N, const float* query, int M, const float* ref, int K, float* dists, int64_t* idx, void* stream);
/* Same results (values AND indices) through an exact uniform-grid search, O(N) instead of O(N*M): the reference set is
 * counting-sorted into ~6-point cells and each query walks rings of cells until its K-th best distance is covered.
 * `scratch`: sgr_knn_grid_scratch_bytes(M) bytes of device memory.  With M >= 50 000 the call reads ONE word back from the device
 * (how many cells of the default grid hold a point) and refines the grid when the set turns out to be a surface rather than a
 * volume (round 5: Gaussians bound to a mesh put ~110 points into every occupied cell of the volume-sized grid); the choice is
 * kept per calling thread for the next 63 reference sets of the same size, which take no round trip.  Smaller sets and sgr_knn
 * never take one. */
size_t sgr_knn_grid_scratch_bytes(int M);
int sgr_knn_grid(int N, const float* query, int M, const float* ref, int K, float* dists, int64_t* idx, char* scratch,
                 void* stream);
int sgr_dist2_grid(int P, const float* points, float* meanDists, char* scratch, void* stream);

/* ---- fused photometric loss of the train step ---------------------------------------------------
 * loss = (1 - lambda) * mean|img - gt| + lambda * (1 - mean(SSIM(img, gt))), window 11, sigma 1.5, zero padding:
 * l1_loss / ssim of sugar_utils/loss_utils.py:17-63 as combined at gaussian_splatting/train.py:88-90 and
 * sugar_trainers/coarse_sdf.py:456-457.  img, gt: [channels, height, width] float32.
 *   forward : loss_out[3] = {loss, l1 mean, ssim mean}; `scratch` (sgr_l1_ssim_scratch_bytes) keeps the per-pixel SSIM
 *             partials for the backward.
 *   backward: grad_img[c,h,w] = grad_loss[0] * dloss/dimg (grad_loss is a device scalar, NULL = 1; gt receives no gradient).
 *   sgr_l1_ssim_forward with loss_out == NULL leaves the value to sgr_l1_ssim_backward_ex(..., loss_out): one spare workgroup of
 *   the backward kernel adds the forward's per-tile sums (one launch less on the train step's chain). */
size_t sgr_l1_ssim_scratch_bytes(int channels, int width, int height);
int sgr_l1_ssim_forward(int channels, int width, int height, const float* img, const float* gt, float lambda,
                        char* scratch, float* loss_out, void* stream);
int sgr_l1_ssim_backward(int channels, int width, int height, const float* img, const float* gt, float lambda,
                         const char* scratch, const float* grad_loss, float* grad_img, void* stream);
int sgr_l1_ssim_backward_ex(int channels, int width, int height, const float* img, const float* gt, float lambda,
                            const char* scratch, const float* grad_loss, float* grad_img, float* loss_out, void* stream);

/* ---- one-launch Adam over a flat parameter buffer -----------------------------------------------
 * torch.optim.Adam semantics (eps inside the bias-corrected denominator, no weight decay / amsgrad) as configured by
 * gaussian_splatting/scene/gaussian_model.py:152-166.  n floats in params / grads / exp_avg / exp_avg_sq (16-byte aligned);
 * the learning rate of element i is given by up to 8 segments k (HOST arrays): for seg_begin[k] <= i < seg_end[k] it is
 * seg_lr_a[k] when (i - seg_begin[k]) % seg_period[k] < seg_split[k], else seg_lr_b[k]; elements in no segment keep
 * lr 0 (their moments still update).  `step` is the 1-based step count used for bias correction; gradients are
 * multiplied by `grad_scale` on the fly (1/world_size after a SUM all-reduce, so the mean needs no extra pass). */
int sgr_adam_step(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_seg,
                  const long long* seg_begin, const long long* seg_end, const float* seg_lr_a, const float* seg_lr_b,
                  const int* seg_period, const int* seg_split, float beta1, float beta2, float eps, int step,
                  float grad_scale, void* stream);
/* The same update for up to 8 SEPARATE tensors in one launch (a drop-in `torch.optim.Adam` over the reference's six parameter
 * tensors: gaussian_model.py:152-166, sugar_optimizer.py:60-85): tensor t has n[t] floats in params[t] / grads[t] / exp_avg[t] /
 * exp_avg_sq[t] (16-byte aligned device pointers; the pointer tables and the per-tensor lr / beta1 / beta2 / eps / 1-based step are
 * HOST arrays).  Bit-identical to one sgr_adam_step per tensor with a single segment of that learning rate. */
int sgr_adam_step_multi(int n_tensors, const long long* n, float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const float* lr, const float* beta1, const float* beta2, const float* eps,
                        const int* step, void* stream);
/* The same step with a second gradient term: the gradient of element i < extra_n is grads[i] + extra[i] (both scaled by
 * grad_scale).  Used for the position gradient of sgr_sh_adam_from_views_ex (positions first in the flat buffer). */
int sgr_adam_step_ex(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_seg,
                     const long long* seg_begin, const long long* seg_end, const float* seg_lr_a, const float* seg_lr_b,
                     const int* seg_period, const int* seg_split, float beta1, float beta2, float eps, int step,
                     float grad_scale, const float* extra, long long extra_n, void* stream);

/* ---- the train step as ONE native call per phase ---------------------------------------------------------------------
 * gaussian_splatting/train.py:86-128 (render -> 0.8 L1 + 0.2 (1 - SSIM) -> backward -> Adam) enqueued by this library on the
 * caller's stream: no Python between the kernels, no host round trip (sync-free forward), no allocation (every buffer is
 * the caller's, sized once).  All pointers are DEVICE pointers owned by the caller unless said otherwise.
 *
 * Parameters: ONE flat float buffer (and equally laid out gradient / moment buffers), raw 3DGS parameters
 * (gaussian_model.py:92-117: log scale, unnormalised quaternion, opacity logit; activated inside the kernels):
 *   flat[off_xyz .. +3P] positions, [off_opacity .. +P], [off_scaling .. +3P], [off_rotation .. +4P] -- these four are the
 *   prefix flat[0 .. n_small) -- and [off_features .. +3MP] the SH tensor [P,M,3].
 * A step has four phases (bit mask), so that a view-sharded trainer can put its collectives between them:
 *   1  forward, loss, loss backward, blend backward; the clamp-masked colour gradients land in colors[0..3P) followed by the
 *      camera centre in colors[3P..3P+3): the send buffer of the all-gather;
 *   2  backward preprocess: gradients of the small parameters into flat_grad (+ the densification statistics, if given);
 *   4  Adam on the SH tensor straight from the per-view colour gradients (sgr_sh_adam_from_views);
 *   8  Adam on flat[0 .. n_small).
 * Validity: the forward is sync-free (binning capacity, optional walk hint).  An invalid forward (header words 0 / 3 / 6, see
 * sgr_forward_ex) turns every later kernel of the step -- backward AND Adam -- into a no-op on the device; the host learns
 * about it from sgr_trainer_forward_valid() whenever it chooses to look (typically one step later, when the header copy has
 * long arrived) and repeats the step with a larger capacity / without the hint. */
typedef struct sgr_train_config {
    int P, D, M, width, height;
    float* flat; float* flat_grad; float* exp_avg; float* exp_avg_sq;
    long long off_xyz, off_opacity, off_scaling, off_rotation, off_features, n_small;
    float lr_xyz, lr_opacity, lr_scaling, lr_rotation, lr_features_dc, lr_features_rest;
    float beta1, beta2, eps, lambda_dssim;
    const float* background;     /* [3] */
    char* geom; size_t geom_bytes;       /* >= sgr_geom_bytes(P) */
    char* img; size_t img_bytes;         /* >= sgr_img_bytes(width, height) + sgr_bin2_bytes(P, width, height) */
    char* binning; size_t binning_bytes; /* >= sgr_binning_bytes(binning_capacity, width, height) */
    int64_t binning_capacity;
    char* loss_scratch;          /* sgr_l1_ssim_scratch_bytes(3, width, height) */
    float* image;                /* [3,H,W] rendered image (output) */
    float* grad_image;           /* [3,H,W] dL/dimage (scratch) */
    float* loss_out;             /* [3]: loss, l1 mean, ssim mean (output) */
    float* colors;               /* [(P + 1) * 3] */
    int* radii;                  /* [P] or NULL */
    uint32_t* header_host;       /* pinned HOST memory, 16 uint32 */
    float* dL_dmean2D;           /* [P,3] or NULL: the gradient the densifier reads (viewspace_points.grad) */
    float* max_radii2D; float* grad_accum; float* denom; /* [P] each or NULL: fused densification statistics */
} sgr_train_config;
typedef struct sgr_train_view {
    const float* viewmatrix; const float* projmatrix; const float* campos; /* device: [16], [16], [3] */
    float tan_fovx, tan_fovy;
    const float* gt_image;       /* [3,H,W] */
    const uint32_t* tile_need;   /* walk hint of this camera (device, [tiles]) or NULL */
    uint32_t* tile_need_out;     /* receives the hint for its next visit, or NULL */
    float hint_margin;           /* see sgr_forward_opts */
    uint32_t chunk_grid;         /* see sgr_forward_opts */
    const uint32_t* tile_order;  /* launch order of this camera's previous visit (device, [tiles]) or NULL */
    uint32_t* tile_order_out;    /* receives this visit's order (also used by this step's backward), or NULL */
    uint32_t flags;              /* SGR_VIEW_* */
} sgr_train_view;
#define SGR_VIEW_DEEP_LISTS 1    /* this camera's hint has tiles above sgr_get_deep_min(): use the eight-wave kernel for them */
typedef struct sgr_train_exchange { /* phases 4 and 8 */
    int n_views;                 /* views whose colour gradients are summed (1: this rank's own) */
    const float* all_colors;     /* [n_views][view_stride][3]; NULL: cfg.colors */
    int64_t view_stride;         /* rows between views (0: P) */
    const float* all_campos;     /* [n_views][3]; NULL: this view's campos */
    float grad_scale;            /* 1 / n_views: the mean over the views is folded into the Adam kernels */
    int step;                    /* 1-based Adam step (bias correction) */
    /* round 6 -- the exchange in CHUNKS (each chunk's Adam launch starts as soon as ITS collective has landed):
     * phase 4 over the Gaussians [g_begin, g_end) only (both 0: all P); all_colors / view_stride then describe the chunk's own
     * block [n_views][view_stride][3] with row 0 = Gaussian g_begin.  g_begin must be a multiple of 64.
     * phase 8 over the floats [f_begin, f_end) of the flat small-parameter range only (both 0: all n_small); multiples of 4. */
    int64_t g_begin, g_end;
    int64_t f_begin, f_end;
} sgr_train_exchange;
typedef struct sgr_trainer sgr_trainer;
sgr_trainer* sgr_trainer_create(const sgr_train_config* cfg); /* copies cfg; NULL on a bad configuration (sgr_last_error) */
void sgr_trainer_destroy(sgr_trainer* t);
int sgr_trainer_set_binning(sgr_trainer* t, char* binning, size_t bytes, int64_t capacity); /* after a capacity overflow */
int sgr_trainer_step(sgr_trainer* t, const sgr_train_view* view, int phases, const sgr_train_exchange* ex, void* stream);
/* chunks of the in-library exchange (sgr_trainer_step_exchange): the colour all-gather and the small all-reduce are cut into
 * `n` pieces (1 <= n <= 16; default 1 -- sugar_amd.train_step.NativeTrainer sets 1 / 2 / 4 for 1 / 2 / more ranks), each followed
 * by its own Adam launch.  Results are bit-identical for every n on one or two ranks. */
int sgr_trainer_set_exchange_chunks(sgr_trainer* t, int n);
/* Waits for the header copy of the most recent phase-1 call (the copy sits behind the forward's blend kernel) and returns
 * 1 if that forward was valid, 0 if the step was a no-op on the device and has to be repeated; header_out[16] (may be NULL)
 * receives the two header copies (word 0: the real num_rendered, word 3: hint miss, word 6: level-1 overflow). */
int sgr_trainer_forward_valid(sgr_trainer* t, uint32_t* header_out);
const char* sgr_trainer_last_error(void);
/* The gradient exchange of the view-sharded step INSIDE the library (extension; SURVEY.md section 8e: one view per GPU, replicas,
 * "RCCL all-reduce of parameter grads over xGMI"): RCCL is bound at run time (dlopen of librccl.so -- the copy PyTorch has already
 * loaded, if any), one communicator per trainer, collectives on a stream of the library's own.
 *   sgr_rccl_unique_id(out[128])      rank 0: ncclGetUniqueId; the caller distributes the 128 bytes (torch.distributed, MPI, a file)
 *   sgr_trainer_comm_init(t, id, world, rank, recv, recv_bytes)   ncclCommInitRank; recv: world x (P + 1) x 3 floats (device)
 *   sgr_trainer_step_exchange(t, view, step, stream)   ONE call per step: phase 1, header check BEFORE anything is sent, all-gather
 *       of the masked colour gradients + camera centres beside phase 2, all-reduce of flat_grad[0 .. n_small) beside phase 4,
 *       phase 8 behind it.  Returns 0, or 1 when this rank's forward was invalid (nothing sent, nothing changed: enlarge the
 *       capacity / drop the hint and call again -- the other ranks wait in the all-gather), or a negative error code. */
int sgr_rccl_unique_id(char* out128);
int sgr_trainer_comm_init(sgr_trainer* t, const char* id128, int world, int rank, float* recv, size_t recv_bytes);
int sgr_trainer_comm_destroy(sgr_trainer* t);
/* A rank that hit an error no repeat of the step repairs calls this before it gives up: the communicator is ABORTED (ncclCommAbort),
 * so that the other ranks' pending collectives fail instead of waiting for ever for a peer that will not join. */
int sgr_trainer_comm_abort(sgr_trainer* t);
int sgr_trainer_step_exchange(sgr_trainer* t, const sgr_train_view* view, int step, void* stream);
double sgr_trainer_last_exchange_wait_ms(sgr_trainer* t); /* host time the last call spent waiting for the forward's header */
size_t sgr_bin2_bytes(int P, int width, int height); /* scratch of the two-level binning appended to the image scratch */

/* ---- SuGaR density field and level-set surface sampler (share the Gaussian buffers) --------------
 * B_g = R_g diag(1 / max(s_g, 1e-8)) is SuGaR's get_covariance(return_full_matrix=True, return_sqrt=True,
 * inverse_scales=True) (sugar_scene/sugar_model.py:730-736), row-major [P,3,3]; nbr_idx is the int64 [N,K] k-NN index
 * table (knn_idx[gaussian_idx]); strengths is [P] (SuGaR's [P,1]).
 *
 * sgr_density_field_forward: sugar_model.py:1270-1276 (get_field_values) -- neighbor_opacities[N,K] (may be NULL) and
 *   density[N] = sum over the K neighbours of  f * strength * exp(-0.5 * clamp(|B^T (x - mu)|^2, 0, 1e8)).
 *   (The `density >= 1 -> density / (density.detach() + 1e-12)` step of :1277-1281 is left to the caller.)
 * sgr_density_field_backward: gradients of  sum(dL_dopacities * opacities) + sum(dL_ddensity * density):
 *   dL_dx[N,3] is written; dL_dcenters[P,3], dL_dinv_scaled_rot[P,9], dL_dstrengths[P] are ACCUMULATED into (zero them).
 * sgr_level_set_points: sugar_model.py:1971-2079 (compute_level_surface_points_from_camera_fast) for N unprojected
 *   pixels: n_range samples in +-range_size * gaussian_std[nbr_idx[n,0]] along normalize(p - cam_center), densities
 *   (normalised where >= 1), per level (HOST array, <= 8) the first crossing with linear interpolation and the normal
 *   -normalize(grad density).  Outputs are level-major: valid[L,N] (0 = the reference's empty_pixels), points[L,N,3],
 *   normals[L,N,3] (may be NULL); rows with valid == 0 are zero. */
/* SuGaR.get_covariance(return_sqrt=True, inverse_scales=...), sugar_scene/sugar_model.py:730-736:
 * out[P,3,3] = quaternion_to_matrix(quaternions[P,4]) * s[:, None],  s = scaling[P,3] or 1 / clamp(scaling, 1e-8)
 * (real part first; any non-zero quaternion: the matrix is scaled by 2 / |q|^2 as in pytorch3d).  16-byte aligned quaternions. */
int sgr_scaled_rotation_forward(int P, const float* quaternions, const float* scaling, int inverse_scales, float* out,
                                void* stream);
int sgr_scaled_rotation_backward(int P, const float* quaternions, const float* scaling, int inverse_scales,
                                 const float* dL_dout, float* dL_dquaternions, float* dL_dscaling, void* stream);

/* Optional `packed` argument of the four functions below: [P][16] floats from sgr_pack_gaussians ({centre, strength,
 * inverse-scaled rotation, pad}: one 64-byte record per Gaussian, four 16-byte loads per neighbour instead of thirteen scalar
 * ones), or NULL to read the three arrays directly.  Must be 16-byte aligned. */
int sgr_pack_gaussians(int P, const float* centers, const float* inv_scaled_rot, const float* strengths, float* packed,
                       void* stream);
int sgr_density_field_forward(int N, int K, const float* x, const int64_t* nbr_idx, const float* centers,
                              const float* inv_scaled_rot, const float* strengths, float density_factor,
                              float* neighbor_opacities, float* density, const float* packed, void* stream);
int sgr_density_field_backward(int N, int K, const float* x, const int64_t* nbr_idx, const float* centers,
                               const float* inv_scaled_rot, const float* strengths, float density_factor,
                               const float* dL_dopacities, const float* dL_ddensity, float* dL_dx, float* dL_dcenters,
                               float* dL_dinv_scaled_rot, float* dL_dstrengths, const float* packed, void* stream);
/* The same gradients without float atomics: every pair takes one integer atomic for its rank among the pairs that reference
 * the same Gaussian, the pairs are laid out per Gaussian and summed in registers.  P = number of Gaussians; the three
 * per-Gaussian outputs are WRITTEN (every row, no zero-fill needed); scratch of sgr_density_field_backward_scratch_bytes. */
size_t sgr_density_field_backward_scratch_bytes(int N, int K, int P);
int sgr_density_field_backward_gather(int N, int K, int P, const float* x, const int64_t* nbr_idx, const float* centers,
                                      const float* inv_scaled_rot, const float* strengths, float density_factor,
                                      const float* dL_dopacities, const float* dL_ddensity, float* dL_dx, float* dL_dcenters,
                                      float* dL_dinv_scaled_rot, float* dL_dstrengths, char* scratch, const float* packed, void* stream);

/* ---- backward of a row gather -------------------------------------------------------------------------------------------------
 * out[p, 0..W) = sum over { n : idx[n] == p } of src[n, 0..W): what autograd computes for `x[idx]` (x: [P, W] float32, idx: N int64
 * entries, src = the incoming gradient [N, W]) -- the scatter-add of SURVEY.md section 8 row a21, met by SuGaR's regulariser at
 * sugar_model.py:922-925 and coarse_sdf.py:690-692.  W in 1..4; every row of `out` is WRITTEN (rows without entries get zeros);
 * negative indices wrap once, entries outside [-P, P) are ignored; N < 2^32.  The sum of a row is deterministic -- the entries are
 * grouped stably by row, sixteen lanes take every sixteenth entry of the group and their partial sums are folded in a fixed order --
 * so the result is reproducible bit for bit from run to run; it is NOT the sequential left-to-right sum of the entries (a CPU
 * `index_add` differs in the last bits).  scratch: sgr_scatter_add_rows_scratch_bytes(N, P) bytes of device memory.  No host synchronisation. */
size_t sgr_scatter_add_rows_scratch_bytes(long long N, int P);
int sgr_scatter_add_rows(long long N, const int64_t* idx, const float* src, int W, int P, float* out, char* scratch, void* stream);

int sgr_level_set_points(int N, int K, const float* world_points, const int64_t* nbr_idx, const float* cam_center,
                         const float* centers, const float* inv_scaled_rot, const float* strengths,
                         const float* gaussian_std, int n_levels, const float* levels_host, int n_range, float range_size,
                         float density_factor, uint8_t* valid, float* points, float* normals, const float* packed, void* stream);

/* The two per-view preparations of sgr_level_set_points, one launch each (as torch ops: ~40 and ~15 launches, and a host check
 * inside the 4 x 4 inverse):
 * sgr_view_std: gaussian_std[P] of sugar_model.py:1971-1972 = | scaling[P,3] (.) R(q)^T normalize(cam_center - centers) | for unit
 *   quaternions[P,4] (real part first, 16-byte aligned); cam_center: 3 floats on the DEVICE.
 * sgr_unproject_pixels: world[n,3] of the pixels picked[n] (int64 row-major indices into depth[H*W], the view-space depth image) through
 *   the reference's NDC pixel tables (sugar_model.py:1934-1941) and the inverse of `viewmatrix` (the rasterizer's 16 floats, on the
 *   DEVICE; inverted in the kernel) -- cameras.unproject_points(..., world_coordinates=True) of :1958-1959. */
int sgr_view_std(int P, const float* centers, const float* quaternions, const float* scaling, const float* cam_center, float* out,
                 void* stream);
/* out[P,3] = the view-space depth of every centre, three times: the colours of the sampler's depth render (sugar_model.py:1901-1911,
 * `depth.expand(-1, 3)`); viewmatrix: the rasterizer's 16 floats on the DEVICE. */
int sgr_view_depth_rgb(int P, const float* centers, const float* viewmatrix, float* out, void* stream);
int sgr_unproject_pixels(int n, const int64_t* picked, const float* depth, int width, int height, float tanfovx, float tanfovy,
                         const float* viewmatrix, float* world, void* stream);

/* ---- the sampling pass without host round trips (round 6; csrc/pick.hip) -----------------------------------------------------------
 * sgr_pick_pixels: sugar_model.py:1929-1957 -- of the n_pix pixels of `depth` (the view-space depth image, "no depth" < 0) those with a
 *   depth are the candidates, and a uniformly random subset of min(k, n_valid) of them is written to picked[k] (int64 row-major pixel
 *   indices, in RASTER order; the reference keeps `torch.randperm(n_valid)[:k]` of them: the same distribution of the SET, another order
 *   of the rows, which nothing behind it depends on).  *count = min(k, n_valid) and *n_valid stay on the device (either may be NULL);
 *   rows behind the count repeat picked[0] (pixel 0 when nothing is valid), so that everything computed from them stays finite.
 *   The subset is a function of (seed, which pixels are valid).  scratch: sgr_pick_pixels_scratch_bytes(n_pix) bytes.  No host wait.
 * sgr_compact_level_rows: behind sgr_level_set_points (valid[L,N], points[L,N,3], normals[L,N,3] or NULL): per level the rows with
 *   valid != 0 among the first *n_rows (device word, or NULL: all N) move to the front of rows_out[L,N] (their row numbers),
 *   points_out[L,N,3], normals_out[L,N,3] and -- for two per-row int64 tags such as the pixel and the front Gaussian of every row --
 *   tag_*_out[L,N]; counts[L] (device) receives how many.  Rows behind the count are left untouched.  scratch:
 *   sgr_compact_level_rows_scratch_bytes(N, L) bytes. */
size_t sgr_pick_pixels_scratch_bytes(int n_pix);
int sgr_pick_pixels(int n_pix, const float* depth, int k, uint32_t seed, int64_t* picked, uint32_t* count, uint32_t* n_valid, char* scratch,
                    void* stream);
int sgr_compact_level_rows(int N, int L, const uint8_t* valid, const uint32_t* n_rows, const float* points, const float* normals,
                           const int64_t* tag_a, const int64_t* tag_b, int64_t* rows_out, float* points_out, float* normals_out,
                           int64_t* tag_a_out, int64_t* tag_b_out, uint32_t* counts, char* scratch, void* stream);
size_t sgr_compact_level_rows_scratch_bytes(int N, int L);

/* ---- SuGaR.get_points_rgb, sugar_scene/sugar_model.py:839-883 (with sugar_utils/spherical_harmonics.py:117-172) -----
 * colors[P,3] = clamp_min(eval_sh(D, sh, dir) + 0.5, 0),  dir = F.normalize(positions - camera_centers) when positions is
 * given (camera_centers[n_centers,3], n_centers = 1 or P), else directions[P,3] as they are.  sh is [P,M,3] (the
 * rasterizer's layout; rows may be a view of a wider tensor: M = row stride in coefficients), D = sh_levels - 1 <= 3 and
 * the first (D+1)^2 coefficients are used.
 * Backward: dL_dsh[P,M,3] (all M rows written, zeros beyond the used ones), dL_dpositions[P,3] (positions mode) or
 * dL_ddirections[P,3] (directions mode); each may be NULL.  The clamp mask is recomputed. */
int sgr_sh_to_rgb_forward(int P, int D, int M, const float* sh, const float* positions, const float* camera_centers,
                          int n_centers, const float* directions, float* colors, void* stream);
int sgr_sh_to_rgb_backward(int P, int D, int M, const float* sh, const float* positions, const float* camera_centers,
                           int n_centers, const float* directions, const float* dL_dcolors, float* dL_dsh,
                           float* dL_dpositions, float* dL_ddirections, void* stream);

/* ---- parameter activations of the train step, gaussian_splatting/scene/gaussian_model.py:92-117 ----------------------
 * scales = exp(scaling_raw)[P,3], rotations = F.normalize(rotation_raw)[P,4] (v / max(|v|, 1e-12)), opacities =
 * sigmoid(opacity_raw)[P,1]; the backward maps gradients w.r.t. the activated values onto the raw parameters (every
 * output element is written).  rotation pointers must be 16-byte aligned. */
int sgr_activations_forward(int P, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                            float* scales, float* rotations, float* opacities, void* stream);
int sgr_activations_backward(int P, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                             const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                             float* dL_dscaling_raw, float* dL_drotation_raw, float* dL_dopacity_raw, void* stream);

/* ---- triangle-mesh z-buffer: pytorch3d.renderer.MeshRasterizer's fragments for SuGaR's level-set sampler --------------------
 * Replaces pytorch3d 0.7.4 `_C.rasterize_meshes` (pytorch3d/csrc/rasterize_meshes/rasterize_meshes.h: RasterizeMeshes(face_verts,
 * mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size, blur_radius, faces_per_pixel, bin_size,
 * max_faces_per_bin, perspective_correct, clip_barycentric_coords, cull_backfaces) -> (pix_to_face, zbuf, bary_coords, dists))
 * for ONE mesh at blur_radius == 0, as reached from sugar_scene/sugar_model.py:1880-1893,1927-1928,1966 (and :1434-1471,
 * :1661-1698, :2616-2661) and sugar_extractors/coarse_mesh.py:216-225.  pytorch3d is not part of /root/reference (a pinned
 * dependency, environment.yml:161): the rule is restated from its published sources in oracle/mesh_rasterizer.c.
 *   face_verts[F,3,3]: per face and vertex (x, y, z): x, y in pytorch3d NDC (+x left, +y up; the SHORTER image side spans
 *     [-1, 1]), z = view-space depth.  Faces with a vertex at z < 1e-8 do not render (clip them first, as pytorch3d's
 *     Python-side clip_faces does: sugar_amd/shims/pytorch3d/renderer/mesh/clip.py).
 *   outputs (row-major [height, width, faces_per_pixel], unfilled slots -1): pix_to_face = index of the face
 *     (+ face_index_base: pytorch3d numbers faces across the meshes of a batch), zbuf = its interpolated depth; per pixel the
 *     faces_per_pixel (<= 16) nearest faces whose projection strictly contains the pixel centre, ascending in z, equal z by
 *     ascending face index.  bary_coords[..,3] and dists (minus the squared NDC distance to the nearest edge) may be NULL.
 *   blur_radius must be 0 and clip_barycentric_coords false (SGR_E_INVALID otherwise); bin_size / max_faces_per_bin have no
 *     counterpart: nothing is binned away.
 *   scratch: sgr_rasterize_meshes_scratch_bytes(F, width, height) bytes of device memory; list_alloc is called once for the
 *     (tile, face) instance list after ONE host synchronisation (as sgr_forward does for its list).
 * Returns the number of (16x16 tile, face) instances, or a negative error code. */
size_t sgr_rasterize_meshes_scratch_bytes(int64_t F, int width, int height);
int64_t sgr_rasterize_meshes(const float* face_verts, int64_t F, int64_t face_index_base, int width, int height,
                             float blur_radius, int faces_per_pixel, int perspective_correct, int clip_barycentric_coords,
                             int cull_backfaces, char* scratch, size_t scratch_bytes, sgr_alloc_fn list_alloc, void* list_user,
                             int64_t* pix_to_face, float* zbuf, float* bary_coords, float* dists, void* stream);

/* The splat mesh of SuGaR's level-set sampler as face_verts for sgr_rasterize_meshes, straight from the Gaussian buffers:
 * SuGaR.triangle_vertices (sugar_scene/sugar_model.py:481-514) + SuGaR.splat_mesh(mode='perspective') (:695-716) + the vertex
 * transform of pytorch3d's MeshRasterizer (world -> view -> NDC x, y; view-space z) in one kernel.
 *   points[P,3], scaling[P,3] (activated), quaternions[P,4] (unit, real part first), primitive_verts[4,3] (the canonical
 *   corners, sugar_model.py:242-249, device), triangle_scale (:262), world_to_view[16] / projection[16]: the camera's
 *   get_world_to_view_transform() / get_projection_transform() matrices in pytorch3d's row-vector convention (device).
 *   face_verts[2P,3,3]: faces 2g and 2g+1 belong to Gaussian g (corner order [0,2,1], [0,3,2], :254-256). */
int sgr_splat_mesh_face_verts(int P, const float* points, const float* scaling, const float* quaternions,
                              const float* primitive_verts, float triangle_scale, const float* world_to_view,
                              const float* projection, float* face_verts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUGAR_RASTER_H_INCLUDED */
