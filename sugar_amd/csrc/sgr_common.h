// sgr_common.h -- shared declarations of the gfx950 rasterizer (private to sugar_amd/csrc).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/sugar_raster.h"  // SGR_HDR_*, the option structs

#define SGR_TILE_X 16  // BLOCK_X, DGR/cuda_rasterizer/config.h:16 (part of the pixel-exact contract)
#define SGR_TILE_Y 16  // BLOCK_Y, DGR/cuda_rasterizer/config.h:17
#define SGR_TILE_PIX 256
#define SGR_NUM_CUS 256              // MI355X
#define SGR_LEGACY_LDS_BYTES (150 * 1024)
#define SGR_BIN_SLICES 1024          // slices of the depth order in the ordered binning (one T-entry LDS histogram each)

// ---- private scratch layouts -----------------------------------------------------------------
// geom  : [ GeomRec rec[P] | acc f32[P][16] | sort scratch ]   48 B / Gaussian record (AoS: one gather = 1-2 lines),
//           the backward's per-Gaussian accumulator (zeroed by each backward), and the depth sort's ping-pong
//           key/value arrays (the sorted Gaussian order stays there for the backward-free forward only)
// img   : [ final_T f32[WH] | n_contrib u32[WH] | tile_start u32[T+1] | tile_count u32[T] |
//           tile_maxc u32[T] | tile_walked u32[T] | blk_nb u32[4T] | header u32[8] | blk_hist u32[n_blocks][T] ]
// binning: [ point_list u32[R] ]
struct GeomRec {
    float x, y, cx, cy;          // pixel-space mean, conic.x, conic.y
    float cz, opacity, depth;    // conic.z, opacity, view-space depth
    int radius;                  // 0 = culled
    float r, g, b;               // colour
    uint32_t clamped;            // bit c set <=> SH colour channel c was clamped at 0
};
static_assert(sizeof(GeomRec) == 48, "GeomRec must be 48 bytes");

static inline size_t sgr_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline size_t sgr_geom_acc_offset(int P) { return sgr_align((size_t)(P > 0 ? P : 1) * 48); }
size_t sgr_sort_scratch_bytes(int P);  // binning.hip
#define SGR_ACC_STRIDE 16  // floats per Gaussian in the backward's accumulator table: 64-byte records, nine used
                           // (one record = one half cache line: the nine atomics of a (block, Gaussian) pair coalesce)
static inline size_t sgr_geom_sort_offset(int P) { return sgr_geom_acc_offset(P) + sgr_align((size_t)(P > 0 ? P : 1) * SGR_ACC_STRIDE * 4); }
size_t sgr_sort_rects_offset(int P);    // binning.hip: offset of the packed rectangles inside the sort scratch
static inline size_t sgr_geom_total(int P) { return sgr_geom_sort_offset(P) + sgr_sort_scratch_bytes(P); }

struct ImgLayout {
    size_t final_T, n_contrib, tile_start, tile_cursor, tile_maxc, tile_walked, blk_nb, header, repair_flag, repair_list, deep_list, blk_hist, total;
    int n_blocks;      // slices of the depth order in the ordered binning
    int gx, gy, T;
};
static inline ImgLayout sgr_img_layout(int W, int H)
{
    ImgLayout L;
    L.gx = (W + SGR_TILE_X - 1) / SGR_TILE_X;
    L.gy = (H + SGR_TILE_Y - 1) / SGR_TILE_Y;
    L.T = L.gx * L.gy;
    size_t off = 0;
    L.final_T = off;     off = sgr_align(off + (size_t)W * H * 4);
    L.n_contrib = off;   off = sgr_align(off + (size_t)W * H * 4);
    L.tile_start = off;  off = sgr_align(off + (size_t)(L.T + 1) * 4);
    L.tile_cursor = off; off = sgr_align(off + (size_t)L.T * 4);
    L.tile_maxc = off;   off = sgr_align(off + (size_t)L.T * 4);
    L.tile_walked = off; off = sgr_align(off + (size_t)L.T * 4);
    L.blk_nb = off;      off = sgr_align(off + (size_t)L.T * 16);  // batches the forward walked, per block
    L.header = off;      off = sgr_align(off + 64);
    // walk-hint repair (round 5): per tile 0 / 0xFFFFFFFF "this tile outran its hint" (read by the second list-write pass as ITS
    // walk hint) and the list of those tiles (the second blend pass's launch order); their count is header word SGR_HDR_REPAIR
    L.repair_flag = off; off = sgr_align(off + (size_t)L.T * 4);
    L.repair_list = off; off = sgr_align(off + (size_t)L.T * 4);
    L.deep_list = off;   off = sgr_align(off + (size_t)L.T * 4);   // tiles whose blocks k_blend_fwd_deep takes (count: header word SGR_HDR_DEEP)
    // the single-level fallback keeps one LDS counter per tile: beyond ~38 000 tiles (8K images) only the two-level path exists
    L.n_blocks = ((size_t)L.T * 4 <= SGR_LEGACY_LDS_BYTES) ? SGR_BIN_SLICES : 0;
    L.blk_hist = off;    off = sgr_align(off + (size_t)L.n_blocks * L.T * 4);
    L.total = off;
    return L;
}
// two-level binning (binning2.hip): scratch appended to the img allocation
#define SGR_SUP 8                // tiles per super-tile edge
#define SGR_SUP_SHIFT 3
#ifndef SGR_B2_SLICES
#define SGR_B2_SLICES 2048       // slices of the depth order in the level-1 ordered scatter
#endif
#define SGR_B2_CHUNK 512         // level-1 list entries per level-2 wave
#define SGR_B2_HDR_R1 0          // level-1 entries (Gaussian x super-tile)
#define SGR_B2_HDR_CHUNKS 1      // level-2 chunks
#define SGR_B2_HDR_OVERFLOW 2    // 1: the level-1 list does not fit its capacity -> the caller falls back to binning.hip
struct Bin2Layout {
    size_t hist1, sup_count, sup_start, chunk_base, hdr, L1, cnt2, chunk_sup, total;
    int sgx, sgy, T1, per_slice;
    uint32_t cap1, chunk_cap;
};
Bin2Layout sgr_bin2_layout(int P, int gx, int gy);
// hdr: three words (SGR_B2_HDR_*) next to the image header, so that one device-to-host copy brings both
void sgr_launch_bin2_count(int P, int gx, int gy, const Bin2Layout& L, char* scratch, uint32_t* hdr, const uint2* rects,
                           const uint32_t* order, uint32_t* tile_count, uint32_t chunk_grid, hipStream_t s);
void sgr_launch_bin2_write(int gx, int gy, const Bin2Layout& L, char* scratch, const uint32_t* hdr, uint32_t n_chunks, const uint2* rects,
                           const uint32_t* order, const uint32_t* tile_start, uint32_t* point_list, uint32_t list_cap,
                           const uint32_t* tile_need, hipStream_t s, const uint32_t* gate = nullptr, uint32_t max_grid = 0u);
// (gate: a device word; the pass is a no-op when it is zero -- the repair pass of the walk hint, see capi.hip)

// binning: [ point_list u32[R] | blk_mask u64[(R/64 + T + 1) * 4] ]   blk_mask: per 64-entry batch of every tile's list and per
//           8x8 block of the tile, the lanes (entries) that survive the block's exact cull -- written by the forward blend
//           for the backward; batch b of tile t (list start r0) sits in slot (r0 >> 6) + t + b
struct BinLayout { size_t point_list, blk_mask, total; };
#define SGR_BIN_MASK_OFFSET(cap) ((((size_t)(cap) * 4 + 255) / 256) * 256)  /* = sgr_bin_layout(cap, .).blk_mask, usable on the device */
static inline BinLayout sgr_bin_layout(int64_t R, int T)
{
    BinLayout L;
    size_t off = 0;
    L.point_list = off; off = sgr_align(off + (size_t)R * 4);
    L.blk_mask = off;   off = sgr_align(off + ((size_t)(R >> 6) + (size_t)T + 2) * 32);
    L.total = off < 256 ? 256 : off;
    return L;
}

// header words: SGR_HDR_* of include/sugar_raster.h (0 R, 1 largest tile count, 2 R high word, 3 hint miss; words 4-6 are
// the two-level binning's SGR_B2_HDR_*, written by k_sup_scan and cleared by the tile scan on the single-level path)

// stage ids of the optional event profile (sgr_profile_read)
enum { SGR_STAGE_PREPROCESS = 0, SGR_STAGE_SCAN /* bin_count + scans */, SGR_STAGE_SCATTER, SGR_STAGE_SORT /* depth sort */, SGR_STAGE_BLEND_FWD,
       SGR_STAGE_BLEND_BWD, SGR_STAGE_PREPROCESS_BWD, SGR_STAGE_HINT_REPAIR /* the two gated launches behind the blend */,
       // the rest of the native train step (train.hip): with these the stages cover every launch of sgr_trainer_step
       SGR_STAGE_FWD_POST /* post-blend bookkeeping when it is a launch of its own */, SGR_STAGE_LOSS_FWD, SGR_STAGE_LOSS_BWD,
       SGR_STAGE_FILL /* the backward's accumulator reset */, SGR_STAGE_SH_ADAM, SGR_STAGE_ADAM, SGR_STAGE_MASKED_COLORS, SGR_STAGE_COUNT };
// HIP events around a stage on the caller's stream when sgr_profile_enable() selected it (capi.hip); a no-op otherwise
struct SgrStageTimer {
    hipStream_t s; int stage; hipEvent_t a = nullptr, b = nullptr; bool live = false;
    SgrStageTimer(hipStream_t s_, int stage_);
    void stop();
    ~SgrStageTimer() { stop(); }
};

// ---- kernel launchers (defined in the .hip translation units) --------------------------------
struct PreprocessArgs {
    int P, D, M;
    const float* means3D; const float* scales; float scale_modifier; const float* rotations;
    const float* opacities; const float* shs; const float* cov3D_precomp; const float* colors_precomp;
    const float* viewmatrix; const float* projmatrix; const float* cam_pos;
    int W, H; float tan_fovx, tan_fovy, focal_x, focal_y;
    int gx, gy;
    int* radii; GeomRec* rec;
    int raw_params;       // scales / rotations / opacities are the RAW parameters (log scale, unnormalised quaternion, logit)
    uint32_t* sort_keys;  // [P] depth bits of visible Gaussians, 0xFFFFFFFF for culled ones (input of the depth sort)
    uint2* rect_by_id;    // [P] packed tile rectangle of every Gaussian (w == 0: culled)
    uint2* key_minmax;    // [ceil(P / 256)] smallest / largest depth key of every workgroup's visible Gaussians
    uint32_t* sort_counters; int n_sort_counters;  // zeroed by workgroup 0 (histograms and tickets of the depth sort)
    int keys_elsewhere = 0;  // 1: sgr_launch_depth_keys writes sort_keys / key_minmax and zeroes the counters (the sort overlaps this kernel)
    uint32_t* zero_words = nullptr; int n_zero_words = 0;  // more words to zero, by the last workgroup (the walk hint's repair flags:
                                                           // as a store stream of the single-workgroup tile scan they cost it 9 us)
};
void sgr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);
void sgr_launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s);
void sgr_launch_depth_keys(const PreprocessArgs& a, hipStream_t s);  // sort keys + key ranges + counter reset only (preprocess.hip)

struct PreprocessBwdArgs {
    int P, D, M;
    const float* means3D; const float* shs; const float* scales; const float* rotations; float scale_modifier;
    const float* cov3D_precomp; const float* viewmatrix; const float* projmatrix; const float* cam_pos;
    int W, H; float tan_fovx, tan_fovy, focal_x, focal_y;
    const GeomRec* rec;
    int raw_params;    // as in PreprocessArgs: dL_dscale / dL_drot / dL_dopacity are then gradients w.r.t. the raw parameters
    int sh_dir_elsewhere;  // compact mode only: skip the SH block (dRGB/d(view direction) -> dL_dmean3D is formed by k_sh_adam_from_views)
    float* campos_row;  // compact mode: receives the camera centre (the row behind the colour gradients in a send buffer) or NULL
    float* dens_max_radii; float* dens_accum; float* dens_denom;  // fused densification statistics (sgr_backward_opts) or NULL
    const uint32_t* header; uint32_t list_cap;  // the forward's device header (or NULL): an INVALID sync-free forward makes the kernel a no-op
    const float* acc;  // [P][SGR_ACC_STRIDE] sums from the blend backward: {dcol r,g,b, S0, Sx, Sy, Sxx, Sxy, Syy, pad x3}
    float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolor;  // written here from acc
    float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale; float* dL_drot;
};
void sgr_launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s);
void sgr_launch_masked_colors(int P, const GeomRec* rec, const float* acc, float* out, const float* cam_pos, float* campos_row, hipStream_t s);
void sgr_launch_sh_grad_from_views(int P, int V, int D, int M, size_t vstride, const float* means3D, const float* campos,
                                   const float* dcolor, float* dL_dsh, hipStream_t s);

// torch.optim.Adam forms `1 - beta` and `1 - beta ** step` in DOUBLE from the Python float the caller wrote (0.999); this C ABI
// carries floats (0.999f = 0.99900001287...), and `1.f - 0.999f` is off by 1.3e-5 relative.  The coefficient helpers recover the
// shortest decimal that rounds to the float -- what repr() would print -- and do the arithmetic in double like torch.
static inline double sgr_intended_double(float x)
{
    static thread_local float memo_x[2] = {0.f, 0.f};
    static thread_local double memo_d[2] = {0.0, 0.0};
    for (int k = 0; k < 2; k++) if (memo_x[k] == x && memo_x[k] != 0.f) return memo_d[k];
    double out = (double)x;
    char buf[48];
    for (int p = 1; p <= 9; p++) {
        snprintf(buf, sizeof buf, "%.*g", p, (double)x);
        const double d = strtod(buf, nullptr);
        if ((float)d == x) { out = d; break; }
    }
    memo_x[1] = memo_x[0]; memo_d[1] = memo_d[0]; memo_x[0] = x; memo_d[0] = out;
    return out;
}
static inline float sgr_one_minus(float beta) { return (float)(1.0 - sgr_intended_double(beta)); }
static inline void sgr_bias_corrections(float beta1, float beta2, int step, float* bc1, float* bc2_sqrt)
{
    *bc1 = (float)(1.0 - pow(sgr_intended_double(beta1), (double)step));
    *bc2_sqrt = (float)sqrt(1.0 - pow(sgr_intended_double(beta2), (double)step));
}

void sgr_launch_sh_adam_from_views(int P, int V, int D, int M, size_t vstride, const float* means3D, const float* campos,
                                   const float* dcolor, float* sh, float* exp_avg, float* exp_avg_sq, float lr_dc, float lr_rest, float b1, float b2,
                                   float eps, float bc1, float bc2_sqrt, float grad_scale, float* dmean_extra, hipStream_t s,
                                   const uint32_t* guard = nullptr, uint32_t guard_cap = 0);
// sgr_adam_step_ex with the same guard (adam.hip)
int sgr_adam_launch(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_seg,
                    const long long* seg_begin, const long long* seg_end, const float* seg_lr_a, const float* seg_lr_b,
                    const int* seg_period, const int* seg_split, float beta1, float beta2, float eps, int step, float grad_scale,
                    const float* extra, long long extra_n, const uint32_t* guard, uint32_t guard_cap, hipStream_t s);

// part: 0 the whole sort; 1 histogram kernel + passes 0 and 1 (keys only); 2 passes 2 and 3 (the last one carries the rectangles)
void sgr_launch_gaussian_sort(int P, char* sort_scratch, const uint32_t** order_out, const uint2* rect_by_id, uint2* rects_sorted,
                              hipStream_t s, int part = 0);
size_t sgr_sort_rect_by_id_offset(int P);  // binning.hip: the by-id rectangles written by the preprocess kernel
size_t sgr_sort_minmax_offset(int P);      // binning.hip: the per-workgroup key ranges written by the preprocess kernel
size_t sgr_sort_counters_offset(int P);    // binning.hip: the counters the preprocess kernel zeroes for the sort
int sgr_sort_counter_words();
void sgr_launch_bin_count(int P, int gx, int gy, int n_slices, int per_slice, const uint32_t* order, const uint2* rects,
                          uint32_t* blk_hist, hipStream_t s);
void sgr_launch_bin_scatter(int P, int gx, int gy, int n_slices, int per_slice, const uint32_t* order, const uint2* rects,
                            const uint32_t* tile_start, uint32_t* blk_hist, uint32_t* point_list, hipStream_t s);
void sgr_launch_hist_scan(int T, int n_blocks, uint32_t* blk_hist, uint32_t* tile_count, hipStream_t s);
void sgr_launch_tile_scan(int T, const uint32_t* tile_count, uint32_t* tile_start, uint32_t* header, uint32_t* tile_maxc,
                          uint32_t* tile_walked, int clear_b2_words, uint32_t* host_a, uint32_t* host_b, hipStream_t s,
                          uint32_t* repair_flag = nullptr);  // host_*: device-mapped pinned memory or NULL; repair_flag[T] is zeroed

// header: the forward's device header; list_cap: instances the list was allocated for (the forward is a no-op when the header
// says it does not fit, the backward when the forward was one); tile_need / tile_need_out: walk hint (sgr_forward_opts)
void sgr_launch_blend_fwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const GeomRec* rec, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc,
                          uint32_t* tile_walked, float* out_color, unsigned long long* blk_mask, uint32_t* blk_nb,
                          uint32_t* header, uint32_t list_cap, const uint32_t* tile_need, const uint32_t* launch_order, hipStream_t s,
                          uint32_t* repair_flag = nullptr, uint32_t* repair_list = nullptr, int exact = 0, uint32_t deep_min = 0u);
// blocks of tiles whose hinted list is longer than deep_min entries: eight waves per block (blend.hip); deep_list: [T] words of scratch
void sgr_launch_deep_list(int gx, int gy, const uint32_t* tile_start, const uint32_t* tile_need, uint32_t deep_min, uint32_t* header,
                          uint32_t list_cap, uint32_t* deep_list, hipStream_t s);
void sgr_launch_blend_fwd_deep(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list, const GeomRec* rec,
                               const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc, uint32_t* tile_walked,
                               float* out_color, unsigned long long* blk_mask, uint32_t* blk_nb, uint32_t* header, uint32_t list_cap,
                               const uint32_t* tile_need, uint32_t* deep_list, uint32_t deep_min, hipStream_t s, uint32_t* repair_flag,
                               uint32_t* repair_list, int exact);
// exact: the exact-alpha kernels (blend.hip: SGR_FWD_BODY_X); process-wide default sgr_exact_alpha() (capi.hip)
int sgr_exact_alpha();
uint32_t sgr_deep_min();   // capi.hip: process-wide threshold of the eight-wave kernel (sgr_set_deep_min)
// Walk-hint repair: the tiles the first pass listed (they outran their hint) once more, over their full lists (written meanwhile
// by a list-write pass gated on the same count).  A no-op launch when the list is empty.
#define SGR_REPAIR_TILES 1024
void sgr_launch_blend_fwd_repair(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                                 const GeomRec* rec, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc,
                                 uint32_t* tile_walked, float* out_color, unsigned long long* blk_mask, uint32_t* blk_nb,
                                 uint32_t* header, uint32_t list_cap, const uint32_t* repair_list, hipStream_t s, int exact = 0);
void sgr_launch_blend_fwd_post(int gx, int gy, const uint32_t* tile_maxc, const uint32_t* tile_walked, uint32_t* header, uint32_t list_cap,
                               uint32_t* tile_need_out, float hint_margin, uint32_t* header_host_dev, uint32_t* order_scratch,
                               uint32_t* order_out, hipStream_t s);
// binning: the binning buffer's base; the survivor masks sit at sgr_bin_layout(header[SGR_HDR_LAYOUT_CAP]).blk_mask (device-side)
void sgr_launch_blend_bwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const char* binning, const uint32_t* blk_nb, const GeomRec* rec, const float* bg,
                          const float* final_T, const uint32_t* n_contrib, const float* dL_dpix, float* acc,
                          const uint32_t* tile_maxc, const uint32_t* header, uint32_t list_cap, uint32_t* tile_order, int order_ready,
                          hipStream_t s, int exact = 0);
// the post-blend bookkeeping as a job another kernel can carry (tile_order.h); capi.hip fills it for a forward that was run with
// SGR_FLAG_DEFER_POST, loss.hip's forward kernel executes it in a spare workgroup
struct SgrTileOrderJob;
extern "C" int sgr_l1_ssim_forward_job(int channels, int width, int height, const float* img, const float* gt, float lambda, char* scratch,
                                       float* loss_out, const SgrTileOrderJob* job, void* stream);
extern "C" int sgr_forward_post_job(int width, int height, char* img_buffer, int64_t R, const sgr_forward_opts* opts, SgrTileOrderJob* job);
// true on the device when the forward that wrote `hdr` must be treated as not having happened
#define SGR_FORWARD_INVALID(hdr, cap) ((hdr)[SGR_HDR_R] > (cap) || (hdr)[SGR_HDR_HINT_MISS] != 0u || (hdr)[4 + SGR_B2_HDR_OVERFLOW] != 0u)
