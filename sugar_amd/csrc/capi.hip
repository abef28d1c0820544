// capi.hip -- host orchestration and the C ABI of include/sugar_raster.h.
//
// Mirrors the call structure of CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (DGR/cuda_rasterizer/rasterizer_impl.cu:141-153, 198-336, 340-434) on a caller-supplied HIP stream.
#include "../../include/sugar_raster.h"
#include "sgr_common.h"
#include "tile_order.h"

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(SGR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

// debug mode = the reference's CHECK_CUDA (auxiliary.h:166-173): synchronise and check after every stage
#define STAGE_CHECK(what)                                                                               \
    do {                                                                                                \
        hipError_t e_ = hipGetLastError();                                                              \
        if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize(s);                                    \
        if (e_ != hipSuccess) return fail(SGR_E_HIP, std::string(what) + ": " + hipGetErrorString(e_)); \
    } while (0)

// one small pinned readback slot per calling thread (the forward reads R and the max tile count once)
struct Pinned {
    uint32_t* p = nullptr;
    ~Pinned() { /* leaked on purpose: the HIP runtime may already be gone at thread exit */ }
};
thread_local Pinned g_pinned;

// Speculative forward (SGR_FLAG_SPECULATIVE): the instance list is laid out for the caller's guess of the capacity, while the call
// returns the true instance count, and the backward is handed that count (rasterizer.h:57-84: "R = the value forward returned").
// The layout is SELF-DESCRIBING: the forward blend kernel leaves the capacity its list was laid out for in the image scratch's
// device header (word SGR_HDR_LAYOUT_CAP) and the backward blend kernel finds the survivor masks from there -- the point list
// itself starts at offset 0 of the binning buffer for every capacity.  (Until round 5 a process-global map keyed by the binning
// buffer's ADDRESS remembered the capacity: a caller that cloned or moved the opaque buffer got silently wrong gradients.)
struct ScanEvent {
    hipEvent_t e = nullptr;
    ~ScanEvent() { /* leaked on purpose, like the pinned slot */ }
};
thread_local ScanEvent g_scan_ev;

// the side stream of the forward (depth keys + the first half of the depth sort beside the preprocess kernel), one per calling
// thread and device; leaked on purpose like the pinned slot
struct SideStream { hipStream_t st = nullptr; hipEvent_t fork = nullptr, join = nullptr; int device = -1; };
thread_local SideStream g_side[16];
SideStream* side_stream()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream& sd = g_side[dev];
    if (!sd.st) {
        // (highest priority: the sort's few, latency-bound workgroups should get their slots ahead of the preprocess kernel's thousands)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&sd.st, hipStreamNonBlocking, hi) != hipSuccess) { sd.st = nullptr; (void)hipGetLastError(); return nullptr; }
        if (hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        sd.device = dev;
    }
    return (sd.fork && sd.join) ? &sd : nullptr;
}

// ---- optional per-stage timing with HIP events on the caller's stream (bench.py's roofline leg) ----------
struct Prof {
    unsigned mask = 0;  // bit s set: time stage s
    struct Rec { int stage; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Prof g_prof;            // process-wide: autograd runs the backward on its own thread
std::mutex g_prof_mu;

}  // namespace

// the per-stage event timer of sgr_common.h (used by capi.hip and train.hip)
SgrStageTimer::SgrStageTimer(hipStream_t s_, int stage_) : s(s_), stage(stage_)
{
    if (g_prof.mask & (1u << stage_)) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof.recs.size() < 65536) {
            a = g_prof.get(); b = g_prof.get();
            live = a && b;
        }
        if (live) (void)hipEventRecord(a, s);
    }
}
void SgrStageTimer::stop()
{
    if (live) {
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.recs.push_back({stage, a, b});
        live = false;
    }
}

// error reporting for the other translation units (mesh_raster.hip): same thread-local message as sgr_last_error()
int sgr_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }

// process-wide exact-alpha mode (include/sugar_raster.h: sgr_set_exact_alpha): ON unless SGR_EXACT_ALPHA=0 is in the environment
static int g_exact_alpha = -1;
int sgr_exact_alpha()
{
    if (g_exact_alpha < 0) { const char* e = getenv("SGR_EXACT_ALPHA"); g_exact_alpha = (e && e[0] == '0') ? 0 : 1; }
    return g_exact_alpha;
}

// hinted list length above which a tile's blocks go to the eight-wave blend kernel (0: never); SGR_DEEP_MIN in the environment
static int g_deep_min = -1;
uint32_t sgr_deep_min()
{
    // (default 0 = off: as measured in round 6 the eight-wave kernel does not beat the one-wave walk yet -- profiles/r06_blend_fwd_deep_lists_ab.txt)
    if (g_deep_min < 0) { const char* e = getenv("SGR_DEEP_MIN"); g_deep_min = e ? atoi(e) : 0; if (g_deep_min < 0) g_deep_min = 0; }
    return (uint32_t)g_deep_min;
}

extern "C" {

int sgr_abi_version(void) { return SGR_ABI_VERSION; }
void sgr_set_deep_min(int entries) { g_deep_min = entries > 0 ? entries : 0; }
int sgr_get_deep_min(void) { return (int)sgr_deep_min(); }
void sgr_set_exact_alpha(int on) { g_exact_alpha = on ? 1 : 0; }
int sgr_get_exact_alpha(void) { return sgr_exact_alpha(); }

const char* sgr_last_error(void) { return g_err.c_str(); }

size_t sgr_geom_bytes(int P) { return sgr_geom_total(P); }
size_t sgr_img_bytes(int width, int height) { return sgr_img_layout(width, height).total; }
size_t sgr_binning_bytes(int64_t R, int width, int height) { return sgr_bin_layout(R, sgr_img_layout(width, height).T).total; }
size_t sgr_geom_rec_offset(int) { return 0; }
size_t sgr_img_final_T_offset(int w, int h) { return sgr_img_layout(w, h).final_T; }
size_t sgr_img_n_contrib_offset(int w, int h) { return sgr_img_layout(w, h).n_contrib; }
size_t sgr_img_tile_start_offset(int w, int h) { return sgr_img_layout(w, h).tile_start; }
size_t sgr_img_tile_maxc_offset(int w, int h) { return sgr_img_layout(w, h).tile_maxc; }
size_t sgr_img_tile_walked_offset(int w, int h) { return sgr_img_layout(w, h).tile_walked; }
size_t sgr_img_header_offset(int w, int h) { return sgr_img_layout(w, h).header; }
size_t sgr_binning_point_list_offset(int64_t R) { return sgr_bin_layout(R, 0).point_list; }

void sgr_profile_enable(int stage_mask) { g_prof.mask = (unsigned)stage_mask; }

int sgr_profile_read(double* ms_sum, int64_t* count, int n_stages)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < n_stages; i++) { ms_sum[i] = 0.0; count[i] = 0; }
    for (auto& r : g_prof.recs) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e == hipSuccess && r.stage < n_stages) { ms_sum[r.stage] += ms; count[r.stage]++; }
        g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b);
    }
    g_prof.recs.clear();
    return 0;
}

int sgr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
    (void)projmatrix;
    hipStream_t s = (hipStream_t)stream;
    if (P < 0) return fail(SGR_E_INVALID, "P < 0");
    sgr_launch_mark_visible(P, means3D, viewmatrix, present, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SGR_E_HIP, std::string("mark_visible: ") + hipGetErrorString(e));
    return 0;
}

int64_t sgr_forward_ex(sgr_alloc_fn geom_alloc, void* geom_user, sgr_alloc_fn binning_alloc, void* binning_user,
                       sgr_alloc_fn img_alloc, void* img_user, int P, int D, int M, const float* background, int width,
                       int height, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                       float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii, int debug,
                       void* stream, const sgr_forward_opts* opts)
{
    // prefiltered: the reference only uses it to abort the process when a Gaussian the caller promised to be visible is
    // culled (auxiliary.h:156-160: printf + __trap); a library must not kill its host, so the promise is not checked
    (void)prefiltered;
    hipStream_t s = (hipStream_t)stream;
    static const sgr_forward_opts no_opts = {0, 0, nullptr, nullptr, nullptr, nullptr, 0.f, 0u, nullptr};
    if (!opts) opts = &no_opts;
    int64_t binning_capacity = opts->binning_capacity;
    const int flags = opts->flags;
    const int binning_mode = (flags & SGR_FLAG_SINGLE_LEVEL_BINNING) ? 1 : 0;
    const int exact = (sgr_exact_alpha() || (flags & SGR_FLAG_EXACT_ALPHA)) ? 1 : 0;
    if (opts->tile_need && binning_mode == 1) return fail(SGR_E_INVALID, "the walk hint needs the two-level binning");
    if (P <= 0 || width <= 0 || height <= 0) return fail(SGR_E_INVALID, "P, width and height must be positive");
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !background || !out_color)
        return fail(SGR_E_INVALID, "null required pointer");
    if (!shs && !colors_precomp) return fail(SGR_E_INVALID, "need SHs or precomputed colours");
    if (shs && !colors_precomp && (M <= 0 || M > 16 || (D + 1) * (D + 1) > M || D < 0 || D > 3))
        return fail(SGR_E_INVALID, "SH degree / coefficient count out of range (D <= 3, (D+1)^2 <= M <= 16)");
    if (shs && !colors_precomp && !cam_pos) return fail(SGR_E_INVALID, "cam_pos required with SHs");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(SGR_E_INVALID, "need scales+rotations or cov3D_precomp");

    const ImgLayout IL = sgr_img_layout(width, height);
    const bool legacy_ok = IL.n_blocks > 0;  // one LDS counter per tile fits (about 38 000 tiles)
    if (IL.gx > 65535 || IL.gy > 65535) return fail(SGR_E_INVALID, "image too large: more than 65535 tiles per axis");
    if (!legacy_ok && binning_mode == 1)
        return fail(SGR_E_INVALID, "image too large for the single-level binning (one LDS counter per tile, about 38 000 tiles)");
    const Bin2Layout B2 = sgr_bin2_layout(P, IL.gx, IL.gy);
    char* geom = geom_alloc(geom_user, sgr_geom_bytes(P));
    char* img = img_alloc(img_user, IL.total + B2.total);  // [ image state | two-level binning scratch ]
    if (!geom || !img) return fail(SGR_E_ALLOC, "geometry/image scratch allocation failed");

    GeomRec* rec = reinterpret_cast<GeomRec*>(geom);
    float* final_T = reinterpret_cast<float*>(img + IL.final_T);
    uint32_t* n_contrib = reinterpret_cast<uint32_t*>(img + IL.n_contrib);
    uint32_t* tile_start = reinterpret_cast<uint32_t*>(img + IL.tile_start);
    uint32_t* tile_cursor = reinterpret_cast<uint32_t*>(img + IL.tile_cursor);
    uint32_t* tile_maxc = reinterpret_cast<uint32_t*>(img + IL.tile_maxc);
    uint32_t* tile_walked = reinterpret_cast<uint32_t*>(img + IL.tile_walked);
    uint32_t* blk_nb = reinterpret_cast<uint32_t*>(img + IL.blk_nb);
    uint32_t* header = reinterpret_cast<uint32_t*>(img + IL.header);
    uint32_t* repair_flag = reinterpret_cast<uint32_t*>(img + IL.repair_flag);
    uint32_t* repair_list = reinterpret_cast<uint32_t*>(img + IL.repair_list);

    uint32_t* blk_hist = reinterpret_cast<uint32_t*>(img + IL.blk_hist);
    char* sort_scratch = geom + sgr_geom_sort_offset(P);
    const int per_block = legacy_ok ? (((P + IL.n_blocks - 1) / IL.n_blocks + 63) / 64) * 64 : 0;  // Gaussians per slice
    uint2* rects = reinterpret_cast<uint2*>(sort_scratch + sgr_sort_rects_offset(P));

    // host-visible header targets: the caller's pinned memory and this thread's own read-back slot, as device pointers
    uint32_t* hh_dev = nullptr;
    if (opts->header_host && hipHostGetDevicePointer(reinterpret_cast<void**>(&hh_dev), opts->header_host, 0) != hipSuccess) {
        hh_dev = nullptr;
        (void)hipGetLastError();  // not pinned / not mapped: the copy-command path below
    }
    if ((flags & SGR_FLAG_DEFER_POST) && opts->header_host && !hh_dev)  // (checked before anything is enqueued)
        return fail(SGR_E_INVALID, "SGR_FLAG_DEFER_POST needs a device-mapped header_host");
    // (SGR_NO_HINT_REPAIR=1: a tile that outruns its walk hint invalidates the forward, as until round 4 -- an A/B switch)
    static const bool hint_repair = getenv("SGR_NO_HINT_REPAIR") == nullptr;
    PreprocessArgs pa;
    pa.P = P; pa.D = D; pa.M = shs ? M : 0;
    pa.means3D = means3D; pa.scales = scales; pa.scale_modifier = scale_modifier; pa.rotations = rotations;
    pa.opacities = opacities; pa.shs = shs; pa.cov3D_precomp = cov3D_precomp; pa.colors_precomp = colors_precomp;
    pa.viewmatrix = viewmatrix; pa.projmatrix = projmatrix; pa.cam_pos = cam_pos;
    pa.W = width; pa.H = height; pa.tan_fovx = tan_fovx; pa.tan_fovy = tan_fovy;
    pa.focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:222-223
    pa.focal_x = width / (2.0f * tan_fovx);
    pa.gx = IL.gx; pa.gy = IL.gy;
    pa.raw_params = (flags & SGR_FLAG_RAW_PARAMS) && !cov3D_precomp;
    pa.radii = radii; pa.rec = rec; pa.sort_keys = reinterpret_cast<uint32_t*>(sort_scratch);
    pa.rect_by_id = reinterpret_cast<uint2*>(sort_scratch + sgr_sort_rect_by_id_offset(P));
    pa.key_minmax = reinterpret_cast<uint2*>(sort_scratch + sgr_sort_minmax_offset(P));
    pa.sort_counters = reinterpret_cast<uint32_t*>(sort_scratch + sgr_sort_counters_offset(P)); pa.n_sort_counters = sgr_sort_counter_words();
    if (opts->tile_need && hint_repair) { pa.zero_words = repair_flag; pa.n_zero_words = IL.T; }  // (the walk hint's repair flags start from zero)
    // Sort beside preprocess (round 6): the depth keys come from a 12-byte-per-Gaussian kernel of their own on a side stream, which
    // also runs the sort's histogram kernel and its first two passes while the HBM-bound preprocess kernel streams on the caller's;
    // the caller's stream joins before the sort's last pass (it carries the rectangles the preprocess kernel writes).
    // (OFF by default: measured on one box over 3 x 100 steps each, the sort stage shrinks from 77 to 40 us but the step does not --
    // 1.083 ms with the overlap against 1.074 without: the two cross-stream event waits and the contention on the preprocess
    // kernel eat what the overlap hides; profiles/r06_sort_overlap_ab.txt.  SGR_SORT_OVERLAP=1 enables it.)
    static const bool overlap_ok = getenv("SGR_SORT_OVERLAP") != nullptr && getenv("SGR_SORT_OVERLAP")[0] == '1';
    SideStream* side = (overlap_ok && !debug && P >= 65536) ? side_stream() : nullptr;
    const uint32_t* order = nullptr;
    if (side) {
        HIP_TRY(hipEventRecord(side->fork, s));
        HIP_TRY(hipStreamWaitEvent(side->st, side->fork, 0));
        sgr_launch_depth_keys(pa, side->st);
        sgr_launch_gaussian_sort(P, sort_scratch, &order, pa.rect_by_id, rects, side->st, 1);
        HIP_TRY(hipEventRecord(side->join, side->st));
        pa.keys_elsewhere = 1;
    }
    { SgrStageTimer t(s, SGR_STAGE_PREPROCESS); sgr_launch_preprocess_fwd(pa, s); }
    STAGE_CHECK("preprocess");
    {
        SgrStageTimer t(s, SGR_STAGE_SORT);
        if (side) HIP_TRY(hipStreamWaitEvent(s, side->join, 0));
        sgr_launch_gaussian_sort(P, sort_scratch, &order, pa.rect_by_id, rects, s, side ? 2 : 0);
    }
    STAGE_CHECK("gaussian_sort");

    // speculative: sync-free launches with the caller's capacity, then ONE wait for the tile scan's header at the END of the call,
    // when the list-write pass and the blend kernel are already queued behind it: the host round trip of rasterizer_impl.cu:280-281
    // without the idle GPU, and the true instance count as the return value
    const bool speculative = (flags & SGR_FLAG_SPECULATIVE) && binning_capacity > 0 && binning_mode == 0 && !(flags & SGR_FLAG_DEFER_POST);
    const bool will_sync = !(binning_capacity > 0 && binning_mode == 0) || speculative;
    uint32_t* pin_dev = nullptr;
    if (will_sync) {
        if (!g_pinned.p) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g_pinned.p), 64, hipHostMallocDefault));
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_dev), g_pinned.p, 0) != hipSuccess) { pin_dev = nullptr; (void)hipGetLastError(); }
    }
    char* bin2 = img + IL.total;
    bool two_level = binning_mode == 0;
    {
        SgrStageTimer t(s, SGR_STAGE_SCAN);
        if (two_level) {
            sgr_launch_bin2_count(P, IL.gx, IL.gy, B2, bin2, header + 4, rects, order, tile_cursor,
                                  binning_capacity > 0 ? opts->chunk_grid : 0u, s);
        } else {
            sgr_launch_bin_count(P, IL.gx, IL.gy, IL.n_blocks, per_block, order, rects, blk_hist, s);
            sgr_launch_hist_scan(IL.T, IL.n_blocks, blk_hist, tile_cursor, s);
        }
        sgr_launch_tile_scan(IL.T, tile_cursor, tile_start, header, tile_maxc, tile_walked, two_level ? 0 : 1, hh_dev, pin_dev, s);
    }
    STAGE_CHECK("bin_count");
    // the header for a caller that checks late: right behind the tile scan (words 0 and 6 are final) -- written by the scan
    // kernel itself when the caller's memory is device-mapped, a copy command otherwise
    if (opts->header_host && !hh_dev) HIP_TRY(hipMemcpyAsync(opts->header_host, header, 32, hipMemcpyDeviceToHost, s));

    // Sync-free mode (binning_capacity > 0, two-level binning): no device-to-host copy of R and no host wait -- the
    // instance list gets the caller's capacity, the write pass clamps to it and the blend kernel returns at once when the
    // header says the forward is invalid (R > capacity, or level-1 overflow).  The caller reads the header later.
    const bool nosync = binning_capacity > 0 && two_level;
    if (binning_capacity > 0xFFFFFFFEll) binning_capacity = 0xFFFFFFFEll;  // (0xFFFFFFFF is the saturated count of k_tile_scan: never valid)
    if (speculative) {
        if (!pin_dev) HIP_TRY(hipMemcpyAsync(g_pinned.p, header, 32, hipMemcpyDeviceToHost, s));
        if (!g_scan_ev.e) HIP_TRY(hipEventCreateWithFlags(&g_scan_ev.e, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g_scan_ev.e, s));
    }
    int64_t R = 0;
    uint32_t n_chunks = 0;
    // everything behind the tile scan: list allocation, list-write pass, blend, post-blend bookkeeping
    char* binning = nullptr;
    auto tail = [&](int64_t R_, bool nosync_, uint32_t n_chunks_) -> int {
        const BinLayout BL = sgr_bin_layout(R_, IL.T);
        binning = binning_alloc(binning_user, BL.total);
        if (!binning) return fail(SGR_E_ALLOC, "binning scratch allocation failed");
        uint32_t* point_list = reinterpret_cast<uint32_t*>(binning + BL.point_list);
        unsigned long long* blk_mask = reinterpret_cast<unsigned long long*>(binning + BL.blk_mask);
        if (R_ > 0) {
            SgrStageTimer t(s, SGR_STAGE_SCATTER);
            if (two_level)
                sgr_launch_bin2_write(IL.gx, IL.gy, B2, bin2, header + 4, n_chunks_, rects, order, tile_start, point_list,
                                      nosync_ ? (uint32_t)R_ : 0xFFFFFFFFu, opts->tile_need, s);
            else
                sgr_launch_bin_scatter(P, IL.gx, IL.gy, IL.n_blocks, per_block, order, rects, tile_start, blk_hist, point_list, s);
        }
        STAGE_CHECK("bin_scatter");
        {
            SgrStageTimer t(s, SGR_STAGE_BLEND_FWD);
            // Long lists: the blocks of tiles whose hinted list exceeds deep_min go to the eight-wave kernel on the side stream, beside
            // the one-wave kernel (which skips exactly those).  Only with a walk hint: without one every tile's full list counts.
            const uint32_t deep_cfg = sgr_deep_min();
            SideStream* dside = (deep_cfg && opts->tile_need && two_level && !debug && !(flags & SGR_FLAG_NO_DEEP)) ? side_stream() : nullptr;
            const uint32_t deep_min = dside ? deep_cfg : 0u;
            if (dside) {
                sgr_launch_deep_list(IL.gx, IL.gy, tile_start, opts->tile_need, deep_min, header, (uint32_t)R_,
                                     reinterpret_cast<uint32_t*>(img + IL.deep_list), s);
                HIP_TRY(hipEventRecord(dside->fork, s));
                HIP_TRY(hipStreamWaitEvent(dside->st, dside->fork, 0));
                sgr_launch_blend_fwd_deep(width, height, IL.gx, IL.gy, tile_start, point_list, rec, background, final_T, n_contrib, tile_maxc,
                                          tile_walked, out_color, blk_mask, blk_nb, header, (uint32_t)R_, opts->tile_need,
                                          reinterpret_cast<uint32_t*>(img + IL.deep_list), deep_min, dside->st,
                                          hint_repair ? repair_flag : nullptr, repair_list, exact);
                HIP_TRY(hipEventRecord(dside->join, dside->st));
            }
            sgr_launch_blend_fwd(width, height, IL.gx, IL.gy, tile_start, point_list, rec, background, final_T, n_contrib,
                                 tile_maxc, tile_walked, out_color, blk_mask, blk_nb, header, (uint32_t)R_, opts->tile_need,
                                 opts->tile_order, s, hint_repair ? repair_flag : nullptr, repair_list, exact, deep_min);
            if (dside) HIP_TRY(hipStreamWaitEvent(s, dside->join, 0));
        }   // (that stage timer -- bench.py's roofline.launch_ms -- brackets k_blend_fwd_w alone; the two gated launches are a stage of their own)
        {
            SgrStageTimer t(s, SGR_STAGE_HINT_REPAIR);
            if (opts->tile_need && two_level && R_ > 0 && hint_repair) {
                // Walk-hint repair: tiles that outran their hint are on the device's repair list now.  The list-write pass once more
                // with the repair flags as ITS hint (0: nothing needed; 0xFFFFFFFF: the whole list) and the blend once more over the
                // listed tiles' full lists -- both gated on the list's count, i.e. two empty launches when every hint held.  A hint
                // that is too short used to invalidate the forward and with it the whole train step.
                sgr_launch_bin2_write(IL.gx, IL.gy, B2, bin2, header + 4, n_chunks_, rects, order, tile_start, point_list,
                                      nosync_ ? (uint32_t)R_ : 0xFFFFFFFFu, repair_flag, s, header + SGR_HDR_REPAIR, 512u);
                sgr_launch_blend_fwd_repair(width, height, IL.gx, IL.gy, tile_start, point_list, rec, background, final_T, n_contrib,
                                            tile_maxc, tile_walked, out_color, blk_mask, blk_nb, header, (uint32_t)R_, repair_list, s, exact);
            }
        }
        if (flags & SGR_FLAG_DEFER_POST) {  // (the caller's next kernel carries the post-blend job: sgr_forward_post_job)
            STAGE_CHECK("blend_fwd");
            return 0;
        }
        {
            SgrStageTimer t(s, SGR_STAGE_FWD_POST);
            sgr_launch_blend_fwd_post(IL.gx, IL.gy, tile_maxc, tile_walked, header, (uint32_t)R_, opts->tile_need_out, opts->hint_margin, hh_dev,
                                      tile_cursor, opts->tile_order_out, s);
        }
        STAGE_CHECK("blend_fwd");
        // ... and once more behind the blend: word 3 (hint miss) is final only now
        if (opts->header_host && !hh_dev) HIP_TRY(hipMemcpyAsync(opts->header_host + 8, header, 32, hipMemcpyDeviceToHost, s));
        if (opts->header_event) HIP_TRY(hipEventRecord((hipEvent_t)opts->header_event, s));
        return 0;
    };
    // the host round trip of the forward (rasterizer_impl.cu:280-281) and what hangs on it: the single-level fallback on a level-1
    // overflow, the instance count, the chunk count
    auto read_header = [&](bool already_waited) -> int {
        if (!already_waited) {
            if (!pin_dev) HIP_TRY(hipMemcpyAsync(g_pinned.p, header, 32, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        if (two_level && g_pinned.p[4 + SGR_B2_HDR_OVERFLOW]) {
            // more (Gaussian, super-tile) pairs than the level-1 list holds (huge splats): the single-level path has no such limit
            if (!legacy_ok)
                return fail(SGR_E_INVALID, "level-1 binning list overflow on an image too large for the single-level fallback");
            two_level = false;
            sgr_launch_bin_count(P, IL.gx, IL.gy, IL.n_blocks, per_block, order, rects, blk_hist, s);
            sgr_launch_hist_scan(IL.T, IL.n_blocks, blk_hist, tile_cursor, s);
            sgr_launch_tile_scan(IL.T, tile_cursor, tile_start, header, tile_maxc, tile_walked, 1, hh_dev, pin_dev, s);
            if (opts->header_host && !hh_dev) HIP_TRY(hipMemcpyAsync(opts->header_host, header, 32, hipMemcpyDeviceToHost, s));
            if (!pin_dev) HIP_TRY(hipMemcpyAsync(g_pinned.p, header, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        R = (int64_t)g_pinned.p[SGR_HDR_R];
        if (R >= 0xFFFFFFFFll) return fail(SGR_E_INVALID, "more than 2^32 - 2 (Gaussian, tile) instances in one view");
        n_chunks = g_pinned.p[4 + SGR_B2_HDR_CHUNKS];
        return 0;
    };
    if (opts->info) { opts->info->speculation = 0; }
    if (nosync) {
        R = binning_capacity;
        // the chunk count lives on the device: the passes are grid-stride loops over it; the caller may know better than the
        // capacity bound how many workgroups are worth launching (idle 512-thread workgroups are not free to dispatch)
        n_chunks = B2.chunk_cap < 8192u ? B2.chunk_cap : 8192u;
        if (opts->chunk_grid && opts->chunk_grid < n_chunks) n_chunks = opts->chunk_grid;
    } else {
        const int rc = read_header(false);
        if (rc < 0) return rc;
    }
    if (opts->info) { opts->info->binning_mode = two_level ? 0 : 1; opts->info->sync_free = nosync ? 1 : 0; }
    if (opts->tile_need && !two_level) return fail(SGR_E_INVALID, "the walk hint needs the two-level binning (level-1 overflow on this view)");
    {
        const int rc = tail(R, nosync, n_chunks);
        if (rc < 0) return rc;
    }
    if (!speculative) return R;
    // ---- speculative: the tile scan's header has long arrived (list-write pass and blend are queued behind it)
    HIP_TRY(hipEventSynchronize(g_scan_ev.e));
    const bool overflow = g_pinned.p[4 + SGR_B2_HDR_OVERFLOW] != 0u;
    const int64_t R_true = (int64_t)g_pinned.p[SGR_HDR_R];
    if (!overflow && R_true <= binning_capacity) {
        if (opts->info) opts->info->speculation = 1;
        return R_true;
    }
    // a miss: the kernels queued above were no-ops (SGR_FORWARD_INVALID); the same tail once more with the true count
    {
        int rc = read_header(true);
        if (rc < 0) return rc;
        if (opts->info) { opts->info->binning_mode = two_level ? 0 : 1; opts->info->sync_free = 0; opts->info->speculation = 2; }
        if (opts->tile_need && !two_level) return fail(SGR_E_INVALID, "the walk hint needs the two-level binning (level-1 overflow on this view)");
        rc = tail(R, false, n_chunks);
        if (rc < 0) return rc;
    }
    return R;
}

int64_t sgr_forward(sgr_alloc_fn geom_alloc, void* geom_user, sgr_alloc_fn binning_alloc, void* binning_user,
                    sgr_alloc_fn img_alloc, void* img_user, int P, int D, int M, const float* background, int width,
                    int height, const float* means3D, const float* shs, const float* colors_precomp,
                    const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                    float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii, int debug,
                    void* stream)
{
    return sgr_forward_ex(geom_alloc, geom_user, binning_alloc, binning_user, img_alloc, img_user, P, D, M, background, width, height,
                          means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                          projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, debug, stream, nullptr);
}

// phase 0: everything; 1: the blend half (accumulator reset, blend backward, and in compact mode the masked colour
// gradients into dL_dcolor); 2: the preprocess half (in compact mode dL_dcolor is left alone: phase 1 wrote it)
// The post-blend bookkeeping of a forward run with SGR_FLAG_DEFER_POST, as a job for another kernel (tile_order.h)
int sgr_forward_post_job(int width, int height, char* img_buffer, int64_t R, const sgr_forward_opts* opts, SgrTileOrderJob* job)
{
    if (!img_buffer || !opts || !job) return SGR_E_INVALID;
    const ImgLayout IL = sgr_img_layout(width, height);
    uint32_t* hh_dev = nullptr;
    if (opts->header_host && hipHostGetDevicePointer(reinterpret_cast<void**>(&hh_dev), opts->header_host, 0) != hipSuccess) return SGR_E_INVALID;
    job->T = IL.T;
    job->list_cap = (uint32_t)(R > 0xFFFFFFFFll ? 0xFFFFFFFFll : R);
    job->tile_maxc = reinterpret_cast<const uint32_t*>(img_buffer + IL.tile_maxc);
    job->tile_walked = opts->tile_need_out ? reinterpret_cast<const uint32_t*>(img_buffer + IL.tile_walked) : nullptr;
    job->header = reinterpret_cast<const uint32_t*>(img_buffer + IL.header);
    job->order = reinterpret_cast<uint32_t*>(img_buffer + IL.tile_cursor);
    job->order_copy = opts->tile_order_out;
    job->need_out = opts->tile_need_out;
    job->header_host = hh_dev;
    job->margin = opts->hint_margin > 0.f ? opts->hint_margin : 0.25f;
    job->gx = IL.gx;
    job->gy = IL.gy;
    return 0;
}

static int backward_impl(int phase, int P, int D, int M, int64_t R, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, char* geom_buffer,
                         char* binning_buffer, char* img_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                         float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                         float* dL_dscale, float* dL_drot, int debug, void* stream, const sgr_backward_opts* opts = nullptr)
{
    hipStream_t s = (hipStream_t)stream;
    const int raw_params = (phase & SGR_MODE_RAW_PARAMS) ? 1 : 0;
    const int sh_dir_elsewhere = (phase & SGR_MODE_SH_DIR_ELSEWHERE) ? 1 : 0;
    phase &= 3;
    if (phase < 0 || phase > 2) return fail(SGR_E_INVALID, "phase must be 0, 1 or 2");
    if (P <= 0 || width <= 0 || height <= 0) return fail(SGR_E_INVALID, "P, width and height must be positive");
    if (!geom_buffer || !binning_buffer || !img_buffer || !dL_dpix) return fail(SGR_E_INVALID, "null scratch / dL_dpix");
    // dL_dmean2D, dL_dconic and dL_dcov3D may be NULL: intermediate results a training step has no use for are then not written
    if (!dL_dopacity || !dL_dcolor || !dL_dmean3D) return fail(SGR_E_INVALID, "null gradient output");
    if (cov3D_precomp && !dL_dcov3D) return fail(SGR_E_INVALID, "dL_dcov3D required with cov3D_precomp");
    const bool use_sh = shs && !colors_precomp;
    // use_sh with dL_dsh == NULL selects the compact mode: dL_dcolor receives the clamp-masked colour gradients and no
    // SH gradient is materialised (see sgr_sh_grad_from_views)
    const bool compact = use_sh && !dL_dsh;
    if (!cov3D_precomp && (!dL_dscale || !dL_drot)) return fail(SGR_E_INVALID, "dL_dscale/dL_drot required");
    if (phase != 0 && !compact) return fail(SGR_E_INVALID, "the two-phase backward is for the compact SH mode");

    const ImgLayout IL = sgr_img_layout(width, height);
    const BinLayout BL = sgr_bin_layout(R, IL.T);  // (only point_list, at offset 0 for every R; the masks' offset is read on the device)
    const GeomRec* rec = reinterpret_cast<const GeomRec*>(geom_buffer);
    const float* final_T = reinterpret_cast<const float*>(img_buffer + IL.final_T);
    const uint32_t* n_contrib = reinterpret_cast<const uint32_t*>(img_buffer + IL.n_contrib);
    const uint32_t* tile_start = reinterpret_cast<const uint32_t*>(img_buffer + IL.tile_start);
    const uint32_t* blk_nb = reinterpret_cast<const uint32_t*>(img_buffer + IL.blk_nb);
    const uint32_t* point_list = reinterpret_cast<const uint32_t*>(binning_buffer + BL.point_list);

    // the blend backward accumulates nine sums per Gaussian with atomics into the private acc[P][12] table
    float* acc = reinterpret_cast<float*>(geom_buffer + sgr_geom_acc_offset(P));
    if (phase != 2) {
        { SgrStageTimer t(s, SGR_STAGE_FILL); HIP_TRY(hipMemsetAsync(acc, 0, (size_t)P * SGR_ACC_STRIDE * 4, s)); }
        if (R > 0) {
            SgrStageTimer t(s, SGR_STAGE_BLEND_BWD);
            // (the forward's per-tile counters are dead by now: their array holds the backward's launch order)
            sgr_launch_blend_bwd(width, height, IL.gx, IL.gy, tile_start, point_list, binning_buffer, blk_nb, rec, background, final_T,
                                 n_contrib, dL_dpix, acc, reinterpret_cast<const uint32_t*>(img_buffer + IL.tile_maxc),
                                 reinterpret_cast<const uint32_t*>(img_buffer + IL.header), (uint32_t)(R > 0xFFFFFFFFll ? 0xFFFFFFFFll : R),
                                 reinterpret_cast<uint32_t*>(img_buffer + IL.tile_cursor),
                                 (opts && (opts->flags & SGR_BWD_TILE_ORDER_READY)) ? 1 : 0, s,
                                 (sgr_exact_alpha() || (opts && (opts->flags & SGR_BWD_EXACT_ALPHA))) ? 1 : 0);
        }
        STAGE_CHECK("blend_bwd");
        if (phase == 1) {
            { SgrStageTimer t(s, SGR_STAGE_MASKED_COLORS); sgr_launch_masked_colors(P, rec, acc, dL_dcolor, cam_pos, opts ? opts->campos_row : nullptr, s); }
            STAGE_CHECK("masked_colors");
            return 0;
        }
    }
    PreprocessBwdArgs pb;
    pb.P = P; pb.D = D; pb.M = use_sh ? M : 0;
    pb.means3D = means3D; pb.shs = use_sh ? shs : nullptr;
    pb.scales = cov3D_precomp ? nullptr : scales; pb.rotations = cov3D_precomp ? nullptr : rotations;
    pb.scale_modifier = scale_modifier; pb.cov3D_precomp = cov3D_precomp;
    pb.viewmatrix = viewmatrix; pb.projmatrix = projmatrix; pb.cam_pos = cam_pos;
    pb.W = width; pb.H = height; pb.tan_fovx = tan_fovx; pb.tan_fovy = tan_fovy;
    pb.focal_y = height / (2.0f * tan_fovy);
    pb.focal_x = width / (2.0f * tan_fovx);
    pb.rec = rec;
    pb.raw_params = raw_params && !cov3D_precomp;
    pb.sh_dir_elsewhere = sh_dir_elsewhere && use_sh && !dL_dsh;  // (compact SH mode only: see sgr_sh_adam_from_views_ex)
    pb.acc = acc;
    pb.header = reinterpret_cast<const uint32_t*>(img_buffer + IL.header);
    pb.list_cap = (uint32_t)(R > 0xFFFFFFFFll ? 0xFFFFFFFFll : R);
    pb.campos_row = (opts && compact && phase == 0) ? opts->campos_row : nullptr;  // (phase 1 wrote it with the colours)
    pb.dens_max_radii = opts ? opts->max_radii2D : nullptr;
    pb.dens_accum = opts ? opts->grad_accum : nullptr;
    pb.dens_denom = opts ? opts->denom : nullptr;
    pb.dL_dmean2D = dL_dmean2D; pb.dL_dconic = dL_dconic; pb.dL_dopacity = dL_dopacity;
    pb.dL_dcolor = phase == 2 ? nullptr : dL_dcolor;  // phase 2: already written (and possibly being sent) by phase 1
    pb.dL_dmean3D = dL_dmean3D; pb.dL_dcov3D = dL_dcov3D; pb.dL_dsh = use_sh ? dL_dsh : nullptr;
    pb.dL_dscale = cov3D_precomp ? nullptr : dL_dscale; pb.dL_drot = cov3D_precomp ? nullptr : dL_drot;
    { SgrStageTimer t(s, SGR_STAGE_PREPROCESS_BWD); sgr_launch_preprocess_bwd(pb, s); }
    STAGE_CHECK("preprocess_bwd");
    return 0;
}

int sgr_backward(int P, int D, int M, int64_t R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                 char* binning_buffer, char* img_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot, int debug, void* stream)
{
    (void)radii;  // the private geometry record carries the radius
    return backward_impl(0, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, geom_buffer, binning_buffer, img_buffer,
                         dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                         debug, stream);
}

int sgr_backward_phase(int phase, int P, int D, int M, int64_t R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                       char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix, float* dL_dmean2D,
                       float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                       float* dL_dscale, float* dL_drot, int debug, void* stream)
{
    (void)radii;
    return backward_impl(phase, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier,
                         rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, geom_buffer, binning_buffer,
                         img_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                         dL_drot, debug, stream);
}

int sgr_backward_ex(int phase, int P, int D, int M, int64_t R, const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                    float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                    char* geom_buffer, char* binning_buffer, char* img_buffer, const float* dL_dpix, float* dL_dmean2D,
                    float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                    float* dL_dscale, float* dL_drot, int debug, void* stream, const sgr_backward_opts* opts)
{
    (void)radii;
    return backward_impl(phase, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier,
                         rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, geom_buffer, binning_buffer,
                         img_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                         dL_drot, debug, stream, opts);
}

// (introspection for the developer scripts: byte offsets of the two-level binning tables inside the image scratch)
void sgr_debug_bin2_offsets(int P, int width, int height, size_t* out)
{
    const ImgLayout IL = sgr_img_layout(width, height);
    const Bin2Layout B = sgr_bin2_layout(P, IL.gx, IL.gy);
    out[0] = IL.total + B.sup_start; out[1] = IL.total + B.chunk_base; out[2] = IL.total + B.cnt2; out[3] = IL.total + B.chunk_sup;
    out[4] = (size_t)B.T1; out[5] = (size_t)B.sgx; out[6] = (size_t)B.chunk_cap; out[7] = IL.header;
}

size_t sgr_bin2_bytes(int P, int width, int height)
{
    const ImgLayout IL = sgr_img_layout(width, height);
    return sgr_bin2_layout(P, IL.gx, IL.gy).total;
}

int sgr_sh_grad_from_views(int P, int n_views, int D, int M, const float* means3D, const float* campos_all,
                           const float* dcolor_all, int64_t view_stride, float* dL_dsh, void* stream)
{
    if (P <= 0) return 0;
    if (n_views <= 0 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || M > 16 || !means3D || !campos_all || !dcolor_all || !dL_dsh)
        return fail(SGR_E_INVALID, "sgr_sh_grad_from_views: bad arguments");
    if (view_stride != 0 && view_stride < P) return fail(SGR_E_INVALID, "sgr_sh_grad_from_views: view_stride < P");
    sgr_launch_sh_grad_from_views(P, n_views, D, M, (size_t)(view_stride ? view_stride : P), means3D, campos_all, dcolor_all, dL_dsh,
                                  (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SGR_E_HIP, std::string("sh_grad_from_views: ") + hipGetErrorString(e));
    return 0;
}

int sgr_sh_adam_from_views_ex(int P, int n_views, int D, int M, const float* means3D, const float* campos_all,
                              const float* dcolor_all, int64_t view_stride, float* sh_params, float* exp_avg, float* exp_avg_sq, float lr_dc,
                              float lr_rest, float beta1, float beta2, float eps, int step, float grad_scale, float* dmean_extra,
                              void* stream)
{
    if (P <= 0) return 0;
    if (n_views <= 0 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || M > 16 || step < 1 || !means3D || !campos_all || !dcolor_all ||
        !sh_params || !exp_avg || !exp_avg_sq)
        return fail(SGR_E_INVALID, "sgr_sh_adam_from_views: bad argument");
    float bc1, bc2_sqrt;
    sgr_bias_corrections(beta1, beta2, step, &bc1, &bc2_sqrt);
    if (view_stride != 0 && view_stride < P) return fail(SGR_E_INVALID, "sgr_sh_adam_from_views: view_stride < P");
    if (dmean_extra && (M != 16 || ((uintptr_t)sh_params & 15)))
        return fail(SGR_E_INVALID, "sgr_sh_adam_from_views_ex: dmean_extra needs M == 16 and 16-byte aligned sh_params");
    sgr_launch_sh_adam_from_views(P, n_views, D, M, (size_t)(view_stride ? view_stride : P), means3D, campos_all, dcolor_all, sh_params, exp_avg, exp_avg_sq, lr_dc, lr_rest,
                                  beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, dmean_extra, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SGR_E_HIP, std::string("sh_adam_from_views: ") + hipGetErrorString(e));
    return 0;
}

int sgr_sh_adam_from_views(int P, int n_views, int D, int M, const float* means3D, const float* campos_all,
                           const float* dcolor_all, int64_t view_stride, float* sh_params, float* exp_avg, float* exp_avg_sq, float lr_dc,
                           float lr_rest, float beta1, float beta2, float eps, int step, float grad_scale, void* stream)
{
    return sgr_sh_adam_from_views_ex(P, n_views, D, M, means3D, campos_all, dcolor_all, view_stride, sh_params, exp_avg, exp_avg_sq, lr_dc,
                                     lr_rest, beta1, beta2, eps, step, grad_scale, nullptr, stream);
}

}  // extern "C"
