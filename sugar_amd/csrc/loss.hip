// loss.hip -- fused photometric loss of the train step for gfx950:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(SSIM(x, y)))
// Restates l1_loss / ssim / _ssim of sugar_utils/loss_utils.py:17-63 as used by the train step
// (gaussian_splatting/train.py:88-90; sugar_trainers/coarse_sdf.py:456-457,536).  The reference evaluates SSIM with five
// grouped 11x11 conv2d calls plus ~25 elementwise kernels and their autograd twins; on an MI355X that costs far more than
// the rasterizer it scores.  Here the window is applied separably through LDS and the backward is analytic:
//
//   forward kernel : per 32x28 tile, stage x, y (+5 px halo, zero padded like conv2d(padding=5)) in LDS, horizontal then
//                    vertical 11-tap pass for {x, y, x^2 + y^2, xy}, SSIM map value S and the three partials
//                    dS/dmu1, dS/dE[x^2], dS/dE[xy]; tile sums of S and |x-y| go to one partial per tile, added in double by
//                    a finishing kernel.
//   backward kernel: dL/dx = gL * [ (1-lambda)/N * sign(x-y) - lambda/N * ( G*(dS/dmu1) + 2x * G*(dS/dE[x^2]) + y * G*(dS/dE[xy]) ) ]
//                    (G is symmetric, so the adjoint of the window is the same separable filter).
// HBM traffic: x, y read twice, three partial maps written and read once: ~9 floats per pixel-channel in total.
#include "../../include/sugar_raster.h"
#include "sgr_common.h"
#include "tile_order.h"

namespace {

#define LT 32             // output tile: 32 columns x 28 rows (33 KB of LDS in the forward: four workgroups per CU)
#define LTY 28
#define LH 5              // window half width (window_size 11)
#define LR (LT + 2 * LH)  // 42 staged columns
#define LRY (LTY + 2 * LH)  // 38 staged rows
#define LRP (LR + 1)      // padded row strides (bank spread)
#define LTP (LT + 1)

struct Win { float w[11]; };

// Workgroup b runs on XCD b % 8 (its own L2).  With the tiles handed out in raster order no two neighbouring tiles ever share
// an L2 and every tile fetches its whole halo from the fabric (rocprofv3 FETCH_SIZE: 2.2x the image bytes in the forward, 1.7x
// the maps in the backward).  Here XCD k takes the k-th eighth of the (channel, row, column) tile order instead, so that the
// tiles it works on at about the same time are neighbours.  Returns false for the padding workgroups.
// `lead` = 8 when the grid starts with eight extra workgroups (the first of them carries a side job: dispatched first, it is
// done long before the tiles are -- as the LAST workgroup of the grid it ran alone after them and its 10 us showed in the step).
__device__ __forceinline__ bool xcd_tile(int tiles_x, int tiles_y, int n_tiles, int& tile, int& c, int& ty, int& tx, int lead = 0)
{
    const int per = (n_tiles + 7) >> 3;
    const int b = (int)blockIdx.x - lead, j = b >> 3;
    if (b < 0) return false;
    tile = (b & 7) * per + j;
    if (j >= per || tile >= n_tiles) return false;
    c = tile / (tiles_x * tiles_y);
    const int r = tile - c * (tiles_x * tiles_y);
    ty = r / tiles_x;
    tx = r - ty * tiles_x;
    return true;
}

// conv2d(padding=5) reads zeros outside the image: loads go to the clamped pixel (always a valid address, so they can
// be issued unconditionally and in a batch) and the value is replaced by zero afterwards
__device__ __forceinline__ bool in_image(int W, int H, int x, int y) { return x >= 0 && x < W && y >= 0 && y < H; }
__device__ __forceinline__ unsigned ld_clamped(int W, int H, int x, int y)  // offset inside one plane (< 2^32 pixels)
{
    return (unsigned)(min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1));
}

// Register-blocked separable window: in the horizontal pass a thread produces 4 adjacent outputs of one staged row from
// 14 staged values; in the vertical pass a thread produces 4 vertically adjacent pixels of one column from 14 rows.
// A 32x32 tile re-reads 1.7x its pixels as halo (a 16x16 tile: 2.6x).
//
// The forward needs FOUR windowed quantities, not five: E[x^2] and E[y^2] only ever enter S through their sum (D below).
// They are carried as two float2 pairs, {x, y} and {x^2 + y^2, xy}: every multiply-add of the two passes is one
// v_pk_fma_f32 on a pair (twice the FP32 rate of v_fma_f32 on gfx950) and every LDS access moves a pair (ds_*_b64), which
// halves the instruction count of a kernel that is bound by how fast its few resident waves issue.
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256, 4) k_l1_ssim_fwd(int W, int H, int tiles_x, int tiles_y, int n_tiles, const float* __restrict__ X,
                                                             const float* __restrict__ Y, Win win, float* __restrict__ dm1,
                                                             float* __restrict__ ds1, float* __restrict__ ds12,
                                                             float2* __restrict__ partial, SgrTileOrderJob job)
{
    // row strides in 8-byte units are odd and a wave's lanes walk down rows (horizontal pass) or along a row (vertical
    // pass), so the 16 lanes an LDS cycle serves fall into 16 different bank pairs
    __shared__ f2 sxy[LRY][LRP];      // {x, y}, zero padded
    __shared__ f2 hm[LRY][LTP];       // rows filtered: {mu1, mu2}
    __shared__ f2 hs[LRY][LTP];       //                {E[x^2 + y^2], E[xy]}
    __shared__ float red[2][4];
    int tile, c, tby, tbx;
    if (!xcd_tile(tiles_x, tiles_y, n_tiles, tile, c, tby, tbx, job.T > 0 ? 8 : 0)) {
        // train step: the rasterizer's post-blend bookkeeping (launch order, walk hint, header copy: tile_order.h) rides in a spare
        // workgroup of this kernel -- the grid then starts with eight workgroups more than tiles need
        if (job.T > 0 && blockIdx.x == 0) sgr_tile_order_block<256>(job);
        return;
    }
    const size_t plane = (size_t)c * W * H;
    const int x0 = tbx * LT, y0 = tby * LTY;
    const int tid = threadIdx.x;
    {
        // Staging: wave w takes the staged rows w, w + 4, ..., lane = staged column.  The row part of every address and of
        // the zero-padding test is then wave-uniform (scalar registers, scalar ALU), the column part is computed once, and a
        // load is one instruction with a scalar base and a 32-bit lane offset.  (Indexing the 38 x 42 region linearly cost
        // ~35 vector instructions per element in divisions, clamps and 64-bit adds: more than the filter itself.)
        // All values are requested before the first is used.
        const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int gx = x0 + lane - LH;
        const bool inx = lane < LR && gx >= 0 && gx < W;
        const unsigned bx = (unsigned)min(max(gx, 0), W - 1) * 4u;
        constexpr int NR = (LRY + 3) / 4;
        float vx[NR], vy[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int gy = min(max(y0 + w + 4 * j - LH, 0), H - 1);
            const size_t row = (plane + (size_t)gy * W) * 4;
            vx[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(X) + row + bx);
            vy[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Y) + row + bx);
        }
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int r = w + 4 * j, gy = y0 + r - LH;
            const bool in = inx && gy >= 0 && gy < H;
            if (r < LRY && lane < LR) { f2 v = {in ? vx[j] : 0.f, in ? vy[j] : 0.f}; sxy[r][lane] = v; }
        }
    }
    __syncthreads();
    // horizontal pass: LRY rows x LT columns; item = (group of 4 columns, row), consecutive lanes on consecutive rows
    for (int i = tid; i < LRY * (LT / 4); i += 256) {
        const int g = i / LRY, r = i - g * LRY, c0 = g * 4;
        f2 am[4], as[4];
#pragma unroll
        for (int o = 0; o < 4; o++) { am[o] = (f2){0.f, 0.f}; as[o] = (f2){0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const f2 v = sxy[r][c0 + k];
            const f2 p = {v.x * v.x + v.y * v.y, v.x * v.y};
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int t = k - o;  // tap index of output o for staged value k
                if (t >= 0 && t < 11) { am[o] += win.w[t] * v; as[o] += win.w[t] * p; }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) { hm[r][c0 + o] = am[o]; hs[r][c0 + o] = as[o]; }
    }
    __syncthreads();
    // vertical pass: thread = (column lx, rows 4*lg .. 4*lg+3)
    const int lx = tid & 31, lg = tid >> 5;
    const int px = x0 + lx;
    float s_val = 0.f, l1_val = 0.f;
    if (4 * lg < LTY) {
        f2 vm[4], vs[4];
#pragma unroll
        for (int o = 0; o < 4; o++) { vm[o] = (f2){0.f, 0.f}; vs[o] = (f2){0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const f2 m = hm[4 * lg + k][lx], q = hs[4 * lg + k][lx];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int t = k - o;
                if (t >= 0 && t < 11) { vm[o] += win.w[t] * m; vs[o] += win.w[t] * q; }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int ly = 4 * lg + o, py = y0 + ly;
            if (px < W && py < H) {
                const float mu1 = vm[o].x, mu2 = vm[o].y, e2 = vs[o].x, exy = vs[o].y;
                const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float sig12 = exy - mu12;
                const float A = 2.f * mu12 + C1, B = 2.f * sig12 + C2, Cc = mu1_sq + mu2_sq + C1;
                const float D = (e2 - mu1_sq - mu2_sq) + C2;  // sigma1^2 + sigma2^2 + C2
                // v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division: the maps feed an 11 x 11 average
                const float inv_d = __builtin_amdgcn_rcpf(D);
                const float inv_cd = __builtin_amdgcn_rcpf(Cc) * inv_d;
                const float S = A * B * inv_cd;
                const unsigned ob = (unsigned)(py * W + px) * 4u;  // byte offset inside the plane: scalar base + 32-bit lane offset
                *reinterpret_cast<float*>(reinterpret_cast<char*>(dm1 + plane) + ob) = 2.f * mu2 * (B - A) * inv_cd - 2.f * mu1 * S * (D - Cc) * inv_cd;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(ds1 + plane) + ob) = -S * inv_d;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(ds12 + plane) + ob) = 2.f * A * inv_cd;
                s_val += S;
                const f2 v = sxy[ly + LH][lx + LH];
                l1_val += fabsf(v.x - v.y);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { s_val += __shfl_xor(s_val, o); l1_val += __shfl_xor(l1_val, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s_val; red[1][tid >> 6] = l1_val; }
    __syncthreads();
    // one partial per workgroup; k_l1_ssim_finish adds them in double (same-address device atomics cost 0.5 ms)
    if (tid == 0) {
        partial[tile] = make_float2(red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// the tile partials added in double by ONE workgroup of NT threads: its own small kernel behind the forward, or -- when the
// backward follows at once, as in the train step -- a spare workgroup of the backward kernel (one launch less on the step's chain)
template <int NT>
__device__ __forceinline__ void l1_ssim_finish(const float2* __restrict__ partial, int n_partial, double n, float lambda,
                                               float* __restrict__ loss)
{
    __shared__ double s0[NT / 64], s1[NT / 64];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += NT) { const float2 v = partial[i]; a += (double)v.x; b += (double)v.y; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = a; s1[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ssim_sum = 0.0, l1_sum = 0.0;
        for (int w = 0; w < NT / 64; w++) { ssim_sum += s0[w]; l1_sum += s1[w]; }
        const double ssim_mean = ssim_sum / n, l1_mean = l1_sum / n;
        loss[0] = (float)((1.0 - (double)lambda) * l1_mean + (double)lambda * (1.0 - ssim_mean));
        loss[1] = (float)l1_mean;
        loss[2] = (float)ssim_mean;
    }
}

__global__ void __launch_bounds__(1024) k_l1_ssim_finish(const float2* __restrict__ partial, int n_partial, double n,
                                                         float lambda, float* __restrict__ loss)
{
    l1_ssim_finish<1024>(partial, n_partial, n, lambda, loss);
}

// (the backward keeps 32 x 32 tiles: with 28 rows it was 11 % slower)  Three maps are windowed: {dS/dmu1, dS/dE[x^2]} travel
// as a float2 pair (packed multiply-adds, 8-byte LDS accesses), dS/dE[xy] on its own.
__global__ void __launch_bounds__(256) k_l1_ssim_bwd(int W, int H, int tiles_x, int tiles_y, int n_tiles, const float* __restrict__ X,
                                                     const float* __restrict__ Y,
                                                     Win win, const float* __restrict__ dm1, const float* __restrict__ ds1,
                                                     const float* __restrict__ ds12, const float* __restrict__ grad_loss,
                                                     float lambda, float inv_n, float* __restrict__ dX,
                                                     const float2* __restrict__ partial, int n_partial, float* __restrict__ loss)
{
    __shared__ f2 sp[LR][LRP];
    __shared__ float sq[LR][LRP];
    __shared__ f2 hp[LR][LTP];
    __shared__ float hq[LR][LTP];
    int tile, c, tby, tbx;
    if (!xcd_tile(tiles_x, tiles_y, n_tiles, tile, c, tby, tbx, loss ? 8 : 0)) {
        // (loss != NULL: the grid starts with eight workgroups more than tiles need; the first one reduces the forward's partials)
        if (loss && blockIdx.x == 0) l1_ssim_finish<256>(partial, n_partial, 1.0 / (double)inv_n, lambda, loss);
        return;
    }
    const size_t plane = (size_t)c * W * H;
    const int x0 = tbx * LT, y0 = tby * LT;
    const int tid = threadIdx.x;
    const int lx = tid & 31, lg = tid >> 5;
    const int px = x0 + lx;
    float xv[4], yv[4];
    {
        // (the forward's row-per-wave staging needs 33 load instructions here instead of 21 and measured 4 % slower: this
        // kernel is not bound by its vector ALU work)
        constexpr int NL = (LR * LR + 255) / 256;
        float v0[NL], v1[NL], v2[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = tid + 256 * j, r = i / LR, cc = i - r * LR;
            const unsigned o = ld_clamped(W, H, x0 + cc - LH, y0 + r - LH);
            v0[j] = (dm1 + plane)[o]; v1[j] = (ds1 + plane)[o]; v2[j] = (ds12 + plane)[o];
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {  // the epilogue's operands, requested with the rest
            const unsigned oo = ld_clamped(W, H, px, y0 + 4 * lg + o);
            xv[o] = (X + plane)[oo]; yv[o] = (Y + plane)[oo];
        }
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = tid + 256 * j, r = i / LR, cc = i - r * LR;
            const bool in = in_image(W, H, x0 + cc - LH, y0 + r - LH);
            if (i < LR * LR) { f2 v = {in ? v0[j] : 0.f, in ? v1[j] : 0.f}; sp[r][cc] = v; sq[r][cc] = in ? v2[j] : 0.f; }
        }
    }
    __syncthreads();
    for (int i = tid; i < LR * (LT / 4); i += 256) {
        const int g = i / LR, r = i - g * LR, c0 = g * 4;  // consecutive lanes on consecutive rows (odd strides: no bank conflicts)
        f2 ap[4];
        float aq[4];
#pragma unroll
        for (int o = 0; o < 4; o++) { ap[o] = (f2){0.f, 0.f}; aq[o] = 0.f; }
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const f2 vp = sp[r][c0 + k];
            const float vq = sq[r][c0 + k];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int t = k - o;
                if (t >= 0 && t < 11) { ap[o] += win.w[t] * vp; aq[o] += win.w[t] * vq; }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) { hp[r][c0 + o] = ap[o]; hq[r][c0 + o] = aq[o]; }
    }
    __syncthreads();
    f2 gp[4];
    float gq[4];
#pragma unroll
    for (int o = 0; o < 4; o++) { gp[o] = (f2){0.f, 0.f}; gq[o] = 0.f; }
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const f2 h = hp[4 * lg + k][lx];
        const float h2 = hq[4 * lg + k][lx];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int t = k - o;
            if (t >= 0 && t < 11) { gp[o] += win.w[t] * h; gq[o] += win.w[t] * h2; }
        }
    }
    const float gl = (grad_loss ? grad_loss[0] : 1.0f) * inv_n;  // (NULL: d loss / d loss = 1)
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int py = y0 + 4 * lg + o;
        if (px < W && py < H) {
            const unsigned oo = (unsigned)(py * W + px);
            const float d = xv[o] - yv[o];
            const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            const float dssim = gp[o].x + 2.f * xv[o] * gp[o].y + yv[o] * gq[o];
            (dX + plane)[oo] = gl * ((1.f - lambda) * sgn - lambda * dssim);
        }
    }
}

Win make_window()
{
    // gaussian(11, 1.5) of sugar_utils/loss_utils.py:23-25, float32 like torch.Tensor([...]) / sum
    Win w;
    float s = 0.f;
    for (int i = 0; i < 11; i++) { w.w[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += w.w[i]; }
    for (int i = 0; i < 11; i++) w.w[i] = w.w[i] / s;
    return w;
}

}  // namespace

extern "C" {

size_t sgr_l1_ssim_scratch_bytes(int channels, int width, int height)
{
    const size_t blocks = (size_t)((width + LT - 1) / LT) * ((height + LTY - 1) / LTY) * channels;
    return sgr_align((size_t)channels * width * height * 4) * 3 + sgr_align(blocks * sizeof(float2));
}

int sgr_l1_ssim_forward(int channels, int width, int height, const float* img, const float* gt, float lambda,
                        char* scratch, float* loss_out, void* stream)
{
    return sgr_l1_ssim_forward_job(channels, width, height, img, gt, lambda, scratch, loss_out, nullptr, stream);
}

// (internal, sgr_common.h: the forward with the rasterizer's post-blend job on board)
int sgr_l1_ssim_forward_job(int channels, int width, int height, const float* img, const float* gt, float lambda,
                            char* scratch, float* loss_out, const SgrTileOrderJob* job, void* stream)
{
    if (channels <= 0 || width <= 0 || height <= 0 || !img || !gt || !scratch) return SGR_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const size_t plane = sgr_align((size_t)channels * width * height * 4);
    float* dm1 = reinterpret_cast<float*>(scratch);
    float* ds1 = reinterpret_cast<float*>(scratch + plane);
    float* ds12 = reinterpret_cast<float*>(scratch + 2 * plane);
    float2* partial = reinterpret_cast<float2*>(scratch + 3 * plane);
    const int tiles_x = (width + LT - 1) / LT, tiles_y = (height + LTY - 1) / LTY, n_tiles = tiles_x * tiles_y * channels;
    SgrTileOrderJob none = {};
    const bool ride = job && job->T > 0;
    hipLaunchKernelGGL(k_l1_ssim_fwd, dim3(8 * ((n_tiles + 7) / 8) + (ride ? 8 : 0)), dim3(256), 0, s, width, height, tiles_x, tiles_y, n_tiles,
                       img, gt, make_window(), dm1, ds1, ds12, partial, ride ? *job : none);
    if (loss_out)  // (NULL: the caller asks sgr_l1_ssim_backward_ex for the value)
        hipLaunchKernelGGL(k_l1_ssim_finish, dim3(1), dim3(1024), 0, s, partial, n_tiles, (double)channels * width * height, lambda,
                           loss_out);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

int sgr_l1_ssim_backward(int channels, int width, int height, const float* img, const float* gt, float lambda,
                         const char* scratch, const float* grad_loss, float* grad_img, void* stream)
{
    return sgr_l1_ssim_backward_ex(channels, width, height, img, gt, lambda, scratch, grad_loss, grad_img, nullptr, stream);
}

int sgr_l1_ssim_backward_ex(int channels, int width, int height, const float* img, const float* gt, float lambda,
                            const char* scratch, const float* grad_loss, float* grad_img, float* loss_out, void* stream)
{
    if (channels <= 0 || width <= 0 || height <= 0 || !img || !gt || !scratch || !grad_img) return SGR_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const size_t plane = sgr_align((size_t)channels * width * height * 4);
    const float* dm1 = reinterpret_cast<const float*>(scratch);
    const float* ds1 = reinterpret_cast<const float*>(scratch + plane);
    const float* ds12 = reinterpret_cast<const float*>(scratch + 2 * plane);
    const int tiles_x = (width + LT - 1) / LT, tiles_y = (height + LT - 1) / LT, n_tiles = tiles_x * tiles_y * channels;
    const float inv_n = (float)(1.0 / ((double)channels * width * height));
    const float2* partial = reinterpret_cast<const float2*>(scratch + 3 * plane);  // (the forward's tiles are 32 x 28)
    const int n_partial = ((width + LT - 1) / LT) * ((height + LTY - 1) / LTY) * channels;
    hipLaunchKernelGGL(k_l1_ssim_bwd, dim3(8 * ((n_tiles + 7) / 8) + (loss_out ? 8 : 0)), dim3(256), 0, s, width, height, tiles_x, tiles_y,
                       n_tiles, img, gt, make_window(), dm1, ds1, ds12, grad_loss, lambda, inv_n, grad_img, partial, n_partial, loss_out);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

}  // extern "C"
