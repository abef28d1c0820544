// adam.hip -- one-launch Adam over the flat parameter buffer of the train step (gfx950).
//
// The reference optimises six tensors with torch.optim.Adam(lr per group, eps=1e-15)
// (gaussian_splatting/scene/gaussian_model.py:152-166; learning rates arguments/__init__.py:74-83), i.e. the update
//     m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
// Here all parameters live in ONE contiguous float buffer (sugar_amd/train_step.py: GaussianParams.flat), so the step is a
// single HBM-streaming kernel: 16 B read + 12 B written per parameter, float4-vectorised, no per-tensor launches.  The
// learning rate is a function of the element index through a small segment table; inside a segment it may alternate
// with a period (the SH tensor [P,M,3] has lr_a for the 3 DC coefficients and lr_b for the other 3*(M-1) of every
// Gaussian: feature_lr and feature_lr/20).
#include "../../include/sugar_raster.h"
#include "sgr_common.h"

namespace {

#define ADAM_MAX_SEG 8
struct AdamSegs {
    int n;
    long long begin[ADAM_MAX_SEG], end[ADAM_MAX_SEG];
    float lr_a[ADAM_MAX_SEG], lr_b[ADAM_MAX_SEG];
    int period[ADAM_MAX_SEG], split[ADAM_MAX_SEG];
};

// The segment table stays in the kernel-argument segment and is read with scalar loads at a run-time index (loops kept rolled): held
// in registers -- what full unrolling makes of it -- its 48 words cost 106 SGPRs, 114 of them spilled to VGPR lanes (round-5 verdict).
__device__ __forceinline__ int seg_of(const AdamSegs& sg, long long i)
{
    int k = -1;
#pragma clang loop unroll(disable)
    for (int j = 0; j < sg.n; j++)
        if (i >= sg.begin[j] && i < sg.end[j]) k = j;
    return k;
}

__device__ __forceinline__ float lr_of1(const AdamSegs& sg, long long e)
{
    const int k = seg_of(sg, e);
    if (k < 0) return 0.f;
    return ((int)((e - sg.begin[k]) % sg.period[k]) < sg.split[k]) ? sg.lr_a[k] : sg.lr_b[k];
}

// Learning rates of the four elements e0 .. e0 + 3.  `wave_first` .. `wave_last`: the element range of the whole WAVE's float4s (wave
// uniform): almost always inside one segment, found once per wave with scalar compares; a float4 that straddles a boundary (at most
// n_seg of them) takes the per-element search.  ONE 32-bit remainder for the four elements of a periodic segment (a per-element
// 64-bit '%' made this kernel ALU-bound).
__device__ __forceinline__ void lr_of4(const AdamSegs& sg, long long e0, long long wave_first, long long wave_last, float* lr)
{
    int ku = -1;
#pragma clang loop unroll(disable)
    for (int j = 0; j < sg.n; j++)
        if (wave_first >= sg.begin[j] && wave_last < sg.end[j]) ku = j;
    if (ku >= 0) {
        const float la = sg.lr_a[ku], lb = sg.lr_b[ku];
        if (la == lb) { lr[0] = lr[1] = lr[2] = lr[3] = la; return; }
        const int period = sg.period[ku], split = sg.split[ku];
        const unsigned long long off = (unsigned long long)(e0 - sg.begin[ku]);
        unsigned r = (off >> 32) ? (unsigned)(off % (unsigned long long)period) : ((unsigned)off % (unsigned)period);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            lr[c] = (int)r < split ? la : lb;
            r = (r + 1 == (unsigned)period) ? 0u : r + 1;
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) lr[c] = lr_of1(sg, e0 + c);
}

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_adam(long long n4, f4* __restrict__ p, const f4* __restrict__ g,
                                              f4* __restrict__ m, f4* __restrict__ v, AdamSegs sg, float b1, float b2, float omb1, float omb2,
                                              float eps, float bc1, float bc2_sqrt, float grad_scale,
                                              const float* __restrict__ extra, long long extra_n,
                                              const uint32_t* __restrict__ guard, uint32_t guard_cap, int n_tail)
{
    if (guard && SGR_FORWARD_INVALID(guard, guard_cap)) return;  // the step's forward was a no-op: so is its optimiser step
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        // g, m and v are touched once per step: stream them past the caches (p is read again by the next forward)
        f4 gg = __builtin_nontemporal_load(&g[i]);
        if (4 * i < extra_n) {  // a second gradient term for the first extra_n elements (sgr_adam_step_ex)
            float* gx = reinterpret_cast<float*>(&gg);
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (4 * i + c < extra_n) gx[c] += extra[4 * i + c];
        }
        gg *= grad_scale;
        f4 pp = p[i], mm = __builtin_nontemporal_load(&m[i]), vv = __builtin_nontemporal_load(&v[i]);
        float* pf = reinterpret_cast<float*>(&pp);
        float* mf = reinterpret_cast<float*>(&mm);
        float* vf = reinterpret_cast<float*>(&vv);
        const float* gf = reinterpret_cast<const float*>(&gg);
        float lr[4];
        {
            // the wave's first float4 (lanes hold consecutive i): a scalar pair
            const long long i0 = i - (long long)(threadIdx.x & 63);
            const long long wf = 4 * (((long long)__builtin_amdgcn_readfirstlane((int)(i0 >> 32)) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)i0));
            lr_of4(sg, 4 * i, wf, wf + 255, lr);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            mf[c] = b1 * mf[c] + omb1 * gf[c];
            vf[c] = b2 * vf[c] + omb2 * gf[c] * gf[c];
            const float denom = sqrtf(vf[c]) / bc2_sqrt + eps;
            pf[c] -= (lr[c] / bc1) * (mf[c] / denom);
        }
        p[i] = pp;
        __builtin_nontemporal_store(mm, &m[i]);
        __builtin_nontemporal_store(vv, &v[i]);
    }
    // the last n % 4 elements of a buffer whose length is no multiple of four (a drop-in optimiser meets [P,3] tensors with odd P)
    if (n_tail > 0 && blockIdx.x == 0 && (int)threadIdx.x < n_tail) {
        const long long e = 4 * n4 + threadIdx.x;
        float* pf = reinterpret_cast<float*>(p); float* mf = reinterpret_cast<float*>(m); float* vf = reinterpret_cast<float*>(v);
        float gg = reinterpret_cast<const float*>(g)[e];
        if (e < extra_n) gg += extra[e];
        gg *= grad_scale;
        const float lr = lr_of1(sg, e);
        const float mm = b1 * mf[e] + omb1 * gg, vv = b2 * vf[e] + omb2 * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pf[e] -= (lr / bc1) * (mm / denom);
        mf[e] = mm; vf[e] = vv;
    }
}

// ---- several tensors, one launch (a drop-in torch.optim.Adam: six parameter tensors of their own) ------------------------------
#define ADAM_MAX_TENSORS 8
struct AdamMulti {
    int n_t;
    float* p[ADAM_MAX_TENSORS]; const float* g[ADAM_MAX_TENSORS]; float* m[ADAM_MAX_TENSORS]; float* v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    float lr_bc1[ADAM_MAX_TENSORS] /* lr / (1 - b1^t) */, bc2_sqrt[ADAM_MAX_TENSORS], b1[ADAM_MAX_TENSORS], b2[ADAM_MAX_TENSORS],
          omb1[ADAM_MAX_TENSORS], omb2[ADAM_MAX_TENSORS], eps[ADAM_MAX_TENSORS];
    unsigned int blk_begin[ADAM_MAX_TENSORS + 1];   // workgroups [blk_begin[t], blk_begin[t + 1]) stream tensor t
};

// the update of k_adam (same operations in the same order: bit-identical to one launch per tensor), the tensor chosen per workgroup
__global__ void __launch_bounds__(256) k_adam_multi(AdamMulti a)
{
    int t = 0;
#pragma unroll
    for (int j = 1; j < ADAM_MAX_TENSORS; j++)
        if (j < a.n_t && blockIdx.x >= a.blk_begin[j]) t = j;
    float *pp_ = nullptr, *mm_ = nullptr, *vv_ = nullptr; const float* gg_ = nullptr;
    long long n = 0; float step_size = 0.f, bc2_sqrt = 1.f, b1 = 0.f, b2 = 0.f, omb1 = 0.f, omb2 = 0.f, eps = 0.f;
    unsigned int first = 0u, last = 0u;
#pragma unroll
    for (int j = 0; j < ADAM_MAX_TENSORS; j++)   // (selects instead of dynamic indexing of the by-value argument: no scratch)
        if (j == t) { pp_ = a.p[j]; gg_ = a.g[j]; mm_ = a.m[j]; vv_ = a.v[j]; n = a.n[j]; step_size = a.lr_bc1[j]; bc2_sqrt = a.bc2_sqrt[j];
                      b1 = a.b1[j]; b2 = a.b2[j]; omb1 = a.omb1[j]; omb2 = a.omb2[j]; eps = a.eps[j]; first = a.blk_begin[j]; last = a.blk_begin[j + 1]; }
    const long long n4 = n / 4;
    f4* p = reinterpret_cast<f4*>(pp_); const f4* g = reinterpret_cast<const f4*>(gg_);
    f4* m = reinterpret_cast<f4*>(mm_); f4* v = reinterpret_cast<f4*>(vv_);
    const unsigned int lb = blockIdx.x - first, nb = last - first;
    for (long long i = (long long)lb * 256 + threadIdx.x; i < n4; i += (long long)nb * 256) {
        const f4 gg = __builtin_nontemporal_load(&g[i]);
        f4 pp = p[i], mm = __builtin_nontemporal_load(&m[i]), vv = __builtin_nontemporal_load(&v[i]);
        float* pf = reinterpret_cast<float*>(&pp);
        float* mf = reinterpret_cast<float*>(&mm);
        float* vf = reinterpret_cast<float*>(&vv);
        const float* gf = reinterpret_cast<const float*>(&gg);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            mf[c] = b1 * mf[c] + omb1 * gf[c];
            vf[c] = b2 * vf[c] + omb2 * gf[c] * gf[c];
            const float denom = sqrtf(vf[c]) / bc2_sqrt + eps;
            pf[c] -= step_size * (mf[c] / denom);
        }
        p[i] = pp;
        __builtin_nontemporal_store(mm, &m[i]);
        __builtin_nontemporal_store(vv, &v[i]);
    }
    if (lb == 0 && (long long)threadIdx.x < n - 4 * n4) {   // the last n % 4 elements
        const long long e = 4 * n4 + threadIdx.x;
        const float gg = gg_[e];
        const float mm = b1 * mm_[e] + omb1 * gg, vv = b2 * vv_[e] + omb2 * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp_[e] -= step_size * (mm / denom);
        mm_[e] = mm; vv_[e] = vv;
    }
}

}  // namespace

int sgr_adam_launch(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_seg,
                    const long long* seg_begin, const long long* seg_end, const float* seg_lr_a, const float* seg_lr_b,
                    const int* seg_period, const int* seg_split, float beta1, float beta2, float eps, int step, float grad_scale,
                    const float* extra, long long extra_n, const uint32_t* guard, uint32_t guard_cap, hipStream_t stream)
{
    if (n <= 0) return 0;
    if (extra_n < 0 || extra_n > n || (extra_n > 0 && !extra)) return SGR_E_INVALID;
    if (!params || !grads || !exp_avg || !exp_avg_sq || n_seg < 0 || n_seg > ADAM_MAX_SEG || step < 1) return SGR_E_INVALID;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return SGR_E_INVALID;  // float4 accesses
    AdamSegs sg;
    sg.n = n_seg;
    for (int k = 0; k < ADAM_MAX_SEG; k++) {
        const bool on = k < n_seg;
        sg.begin[k] = on ? seg_begin[k] : 0; sg.end[k] = on ? seg_end[k] : 0;
        sg.lr_a[k] = on ? seg_lr_a[k] : 0.f; sg.lr_b[k] = on ? seg_lr_b[k] : 0.f;
        sg.period[k] = on && seg_period[k] > 0 ? seg_period[k] : 1; sg.split[k] = on ? seg_split[k] : 1;
    }
    float bc1, bc2_sqrt;
    sgr_bias_corrections(beta1, beta2, step, &bc1, &bc2_sqrt);
    const long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, stream, n4, reinterpret_cast<f4*>(params),
                       reinterpret_cast<const f4*>(grads), reinterpret_cast<f4*>(exp_avg),
                       reinterpret_cast<f4*>(exp_avg_sq), sg, beta1, beta2, sgr_one_minus(beta1), sgr_one_minus(beta2), eps, bc1, bc2_sqrt, grad_scale, extra, extra_n, guard,
                       guard_cap, (int)(n & 3));
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

extern "C" int sgr_adam_step_ex(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_seg,
                                const long long* seg_begin, const long long* seg_end, const float* seg_lr_a,
                                const float* seg_lr_b, const int* seg_period, const int* seg_split, float beta1, float beta2,
                                float eps, int step, float grad_scale, const float* extra, long long extra_n, void* stream)
{
    return sgr_adam_launch(n, params, grads, exp_avg, exp_avg_sq, n_seg, seg_begin, seg_end, seg_lr_a, seg_lr_b, seg_period, seg_split,
                           beta1, beta2, eps, step, grad_scale, extra, extra_n, nullptr, 0, (hipStream_t)stream);
}

extern "C" int sgr_adam_step(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_seg,
                             const long long* seg_begin, const long long* seg_end, const float* seg_lr_a,
                             const float* seg_lr_b, const int* seg_period, const int* seg_split, float beta1, float beta2,
                             float eps, int step, float grad_scale, void* stream)
{
    return sgr_adam_step_ex(n, params, grads, exp_avg, exp_avg_sq, n_seg, seg_begin, seg_end, seg_lr_a, seg_lr_b, seg_period, seg_split,
                            beta1, beta2, eps, step, grad_scale, nullptr, 0, stream);
}

extern "C" int sgr_adam_step_multi(int n_tensors, const long long* n, float* const* params, const float* const* grads,
                                   float* const* exp_avg, float* const* exp_avg_sq, const float* lr, const float* beta1,
                                   const float* beta2, const float* eps, const int* step, void* stream)
{
    if (n_tensors <= 0) return 0;
    if (n_tensors > ADAM_MAX_TENSORS || !n || !params || !grads || !exp_avg || !exp_avg_sq || !lr || !beta1 || !beta2 || !eps || !step)
        return SGR_E_INVALID;
    AdamMulti a;
    a.n_t = n_tensors;
    unsigned int blk = 0;
    for (int t = 0; t < ADAM_MAX_TENSORS; t++) {
        const bool on = t < n_tensors;
        if (on) {
            if (n[t] < 0 || step[t] < 1 || !params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t]) return SGR_E_INVALID;
            if (((uintptr_t)params[t] | (uintptr_t)grads[t] | (uintptr_t)exp_avg[t] | (uintptr_t)exp_avg_sq[t]) & 15) return SGR_E_INVALID;
        }
        a.p[t] = on ? params[t] : nullptr; a.g[t] = on ? grads[t] : nullptr; a.m[t] = on ? exp_avg[t] : nullptr; a.v[t] = on ? exp_avg_sq[t] : nullptr;
        a.n[t] = on ? n[t] : 0;
        float bc1 = 1.f, bc2s = 1.f;
        if (on) sgr_bias_corrections(beta1[t], beta2[t], step[t], &bc1, &bc2s);
        // (k_adam forms lr / bc1 per element; the same float division here, once)
        a.lr_bc1[t] = on ? lr[t] / bc1 : 0.f; a.bc2_sqrt[t] = bc2s;
        a.b1[t] = on ? beta1[t] : 0.f; a.b2[t] = on ? beta2[t] : 0.f;
        a.omb1[t] = on ? sgr_one_minus(beta1[t]) : 0.f; a.omb2[t] = on ? sgr_one_minus(beta2[t]) : 0.f; a.eps[t] = on ? eps[t] : 0.f;
        a.blk_begin[t] = blk;
        if (on) {
            const long long n4 = n[t] / 4;
            long long b = (n4 + 255) / 256;
            if (b > 4096) b = 4096;
            if (b < 1) b = 1;
            blk += (unsigned int)b;
        }
    }
    a.blk_begin[ADAM_MAX_TENSORS] = blk;
    for (int t = n_tensors; t < ADAM_MAX_TENSORS; t++) a.blk_begin[t] = blk;
    hipLaunchKernelGGL(k_adam_multi, dim3(blk), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}
