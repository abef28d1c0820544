// preprocess.hip -- per-Gaussian kernels for gfx950: frustum test, forward preprocess (projection, EWA
// covariance, tile rectangle, SH colour, depth-sort key) and the fused backward preprocess.
//
// Replaces (does not translate) the reference kernels
//   checkFrustum        DGR/cuda_rasterizer/rasterizer_impl.cu:54-66
//   preprocessCUDA fwd  DGR/cuda_rasterizer/forward.cu:155-256  (+ computeCov3D :118-152, computeCov2D :74-113,
//                       computeColorFromSH :20-71, getRect auxiliary.h:46-56, ndc2Pix :41-44)
//   computeCov2DCUDA    DGR/cuda_rasterizer/backward.cu:144-274   } fused into ONE kernel here: the reference
//   preprocessCUDA bwd  DGR/cuda_rasterizer/backward.cu:346-396   } round-trips dL_dmeans / dL_dcov3D through HBM
//
// Arithmetic contract: this translation unit is compiled with -ffp-contract=off; every float op is an
// individually rounded IEEE operation in the order written, which is the order of oracle/cpu_rasterizer.c.
// Radii, tile rectangles and depth keys therefore match the oracle bit for bit.
//
// MI355X notes: one lane per Gaussian, 256-thread blocks (4 waves).  All per-Gaussian state that the
// blend kernels gather later is emitted as ONE 48-byte record (GeomRec), so a gather touches one or two
// 128-B lines instead of the reference's three separate arrays; cov3D is recomputed in the backward
// instead of being stored (24 B/Gaussian of HBM traffic each way for ~40 flops).
#include "../../include/sugar_raster.h"
#include "sgr_common.h"

namespace {

__device__ __constant__ const float SH_C0 = 0.28209479177387814f;
__device__ __constant__ const float SH_C1 = 0.4886025119029199f;
__device__ __constant__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                                 -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                                 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                                 -0.5900435899266435f};

__device__ __forceinline__ float fmin_(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float fmax_(float a, float b) { return a > b ? a : b; }

// float -> int with saturation and NaN -> 0 (what v_cvt_i32_f32 does; spelled out so it cannot be UB)
__device__ __forceinline__ int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

__device__ __forceinline__ float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

}  // namespace

// shared with binning.hip through sgr_device.h
#include "sgr_device.h"

namespace {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 xform4x3(V3 p, const float* m)
{
    V3 t;
    t.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    t.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    t.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return t;
}

__device__ __forceinline__ void quat_to_glmR(const float* q, float R[3][3])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float* scale, float mod, const float* rot, float c6[6])
{
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float R[3][3], M[3][3];
    quat_to_glmR(rot, R);
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) M[c][k] = s[k] * R[c][k];
#define SIG(c, r) (M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2])
    c6[0] = SIG(0, 0); c6[1] = SIG(0, 1); c6[2] = SIG(0, 2);
    c6[3] = SIG(1, 1); c6[4] = SIG(1, 2); c6[5] = SIG(2, 2);
#undef SIG
}

// T = J * Rw2c (2x3), with the 1.3*tanfov clamp of forward.cu:80-87
__device__ __forceinline__ void compute_T(V3 mean, float fx, float fy, float tanx, float tany, const float* v,
                                          float T[2][3], V3& t, float& txtz, float& tytz)
{
    t = xform4x3(mean, v);
    const float limx = 1.3f * tanx;
    const float limy = 1.3f * tany;
    txtz = t.x / t.z;
    tytz = t.y / t.z;
    t.x = fmin_(limx, fmax_(-limx, txtz)) * t.z;
    t.y = fmin_(limy, fmax_(-limy, tytz)) * t.z;
    float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
    float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        T[0][r] = v[4 * r + 0] * J00 + v[4 * r + 2] * J02;
        T[1][r] = v[4 * r + 1] * J11 + v[4 * r + 2] * J12;
    }
}

__device__ __forceinline__ void cov2d_from_T(const float T[2][3], const float* c3, float& a, float& b, float& c)
{
    float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float A[3][2];
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
#pragma unroll
        for (int r = 0; r < 2; r++) A[cc][r] = T[r][0] * V[cc][0] + T[r][1] * V[cc][1] + T[r][2] * V[cc][2];
    a = A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2];
    b = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
    c = A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2];
}

// SH -> RGB for channel c; sh points at this Gaussian's [M][3] coefficient block
__device__ __forceinline__ float sh_channel(int deg, const float* sh, int c, float x, float y, float z)
{
#define SH(k) sh[3 * (k) + c]
    float r = SH_C0 * SH(0);
    if (deg > 0) {
        r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
            if (deg > 2) {
                r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                    SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                    SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                    SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                    SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
            }
        }
    }
#undef SH
    return r + 0.5f;
}

// Load this Gaussian's SH block into registers.  The [P,M,3] layout gives each lane 12*M contiguous
// bytes; with M == 16 and a 16-B aligned base (FAST: decided by the launcher, so the kernel holds ONE load path and the
// loaded values need not meet another path's at a join, which would make the compiler wait for them on the spot) that is
// twelve dwordx4 loads per lane.
template <int MAXC, bool FAST = false>
__device__ __forceinline__ void load_sh(const float* shs, size_t idx, int M, float* sh)
{
    const float* src = shs + idx * (size_t)M * 3;
    const int n = M * 3;
    if (FAST) {
        const float4* s4 = reinterpret_cast<const float4*>(shs) + idx * 12;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            float4 v = s4[i];
            sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
        }
    } else if (MAXC == 48 && n == 48 && ((uintptr_t)src & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int i = 0; i < 12; i++) {
            float4 v = s4[i];
            sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < MAXC; i++) sh[i] = (i < n) ? src[i] : 0.0f;
    }
}
__host__ __device__ __forceinline__ bool sh_fast_layout(const float* shs, int M) { return shs && M == 16 && ((uintptr_t)shs & 15) == 0; }

// d RGB_c / d(view direction) for one colour channel from its 16 coefficients h[k] (backward.cu:99-131): the derivative half
// of computeColorFromSH's backward, shared by the backward preprocess kernel and the SH-Adam kernel.
__device__ __forceinline__ void sh_dir_channel(const int deg, const float* h, const float x, const float y, const float z,
                                               float& dRGBdx, float& dRGBdy, float& dRGBdz)
{
    dRGBdx = 0; dRGBdy = 0; dRGBdz = 0;
    if (deg > 0) {
        dRGBdx = -SH_C1 * h[3]; dRGBdy = -SH_C1 * h[1]; dRGBdz = SH_C1 * h[2];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dRGBdx += SH_C2[0] * y * h[4] + SH_C2[2] * 2.f * -x * h[6] + SH_C2[3] * z * h[7] + SH_C2[4] * 2.f * x * h[8];
            dRGBdy += SH_C2[0] * x * h[4] + SH_C2[1] * z * h[5] + SH_C2[2] * 2.f * -y * h[6] + SH_C2[4] * 2.f * -y * h[8];
            dRGBdz += SH_C2[1] * y * h[5] + SH_C2[2] * 2.f * 2.f * z * h[6] + SH_C2[3] * x * h[7];
            if (deg > 2) {
                dRGBdx += (SH_C3[0] * h[9] * 3.f * 2.f * xy + SH_C3[1] * h[10] * yz + SH_C3[2] * h[11] * -2.f * xy +
                           SH_C3[3] * h[12] * -3.f * 2.f * xz + SH_C3[4] * h[13] * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * h[14] * 2.f * xz + SH_C3[6] * h[15] * 3.f * (xx - yy));
                dRGBdy += (SH_C3[0] * h[9] * 3.f * (xx - yy) + SH_C3[1] * h[10] * xz +
                           SH_C3[2] * h[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * h[12] * -3.f * 2.f * yz +
                           SH_C3[4] * h[13] * -2.f * xy + SH_C3[5] * h[14] * -2.f * yz + SH_C3[6] * h[15] * -3.f * 2.f * xy);
                dRGBdz += (SH_C3[1] * h[10] * xy + SH_C3[2] * h[11] * 4.f * 2.f * yz +
                           SH_C3[3] * h[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * h[13] * 4.f * 2.f * xz +
                           SH_C3[5] * h[14] * (xx - yy));
            }
        }
    }
}

// The 16 SH basis values of computeColorFromSH's backward (backward.cu:47-97: dRGBdsh0 ... dRGBdsh15), zero beyond the active
// degree; the very expressions sh_backward multiplies by dL/dRGB, so basis[k] * dL_dRGB[c] is bit-identical to its DSH(k).
__device__ __forceinline__ void sh_basis16(const int deg, const float x, const float y, const float z, float* b)
{
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = 0.0f;
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.f * zz - xx - yy); b[7] = SH_C2[3] * xz;
            b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.f * xx - yy); b[10] = SH_C3[1] * xy * z; b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy); b[15] = SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

// computeColorFromSH backward for one Gaussian (backward.cu:47-137): dsh[k][c] = basis_k(dir) * masked dL/dRGB[c] and the
// gradient w.r.t. the (normalised) view direction.  `clamped` bit c set = channel c was clamped to 0 by the forward.
__device__ __forceinline__ void sh_backward(const int deg, const float* sh, const float* dcol, const uint32_t clamped,
                                            const float x, const float y, const float z, float* dsh, float* dL_ddir)
{
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define DSH(k) dsh[3 * (k) + c]
        float dL_dRGB = dcol[c] * (((clamped >> c) & 1u) ? 0.0f : 1.0f);
        DSH(0) = SH_C0 * dL_dRGB;
        if (deg > 0) {
            DSH(1) = (-SH_C1 * y) * dL_dRGB; DSH(2) = (SH_C1 * z) * dL_dRGB; DSH(3) = (-SH_C1 * x) * dL_dRGB;
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                DSH(4) = (SH_C2[0] * xy) * dL_dRGB; DSH(5) = (SH_C2[1] * yz) * dL_dRGB;
                DSH(6) = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB; DSH(7) = (SH_C2[3] * xz) * dL_dRGB;
                DSH(8) = (SH_C2[4] * (xx - yy)) * dL_dRGB;
                if (deg > 2) {
                    DSH(9) = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB; DSH(10) = (SH_C3[1] * xy * z) * dL_dRGB;
                    DSH(11) = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
                    DSH(12) = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
                    DSH(13) = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
                    DSH(14) = (SH_C3[5] * z * (xx - yy)) * dL_dRGB; DSH(15) = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
                }
            }
        }
#undef DSH
        float h[16];
#pragma unroll
        for (int k = 0; k < 16; k++) h[k] = sh[3 * k + c];
        float dRGBdx, dRGBdy, dRGBdz;
        sh_dir_channel(deg, h, x, y, z, dRGBdx, dRGBdy, dRGBdz);
        dL_ddir[0] += dRGBdx * dL_dRGB; dL_ddir[1] += dRGBdy * dL_dRGB; dL_ddir[2] += dRGBdz * dL_dRGB;
    }
}

__global__ void __launch_bounds__(256) k_mark_visible(int P, const float* __restrict__ means3D,
                                                      const float* __restrict__ viewmatrix, uint8_t* __restrict__ present)
{
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    V3 p = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
    V3 pv = xform4x3(p, viewmatrix);
    present[idx] = (pv.z <= 0.2f) ? 0 : 1;
}

// The parameter activations of gaussian_model.py:92-117 (exp / F.normalize / sigmoid), applied on the fly in the raw-parameter
// mode.  Same arithmetic, operation for operation, as the stand-alone kernels of activations.hip (which is compiled with FMA
// contraction: the fused multiply-adds are spelled out here).
__device__ __forceinline__ float quat_norm_contracted(const float4 q)
{
    return sqrtf(__builtin_fmaf(q.w, q.w, __builtin_fmaf(q.z, q.z, __builtin_fmaf(q.y, q.y, q.x * q.x))));
}
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

// The camera's 16 + 16 + 3 floats as wave-uniform values: ONE vector load (lane k fetches element k) and a v_readlane per
// element.  Read through the pointers at every use they are per-lane loads (the pointers arrive in a by-value struct, so
// the compiler has no aliasing information to make them scalar loads), each one another round trip in a kernel whose cost
// is round trips.  Call with all 64 lanes active; `wait` the value returned by cam_request.
struct Cam { float vm[16], pm[16], cp[3]; };
__device__ __forceinline__ float cam_request(const float* vmat, const float* pmat, const float* campos)
{
    const int lane = threadIdx.x & 63;
    const float* src = lane < 16 ? vmat + lane : (lane < 32 ? pmat + (lane - 16) : (campos ? campos + min(lane - 32, 2) : vmat));
    return *src;
}
__device__ __forceinline__ void cam_unpack(const float v, Cam& c)
{
    const int iv = __float_as_int(v);
#pragma unroll
    for (int k = 0; k < 16; k++) c.vm[k] = __int_as_float(__builtin_amdgcn_readlane(iv, k));
#pragma unroll
    for (int k = 0; k < 16; k++) c.pm[k] = __int_as_float(__builtin_amdgcn_readlane(iv, 16 + k));
#pragma unroll
    for (int k = 0; k < 3; k++) c.cp[k] = __int_as_float(__builtin_amdgcn_readlane(iv, 32 + k));
}

// Forward preprocess, one lane per Gaussian: GeomRec, radius, tile rectangle and depth-sort key.
// The kernel is a chain of dependent HBM round trips (a few hundred bytes per Gaussian, a few hundred flops), so the
// loads are issued as early as their addresses are known: the mean, scale, rotation, opacity and the camera in one batch
// at the top (lanes past the end re-read the last Gaussian, so nothing is predicated), the 192-byte SH block as soon as the
// Gaussian is known to be in front of the camera, overlapping the covariance arithmetic.
// COLOR: 0 = SH in the fast layout, 1 = SH in any layout, 2 = precomputed colours.  A compile-time mode keeps the SH
// loads free of joins with other paths (at a join the compiler copies the loaded registers, i.e. waits for the loads).
template <int COLOR>
__global__ void __launch_bounds__(256) k_preprocess_fwd(PreprocessArgs a)
{
    __shared__ uint32_t s_mm[2][4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < a.P;
    const int li = valid ? idx : a.P - 1;
    const size_t i3 = 3 * (size_t)li;

    if (blockIdx.x == 0 && !a.keys_elsewhere)  // the depth sort's histograms and chunk tickets start from zero (binning.hip)
        for (int i = threadIdx.x; i < a.n_sort_counters; i += 256) a.sort_counters[i] = 0u;
    if (a.zero_words && blockIdx.x == gridDim.x - 1)
        for (int i = threadIdx.x; i < a.n_zero_words; i += 256) a.zero_words[i] = 0u;
    const float cam_raw = cam_request(a.viewmatrix, a.projmatrix, a.cam_pos);
    const V3 p_orig = {a.means3D[i3], a.means3D[i3 + 1], a.means3D[i3 + 2]};
    float c6[6], sc[3] = {0.f, 0.f, 0.f};
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * (size_t)li + k];
    } else {
        sc[0] = a.scales[i3]; sc[1] = a.scales[i3 + 1]; sc[2] = a.scales[i3 + 2];
        q4 = *reinterpret_cast<const float4*>(a.rotations + 4 * (size_t)li);
    }
    const float op_in = a.opacities[li];
    float pre[3] = {0.f, 0.f, 0.f};
    if (COLOR == 2) { pre[0] = a.colors_precomp[i3]; pre[1] = a.colors_precomp[i3 + 1]; pre[2] = a.colors_precomp[i3 + 2]; }
    Cam cam;
    cam_unpack(cam_raw, cam);

    GeomRec rec;
    rec.radius = 0;
    rec.clamped = 0;
    rec.x = rec.y = rec.cx = rec.cy = rec.cz = rec.opacity = rec.r = rec.g = rec.b = rec.depth = 0.0f;
    int out_radius = 0;
    uint2 rect = make_uint2(0u, 0u);  // {minx | miny << 16, w | h << 16} (w == 0: culled); the last depth-sort pass carries it along

    const float* vm = cam.vm;
    const float* pm = cam.pm;
    V3 p_view = xform4x3(p_orig, vm);
    const bool alive = valid && !(p_view.z <= 0.2f);  // in_frustum, auxiliary.h:152-163

    if (alive) {
        float sh[48];
        constexpr bool use_sh = COLOR != 2;
        if (use_sh) load_sh<48, COLOR == 0>(a.shs, li, a.M, sh);

        float hx = pm[0] * p_orig.x + pm[4] * p_orig.y + pm[8] * p_orig.z + pm[12];
        float hy = pm[1] * p_orig.x + pm[5] * p_orig.y + pm[9] * p_orig.z + pm[13];
        float hw = pm[3] * p_orig.x + pm[7] * p_orig.y + pm[11] * p_orig.z + pm[15];
        float p_w = 1.0f / (hw + 0.0000001f);
        float proj_x = hx * p_w, proj_y = hy * p_w;

        if (!a.cov3D_precomp) {
            if (a.raw_params) {
                sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]);
                const float inv = 1.0f / fmaxf(quat_norm_contracted(q4), 1e-12f);
                q4 = make_float4(q4.x * inv, q4.y * inv, q4.z * inv, q4.w * inv);
            }
            float q[4] = {q4.x, q4.y, q4.z, q4.w};
            cov3d_from_scale_rot(sc, a.scale_modifier, q, c6);
        }
        float T[2][3], txtz, tytz; V3 t;
        compute_T(p_orig, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, vm, T, t, txtz, tytz);
        float cx, cy, cz;
        cov2d_from_T(T, c6, cx, cy, cz);
        cx += 0.3f; cz += 0.3f;
        float det = (cx * cz - cy * cy);
        if (det != 0.0f) {
            float det_inv = 1.f / det;
            float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;
            float mid = 0.5f * (cx + cz);
            float lambda1 = mid + sqrtf(fmax_(0.1f, mid * mid - det));
            float lambda2 = mid - sqrtf(fmax_(0.1f, mid * mid - det));
            float my_radius = ceilf(3.f * sqrtf(fmax_(lambda1, lambda2)));
            float pix_x = ndc2pix(proj_x, a.W), pix_y = ndc2pix(proj_y, a.H);
            int r_int = f2i_sat(my_radius);
            int rx0, ry0, rx1, ry1;
            sgr_get_rect(pix_x, pix_y, r_int, a.gx, a.gy, rx0, ry0, rx1, ry1);
            if ((rx1 - rx0) * (ry1 - ry0) != 0) {
                if (rx1 > rx0 && ry1 > ry0) {
                    rect.x = (uint32_t)rx0 | ((uint32_t)ry0 << 16);
                    rect.y = (uint32_t)(rx1 - rx0) | ((uint32_t)(ry1 - ry0) << 16);
                }
                if (!use_sh) {
                    rec.r = pre[0]; rec.g = pre[1]; rec.b = pre[2];
                } else {
                    float dx = p_orig.x - cam.cp[0], dy = p_orig.y - cam.cp[1], dz = p_orig.z - cam.cp[2];
                    float len = sqrtf(dx * dx + dy * dy + dz * dz);
                    dx = dx / len; dy = dy / len; dz = dz / len;
                    float c0 = sh_channel(a.D, sh, 0, dx, dy, dz);
                    float c1 = sh_channel(a.D, sh, 1, dx, dy, dz);
                    float c2 = sh_channel(a.D, sh, 2, dx, dy, dz);
                    rec.clamped = (c0 < 0 ? 1u : 0u) | (c1 < 0 ? 2u : 0u) | (c2 < 0 ? 4u : 0u);
                    rec.r = fmax_(c0, 0.0f); rec.g = fmax_(c1, 0.0f); rec.b = fmax_(c2, 0.0f);
                }
                rec.x = pix_x; rec.y = pix_y; rec.cx = conx; rec.cy = cony; rec.cz = conz;
                rec.opacity = a.raw_params ? sigmoid_(op_in) : op_in;
                rec.depth = p_view.z;
                rec.radius = r_int;
                out_radius = r_int;
            }
        }
    }
    // key of the global depth sort (binning.hip): depth > 0.2, so the float bit pattern orders like the value
    const uint32_t key = out_radius > 0 ? __float_as_uint(rec.depth) : 0xFFFFFFFFu;
    if (valid) {
        float4* dst = reinterpret_cast<float4*>(a.rec + idx);
        const float4* srcv = reinterpret_cast<const float4*>(&rec);
        dst[0] = srcv[0]; dst[1] = srcv[1]; dst[2] = srcv[2];
        if (a.radii) a.radii[idx] = out_radius;
        if (!a.keys_elsewhere) a.sort_keys[idx] = key;
        a.rect_by_id[idx] = rect;
    }
    if (a.keys_elsewhere) return;   // (uniform: k_depth_keys wrote the keys and their ranges, the sort is already under way)
    // smallest and largest depth key of the workgroup's visible Gaussians: the depth sort works on key - min and drops its
    // fourth pass when the range fits 24 bits (binning.hip)
    uint32_t kmin = key, kmax = key == 0xFFFFFFFFu ? 0u : key;
    for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o));
    }
    if ((threadIdx.x & 63) == 0) { s_mm[0][threadIdx.x >> 6] = kmin; s_mm[1][threadIdx.x >> 6] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0)
        a.key_minmax[blockIdx.x] = make_uint2(min(min(s_mm[0][0], s_mm[0][1]), min(s_mm[0][2], s_mm[0][3])),
                                              max(max(s_mm[1][0], s_mm[1][1]), max(s_mm[1][2], s_mm[1][3])));
}

// The depth-sort keys on their own (round 6).  The sort needs nothing of a Gaussian but its view-space depth, 12 bytes in and 4 out,
// while k_preprocess_fwd is a 59 us HBM-bound stream of 350 bytes per Gaussian: sgr_forward_ex runs this kernel, the sort's histogram
// kernel and its first two passes on a stream of its own BESIDE the preprocess kernel and joins before the last pass (which carries
// the tile rectangles the preprocess kernel writes).  Same arithmetic as above (xform4x3 on the same camera words), so the key of a
// rendered Gaussian is bit-identical to rec.depth.  A Gaussian the preprocess kernel culls for another reason than its depth (zero
// radius, empty tile rectangle) keeps its depth key here -- it is sorted among the others and skipped by the binning passes (its
// rectangle has zero width), so the order of the rendered ones, and with it every tile list, is unchanged.
__global__ void __launch_bounds__(256) k_depth_keys(PreprocessArgs a)
{
    __shared__ uint32_t s_mm[2][4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < a.P;
    const size_t i3 = 3 * (size_t)(valid ? idx : a.P - 1);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.n_sort_counters; i += 256) a.sort_counters[i] = 0u;
    const float cam_raw = cam_request(a.viewmatrix, a.projmatrix, a.cam_pos);
    const V3 p_orig = {a.means3D[i3], a.means3D[i3 + 1], a.means3D[i3 + 2]};
    Cam cam;
    cam_unpack(cam_raw, cam);
    const V3 p_view = xform4x3(p_orig, cam.vm);
    const uint32_t key = (valid && !(p_view.z <= 0.2f)) ? __float_as_uint(p_view.z) : 0xFFFFFFFFu;
    if (valid) a.sort_keys[idx] = key;
    uint32_t kmin = key, kmax = key == 0xFFFFFFFFu ? 0u : key;
    for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o));
    }
    if ((threadIdx.x & 63) == 0) { s_mm[0][threadIdx.x >> 6] = kmin; s_mm[1][threadIdx.x >> 6] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0)
        a.key_minmax[blockIdx.x] = make_uint2(min(min(s_mm[0][0], s_mm[0][1]), min(s_mm[0][2], s_mm[0][3])),
                                              max(max(s_mm[1][0], s_mm[1][1]), max(s_mm[1][2], s_mm[1][3])));
}

// ---------------------------------------------------------------------------------------------
// Fused backward preprocess: K9 (conic grad -> cov2D -> cov3D + mean) and K10 (mean2D -> mean3D, SH backward,
// cov3D -> scale/rotation).  Writes every output row (zeros for culled Gaussians).
// Like the forward, a chain of round trips: everything whose address is known at entry (the record, the nine sums of the
// blend backward, mean, scale, rotation, the camera) is requested in one batch, the SH block as soon as the radius says the
// Gaussian was rendered.  STORE_SH = false is the compact mode (dL_dsh == NULL): the 48 basis products are not formed at all.
// SH: 0 = fast layout, 1 = any layout, 2 = no SH (precomputed colours), compile-time for the reason given at the forward.
#ifndef SGR_PRE_BWD_BLOCKS
#define SGR_PRE_BWD_BLOCKS 3
#endif
// `stage` (STORE_SH only): this lane's column of its wave's LDS staging panel -- element e of the Gaussian's 48 SH-gradient floats at
// stage[65 e] -- or NULL for the direct stores (see k_preprocess_bwd below)
template <bool STORE_SH, int SH>
__device__ __forceinline__ void preprocess_bwd_lane(const PreprocessBwdArgs& a, float* __restrict__ stage)
{
    const int idx0 = blockIdx.x * 256 + threadIdx.x;
    if (a.campos_row && idx0 < 3) a.campos_row[idx0] = a.cam_pos[idx0];  // (see sgr_backward_opts)
    // a sync-free forward whose list outgrew its capacity (or missed its walk hint) did not happen: the gradients are written as
    // zeros (a caller of the autograd API must never see uninitialised memory) and the densification statistics stay untouched
    // (denom would count the repeated step twice); the caller repeats the step.  (The three header words travel with the batch
    // of loads below.)
    uint32_t hdr_r = 0u, hdr_miss = 0u, hdr_ovf = 0u;
    if (a.header) { hdr_r = a.header[SGR_HDR_R]; hdr_miss = a.header[SGR_HDR_HINT_MISS]; hdr_ovf = a.header[4 + SGR_B2_HDR_OVERFLOW]; }
    const bool valid = idx0 < a.P;
    const int idx = valid ? idx0 : a.P - 1;  // lanes past the end re-read the last Gaussian and store nothing
    const size_t i3 = 3 * (size_t)idx;
    const float cam_raw = cam_request(a.viewmatrix, a.projmatrix, a.cam_pos);
    const float4* rp4 = reinterpret_cast<const float4*>(a.rec + idx);
    const float4 rec0 = rp4[0], rec1 = rp4[1], rec2 = rp4[2];  // {x,y,cx,cy} {cz,opacity,depth,radius} {r,g,b,clamped}
    const float4* ap = reinterpret_cast<const float4*>(a.acc + SGR_ACC_STRIDE * (size_t)idx);
    const float4 s0 = ap[0], s1 = ap[1];  // {k0,k1,k2,S0} {Sx,Sy,Sxx,Sxy} {Syy,-,-,-}
    const float s2x = a.acc[SGR_ACC_STRIDE * (size_t)idx + 8];
    const V3 mean = {a.means3D[i3], a.means3D[i3 + 1], a.means3D[i3 + 2]};
    float sc[3] = {0, 0, 0}, q[4] = {0, 0, 0, 0};
    float4 q_raw = make_float4(0.f, 0.f, 0.f, 0.f);
    float c6[6];
    if (a.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * (size_t)idx + k];
    } else {
        sc[0] = a.scales[i3]; sc[1] = a.scales[i3 + 1]; sc[2] = a.scales[i3 + 2];
        q_raw = *reinterpret_cast<const float4*>(a.rotations + 4 * (size_t)idx);
    }
    const int radius = __float_as_int(rec1.w);
    const uint32_t clamped = __float_as_uint(rec2.w);
    Cam cam;
    cam_unpack(cam_raw, cam);
    if (!valid) return;
    const bool fwd_invalid = hdr_r > a.list_cap || hdr_miss != 0u || hdr_ovf != 0u;  // SGR_FORWARD_INVALID: zero gradients, no statistics
    const float* v = cam.vm;

    float dmean[3] = {0, 0, 0}, dcov[6] = {0, 0, 0, 0, 0, 0};
    const int n_sh = a.M * 3;
    if (fwd_invalid || !(radius > 0)) {
        if (a.dL_dmean2D) { a.dL_dmean2D[i3] = 0; a.dL_dmean2D[i3 + 1] = 0; a.dL_dmean2D[i3 + 2] = 0; }
        if (a.dL_dconic) { float4 z4 = {0, 0, 0, 0}; *reinterpret_cast<float4*>(a.dL_dconic + 4 * (size_t)idx) = z4; }
        a.dL_dopacity[idx] = 0;
        if (a.dL_dcolor) { a.dL_dcolor[i3] = 0; a.dL_dcolor[i3 + 1] = 0; a.dL_dcolor[i3 + 2] = 0; }
        a.dL_dmean3D[i3] = 0; a.dL_dmean3D[i3 + 1] = 0; a.dL_dmean3D[i3 + 2] = 0;
        if (a.dL_dcov3D) for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)idx + k] = 0;
        if (STORE_SH) {
            if (stage) {
#pragma unroll
                for (int k = 0; k < 48; k++) stage[65 * k] = 0.0f;
            } else {
                for (int k = 0; k < n_sh; k++) a.dL_dsh[(size_t)idx * n_sh + k] = 0;
            }
        }
        if (a.dL_dscale) { a.dL_dscale[i3] = 0; a.dL_dscale[i3 + 1] = 0; a.dL_dscale[i3 + 2] = 0; }
        if (a.dL_drot) { float4 z = {0, 0, 0, 0}; *reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx) = z; }
        return;
    }
    float sh[48];
    if (SH != 2) load_sh<48, SH == 0>(a.shs, idx, a.M, sh);
    if (!a.cov3D_precomp) {
        float4 q4 = q_raw;
        if (a.raw_params) {
            sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]);
            const float inv = 1.0f / fmaxf(quat_norm_contracted(q4), 1e-12f);
            q4 = make_float4(q4.x * inv, q4.y * inv, q4.z * inv, q4.w * inv);
        }
        q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
        cov3d_from_scale_rot(sc, a.scale_modifier, q, c6);
    }
    // ---- finish the blend backward: apply the per-Gaussian coefficients of backward.cu:538-554 to the nine sums
    float dcol[3], dm2x, dm2y;
    float g0, g1, g3;
    {
        const float op = rec1.y, cx = rec0.z, cy = rec0.w, cz = rec1.x;
        dcol[0] = s0.x; dcol[1] = s0.y; dcol[2] = s0.z;
        const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
        dm2x = -(op * ddelx_dx) * (cx * s1.x + cy * s1.y);
        dm2y = -(op * ddely_dy) * (cz * s1.y + cy * s1.x);
        g0 = -0.5f * op * s1.z; g1 = -0.5f * op * s1.w; g3 = -0.5f * op * s2x;
        if (a.dL_dmean2D) { a.dL_dmean2D[i3] = dm2x; a.dL_dmean2D[i3 + 1] = dm2y; a.dL_dmean2D[i3 + 2] = 0; }
        // densification statistics of the train loop, fused (train.py:111-123 / sugar_densifier.py:156-164: over the
        // visibility filter radii > 0 -- this branch -- max_radii2D = max(., radii), accum += |viewspace grad .xy|, denom += 1)
        if (a.dens_accum) a.dens_accum[idx] += sqrtf(dm2x * dm2x + dm2y * dm2y);
        if (a.dens_denom) a.dens_denom[idx] += 1.0f;
        if (a.dens_max_radii) a.dens_max_radii[idx] = fmaxf(a.dens_max_radii[idx], (float)radius);
        float4 gc = {g0, g1, 0.f, g3};
        if (a.dL_dconic) *reinterpret_cast<float4*>(a.dL_dconic + 4 * (size_t)idx) = gc;
        // raw mode: d sigmoid = s (1 - s), the activated opacity is in the record (sgr_activations_backward's arithmetic)
        a.dL_dopacity[idx] = a.raw_params ? s0.w * op * (1.0f - op) : s0.w;
        if (a.dL_dcolor) {
            // (sh_dir_elsewhere: compact mode without the SH block -- the clamp-masked colour gradients leave here and
            // k_sh_adam_from_views forms dRGB/d(view direction) -> dL/dmean next to the SH coefficients it loads anyway)
            const bool mk = a.sh_dir_elsewhere != 0;
            a.dL_dcolor[i3] = (mk && (clamped & 1u)) ? 0.f : dcol[0];
            a.dL_dcolor[i3 + 1] = (mk && (clamped & 2u)) ? 0.f : dcol[1];
            a.dL_dcolor[i3 + 2] = (mk && (clamped & 4u)) ? 0.f : dcol[2];
        }
    }
    // ---- K9, backward.cu:144-274
    {
        float T[2][3], txtz, tytz; V3 t;
        compute_T(mean, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, v, T, t, txtz, tytz);
        const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
        float ca, cb, cc;
        cov2d_from_T(T, c6, ca, cb, cc);
        ca += 0.3f; cc += 0.3f;
        float denom = ca * cc - cb * cb;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * g0 + 2 * cb * cc * g1 + (denom - ca * cc) * g3);
            dL_dc = denom2inv * (-ca * ca * g3 + 2 * ca * cb * g1 + (denom - ca * cc) * g0);
            dL_db = denom2inv * 2 * (cb * cc * g0 - (denom + 2 * cb * cb) * g1 + ca * cb * g3);
            dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
            dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
            dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
            dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        }
        float dL_dT00 = 2 * (T[0][0] * V[0][0] + T[0][1] * V[0][1] + T[0][2] * V[0][2]) * dL_da + (T[1][0] * V[0][0] + T[1][1] * V[0][1] + T[1][2] * V[0][2]) * dL_db;
        float dL_dT01 = 2 * (T[0][0] * V[1][0] + T[0][1] * V[1][1] + T[0][2] * V[1][2]) * dL_da + (T[1][0] * V[1][0] + T[1][1] * V[1][1] + T[1][2] * V[1][2]) * dL_db;
        float dL_dT02 = 2 * (T[0][0] * V[2][0] + T[0][1] * V[2][1] + T[0][2] * V[2][2]) * dL_da + (T[1][0] * V[2][0] + T[1][1] * V[2][1] + T[1][2] * V[2][2]) * dL_db;
        float dL_dT10 = 2 * (T[1][0] * V[0][0] + T[1][1] * V[0][1] + T[1][2] * V[0][2]) * dL_dc + (T[0][0] * V[0][0] + T[0][1] * V[0][1] + T[0][2] * V[0][2]) * dL_db;
        float dL_dT11 = 2 * (T[1][0] * V[1][0] + T[1][1] * V[1][1] + T[1][2] * V[1][2]) * dL_dc + (T[0][0] * V[1][0] + T[0][1] * V[1][1] + T[0][2] * V[1][2]) * dL_db;
        float dL_dT12 = 2 * (T[1][0] * V[2][0] + T[1][1] * V[2][1] + T[1][2] * V[2][2]) * dL_dc + (T[0][0] * V[2][0] + T[0][1] * V[2][1] + T[0][2] * V[2][2]) * dL_db;
        float dL_dJ00 = v[0] * dL_dT00 + v[4] * dL_dT01 + v[8] * dL_dT02;
        float dL_dJ02 = v[2] * dL_dT00 + v[6] * dL_dT01 + v[10] * dL_dT02;
        float dL_dJ11 = v[1] * dL_dT10 + v[5] * dL_dT11 + v[9] * dL_dT12;
        float dL_dJ12 = v[2] * dL_dT10 + v[6] * dL_dT11 + v[10] * dL_dT12;
        float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        float h_x = a.focal_x, h_y = a.focal_y;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
        dmean[0] = v[0] * dL_dtx + v[1] * dL_dty + v[2] * dL_dtz;
        dmean[1] = v[4] * dL_dtx + v[5] * dL_dty + v[6] * dL_dtz;
        dmean[2] = v[8] * dL_dtx + v[9] * dL_dty + v[10] * dL_dtz;
    }
    // ---- K10, backward.cu:346-396
    {
        const float* proj = cam.pm;
        float hw = proj[3] * mean.x + proj[7] * mean.y + proj[11] * mean.z + proj[15];
        float m_w = 1.0f / (hw + 0.0000001f);
        float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        float gx = dm2x, gy = dm2y;
        float ddx = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
        float ddy = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
        float ddz = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        dmean[0] += ddx; dmean[1] += ddy; dmean[2] += ddz;
    }
    if (SH != 2) {  // computeColorFromSH backward, backward.cu:20-139
        V3 dir_orig = {mean.x - cam.cp[0], mean.y - cam.cp[1], mean.z - cam.cp[2]};
        float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
        float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
        const int deg = a.D;
        // masked dL/dRGB (backward.cu:42-45): the clamped channels pass no gradient
        const float dLm[3] = {dcol[0] * ((clamped & 1u) ? 0.0f : 1.0f), dcol[1] * ((clamped & 2u) ? 0.0f : 1.0f),
                              dcol[2] * ((clamped & 4u) ? 0.0f : 1.0f)};
        if (STORE_SH) {
            // dL/dsh[k][c] = basis_k(dir) * masked dL/dRGB[c] (backward.cu:47-97) does not read the coefficients: it is formed from
            // the 16 basis values and stored four floats at a time BEFORE the 48 coefficients are consumed below.  (Held as 48
            // values until the end of the kernel, next to the 48 coefficients in flight, this variant needed 168 VGPRs + 24 spilled
            // ones -- 100 bytes of scratch per lane in the kernel the unmodified 3DGS / SuGaR callers run every step.)
            float b[16];
            sh_basis16(deg, x, y, z, b);
            const int n_act = (deg + 1) * (deg + 1);
            float* dst = a.dL_dsh + (size_t)idx * n_sh;
            if (stage) {
                // (round 5) through the wave's LDS panel: the kernel's epilogue writes the wave's 64 rows as ONE contiguous 12 KB
                // stream.  Stored from here, every instruction put 16 bytes into each of 64 rows 192 bytes apart: the kernel ran at
                // 2.6 TB/s for its 430 MB where the compact variant, which does not write the 192 MB, runs at 4.8
#pragma unroll
                for (int e = 0; e < 48; e++) stage[65 * e] = (e / 3) < n_act ? b[e / 3] * dLm[e % 3] : 0.0f;
            } else if (n_sh == 48 && ((uintptr_t)dst & 15) == 0) {
                float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int e = 4 * i + j, k = e / 3, c = e % 3;
                        o[j] = k < n_act ? b[k] * dLm[c] : 0.0f;
                    }
                    d4[i] = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 48; e++)
                    if (e < n_sh) dst[e] = (e / 3) < n_act ? b[e / 3] * dLm[e % 3] : 0.0f;
            }
        }
        float dL_ddir[3] = {0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {  // the view-direction half (backward.cu:99-131), one colour channel at a time
            float h[16];
#pragma unroll
            for (int k = 0; k < 16; k++) h[k] = sh[3 * k + c];
            float dRGBdx, dRGBdy, dRGBdz;
            sh_dir_channel(deg, h, x, y, z, dRGBdx, dRGBdy, dRGBdz);
            dL_ddir[0] += dRGBdx * dLm[c]; dL_ddir[1] += dRGBdy * dLm[c]; dL_ddir[2] += dRGBdz * dLm[c];
        }
        // dnormvdv, auxiliary.h:107-117
        {
            V3 vv = dir_orig;
            float sum2 = vv.x * vv.x + vv.y * vv.y + vv.z * vv.z;
            float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            float ox = ((+sum2 - vv.x * vv.x) * dL_ddir[0] - vv.y * vv.x * dL_ddir[1] - vv.z * vv.x * dL_ddir[2]) * invsum32;
            float oy = (-vv.x * vv.y * dL_ddir[0] + (sum2 - vv.y * vv.y) * dL_ddir[1] - vv.z * vv.y * dL_ddir[2]) * invsum32;
            float oz = (-vv.x * vv.z * dL_ddir[0] - vv.y * vv.z * dL_ddir[1] + (sum2 - vv.z * vv.z) * dL_ddir[2]) * invsum32;
            dmean[0] += ox; dmean[1] += oy; dmean[2] += oz;
        }
        if (!STORE_SH && a.dL_dcolor) {
            // compact mode (view-sharded training): the SH gradient is the outer product basis(dir) x (masked dL/dRGB), so
            // only the 3 masked colour gradients leave this kernel; sgr_sh_grad_from_views rebuilds sum over views later
            a.dL_dcolor[i3] = ((clamped >> 0) & 1u) ? 0.f : dcol[0];
            a.dL_dcolor[i3 + 1] = ((clamped >> 1) & 1u) ? 0.f : dcol[1];
            a.dL_dcolor[i3 + 2] = ((clamped >> 2) & 1u) ? 0.f : dcol[2];
        }
    }
    if (a.scales) {  // computeCov3D backward, backward.cu:278-341
        float r = q[0], x = q[1], y = q[2], z = q[3];
        float R[3][3], M[3][3];
        quat_to_glmR(q, R);
        float s[3] = {a.scale_modifier * sc[0], a.scale_modifier * sc[1], a.scale_modifier * sc[2]};
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int k = 0; k < 3; k++) M[c][k] = s[k] * R[c][k];
        const float* d = dcov;
        float Dm[3][3] = {{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}};
        float X[3][3], dM[3][3], dMt[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int k = 0; k < 3; k++) X[c][k] = 2.0f * M[c][k];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) dM[c][rr] = X[0][rr] * Dm[c][0] + X[1][rr] * Dm[c][1] + X[2][rr] * Dm[c][2];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) dMt[c][rr] = dM[rr][c];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float ds = R[0][k] * dMt[k][0] + R[1][k] * dMt[k][1] + R[2][k] * dMt[k][2];
            a.dL_dscale[i3 + k] = a.raw_params ? ds * sc[k] : ds;  // raw mode: d exp = exp
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) dMt[k][rr] *= s[k];
        float4 dq;
        dq.x = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
        dq.y = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
        dq.z = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
        dq.w = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
        if (a.raw_params) {
            // d (v / |v|) = (I - n n^T) / |v|   (sgr_activations_backward's arithmetic, its FMA contraction spelled out)
            const float n = quat_norm_contracted(q_raw);
            if (n > 1e-12f) {
                const float inv = 1.0f / n;
                const float nx = q_raw.x * inv, ny = q_raw.y * inv, nz = q_raw.z * inv, nw = q_raw.w * inv;
                const float dot = __builtin_fmaf(nw, dq.w, __builtin_fmaf(nz, dq.z, __builtin_fmaf(ny, dq.y, nx * dq.x)));
                dq = make_float4(__builtin_fmaf(-nx, dot, dq.x) * inv, __builtin_fmaf(-ny, dot, dq.y) * inv,
                                 __builtin_fmaf(-nz, dot, dq.z) * inv, __builtin_fmaf(-nw, dot, dq.w) * inv);
            } else {
                dq = make_float4(dq.x * 1e12f, dq.y * 1e12f, dq.z * 1e12f, dq.w * 1e12f);
            }
        }
        *reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx) = dq;
    }
    a.dL_dmean3D[i3] = dmean[0]; a.dL_dmean3D[i3 + 1] = dmean[1]; a.dL_dmean3D[i3 + 2] = dmean[2];
    if (a.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)idx + k] = dcov[k];
    }
}

template <bool STORE_SH, int SH>
__global__ void __launch_bounds__(256, SGR_PRE_BWD_BLOCKS) k_preprocess_bwd(PreprocessBwdArgs a)
{
    // The full SH gradient (dL_dsh[P,16,3]: what every caller of the reference-shaped API receives) leaves through LDS: a lane
    // owns a Gaussian, i.e. a 192-byte row; the wave's 64 rows are contiguous in memory, so each lane drops its 48 values into its
    // column of a [48][65] panel (conflict-free both ways) and the wave then streams the 12 KB out as 16-byte stores of 64
    // consecutive lanes.  4 panels of 12.5 KB per block, 3 blocks per CU: 150 KB of the 160.
    __shared__ float s_stage[STORE_SH ? 4 * 48 * 65 : 1];
    const bool staged = STORE_SH && a.M == 16 && (((uintptr_t)a.dL_dsh) & 15) == 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* panel = s_stage + (STORE_SH ? wave * (48 * 65) : 0);
    preprocess_bwd_lane<STORE_SH, SH>(a, staged ? panel + lane : nullptr);
    if (STORE_SH && staged) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const long long row0 = (long long)blockIdx.x * 256 + wave * 64;
        const int rows = (int)min(64ll, (long long)a.P - row0);          // (<= 0: the wave lies past the last Gaussian)
        float4* out4 = reinterpret_cast<float4*>(a.dL_dsh + (size_t)(row0 > 0 ? row0 : 0) * 48);
#pragma unroll
        for (int it = 0; it < 12; it++) {
            const int j4 = it * 64 + lane;                                // float4 index in the wave's stream
            if (j4 < rows * 12) {
                const int row = j4 / 12, e0 = (j4 % 12) * 4;
                const float* src = panel + 65 * e0 + row;
                out4[j4] = make_float4(src[0], src[65], src[130], src[195]);
            }
        }
    }
}

// dL/dsh[k][c] = sum over views v of  basis_k(dir_v) * g_v[c]   with dir_v = normalize(mean - campos_v) and g_v the
// clamp-masked dL/dRGB of view v -- exactly the per-view SH backward (backward.cu:47-97) summed over views, but the views
// exchange 3 floats per Gaussian instead of 3*M.  dirs use the same arithmetic as the forward (glm::length, division).
// The clamp-masked colour gradients of the compact mode on their own: they only need the blend backward's sums, so a
// trainer can start exchanging them while the backward preprocess is still running (sgr_backward_phase).
__global__ void __launch_bounds__(256) k_masked_colors(int P, const GeomRec* __restrict__ rec, const float* __restrict__ acc,
                                                       float* __restrict__ out, const float* __restrict__ cam_pos,
                                                       float* __restrict__ campos_row)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (campos_row && idx < 3) campos_row[idx] = cam_pos[idx];  // the camera centre travels with the colours (one send buffer)
    if (idx >= P) return;
    const GeomRec* rp = rec + idx;
    const size_t i3 = 3 * (size_t)idx;
    if (!(rp->radius > 0)) { out[i3] = 0.f; out[i3 + 1] = 0.f; out[i3 + 2] = 0.f; return; }
    const uint32_t clamped = rp->clamped;
    const float4 s0 = *reinterpret_cast<const float4*>(acc + SGR_ACC_STRIDE * (size_t)idx);
    out[i3] = (clamped & 1u) ? 0.f : s0.x;
    out[i3 + 1] = (clamped & 2u) ? 0.f : s0.y;
    out[i3 + 2] = (clamped & 4u) ? 0.f : s0.z;
}

// dL/dmean3D through the view direction, summed over the views (the tail of the backward preprocess kernel,
// backward.cu:99-139 + dnormvdv, moved next to the only other reader of the SH coefficients; `row` = this Gaussian's 48
// coefficients in LDS).  Same arithmetic, operation for operation, as k_preprocess_bwd (sh_backward).  One colour channel at
// a time, its 16 coefficients fetched from the LDS row: with all 48 in registers next to the expanded polynomial the kernel
// needed 190 VGPRs, two waves per SIMD, and lost in bandwidth what the backward preprocess kernel saved.
__device__ __forceinline__ void sh_dir_sum_over_views(int idx, size_t P, int V, int D, const float* __restrict__ means3D,
                                                      const float* __restrict__ campos, const float* __restrict__ dcolor,
                                                      const float* row, float* dm)
{
    const size_t i3 = 3 * (size_t)idx;
    const float mx = means3D[i3], my = means3D[i3 + 1], mz = means3D[i3 + 2];
    dm[0] = 0.f; dm[1] = 0.f; dm[2] = 0.f;
    for (int v = 0; v < V; v++) {
        const float* g = dcolor + ((size_t)v * P + idx) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;  // culled or fully clamped in this view
        const float dx = mx - campos[3 * v], dy = my - campos[3 * v + 1], dz = mz - campos[3 * v + 2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx / len, y = dy / len, z = dz / len;
        float dL_ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c = 0; c < 3; c++) {
            float h[16];
#pragma unroll
            for (int k = 0; k < 16; k++) h[k] = row[3 * k + c];
            const float dL_dRGB = (c == 0 ? g0 : (c == 1 ? g1 : g2)) * 1.0f;  // (sh_backward's mask factor: the colours arrive masked)
            float dRGBdx, dRGBdy, dRGBdz;
            sh_dir_channel(D, h, x, y, z, dRGBdx, dRGBdy, dRGBdz);
            dL_ddir[0] += dRGBdx * dL_dRGB; dL_ddir[1] += dRGBdy * dL_dRGB; dL_ddir[2] += dRGBdz * dL_dRGB;
        }
        // dnormvdv, auxiliary.h:107-117
        const float sum2 = dx * dx + dy * dy + dz * dz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dm[0] += ((+sum2 - dx * dx) * dL_ddir[0] - dy * dx * dL_ddir[1] - dz * dx * dL_ddir[2]) * invsum32;
        dm[1] += (-dx * dy * dL_ddir[0] + (sum2 - dy * dy) * dL_ddir[1] - dz * dy * dL_ddir[2]) * invsum32;
        dm[2] += (-dx * dz * dL_ddir[0] - dy * dz * dL_ddir[1] + (sum2 - dz * dz) * dL_ddir[2]) * invsum32;
    }
}

__device__ __forceinline__ void sh_grad_sum_over_views(int idx, size_t P, int V, int D, const float* __restrict__ means3D,
                                                       const float* __restrict__ campos, const float* __restrict__ dcolor,
                                                       float* acc)
{
    const size_t i3 = 3 * (size_t)idx;
    const float mx = means3D[i3], my = means3D[i3 + 1], mz = means3D[i3 + 2];
#pragma unroll
    for (int k = 0; k < 48; k++) acc[k] = 0.f;
    for (int v = 0; v < V; v++) {
        const float* g = dcolor + ((size_t)v * P + idx) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;  // culled or fully clamped in this view
        float dx = mx - campos[3 * v], dy = my - campos[3 * v + 1], dz = mz - campos[3 * v + 2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx / len, y = dy / len, z = dz / len;
        float b[16];
        b[0] = SH_C0;
        if (D > 0) {
            b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.f * zz - xx - yy);
                b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
                if (D > 2) {
                    b[9] = SH_C3[0] * y * (3.f * xx - yy); b[10] = SH_C3[1] * xy * z;
                    b[11] = SH_C3[2] * y * (4.f * zz - xx - yy); b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                    b[13] = SH_C3[4] * x * (4.f * zz - xx - yy); b[14] = SH_C3[5] * z * (xx - yy);
                    b[15] = SH_C3[6] * x * (xx - 3.f * yy);
                }
            }
        }
        const int nb = (D + 1) * (D + 1);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < nb) { acc[3 * k] += b[k] * g0; acc[3 * k + 1] += b[k] * g1; acc[3 * k + 2] += b[k] * g2; }
        }
    }
}

__global__ void __launch_bounds__(256) k_sh_grad_from_views(int P, int V, int D, int M, size_t vstride,
                                                            const float* __restrict__ means3D,
                                                            const float* __restrict__ campos, const float* __restrict__ dcolor,
                                                            float* __restrict__ dL_dsh)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    float acc[48];
    sh_grad_sum_over_views(idx, vstride, V, D, means3D, campos, dcolor, acc);
    const int n_sh = 3 * M;
    float* dst = dL_dsh + (size_t)idx * n_sh;
    if (n_sh == 48 && ((uintptr_t)dst & 15) == 0) {
        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int i = 0; i < 12; i++) { float4 o = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]}; d4[i] = o; }
    } else {
#pragma unroll
        for (int i = 0; i < 48; i++) if (i < n_sh) dst[i] = acc[i];
    }
}

// The same sum, consumed on the spot: Adam on the Gaussian's SH coefficients (adam.hip's update, lr_dc for the three DC
// values and lr_rest for the others: gaussian_model.py:157-158) without ever writing the 48-float gradient to memory.
struct ShAdamArgs {
    float lr_dc, lr_rest, b1, b2, omb1, omb2, eps, bc1, bc2_sqrt, grad_scale;
    const uint32_t* guard; uint32_t guard_cap;  // forward header + list capacity: the step is a no-op if that forward was invalid (NULL: no check)
};

// One wave per 64 Gaussians.  Phase 1: lane = Gaussian, the 48 sums in registers, dropped into an LDS panel (row stride 52
// floats: conflict-free for 16-byte accesses).  Phase 2: the wave walks the 64 x 48 floats of parameters and moments as
// one contiguous stream, one float4 per lane and step (a lane-per-Gaussian walk touches 64 cache lines per instruction
// and ran at 2 TB/s).
#define SHA_STRIDE 52
// DIR (dmean_extra != NULL, M == 16): phase 1 also loads the Gaussian's 48 coefficients (the wave's 12 KB block, which
// phase 2 then streams out of the cache) and writes the view-direction part of dL/dmean3D, summed over the views, to
// dmean_extra[P][3]; the flat Adam kernel adds it to the position gradient (sgr_adam_step_ex).
template <bool DIR>
__global__ void __launch_bounds__(64) k_sh_adam_from_views(int P, int V, int D, int M, size_t vstride,
                                                           const float* __restrict__ means3D,
                                                           const float* __restrict__ campos, const float* __restrict__ dcolor,
                                                           float* __restrict__ sh, float* __restrict__ exp_avg,
                                                           float* __restrict__ exp_avg_sq, ShAdamArgs a,
                                                           float* __restrict__ dmean_extra)
{
    __shared__ __attribute__((aligned(16))) float s_g[64 * SHA_STRIDE];
    if (a.guard && SGR_FORWARD_INVALID(a.guard, a.guard_cap)) return;
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const int idx = g0 + lane;
    float4 pkeep[12];  // DIR: the wave's 64 x 48 coefficients in the streaming layout of phase 2 (read from memory once)
    if (DIR) {
        // coalesced into the panel (row = Gaussian) for the direction term, and kept in registers for the Adam update
        // (12 loads at a 192-byte lane stride straight from memory cost this streaming kernel 25 us; re-reading the block in
        // phase 2 gives back what the backward preprocess kernel saved)
        const float4* p4 = reinterpret_cast<const float4*>(sh + (size_t)g0 * 48);
        const int n4 = min(64, P - g0) * 12;
#pragma unroll
        for (int st = 0; st < 12; st++) pkeep[st] = p4[min(st * 64 + lane, n4 - 1)];
#pragma unroll
        for (int st = 0; st < 12; st++) {
            const int e4 = st * 64 + lane;
            if (e4 < n4) { const int g = e4 / 12; *reinterpret_cast<float4*>(s_g + g * SHA_STRIDE + 4 * (e4 - g * 12)) = pkeep[st]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (DIR && idx < P) {
        float dm[3];
        sh_dir_sum_over_views(idx, vstride, V, D, means3D, campos, dcolor, s_g + lane * SHA_STRIDE, dm);
        dmean_extra[3 * (size_t)idx] = dm[0]; dmean_extra[3 * (size_t)idx + 1] = dm[1]; dmean_extra[3 * (size_t)idx + 2] = dm[2];
    }
    if (DIR) {  // every lane is done with its coefficients before any lane overwrites a row with gradient sums
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    {
        float acc[48];
        if (idx < P) {
            sh_grad_sum_over_views(idx, vstride, V, D, means3D, campos, dcolor, acc);
        } else {
#pragma unroll
            for (int k = 0; k < 48; k++) acc[k] = 0.f;
        }
        float4* row = reinterpret_cast<float4*>(s_g + lane * SHA_STRIDE);
#pragma unroll
        for (int i = 0; i < 12; i++) row[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n_sh = 3 * M;
    const int live = min(64, P - g0);
    auto upd = [&](int c, float g, float& p, float& m, float& v) {
        g *= a.grad_scale;
        m = a.b1 * m + a.omb1 * g;
        v = a.b2 * v + a.omb2 * g * g;
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        p -= ((c < 3 ? a.lr_dc : a.lr_rest) / a.bc1) * (m / denom);
    };
    const size_t base = (size_t)g0 * n_sh;
    if (n_sh == 48 && ((((uintptr_t)sh | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0)) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4* p4 = reinterpret_cast<f4*>(sh + base);
        f4* m4 = reinterpret_cast<f4*>(exp_avg + base);
        f4* v4 = reinterpret_cast<f4*>(exp_avg_sq + base);
        const int n4 = live * 12;
        // four steps at a time: their twelve loads go out together (one at a time -- loads under the lane predicate, every
        // step waiting for its own -- a wave made twelve serial round trips; addresses are clamped instead)
        constexpr int NB = DIR ? 2 : 4;  // (DIR holds the parameters in registers already: smaller batches keep it at 3 waves per SIMD)
#pragma unroll
        for (int st0 = 0; st0 < 12; st0 += NB) {
            f4 pv[NB], mv[NB], vv[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int e4 = min((st0 + u) * 64 + lane, n4 - 1);  // float4 index inside the wave's 64 x 48 block
                // the moments are touched once per step: stream them past the caches, the parameters are read again by
                // the next forward and should stay in the last-level cache
                if (DIR) { const float4 k = pkeep[st0 + u]; pv[u] = (f4){k.x, k.y, k.z, k.w}; }
                else pv[u] = p4[e4];
                mv[u] = __builtin_nontemporal_load(&m4[e4]); vv[u] = __builtin_nontemporal_load(&v4[e4]);
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int e4 = (st0 + u) * 64 + lane;
                if (e4 < n4) {
                    const int g = e4 / 12, c4 = e4 - g * 12;
                    const float4 gr = *reinterpret_cast<const float4*>(s_g + g * SHA_STRIDE + 4 * c4);
                    float p[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w}, m[4] = {mv[u].x, mv[u].y, mv[u].z, mv[u].w};
                    float v[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
                    const float gq[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) upd(4 * c4 + k, gq[k], p[k], m[k], v[k]);
                    const f4 po = {p[0], p[1], p[2], p[3]}, mo = {m[0], m[1], m[2], m[3]}, vo = {v[0], v[1], v[2], v[3]};
                    p4[e4] = po;
                    __builtin_nontemporal_store(mo, &m4[e4]);
                    __builtin_nontemporal_store(vo, &v4[e4]);
                }
            }
        }
    } else {
        const int n = live * n_sh;
        for (int e = lane; e < n; e += 64) {
            const int g = e / n_sh, c = e - g * n_sh;
            float p = sh[base + e], m = exp_avg[base + e], v = exp_avg_sq[base + e];
            upd(c, s_g[g * SHA_STRIDE + c], p, m, v);
            sh[base + e] = p; exp_avg[base + e] = m; exp_avg_sq[base + e] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SuGaR.get_points_rgb (sugar_scene/sugar_model.py:839-883): colours = clamp_min(eval_sh(levels-1, sh, dir) + 0.5, 0) with
// dir = F.normalize(positions - camera_centers) (or given directions).  The reference spends ~30 elementwise launches and
// their autograd twins on this every training step (it feeds `colors_precomp`); here it is one lane per point, forward
// and backward, on the same SH basis code as the rasterizer.  sh rows have a stride of `M` coefficients; `n` are used.
struct ShRgbArgs {
    int P, D, M, n;            // D = sh_levels - 1, n = (D+1)^2
    const float* sh;           // [P, M, 3]
    const float* positions;    // [P, 3] or null (then `directions` is used as is)
    const float* centers;      // [n_centers, 3], n_centers in {1, P}
    int n_centers;
    const float* directions;   // [P, 3] or null
};

__device__ __forceinline__ void sh_rgb_dir(const ShRgbArgs& a, int idx, float& x, float& y, float& z, float& vx, float& vy,
                                           float& vz, float& len)
{
    const size_t i3 = 3 * (size_t)idx;
    if (a.positions) {
        const size_t c3 = a.n_centers == 1 ? 0 : i3;
        vx = a.positions[i3] - a.centers[c3]; vy = a.positions[i3 + 1] - a.centers[c3 + 1]; vz = a.positions[i3 + 2] - a.centers[c3 + 2];
        len = fmax_(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);  // F.normalize: v / max(|v|, eps)
        x = vx / len; y = vy / len; z = vz / len;
    } else {
        x = a.directions[i3]; y = a.directions[i3 + 1]; z = a.directions[i3 + 2];
        vx = x; vy = y; vz = z; len = 1.f;
    }
}

__device__ __forceinline__ void load_sh_n(const float* shs, size_t idx, int M, int n, float* sh)
{
    if (n == M) { load_sh<48>(shs, idx, M, sh); return; }
    const float* src = shs + idx * (size_t)M * 3;
#pragma unroll
    for (int i = 0; i < 48; i++) sh[i] = (i < 3 * n) ? src[i] : 0.0f;
}

__global__ void __launch_bounds__(256) k_sh_to_rgb_fwd(ShRgbArgs a, float* __restrict__ colors)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.P) return;
    float x, y, z, vx, vy, vz, len;
    sh_rgb_dir(a, idx, x, y, z, vx, vy, vz, len);
    float sh[48];
    load_sh_n(a.sh, idx, a.M, a.n, sh);
    const size_t i3 = 3 * (size_t)idx;
    colors[i3] = fmax_(sh_channel(a.D, sh, 0, x, y, z), 0.0f);
    colors[i3 + 1] = fmax_(sh_channel(a.D, sh, 1, x, y, z), 0.0f);
    colors[i3 + 2] = fmax_(sh_channel(a.D, sh, 2, x, y, z), 0.0f);
}

// dL_dsh gets every one of the M rows (zeros beyond the n used ones: it is the gradient of the full coefficient tensor);
// dL_dpositions / dL_ddirections are optional.
__global__ void __launch_bounds__(256) k_sh_to_rgb_bwd(ShRgbArgs a, const float* __restrict__ dL_dcolors,
                                                       float* __restrict__ dL_dsh, float* __restrict__ dL_dpositions,
                                                       float* __restrict__ dL_ddirections)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.P) return;
    float x, y, z, vx, vy, vz, len;
    sh_rgb_dir(a, idx, x, y, z, vx, vy, vz, len);
    float sh[48], dsh[48];
    load_sh_n(a.sh, idx, a.M, a.n, sh);
#pragma unroll
    for (int k = 0; k < 48; k++) dsh[k] = 0.0f;
    const size_t i3 = 3 * (size_t)idx;
    const float dcol[3] = {dL_dcolors[i3], dL_dcolors[i3 + 1], dL_dcolors[i3 + 2]};
    // torch.clamp_min passes the gradient where the input is >= the bound
    const uint32_t clamped = (sh_channel(a.D, sh, 0, x, y, z) < 0.f ? 1u : 0u) | (sh_channel(a.D, sh, 1, x, y, z) < 0.f ? 2u : 0u) |
                             (sh_channel(a.D, sh, 2, x, y, z) < 0.f ? 4u : 0u);
    float dL_ddir[3] = {0, 0, 0};
    sh_backward(a.D, sh, dcol, clamped, x, y, z, dsh, dL_ddir);
    if (dL_dsh) {
        float* dst = dL_dsh + (size_t)idx * a.M * 3;
        const int n_sh = 3 * a.M;
        if (n_sh == 48 && ((uintptr_t)dst & 15) == 0) {
            float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
            for (int i = 0; i < 12; i++) { float4 o = {dsh[4 * i], dsh[4 * i + 1], dsh[4 * i + 2], dsh[4 * i + 3]}; d4[i] = o; }
        } else {
#pragma unroll
            for (int i = 0; i < 48; i++) if (i < n_sh) dst[i] = dsh[i];
        }
    }
    if (a.positions) {
        if (dL_dpositions) {
            // d normalize(v) / dv = (I - n n^T) / |v|   (|v| above the eps of F.normalize)
            const float dot = x * dL_ddir[0] + y * dL_ddir[1] + z * dL_ddir[2];
            const float inv = 1.0f / len;
            dL_dpositions[i3] = (dL_ddir[0] - x * dot) * inv;
            dL_dpositions[i3 + 1] = (dL_ddir[1] - y * dot) * inv;
            dL_dpositions[i3 + 2] = (dL_ddir[2] - z * dot) * inv;
        }
    } else if (dL_ddirections) {
        dL_ddirections[i3] = dL_ddir[0]; dL_ddirections[i3 + 1] = dL_ddir[1]; dL_ddirections[i3 + 2] = dL_ddir[2];
    }
}

}  // namespace

void sgr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}

void sgr_launch_depth_keys(const PreprocessArgs& a, hipStream_t s)
{
    if (a.P <= 0) return;
    hipLaunchKernelGGL(k_depth_keys, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
}

void sgr_launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s)
{
    if (a.P <= 0) return;
    const dim3 grid((a.P + 255) / 256);
    if (a.colors_precomp) hipLaunchKernelGGL(k_preprocess_fwd<2>, grid, dim3(256), 0, s, a);
    else if (sh_fast_layout(a.shs, a.M)) hipLaunchKernelGGL(k_preprocess_fwd<0>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_preprocess_fwd<1>, grid, dim3(256), 0, s, a);
}

void sgr_launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s)
{
    if (a.P <= 0) return;
    const dim3 grid((a.P + 255) / 256);
    const int mode = (!a.shs || a.sh_dir_elsewhere) ? 2 : (sh_fast_layout(a.shs, a.M) ? 0 : 1);
    if (a.dL_dsh) {
        if (mode == 0) hipLaunchKernelGGL((k_preprocess_bwd<true, 0>), grid, dim3(256), 0, s, a);
        else if (mode == 1) hipLaunchKernelGGL((k_preprocess_bwd<true, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_preprocess_bwd<true, 2>), grid, dim3(256), 0, s, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((k_preprocess_bwd<false, 0>), grid, dim3(256), 0, s, a);
        else if (mode == 1) hipLaunchKernelGGL((k_preprocess_bwd<false, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_preprocess_bwd<false, 2>), grid, dim3(256), 0, s, a);
    }
}

void sgr_launch_masked_colors(int P, const GeomRec* rec, const float* acc, float* out, const float* cam_pos, float* campos_row,
                              hipStream_t s)
{
    hipLaunchKernelGGL(k_masked_colors, dim3((P + 255) / 256), dim3(256), 0, s, P, rec, acc, out, cam_pos, campos_row);
}

void sgr_launch_sh_adam_from_views(int P, int V, int D, int M, size_t vstride, const float* means3D, const float* campos,
                                   const float* dcolor, float* sh, float* exp_avg, float* exp_avg_sq, float lr_dc, float lr_rest, float b1, float b2,
                                   float eps, float bc1, float bc2_sqrt, float grad_scale, float* dmean_extra, hipStream_t s,
                                   const uint32_t* guard, uint32_t guard_cap)
{
    ShAdamArgs a = {lr_dc, lr_rest, b1, b2, sgr_one_minus(b1), sgr_one_minus(b2), eps, bc1, bc2_sqrt, grad_scale, guard, guard_cap};
    if (dmean_extra)
        hipLaunchKernelGGL(k_sh_adam_from_views<true>, dim3((P + 63) / 64), dim3(64), 0, s, P, V, D, M, vstride, means3D, campos, dcolor,
                           sh, exp_avg, exp_avg_sq, a, dmean_extra);
    else
        hipLaunchKernelGGL(k_sh_adam_from_views<false>, dim3((P + 63) / 64), dim3(64), 0, s, P, V, D, M, vstride, means3D, campos, dcolor,
                           sh, exp_avg, exp_avg_sq, a, dmean_extra);
}

void sgr_launch_sh_grad_from_views(int P, int V, int D, int M, size_t vstride, const float* means3D, const float* campos,
                                   const float* dcolor, float* dL_dsh, hipStream_t s)
{
    hipLaunchKernelGGL(k_sh_grad_from_views, dim3((P + 255) / 256), dim3(256), 0, s, P, V, D, M, vstride, means3D, campos, dcolor,
                       dL_dsh);
}

static int sh_rgb_args(ShRgbArgs& a, int P, int D, int M, const float* sh, const float* positions, const float* centers,
                       int n_centers, const float* directions)
{
    if (D < 0 || D > 3 || M < (D + 1) * (D + 1) || M > 16 || !sh) return SGR_E_INVALID;
    if (positions ? (!centers || (n_centers != 1 && n_centers != P)) : !directions) return SGR_E_INVALID;
    a.P = P; a.D = D; a.M = M; a.n = (D + 1) * (D + 1); a.sh = sh; a.positions = positions; a.centers = centers;
    a.n_centers = n_centers; a.directions = directions;
    return 0;
}

extern "C" int sgr_sh_to_rgb_forward(int P, int D, int M, const float* sh, const float* positions, const float* camera_centers,
                                     int n_centers, const float* directions, float* colors, void* stream)
{
    if (P <= 0) return 0;
    ShRgbArgs a;
    if (!colors || sh_rgb_args(a, P, D, M, sh, positions, camera_centers, n_centers, directions) < 0) return SGR_E_INVALID;
    hipLaunchKernelGGL(k_sh_to_rgb_fwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, colors);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

extern "C" int sgr_sh_to_rgb_backward(int P, int D, int M, const float* sh, const float* positions, const float* camera_centers,
                                      int n_centers, const float* directions, const float* dL_dcolors, float* dL_dsh,
                                      float* dL_dpositions, float* dL_ddirections, void* stream)
{
    if (P <= 0) return 0;
    ShRgbArgs a;
    if (!dL_dcolors || sh_rgb_args(a, P, D, M, sh, positions, camera_centers, n_centers, directions) < 0) return SGR_E_INVALID;
    hipLaunchKernelGGL(k_sh_to_rgb_bwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, dL_dcolors, dL_dsh,
                       dL_dpositions, dL_ddirections);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}
