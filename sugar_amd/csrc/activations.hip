// activations.hip -- the parameter activations of the 3DGS train step as one gfx950 kernel each way.
//
// The reference keeps raw parameters and activates them on every access (gaussian_splatting/scene/gaussian_model.py:92-117:
// get_scaling = exp(_scaling), get_rotation = F.normalize(_rotation), get_opacity = sigmoid(_opacity)); with autograd that
// is ~20 elementwise / reduction launches per step over [P,3], [P,4] and [P,1] tensors (0.2 ms at 1M Gaussians, more than
// the forward preprocess kernel).  Here: one lane per Gaussian, 32 B in / 32 B out forward, and a backward that maps the
// rasterizer's gradients w.r.t. the activated values straight onto the raw parameters.
#include "../../include/sugar_raster.h"
#include "sgr_common.h"

namespace {

__global__ void __launch_bounds__(256) k_activations_fwd(int P, const float* __restrict__ scaling_raw,
                                                         const float* __restrict__ rotation_raw,
                                                         const float* __restrict__ opacity_raw, float* __restrict__ scales,
                                                         float* __restrict__ rotations, float* __restrict__ opacities)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i;
    scales[i3] = expf(scaling_raw[i3]); scales[i3 + 1] = expf(scaling_raw[i3 + 1]); scales[i3 + 2] = expf(scaling_raw[i3 + 2]);
    const float4 q = reinterpret_cast<const float4*>(rotation_raw)[i];
    // F.normalize(v, dim=-1): v / max(|v|, 1e-12)
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    reinterpret_cast<float4*>(rotations)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    opacities[i] = 1.0f / (1.0f + expf(-opacity_raw[i]));
}

__global__ void __launch_bounds__(256) k_activations_bwd(int P, const float* __restrict__ scaling_raw,
                                                         const float* __restrict__ rotation_raw,
                                                         const float* __restrict__ opacity_raw,
                                                         const float* __restrict__ dL_dscales, const float* __restrict__ dL_drot,
                                                         const float* __restrict__ dL_dopac, float* __restrict__ d_scaling_raw,
                                                         float* __restrict__ d_rotation_raw, float* __restrict__ d_opacity_raw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 3; k++) d_scaling_raw[i3 + k] = dL_dscales[i3 + k] * expf(scaling_raw[i3 + k]);  // d exp = exp
    const float4 q = reinterpret_cast<const float4*>(rotation_raw)[i];
    const float4 g = reinterpret_cast<const float4*>(dL_drot)[i];
    const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float4 d;
    if (n > 1e-12f) {
        // d (v/|v|) = (I - n n^T) / |v|
        const float inv = 1.0f / n;
        const float nx = q.x * inv, ny = q.y * inv, nz = q.z * inv, nw = q.w * inv;
        const float dot = nx * g.x + ny * g.y + nz * g.z + nw * g.w;
        d = make_float4((g.x - nx * dot) * inv, (g.y - ny * dot) * inv, (g.z - nz * dot) * inv, (g.w - nw * dot) * inv);
    } else {
        d = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);  // clamped denominator: v / 1e-12
    }
    reinterpret_cast<float4*>(d_rotation_raw)[i] = d;
    const float s = 1.0f / (1.0f + expf(-opacity_raw[i]));
    d_opacity_raw[i] = dL_dopac[i] * s * (1.0f - s);
}

}  // namespace

extern "C" int sgr_activations_forward(int P, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                       float* scales, float* rotations, float* opacities, void* stream)
{
    if (P <= 0) return 0;
    if (!scaling_raw || !rotation_raw || !opacity_raw || !scales || !rotations || !opacities) return SGR_E_INVALID;
    if (((uintptr_t)rotation_raw | (uintptr_t)rotations) & 15) return SGR_E_INVALID;
    hipLaunchKernelGGL(k_activations_fwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scaling_raw, rotation_raw,
                       opacity_raw, scales, rotations, opacities);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

extern "C" int sgr_activations_backward(int P, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                        const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                                        float* dL_dscaling_raw, float* dL_drotation_raw, float* dL_dopacity_raw, void* stream)
{
    if (P <= 0) return 0;
    if (!scaling_raw || !rotation_raw || !opacity_raw || !dL_dscales || !dL_drotations || !dL_dopacities || !dL_dscaling_raw ||
        !dL_drotation_raw || !dL_dopacity_raw)
        return SGR_E_INVALID;
    if (((uintptr_t)rotation_raw | (uintptr_t)dL_drotations | (uintptr_t)dL_drotation_raw) & 15) return SGR_E_INVALID;
    hipLaunchKernelGGL(k_activations_bwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scaling_raw, rotation_raw,
                       opacity_raw, dL_dscales, dL_drotations, dL_dopacities, dL_dscaling_raw, dL_drotation_raw,
                       dL_dopacity_raw);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}
