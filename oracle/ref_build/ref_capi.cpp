// ref_capi.cpp -- TEST INFRASTRUCTURE.  C entry points around the REFERENCE's own CudaRasterizer::Rasterizer
// (DGR/cuda_rasterizer/rasterizer.h:24-84), whose unmodified sources are compiled by hipcc for gfx950 (see build_ref.sh).
// Same calling shape as include/sugar_raster.h so the parity tests can drive both with one harness.  The reference
// launches on the null stream and synchronises through its blocking hipMemcpy; callers must synchronise around it.
#include <cstddef>
#include <cstdint>
#include <functional>
#include "rasterizer.h"

extern "C" {

typedef char* (*ref_alloc_fn)(void* user, size_t bytes);

int ref_forward(ref_alloc_fn geom_alloc, void* geom_user, ref_alloc_fn binning_alloc, void* binning_user,
                ref_alloc_fn img_alloc, void* img_user, int P, int D, int M, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                int prefiltered, float* out_color, int* radii, int debug)
{
    std::function<char*(size_t)> g = [&](size_t n) { return geom_alloc(geom_user, n); };
    std::function<char*(size_t)> b = [&](size_t n) { return binning_alloc(binning_user, n); };
    std::function<char*(size_t)> i = [&](size_t n) { return img_alloc(img_user, n); };
    return CudaRasterizer::Rasterizer::forward(g, b, i, P, D, M, background, width, height, means3D, shs, colors_precomp,
                                               opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                                               projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered != 0, out_color, radii,
                                               debug != 0);
}

void ref_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                  char* binning_buffer, char* img_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                  float* dL_dscale, float* dL_drot, int debug)
{
    CudaRasterizer::Rasterizer::backward(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales,
                                         scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx,
                                         tan_fovy, radii, geom_buffer, binning_buffer, img_buffer, dL_dpix, dL_dmean2D,
                                         dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                                         debug != 0);
}

void ref_mark_visible(int P, float* means3D, float* viewmatrix, float* projmatrix, bool* present)
{
    CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, present);
}

}  // extern "C"
