// sgr_device.h -- device helpers shared by preprocess.hip and binning.hip.
#pragma once
#include "sgr_common.h"

__device__ __forceinline__ int sgr_f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// getRect, DGR/cuda_rasterizer/auxiliary.h:46-56: float divide, truncation toward zero, clamp to the grid.
// Must stay bit-identical between the counting pass (preprocess) and the scatter pass (binning): both
// translation units are built with -ffp-contract=off.
__device__ __forceinline__ void sgr_get_rect(float px, float py, int max_radius, int gx, int gy,
                                             int& minx, int& miny, int& maxx, int& maxy)
{
    const float r = (float)max_radius;
    minx = min(gx, max(0, sgr_f2i_sat((px - r) / (float)SGR_TILE_X)));
    miny = min(gy, max(0, sgr_f2i_sat((py - r) / (float)SGR_TILE_Y)));
    maxx = min(gx, max(0, sgr_f2i_sat((px + r + (float)SGR_TILE_X - 1.0f) / (float)SGR_TILE_X)));
    maxy = min(gy, max(0, sgr_f2i_sat((py + r + (float)SGR_TILE_Y - 1.0f) / (float)SGR_TILE_Y)));
}
