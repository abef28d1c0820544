// Shim so that the reference's UNMODIFIED .cu sources (read in place from /root/reference) compile with hipcc for
// gfx950.  TEST INFRASTRUCTURE: builds oracle/_ref/libref_rasterizer.so, the reference itself running on the MI355X,
// used only as a parity checker and A/B timing baseline.  Nothing here is product code.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetErrorString hipGetErrorString
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemset hipMemset
#define cudaSuccess hipSuccess
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define __trap() abort()
