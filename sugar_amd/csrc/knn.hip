// knn.hip -- exact k-nearest-neighbour kernels for gfx950 (v1: LDS-tiled exhaustive search).
//
//   sgr_dist2 : distCUDA2 of simple-knn (simple-knn/simple_knn.cu:147-221): mean of the three smallest squared
//               distances to the OTHER points.  The reference gets there with a Morton sort + box pruning; the
//               value it defines is exact, so any exact search reproduces it.
//   sgr_knn   : pytorch3d.ops.knn_points semantics as used by SuGaR (sugar_scene/sugar_model.py:49,235,1028,1342).
//
// One lane per query point; the reference set streams through LDS in 1024-point tiles (12 KB), every lane reads
// the same address (broadcast, conflict-free).  The running K-best list lives in registers, sorted ascending;
// a candidate is rejected with a single compare against the current worst.  Squared distances are evaluated as
// (dx*dx + dy*dy) + dz*dz with individually rounded ops (-ffp-contract=off), identical to the oracle.
#include "../../include/sugar_raster.h"
#include "sgr_common.h"
#include <string>

namespace {

#define KNN_TILE 1024

template <int K, bool EXCLUDE_SELF>
__global__ void __launch_bounds__(256) k_knn(int N, const float* __restrict__ query, int M, const float* __restrict__ ref,
                                             float* __restrict__ out_d, int64_t* __restrict__ out_i,
                                             float* __restrict__ out_mean)
{
    __shared__ float s_ref[KNN_TILE * 3];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < N;
    float qx = 0, qy = 0, qz = 0;
    if (live) { qx = query[3 * (size_t)q]; qy = query[3 * (size_t)q + 1]; qz = query[3 * (size_t)q + 2]; }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) { bd[k] = 3.402823466e+38f; bi[k] = -1; }
    for (int base = 0; base < M; base += KNN_TILE) {
        const int nt = min(KNN_TILE, M - base);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * 3; i += 256) s_ref[i] = ref[3 * (size_t)base + i];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < nt; j++) {
            const int gi = base + j;
            if (EXCLUDE_SELF && gi == q) continue;
            const float dx = s_ref[3 * j] - qx, dy = s_ref[3 * j + 1] - qy, dz = s_ref[3 * j + 2] - qz;
            float d = dx * dx + dy * dy + dz * dz;
            if (!(d < bd[K - 1])) continue;
            int id = gi;
#pragma unroll
            for (int k = 0; k < K; k++) {  // insertion into the ascending list (strict <: earlier index wins ties)
                if (d < bd[k]) {
                    const float td = bd[k]; const int ti = bi[k];
                    bd[k] = d; bi[k] = id; d = td; id = ti;
                }
            }
        }
    }
    if (!live) return;
    if (out_mean) {
        out_mean[q] = (bd[0] + bd[1] + bd[2]) / 3.0f;  // simple_knn.cu:182
    } else {
#pragma unroll
        for (int k = 0; k < K; k++) { out_d[(size_t)q * K + k] = bd[k]; out_i[(size_t)q * K + k] = (int64_t)bi[k]; }
    }
}

thread_local std::string g_knn_err;

template <int K>
void launch_knn(int N, const float* query, int M, const float* ref, float* d, int64_t* i, hipStream_t s)
{
    hipLaunchKernelGGL((k_knn<K, false>), dim3((N + 255) / 256), dim3(256), 0, s, N, query, M, ref, d, i, (float*)nullptr);
}

}  // namespace

extern "C" {

int sgr_dist2(int P, const float* points, float* meanDists, void* stream)
{
    if (P <= 0) return 0;
    if (!points || !meanDists) return SGR_E_INVALID;
    hipLaunchKernelGGL((k_knn<3, true>), dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, points, P, points,
                       (float*)nullptr, (int64_t*)nullptr, meanDists);
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

int sgr_knn(int N, const float* query, int M, const float* ref, int K, float* dists, int64_t* idx, void* stream)
{
    if (N <= 0) return 0;
    if (!query || !ref || !dists || !idx || M <= 0) return SGR_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 1: launch_knn<1>(N, query, M, ref, dists, idx, s); break;
        case 2: launch_knn<2>(N, query, M, ref, dists, idx, s); break;
        case 3: launch_knn<3>(N, query, M, ref, dists, idx, s); break;
        case 4: launch_knn<4>(N, query, M, ref, dists, idx, s); break;
        case 8: launch_knn<8>(N, query, M, ref, dists, idx, s); break;
        case 16: launch_knn<16>(N, query, M, ref, dists, idx, s); break;
        case 32: launch_knn<32>(N, query, M, ref, dists, idx, s); break;
        default: return SGR_E_INVALID;  // supported K: 1,2,3,4,8,16,32
    }
    return hipGetLastError() == hipSuccess ? 0 : SGR_E_HIP;
}

}  // extern "C"
