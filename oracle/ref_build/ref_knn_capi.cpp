// ref_knn_capi.cpp -- TEST INFRASTRUCTURE.  C entry point around the REFERENCE's own SimpleKNN::knn
// (gaussian_splatting/submodules/simple-knn/simple_knn.cu:185-221, what `simple_knn._C.distCUDA2` calls through
// spatial.cu:15-26), whose unmodified source is compiled by hipcc for gfx950 (build_ref.sh; cub -> hipCUB, thrust -> rocThrust).
// The reference runs on the null stream; callers synchronise around it.
#include <hip/hip_runtime.h>
#include "simple_knn.h"

extern "C" void ref_dist2(int P, float* points, float* mean_dists)
{
    SimpleKNN::knn(P, reinterpret_cast<float3*>(points), mean_dists);
}
