// launch_rate.hip -- how long does the chip take to START n small workgroups?  (developer yardstick for the one-wave-per-block
// blend kernels: 32640 workgroups of 64 threads per launch)   hipcc -O2 --offload-arch=gfx950 launch_rate.hip -o launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES>
__global__ void __launch_bounds__(256) k_empty(int* out, int spin)
{
    __shared__ int s[LDS_BYTES / 4];
    s[threadIdx.x] = threadIdx.x;
    int v = 0;
    for (int i = 0; i < spin; i++) v += __builtin_amdgcn_s_memtime() & 1;  // ~spin x 40+ cycles of scalar waiting
    if (v == 123456789) out[0] = s[(threadIdx.x + 1) & 63];
}
template <int LDS_BYTES>
static void run(const char* name, int blocks, int threads, int spin)
{
    int* out; hipMalloc(&out, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(blocks), dim3(threads), 0, 0, out, spin);
    hipEventRecord(a);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(blocks), dim3(threads), 0, 0, out, spin);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks=%6d threads=%3d spin=%4d : %7.1f us per launch\n", name, blocks, threads, spin, 1e3 * ms / 20);
    hipFree(out);
}
int main()
{
    run<1024>("1 KB LDS", 32640, 64, 0);
    run<3328>("3.3 KB LDS", 32640, 64, 0);
    run<12288>("12 KB LDS", 32640, 64, 0);
    run<3328>("3.3 KB LDS", 8160, 256, 0);
    run<3328>("3.3 KB LDS", 32640, 64, 100);
    run<3328>("3.3 KB LDS", 32640, 64, 400);
    run<12288>("12 KB LDS", 32640, 64, 400);
    run<3328>("3.3 KB LDS", 8160, 256, 400);
    return 0;
}
