// sort_ref.hip -- how fast does rocPRIM sort 1M (u32 key, u32 value) pairs on this GPU?  (developer yardstick for binning.hip's
// depth sort; not part of the product)   hipcc -O2 --offload-arch=gfx950 sort_ref.hip -o sort_ref
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1000000;
    std::vector<unsigned> hk(n), hv(n);
    std::mt19937 rng(1);
    for (int i = 0; i < n; i++) { float d = 1.3f + 3.4f * (rng() / 4294967296.0f); memcpy(&hk[i], &d, 4); hv[i] = i; }
    unsigned *k0, *k1, *v0, *v1; void* tmp = nullptr; size_t tb = 0;
    hipMalloc(&k0, 4 * n); hipMalloc(&k1, 4 * n); hipMalloc(&v0, 4 * n); hipMalloc(&v1, 4 * n);
    hipMemcpy(k0, hk.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(v0, hv.data(), 4 * n, hipMemcpyHostToDevice);
    for (int bits : {32, 24}) {
        rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0, bits, 0, false);
        if (!tmp) hipMalloc(&tmp, tb * 2);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int r = 0; r < 3; r++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0, false);
        hipEventRecord(a);
        for (int r = 0; r < 20; r++) rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, bits, 0, false);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("rocprim radix_sort_pairs n=%d bits=%d: %.1f us per sort (temp %zu bytes)\n", n, bits, 1e3 * ms / 20, tb);
    }
    return 0;
}
