// loss_bench.hip -- times the fused L1 + D-SSIM kernels of csrc/loss.hip in isolation (developer tool for tile-shape
// experiments; the translation unit is included, so -D overrides of its tile macros apply).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DLTY=22 ...] loss_bench.hip -o loss_bench_x
#include "../../sugar_amd/csrc/loss.hip"
#include <cstdio>
#include <vector>
#include <random>
int main(int argc, char** argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080, C = 3;
    const size_t n = (size_t)C * W * H;
    std::vector<float> hx(n), hy(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; i++) { hx[i] = rng() / 4294967296.0f; hy[i] = rng() / 4294967296.0f; }
    float *x, *y, *g, *loss, *gl; char* scratch;
    hipMalloc(&x, 4 * n); hipMalloc(&y, 4 * n); hipMalloc(&g, 4 * n); hipMalloc(&loss, 16); hipMalloc(&gl, 4);
    hipMalloc(&scratch, sgr_l1_ssim_scratch_bytes(C, W, H));
    hipMemcpy(x, hx.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(y, hy.data(), 4 * n, hipMemcpyHostToDevice);
    const float one = 1.f; hipMemcpy(gl, &one, 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms_f = 0, ms_b = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        for (int r = 0; r < 50; r++) sgr_l1_ssim_forward(C, W, H, x, y, 0.2f, scratch, loss, nullptr);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms_f, a, b);
        hipEventRecord(a);
        for (int r = 0; r < 50; r++) sgr_l1_ssim_backward(C, W, H, x, y, 0.2f, scratch, gl, g, nullptr);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms_b, a, b);
    }
    float hl[3]; hipMemcpy(hl, loss, 12, hipMemcpyDeviceToHost);
    std::vector<float> hg(n); hipMemcpy(hg.data(), g, 4 * n, hipMemcpyDeviceToHost);
    double cs = 0; for (size_t i = 0; i < n; i += 97) cs += (double)hg[i] * (double)((i % 13) + 1);
    printf("fwd(+finish) %.1f us  bwd %.1f us   loss %.7f l1 %.7f ssim %.7f  grad checksum %.9e\n", 1e3 * ms_f / 50, 1e3 * ms_b / 50,
           hl[0], hl[1], hl[2], cs);
    return 0;
}
