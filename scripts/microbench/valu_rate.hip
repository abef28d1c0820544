// valu_rate.hip -- gfx950 VALU issue-rate microbenchmark (developer tool, not part of the product).
// Question it answers: how many shader cycles does one wave64 VALU instruction occupy its SIMD for, as a function of the
// instruction, the number of independent dependency chains per wave, and the number of resident waves per SIMD?
// Build: hipcc -O2 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

enum Op { FMA, MUL, PKFMA, PKMUL, EXP, RCP, CNDMASK, CMP, MAXF, MOV, FMA_EXP_MIX, LDS_B128, LDS_B32, OP_COUNT };
static const char* op_name[] = {"v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "v_cndmask_b32",
                                "v_cmp_gt_f32", "v_max_f32", "v_mov_b32", "7fma+1exp", "ds_read_b128(bcast)", "ds_read_b32(bcast)"};

template <int OP, int NCH>
__global__ void __launch_bounds__(256) k(int iters, float a, float b, float* out, long long* cyc)
{
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(a, b, a, b);
    __syncthreads();
    float x[NCH];
    float2 p[NCH];
    float4 q[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { x[c] = a + c + threadIdx.x * 1e-3f; p[c] = make_float2(x[c], x[c] + 1.f); q[c] = make_float4(0, 0, 0, 0); }
    const float2 pa = make_float2(a, a), pb = make_float2(b, b);
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if constexpr (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pa), "v"(pb));
                if constexpr (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pa));
                if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[c]));
                if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
                if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a));
                if constexpr (OP == CMP) asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(x[c]), "v"(a) : "vcc");
                if constexpr (OP == MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[c]) : "v"(a));
                if constexpr (OP == FMA_EXP_MIX) {
                    if (u == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(x[c]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                }
                if constexpr (OP == LDS_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(q[c]) : "v"(u * 16 + c * 128));
                if constexpr (OP == LDS_B32) asm volatile("ds_read_b32 %0, %1" : "=v"(x[c]) : "v"(u * 16 + c * 128));
            }
        }
        if constexpr (OP == LDS_B128 || OP == LDS_B32) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += x[c] + p[c].x + p[c].y + q[c].x + q[c].w;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP, int NCH>
void run(int waves_per_simd)
{
    static float* out = nullptr;
    static long long* cyc = nullptr;
    if (!out) { hipMalloc(&out, 64); hipMalloc(&cyc, 64); }
    const int iters = 2048;
    const int grid = 256 * waves_per_simd;  // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, NCH>), dim3(grid), dim3(256), 0, 0, 16, 1.0001f, 0.5f, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, NCH>), dim3(grid), dim3(256), 0, 0, iters, 1.0001f, 0.5f, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double inst_per_wave = (double)iters * 8 * NCH;
    // clock64 on gfx950 = s_memtime (shader clock domain per the guide); per-wave cycles per instruction, and per SIMD
    const double cyc_per_inst_wave = (double)c / inst_per_wave;
    const double cyc_per_inst_simd = cyc_per_inst_wave / waves_per_simd;
    const double eff_ghz = (double)c / (ms * 1e6);
    printf("%-22s chains=%d waves/SIMD=%d  cyc/inst(wave)=%7.2f  cyc/inst(SIMD)=%6.2f  wall=%.3f ms  clk~%.2f GHz\n", op_name[OP], NCH,
           waves_per_simd, cyc_per_inst_wave, cyc_per_inst_simd, ms, eff_ghz);
}

template <int OP>
void sweep()
{
    for (int w : {1, 2, 4, 8}) {
        run<OP, 1>(w);
        run<OP, 2>(w);
        run<OP, 4>(w);
        run<OP, 8>(w);
    }
}

int main()
{
    sweep<FMA>(); sweep<MUL>(); sweep<PKFMA>(); sweep<PKMUL>(); sweep<EXP>(); sweep<RCP>(); sweep<CNDMASK>(); sweep<CMP>(); sweep<MAXF>();
    sweep<MOV>(); sweep<FMA_EXP_MIX>(); sweep<LDS_B128>(); sweep<LDS_B32>();
    return 0;
}
