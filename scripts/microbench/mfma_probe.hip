// mfma_probe.hip -- gfx950 probe for the matrix-pipe forms the blend kernels use (developer tool, not part of the product).
//   1. operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1) incl. the cbsz / abid broadcast of A,
//      and of v_mfma_f32_16x16x4_f32, against a host computation;
//   2. issue cost: cycles per instruction of the 4x4x1 form alone, of a VALU stream alone and of both interleaved (does the
//      matrix pipe run beside the vector ALUs from ONE wave and from several?), v_mov_b64, v_pk_fma_f32 with op_sel broadcast;
//   3. how soon a VALU instruction may read an MFMA result (hazard window) -- wrong values flag a missing wait state.
// Build: hipcc -O2 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));

// ---- 1. layouts -------------------------------------------------------------------------------------------------------
template <int CBSZ, int ABID>
__global__ void k_layout_4x4(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; i++) d[4 * l + i] = c[i];
}

__global__ void k_layout_16x16(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[4 * l + i] = c[i];
}

// ---- 3. hazard window: MFMA result read by a VALU instruction after `NOPS` independent VALU instructions ----------------
template <int NOPS>
__global__ void k_hazard(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    float av = a[l], bv = b[l];
    float r0, r1, r2, r3;
    float dummy = 1.0f;
    asm volatile(
        "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
        "s_nop 4\n"
        "v_mfma_f32_4x4x1_16b_f32 v[20:23], %[a], %[b], v[20:23]\n"
        "v_mfma_f32_4x4x1_16b_f32 v[20:23], %[a], %[b], v[20:23]\n"
        ".rept %c[nops]\n v_add_f32 %[dm], %[dm], %[dm]\n .endr\n"
        "v_add_f32 %[r0], v20, v20\n v_add_f32 %[r1], v21, v21\n v_add_f32 %[r2], v22, v22\n v_add_f32 %[r3], v23, v23\n"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [dm] "+v"(dummy)
        : [a] "v"(av), [b] "v"(bv), [nops] "i"(NOPS)
        : "v20", "v21", "v22", "v23");
    d[4 * l + 0] = r0; d[4 * l + 1] = r1; d[4 * l + 2] = r2; d[4 * l + 3] = r3;
    if (dummy == 12345.f) d[0] = dummy;
}

// ---- 2. issue rates ---------------------------------------------------------------------------------------------------
enum Mode { MFMA_ONLY, VALU_ONLY, MIX_6_TO_48, MIX_6_TO_24, MOV64, MOV32, PKFMA_OPSEL, EXP_ONLY, MFMA16_ONLY, MIX16_4_TO_32, MODE_COUNT };
static const char* mode_name[] = {"6 x mfma_4x4x1 (one acc chain)", "48 x v_fma_f32", "6 mfma_4x4x1 + 48 v_fma", "6 mfma_4x4x1 + 24 v_fma",
                                  "48 x v_mov_b64", "48 x v_mov_b32", "48 x v_pk_fma_f32 op_sel", "48 x v_exp_f32",
                                  "4 x mfma_16x16x4 (one acc chain)", "4 mfma_16x16x4 + 32 v_fma"};

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(int iters, float a, float b, float* out, long long* cyc)
{
    float x[8];
    for (int c = 0; c < 8; c++) x[c] = a + c + threadIdx.x * 1e-3f;
    float av = a + threadIdx.x, bv = b;
    double m64 = 1.0;
    float2 p = make_float2(a, b), pw = make_float2(0.5f, 0.25f), pc = make_float2(0.f, 0.f);
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == MFMA_ONLY || MODE == MIX_6_TO_48 || MODE == MIX_6_TO_24)
            asm volatile("v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, %1, 0\n"
                         "v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, %1, v[20:23]\n"
                         "v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, %1, v[20:23]\n"
                         "v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, %1, v[20:23]\n"
                         "v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, %1, v[20:23]\n"
                         "v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, %1, v[20:23]\n" ::"v"(av), "v"(bv) : "v20", "v21", "v22", "v23");
        if constexpr (MODE == MFMA16_ONLY || MODE == MIX16_4_TO_32)
            asm volatile("v_mfma_f32_16x16x4_f32 v[20:23], %0, %1, 0\n"
                         "v_mfma_f32_16x16x4_f32 v[20:23], %0, %1, v[20:23]\n"
                         "v_mfma_f32_16x16x4_f32 v[20:23], %0, %1, v[20:23]\n"
                         "v_mfma_f32_16x16x4_f32 v[20:23], %0, %1, v[20:23]\n" ::"v"(av), "v"(bv) : "v20", "v21", "v22", "v23");
        constexpr int NV = (MODE == VALU_ONLY || MODE == MIX_6_TO_48) ? 6 : (MODE == MIX_6_TO_24 ? 3 : (MODE == MIX16_4_TO_32 ? 4 : 0));
#pragma unroll
        for (int u = 0; u < NV; u++)
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        if constexpr (MODE == MOV64)
#pragma unroll
            for (int u = 0; u < 48; u++) asm volatile("v_mov_b64 %0, %1" : "=v"(m64) : "v"(m64 + 0.0));
        if constexpr (MODE == MOV32)
#pragma unroll
            for (int u = 0; u < 48; u++) asm volatile("v_mov_b32 %0, %1" : "=v"(x[u & 7]) : "v"(a));
        if constexpr (MODE == PKFMA_OPSEL)
#pragma unroll
            for (int u = 0; u < 48; u++)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(pc) : "v"(p), "v"(pw));
        if constexpr (MODE == EXP_ONLY)
#pragma unroll
            for (int u = 0; u < 48; u++) asm volatile("v_exp_f32 %0, %0" : "+v"(x[u & 7]));
    }
    const long long t1 = clock64();
    float s = (float)m64 + pc.x + pc.y;
    for (int c = 0; c < 8; c++) s += x[c];
    if constexpr (MODE == MFMA_ONLY || MODE == MIX_6_TO_48 || MODE == MIX_6_TO_24 || MODE == MFMA16_ONLY || MODE == MIX16_4_TO_32) {
        float r;
        asm volatile("s_nop 15\n s_nop 15\n v_add_f32 %0, v20, v21" : "=v"(r)::"v20", "v21");
        s += r;
    }
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void rate(int waves_per_simd)
{
    static float* out = nullptr;
    static long long* cyc = nullptr;
    if (!out) { hipMalloc(&out, 64); hipMalloc(&cyc, 64); }
    const int iters = 4096;
    const int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), 0, 0, 16, 1.0001f, 0.5f, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(grid), dim3(256), 0, 0, iters, 1.0001f, 0.5f, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-36s waves/SIMD=%d  cycles per iteration: wave %8.1f  SIMD %8.1f   wall %.3f ms\n", mode_name[MODE], waves_per_simd,
           (double)c / iters, (double)c / iters / waves_per_simd, ms);
}

template <int MODE>
void sweep() { for (int w : {1, 2, 4, 8}) rate<MODE>(w); }

template <int NOPS>
int hazard(const float* da, const float* db, float* dd, const std::vector<float>& a, const std::vector<float>& b)
{
    hipLaunchKernelGGL((k_hazard<NOPS>), dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<float> d(256);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            const float ref = 2.f * (2.f * a[4 * (l / 4) + i] * b[l]);
            if (fabsf(d[4 * l + i] - ref) > 1e-4f * fabsf(ref)) bad++;
        }
    printf("hazard: VALU read %d VALU instructions after the MFMA: %s (%d wrong of 256)\n", NOPS, bad ? "WRONG" : "ok", bad);
    return bad;
}

int main()
{
    std::vector<float> a(64), b(64), d(256);
    for (int i = 0; i < 64; i++) { a[i] = 1.0f + i; b[i] = 0.5f + 0.01f * i; }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    // hypothesis for 4x4x1_16b: D[block][i][j] = A[block][i] * B[block][j]; A[block][i] from lane 4 block + i, B[block][j] from
    // lane 4 block + j, D[block][i][j] in lane 4 block + j, register i.  cbsz = 4: every block takes block abid's A.
    auto check4 = [&](const char* name, int cbsz, int abid) {
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 4; i++) {
                const int blk = l / 4, ablk = cbsz == 4 ? abid : blk;
                const float ref = a[4 * ablk + i] * b[l];
                if (fabsf(d[4 * l + i] - ref) > 1e-5f * fabsf(ref)) bad++;
            }
        printf("layout 4x4x1_16b %-18s: %s (%d of 256 differ from D[lane 4b+j][reg i] = A[lane 4b'+i] * B[lane 4b+j])\n", name,
               bad ? "MISMATCH" : "as assumed", bad);
        if (bad) for (int l = 0; l < 8; l++) printf("   lane %d: %g %g %g %g\n", l, d[4 * l], d[4 * l + 1], d[4 * l + 2], d[4 * l + 3]);
    };
    hipLaunchKernelGGL((k_layout_4x4<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); check4("cbsz=0", 0, 0);
    hipLaunchKernelGGL((k_layout_4x4<4, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); check4("cbsz=4 abid=0", 4, 0);
    hipLaunchKernelGGL((k_layout_4x4<4, 5>), dim3(1), dim3(64), 0, 0, da, db, dd); check4("cbsz=4 abid=5", 4, 5);
    hipLaunchKernelGGL((k_layout_4x4<4, 15>), dim3(1), dim3(64), 0, 0, da, db, dd); check4("cbsz=4 abid=15", 4, 15);
    // hypothesis for 16x16x4: A[i][k] from lane i + 16 k, B[k][j] from lane j + 16 k, D[i][j] in lane j + 16 (i / 4), register i % 4
    {
        hipLaunchKernelGGL(k_layout_16x16, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int j = l % 16, i = 4 * (l / 16) + r;
                float ref = 0.f;
                for (int k = 0; k < 4; k++) ref = fmaf(a[i + 16 * k], b[j + 16 * k], ref);
                if (fabsf(d[4 * l + r] - ref) > 1e-5f * fabsf(ref)) bad++;
            }
        printf("layout 16x16x4: %s (%d of 256 differ from D[lane j+16(i/4)][reg i%%4] = sum_k A[lane i+16k] B[lane j+16k])\n",
               bad ? "MISMATCH" : "as assumed", bad);
    }
    hazard<0>(da, db, dd, a, b); hazard<1>(da, db, dd, a, b); hazard<2>(da, db, dd, a, b); hazard<3>(da, db, dd, a, b);
    hazard<4>(da, db, dd, a, b); hazard<6>(da, db, dd, a, b); hazard<8>(da, db, dd, a, b); hazard<12>(da, db, dd, a, b);
    sweep<MFMA_ONLY>(); sweep<VALU_ONLY>(); sweep<MIX_6_TO_48>(); sweep<MIX_6_TO_24>(); sweep<MOV64>(); sweep<MOV32>();
    sweep<PKFMA_OPSEL>(); sweep<EXP_ONLY>(); sweep<MFMA16_ONLY>(); sweep<MIX16_4_TO_32>();
    return 0;
}
