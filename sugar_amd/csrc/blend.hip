// blend.hip -- front-to-back alpha compositing (forward) and its backward for gfx950.
//
// Replaces renderCUDA forward  DGR/cuda_rasterizer/forward.cu:261-374
//      and renderCUDA backward DGR/cuda_rasterizer/backward.cu:399-557.
//
// The reference runs one 256-thread block per 16 x 16 tile in lock step.  Here every 8 x 8 pixel BLOCK of a tile is its own
// 64-thread workgroup (one wave64, lane l = pixel (l & 7, l >> 3)): no workgroup barriers anywhere, a block stops when ITS
// 64 pixels are done, and 4 x more (smaller) work items balance better over the 256 CUs.  The four blocks of a tile are
// placed on the same XCD (workgroup b runs on XCD b % 8), so the tile's Gaussian records are fetched from HBM once.
//
// Forward (k_blend_fwd_w): the wave gathers 64 entries of the tile's depth-sorted list (one 48-byte GeomRec per lane),
// culls them against its block with an EXACT ellipse-vs-rectangle test, compacts the survivors into LDS and walks them with
// a hand-scheduled loop (the reference's skip tests as EXEC masks).  The survivors' (id, list position) pairs are also
// appended to the block's own list in global memory: the backward starts from those and neither culls nor compacts again.
//
// Backward (k_blend_bwd_w): per-pair terms are transposed through a wave-private LDS panel and summed in registers
// (phase A lane = pixel, phase B lane = (Gaussian, two pixel rows)), so a (block, Gaussian) pair costs ONE coalesced
// 36-byte atomic request into a 64-byte accumulator record instead of the reference's up to 64 x 9 float atomics
// (backward.cu:523-554).
#include "sgr_common.h"
#include "tile_order.h"

namespace {

#define LOG2E 1.4426950408889634f

// Sums over the four 16-lane rows of the wave, TWO values per gfx950 lane swap.  v_permlane16_swap exchanges the odd rows of its
// first operand with the even rows of its second, so after one swap and one add rows 0 and 2 hold a's sums over the row pairs
// (0,1) and (2,3) and rows 1 and 3 hold b's; v_permlane32_swap exchanges the upper half of the first operand with the lower half
// of the second, so the same step on two such results leaves the total of a in row 0, of b in row 1, and those of the second
// pair in rows 2 and 3.  Four totals for three swaps and three adds (one value at a time: four instructions and two copies each).
__device__ __forceinline__ float pairsum16(float a, float b)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pairsum32(float c, float d)
{
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// the value of the lane 8 lanes away inside the same 16-lane row (DPP row rotate: folds into the add)
__device__ __forceinline__ float pair_in_row(float x)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x128, 0xF, 0xF, false));
}

// ---------------------------------------------------------------------------------------------
// Forward blend.
//
// Which entries of the tile's list can touch this block?  A pair contributes only if power <= 0 and
// opacity * exp(power) >= 1/255 (forward.cu:336-345), i.e. the pixel lies inside the ellipse
//     q(d) = cx dx^2 + 2 cy dx dy + cz dy^2 <= tau^2 = 2 ln(255 opacity)        (d = pixel - centre).
// The test is exact for the block's rectangle: q is convex, so unless the centre lies in the rectangle its minimum over
// the rectangle is attained on one of the four edges, where q is a 1-D parabola,
//     q(dx, .) = (det/cz) dx^2 + cz (dy - dy*)^2,  dy* = -cy dx / cz      (and symmetrically for a horizontal edge),
// minimised by clamping dy* to the edge.  tau^2 is inflated by 0.2 % + 1e-3 against rounding (at the threshold that is a
// 1 % margin in alpha, the float evaluation of the pair differs from the exact one by ~1e-6).  Skipping an entry for a
// block is therefore exact: no pixel of the block would have passed the reference's tests.  Elongated, diagonal splats
// miss most of the blocks of their bounding box (~70 % of the (entry, block) pairs go).  Degenerate conics: "hit".
__device__ __forceinline__ bool block_hit(float gxc, float gyc, float cx, float cy, float cz, float op, float bx0, float by0)
{
    if (op < 1.0f / 255.0f) return false;
    const float det = cx * cz - cy * cy;
    if (!(det > 0.f) || !(cx > 0.f) || !(cz > 0.f)) return true;
    const float tau2 = 2.0f * __logf(255.0f * op) * 1.002f + 1e-3f;
    const float icx = __builtin_amdgcn_rcpf(cx), icz = __builtin_amdgcn_rcpf(cz);
    const float det_cz = det * icz, det_cx = det * icx;
    const float kyx = -cy * icz, kxy = -cy * icx;
    const float dxl = bx0 - gxc, dxr = dxl + 7.0f;
    const float dyl = by0 - gyc, dyh = dyl + 7.0f;
    const float sL = kyx * dxl, sR = kyx * dxr;
    const float tL = fminf(fmaxf(sL, dyl), dyh) - sL, tR = fminf(fmaxf(sR, dyl), dyh) - sR;
    float q = fminf(det_cz * dxl * dxl + cz * tL * tL, det_cz * dxr * dxr + cz * tR * tR);
    const float sT = kxy * dyl, sB = kxy * dyh;
    const float tT = fminf(fmaxf(sT, dxl), dxr) - sT, tB = fminf(fmaxf(sB, dxl), dxr) - sB;
    q = fminf(q, fminf(det_cx * dyl * dyl + cx * tT * tT, det_cx * dyh * dyh + cx * tB * tB));
    const bool inside = dxl <= 0.f && dxr >= 0.f && dyl <= 0.f && dyh >= 0.f;
    return inside || q <= tau2;
}

// The walk over a compacted batch, hand-scheduled (the compiler's version of this loop carries ~30 VALU and ~23 SALU
// instructions per entry; this one 21 and 9).  Lanes whose pixel is finished are simply removed from EXEC for the whole walk,
// the reference's three skip tests (forward.cu:336-351) narrow EXEC further inside an iteration, and the state updates are
// plain moves under that mask -- no v_cndmask, no per-lane bookkeeping.  The 10 dwords of entry k+1 are fetched from LDS
// (uniform address: broadcast) while entry k is evaluated (two register sets, A = v[40:49], B = v[50:59]).
//   in : exec-independent; live = lanes still accumulating, n = entries to walk (> 0), addr = LDS byte address of entry 0
//   out: live updated, n = entries not yet started when every lane had finished (0: the batch was walked to its end)
// gfx950 hazards observed by hand (the compiler does not look inside): one independent instruction between v_exp_f32 and the
// use of its result; EXEC is only ever written by SALU instructions before VALU instructions depend on it.
#define SGR_FWD_BODY(X, Y, A, B, CZ, OP, R, G, BL, POS, XY, AB)                                     \
    "v_sub_f32 v60, " X ", %[px]\n"                                                                 \
    "v_sub_f32 v61, " Y ", %[py]\n"                                                                 \
    "v_mul_f32 v62, " B ", v61\n"                                                                   \
    "v_fmac_f32 v62, " A ", v60\n"                                                                  \
    "v_mul_f32 v63, " CZ ", v61\n"                                                                  \
    "v_mul_f32 v63, v63, v61\n"                                                                     \
    "v_fmac_f32 v63, v60, v62\n"        /* log2(e) * power */                                       \
    "v_exp_f32 v62, v63\n"                                                                          \
    "s_mov_b64 %[live], exec\n"                                                                     \
    "v_cmp_nlt_f32 vcc, 0, v63\n"       /* !(power > 0) */                                          \
    "v_mul_f32 v62, " OP ", v62\n"                                                                  \
    "v_min_f32 v62, 0x3f7d70a4, v62\n"  /* alpha = min(0.99, opacity * G) */                        \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v62\n" /* !(alpha < 1/255) */                                   \
    "v_mul_f32 v61, v62, %[T]\n"        /* alpha * T */                                             \
    "v_sub_f32 v60, %[T], v61\n"        /* test_T = T - alpha T  (forward.cu:347: T (1 - alpha), one rounding apart) */ \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_gt_f32 vcc, 0x38d1b717, v60\n" /* test_T < 0.0001: this lane is finished */              \
    "s_andn2_b64 %[live], %[live], vcc\n"                                                           \
    "s_andn2_b64 exec, exec, vcc\n"                                                                 \
    "v_mov_b32 %[T], v60\n"                                                                         \
    "v_fmac_f32 %[C0], " R ", v61\n"                                                                \
    "v_fmac_f32 %[C1], " G ", v61\n"                                                                \
    "v_fmac_f32 %[C2], " BL ", v61\n"                                                               \
    "v_mov_b32 %[last], " POS "\n"                                                                  \
    "s_mov_b64 exec, %[live]\n"

// EXACT-ALPHA variant (round 6; sgr_set_exact_alpha / SGR_FLAG_EXACT_ALPHA).  tests/test_gpu_fullsize.py's gradient differences from
// the reference (5e-5 .. 8e-5 norm-wise on rotations, against a bar of 1e-4) were traced switch by switch
// (profiles/r06_grad_switch_table.txt) to ONE place: alpha.  Both sides evaluate it to ~1 ulp, but with different roundings -- the
// pre-scaled conic and the folded log2(e) above against `power` rounded operation by operation and the two-term expf of the
// device library -- and the sums over a splat's pixels cancel so heavily that 1e-7 relative in alpha is 5e-5 in dL/drotation (the
// reference's own float-atomic order alone is 2e-6 there).  This body follows forward.cu:333-347 operation for operation:
//   power = -0.5 (cx dx dx + cz dy dy) - cy dx dy    individually rounded, in that order (the entry carries -0.5 cx and -0.5 cz: exact scalings)
//   G     = expf(power)                              the device library's algorithm: ph = x c, e = rint(ph), a = (ph - e) + (fma(x, c, -ph)
//                                                    + x c_lo), G = ldexp(exp2(a), e); its range checks are dead here (power <= 0 survives,
//                                                    an underflowing G fails the 1/255 test either way)
//   test_T = T (1 - alpha)
// With it alpha, T, final_T and n_contrib are bit-identical to the reference's (compiled without contraction) and every gradient
// tensor is within the reference's own run-to-run spread; 12 more VALU instructions per entry (33 against 21).
#define SGR_FWD_BODY_X(X, Y, A, B, CZ, OP, R, G, BL, POS, XY, AB)                                   \
    "v_sub_f32 v60, " X ", %[px]\n"                                                                 \
    "v_sub_f32 v61, " Y ", %[py]\n"                                                                 \
    "v_mul_f32 v62, " A ", v60\n"                                                                   \
    "v_mul_f32 v63, " CZ ", v61\n"                                                                  \
    "v_mul_f32 v62, v62, v60\n"                                                                     \
    "v_mul_f32 v63, v63, v61\n"                                                                     \
    "v_mul_f32 v60, " B ", v60\n"                                                                   \
    "v_add_f32 v62, v62, v63\n"                                                                     \
    "v_mul_f32 v60, v60, v61\n"                                                                     \
    "v_sub_f32 v63, v62, v60\n"         /* power */                                                 \
    "v_mul_f32 v60, 0x3fb8aa3b, v63\n"  /* ph = power * log2(e) */                                  \
    "v_rndne_f32 v61, v60\n"            /* e */                                                     \
    "v_fma_f32 v62, v63, %[chi], -v60\n" /* the product's rounding error */                         \
    "v_fmac_f32 v62, 0x32a5705f, v63\n" /* + power * (log2(e) - float(log2(e))) */                  \
    "v_sub_f32 v60, v60, v61\n"                                                                     \
    "v_add_f32 v60, v60, v62\n"         /* a */                                                     \
    "v_exp_f32 v60, v60\n"                                                                          \
    "v_cvt_i32_f32 v61, v61\n"                                                                      \
    "s_mov_b64 %[live], exec\n"                                                                     \
    "v_cmp_nlt_f32 vcc, 0, v63\n"       /* !(power > 0) */                                          \
    "v_ldexp_f32 v62, v60, v61\n"       /* G = expf(power) */                                       \
    "v_mul_f32 v62, " OP ", v62\n"                                                                  \
    "v_min_f32 v62, 0x3f7d70a4, v62\n"  /* alpha = min(0.99, opacity * G) */                        \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v62\n" /* !(alpha < 1/255) */                                   \
    "v_sub_f32 v60, 1.0, v62\n"         /* 1 - alpha */                                             \
    "v_mul_f32 v61, v62, %[T]\n"        /* alpha * T */                                             \
    "v_mul_f32 v60, %[T], v60\n"        /* test_T = T (1 - alpha), forward.cu:347 */                \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_gt_f32 vcc, 0x38d1b717, v60\n" /* test_T < 0.0001: this lane is finished */              \
    "s_andn2_b64 %[live], %[live], vcc\n"                                                           \
    "s_andn2_b64 exec, exec, vcc\n"                                                                 \
    "v_mov_b32 %[T], v60\n"                                                                         \
    "v_fmac_f32 %[C0], " R ", v61\n"                                                                \
    "v_fmac_f32 %[C1], " G ", v61\n"                                                                \
    "v_fmac_f32 %[C2], " BL ", v61\n"                                                               \
    "v_mov_b32 %[last], " POS "\n"                                                                  \
    "s_mov_b64 exec, %[live]\n"

// A/B build option -DSGR_FWD_PK (round-5 verdict item 5, "build one instruction-diet variant"): the exact body with `power` on
// packed arithmetic -- (dx, dy), (cx' dx, cz' dy) and their squares as three v_pk_*_f32 instead of six scalar operations; the entry
// then carries (-0.5 cx, -0.5 cz) as a register pair and cy behind them.  Same operations on the same values: bit-identical.
// 30 instead of 33 vector instructions per entry; profiles/r06_fwd_pk_ab.txt has what that bought.
#define SGR_FWD_BODY_XPK(X, Y, A, B, CZ, OP, R, G, BL, POS, XY, AB)                                  \
    "v_pk_add_f32 v[60:61], " XY ", %[pxy] neg_lo:[0,1] neg_hi:[0,1]\n"                              \
    "v_pk_mul_f32 v[62:63], " AB ", v[60:61]\n"                                                      \
    "v_pk_mul_f32 v[62:63], v[62:63], v[60:61]\n"                                                    \
    "v_mul_f32 v60, " CZ ", v60\n"      /* cy dx  (CZ holds cy in this layout) */                   \
    "v_add_f32 v62, v62, v63\n"                                                                     \
    "v_mul_f32 v60, v60, v61\n"                                                                     \
    "v_sub_f32 v63, v62, v60\n"         /* power */                                                 \
    "v_mul_f32 v60, 0x3fb8aa3b, v63\n"                                                              \
    "v_rndne_f32 v61, v60\n"                                                                        \
    "v_fma_f32 v62, v63, %[chi], -v60\n"                                                            \
    "v_fmac_f32 v62, 0x32a5705f, v63\n"                                                             \
    "v_sub_f32 v60, v60, v61\n"                                                                     \
    "v_add_f32 v60, v60, v62\n"                                                                     \
    "v_exp_f32 v60, v60\n"                                                                          \
    "v_cvt_i32_f32 v61, v61\n"                                                                      \
    "s_mov_b64 %[live], exec\n"                                                                     \
    "v_cmp_nlt_f32 vcc, 0, v63\n"                                                                   \
    "v_ldexp_f32 v62, v60, v61\n"                                                                   \
    "v_mul_f32 v62, " OP ", v62\n"                                                                  \
    "v_min_f32 v62, 0x3f7d70a4, v62\n"                                                              \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_ngt_f32 vcc, 0x3b808081, v62\n"                                                          \
    "v_sub_f32 v60, 1.0, v62\n"                                                                     \
    "v_mul_f32 v61, v62, %[T]\n"                                                                    \
    "v_mul_f32 v60, %[T], v60\n"                                                                    \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    "v_cmp_gt_f32 vcc, 0x38d1b717, v60\n"                                                           \
    "s_andn2_b64 %[live], %[live], vcc\n"                                                           \
    "s_andn2_b64 exec, exec, vcc\n"                                                                 \
    "v_mov_b32 %[T], v60\n"                                                                         \
    "v_fmac_f32 %[C0], " R ", v61\n"                                                                \
    "v_fmac_f32 %[C1], " G ", v61\n"                                                                \
    "v_fmac_f32 %[C2], " BL ", v61\n"                                                               \
    "v_mov_b32 %[last], " POS "\n"                                                                  \
    "s_mov_b64 exec, %[live]\n"

#define SGR_FWD_ENTRY_BYTES 48


#ifndef SGR_BLEND_CXX
#define SGR_FWD_WALK_ASM(BODY)                                                                                                 \
        "s_mov_b64 %[full], exec\n"                                                                                            \
        "s_and_b64 exec, exec, %[live]\n"                                                                                      \
        "s_waitcnt lgkmcnt(0)\n"  /* scalar loads return out of order: none may be pending while LDS reads are counted */      \
        "ds_read_b128 v[40:43], %[addr]\n"                                                                                     \
        "ds_read_b128 v[44:47], %[addr] offset:16\n"                                                                           \
        "ds_read_b64 v[48:49], %[addr] offset:32\n"                                                                            \
        "1:\n"                                                                                                                 \
        "ds_read_b128 v[50:53], %[addr] offset:48\n"                                                                           \
        "ds_read_b128 v[54:57], %[addr] offset:64\n"                                                                           \
        "ds_read_b64 v[58:59], %[addr] offset:80\n"                                                                            \
        "s_waitcnt lgkmcnt(3)\n"                                                                                               \
        BODY("v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v[40:41]", "v[42:43]")                      \
        "s_cbranch_execz 3f\n"                                                                                                 \
        "s_add_i32 %[n], %[n], -1\n"                                                                                           \
        "s_cmp_eq_u32 %[n], 0\n"                                                                                               \
        "s_cbranch_scc1 3f\n"                                                                                                  \
        "ds_read_b128 v[40:43], %[addr] offset:96\n"                                                                           \
        "ds_read_b128 v[44:47], %[addr] offset:112\n"                                                                          \
        "ds_read_b64 v[48:49], %[addr] offset:128\n"                                                                           \
        "v_add_u32 %[addr], 96, %[addr]\n"                                                                                     \
        "s_waitcnt lgkmcnt(3)\n"                                                                                               \
        BODY("v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v[50:51]", "v[52:53]")                      \
        "s_cbranch_execz 3f\n"                                                                                                 \
        "s_add_i32 %[n], %[n], -1\n"                                                                                           \
        "s_cmp_eq_u32 %[n], 0\n"                                                                                               \
        "s_cbranch_scc0 1b\n"                                                                                                  \
        "3:\n"                                                                                                                 \
        "s_waitcnt lgkmcnt(0)\n"                                                                                               \
        "s_mov_b64 exec, %[full]\n"
#define SGR_FWD_WALK_OPERANDS                                                                                                  \
        : [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [last] "+v"(last), [addr] "+v"(addr), [n] "+s"(n),          \
          [live] "+s"(live), [full] "=&s"(full)                                                                                \
        : [px] "v"(pixfx), [py] "v"(pixfy), [chi] "s"(LOG2E), [pxy] "v"(pixxy)                                                 \
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", \
          "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc", "scc", "memory"
template <bool EXACT>
__device__ __forceinline__ void fwd_walk(uint32_t addr, int& n, unsigned long long& live, float pixfx, float pixfy, float& T,
                                         float& C0, float& C1, float& C2, uint32_t& last)
{
    unsigned long long full;
    const unsigned long long pixxy = ((unsigned long long)__float_as_uint(pixfy) << 32) | (unsigned long long)__float_as_uint(pixfx);  // (px, py) as a register pair
#ifdef SGR_FWD_PK
    if constexpr (EXACT) asm volatile(SGR_FWD_WALK_ASM(SGR_FWD_BODY_XPK) SGR_FWD_WALK_OPERANDS);
#else
    if constexpr (EXACT) asm volatile(SGR_FWD_WALK_ASM(SGR_FWD_BODY_X) SGR_FWD_WALK_OPERANDS);
#endif
    else asm volatile(SGR_FWD_WALK_ASM(SGR_FWD_BODY) SGR_FWD_WALK_OPERANDS);
}
#endif

#ifdef SGR_BLEND_CXX
// ---- ANALYSIS BUILD (SGR_BLEND_DEFS="-DSGR_BLEND_CXX=<mask>", scripts/r06_grad_switches.sh): the two hand-scheduled walks as plain
// C++, with the places where their arithmetic departs from the reference's (forward.cu:330-366, backward.cu:486-554, compiled without
// contraction) switchable one by one, to find which departure carries the gradient error of tests/test_gpu_fullsize.py.  Mask 0
// reproduces the assembly bit for bit.  Not a product path: slower, and never built by default.
//   1  power from the raw conic in the reference's operation order (ours: conic pre-scaled by log2 e, two FMAs)
//   2  G = expf(power) (needs 1; ours: v_exp_f32 of the pre-scaled power)
//   4  test_T = T * (1 - alpha)  (ours: T - alpha * T)
//   8  C += (c * alpha) * T, unfused  (ours: fma(c, alpha * T, C))
//  16  T = T / (1 - alpha)  (ours: T * v_rcp_f32(1 - alpha))
//  32  accum_rec per channel and the background term as in backward.cu:514-534 (ours: the scalar recurrence on colour . g)
//  64  phase B's moment sums in double
#define SGR_X(bit) ((SGR_BLEND_CXX) & (bit))
#pragma clang fp contract(off)
__device__ __forceinline__ float x_power_G(const float* e, float dx, float dy, float& power_sign)
{
#if SGR_X(1)
    const float power = -0.5f * (e[2] * dx * dx + e[4] * dy * dy) - e[3] * dx * dy;
    power_sign = power;
#if SGR_X(2)
    return expf(power);
#else
    return __builtin_amdgcn_exp2f(power * LOG2E);
#endif
#else
    const float t = __builtin_fmaf(e[2], dx, e[3] * dy);
    const float p2 = __builtin_fmaf(dx, t, (e[4] * dy) * dy);
    power_sign = p2;
    return __builtin_amdgcn_exp2f(p2);
#endif
}
__device__ __forceinline__ void fwd_walk_cxx(const float* s_e, int& n, unsigned long long& live, float pixfx, float pixfy, float& T,
                                             float& C0, float& C1, float& C2, uint32_t& last)
{
    const int lane = threadIdx.x;
    const int n0 = n;
    for (int k = 0; k < n0; k++) {
        const float* e = s_e + k * (SGR_FWD_ENTRY_BYTES / 4);
        const bool act = (live >> lane) & 1ull;
        bool fin = false;
        if (act) {
            const float dx = e[0] - pixfx, dy = e[1] - pixfy;
            float ps;
            const float G = x_power_G(e, dx, dy, ps);
            if (!(ps > 0.f)) {
                const float alpha = fminf(0.99f, e[5] * G);
                if (!(alpha < 1.0f / 255.0f)) {
#if SGR_X(4)
                    const float test_T = T * (1.f - alpha);
                    const float aT = alpha * T;
#else
                    const float aT = alpha * T;
                    const float test_T = T - aT;
#endif
                    if (test_T < 0.0001f) fin = true;
                    else {
#if SGR_X(8)
                        C0 += e[6] * alpha * T; C1 += e[7] * alpha * T; C2 += e[8] * alpha * T;
#else
                        C0 = __builtin_fmaf(e[6], aT, C0); C1 = __builtin_fmaf(e[7], aT, C1); C2 = __builtin_fmaf(e[8], aT, C2);
#endif
                        T = test_T;
                        last = __float_as_uint(e[9]);
                    }
                }
            }
        }
        live &= ~__ballot(fin);
        if (live == 0ull) { n = n0 - k; return; }
    }
    n = 0;
}
#pragma clang fp contract(fast)
#endif  // SGR_BLEND_CXX

template <bool REPAIR, bool EXACT>
__device__ __forceinline__ void blend_fwd_body(int W, int H, int gx, int T_tiles, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ point_list, const GeomRec* __restrict__ rec,
                                                    const float* __restrict__ bg, float* __restrict__ final_T,
                                                    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_maxc,
                                                    uint32_t* __restrict__ tile_walked, float* __restrict__ out_color,
                                                    unsigned long long* __restrict__ blk_mask, uint32_t* __restrict__ blk_nb,
                                                    uint32_t* __restrict__ header, uint32_t list_cap,
                                                    const uint32_t* __restrict__ tile_need, const uint32_t* __restrict__ launch_order,
                                                    uint32_t* __restrict__ repair_flag, uint32_t* __restrict__ repair_list, uint32_t deep_min = 0u)
{
    // sync-free forward: the list did not fit the caller's capacity (or the level-1 binning overflowed) -> leave everything
    // untouched; the caller repeats the forward and every later kernel of the step reads the same header
    if (!REPAIR && blockIdx.x == 0 && threadIdx.x == 0) header[SGR_HDR_LAYOUT_CAP] = list_cap;  // the layout of THIS forward's binning buffer, for its backward
    if (header[SGR_HDR_R] > list_cap || header[4 + SGR_B2_HDR_OVERFLOW]) return;
    // entry k: {x, y, -0.5*conic.x*log2e, -conic.y*log2e | -0.5*conic.z*log2e, opacity, r, g | b, bitcast(1-based list position), -, -}
    // (one spare entry: the walk's look-ahead reads one entry past the last)
    __shared__ __attribute__((aligned(16))) float s_e[65 * (SGR_FWD_ENTRY_BYTES / 4)];
    // workgroup b runs on XCD b % 8: the four blocks of a tile share an XCD (and its L2)
    int slot, sub;
    sgr_slot_of_workgroup((int)blockIdx.x, slot, sub);
    if (slot >= T_tiles) return;
    if (REPAIR) {
        // the repair pass of the walk hint: launch_order = the tiles the first pass listed, header word SGR_HDR_REPAIR their count
        const uint32_t n_rep = header[SGR_HDR_REPAIR];
        if (n_rep > (uint32_t)SGR_REPAIR_TILES) {  // more than this launch covers: the forward is invalid after all
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&header[SGR_HDR_HINT_MISS], 1u);
            return;
        }
        if ((uint32_t)slot >= n_rep) return;
    }
    // (launch_order: the camera's previous visit, deepest tiles first -- sgr_forward_opts.tile_order; clamped, so a buffer that
    // is not a permutation costs tiles, not memory safety)
    const int tile = launch_order ? (int)min(launch_order[slot], (uint32_t)(T_tiles - 1)) : slot;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int bx0 = tx * SGR_TILE_X + 8 * (sub & 1), by0 = ty * SGR_TILE_Y + 8 * (sub >> 1);
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile];
    const int total_all = (int)(tile_start[tile + 1] - r0);
    // walk hint: only the first tile_need[tile] entries of the list are guaranteed to have been written
    const int total = tile_need ? (int)min((uint32_t)total_all, tile_need[tile]) : total_all;
    // a block whose hinted list is longer than deep_min entries belongs to k_blend_fwd_deep (eight waves per block, below)
    if (!REPAIR && deep_min && tile_need && (uint32_t)total > deep_min) return;

    float T = 1.0f;
    uint32_t last_contributor = 0;
    uint32_t walked = (uint32_t)total;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    unsigned long long live = __ballot(inside);  // lanes still accumulating

    // The ids of the next batch are fetched while this one is walked; the records are gathered at the top of the iteration
    // (compiler-scheduled loads: other resident waves cover the latency.  Loads issued from inline asm into registers that
    // stay in flight across the loop edge are NOT safe: the compiler may copy or reuse the destination registers before
    // the data lands.)
    // (loads are unconditional with clamped indices: a load under a lane predicate makes the compiler's wait counts
    // path-dependent and it then drains everything at the first use)
    // (round 5: the ids run TWO batches ahead.  With one batch of lead the id load of batch b + 1, issued at the top of iteration b,
    // had only that iteration's cull and walk to come back in -- a fraction of a microsecond when few entries survive the cull --
    // so every batch paid two dependent memory round trips, ids then records: a tile that walks thousands of entries, a silhouette
    // tile of BASELINE config 4's flat splats walks 4 700, is a chain of ~3 us links and the whole launch waits for it.)
    uint32_t id_next = 0u, id_next2 = 0u;
    if (total > 0) {  // (uniform branch)
        id_next = point_list[r0 + (uint32_t)min(lane, total - 1)];
        id_next2 = point_list[r0 + (uint32_t)min(64 + lane, total - 1)];
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)s_e;  // LDS byte address of the staging area
    // for the backward: which lanes of every 64-entry batch survived this block's cull (one 64-bit mask per batch and block;
    // batch b of the tile sits in slot (r0 >> 6) + tile + b: slots of different tiles never overlap)
    unsigned long long* my_mask = blk_mask + 4 * ((size_t)(r0 >> 6) + (size_t)tile) + sub;
    int n_batches = 0;
#ifdef SGR_FWD_PREFETCH_RECORDS
    // (A/B build option, round 5: the RECORDS of batch b + 1 are requested before batch b is culled and walked -- twelve more live
    // registers across the walk.  For the one wave of a tile that walks thousands of entries the gather's round trip is then off
    // the chain; see DESIGN.md for what it did to the metric workload and to config 4)
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
    if (total > 0) {
        const float4* rp0 = reinterpret_cast<const float4*>(rec + id_next);
        p0 = rp0[0]; p1 = rp0[1]; p2 = rp0[2];
        id_next = id_next2;
        id_next2 = point_list[r0 + (uint32_t)min(128 + lane, total - 1)];
    }
#endif
    for (int base = 0; base < total && live != 0ull; base += 64) {
#ifdef SGR_FWD_PREFETCH_RECORDS
        const float4 v0 = p0, v1 = p1, v2 = p2;
        {
            const float4* rp = reinterpret_cast<const float4*>(rec + id_next);   // batch b + 1 (clamped ids past the end: a cached line)
            p0 = rp[0]; p1 = rp[1]; p2 = rp[2];
        }
        id_next = id_next2;
        id_next2 = point_list[r0 + (uint32_t)min(base + 192 + lane, total - 1)];
#else
        const float4* rp = reinterpret_cast<const float4*>(rec + id_next);
        const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
        id_next = id_next2;
        id_next2 = point_list[r0 + (uint32_t)min(base + 128 + lane, total - 1)];
#endif
        const bool hit = (base + lane < total) && block_hit(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, (float)bx0, (float)by0);
        const unsigned long long m = __ballot(hit);
        int n = __popcll(m);
        if (hit) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            float4* e = reinterpret_cast<float4*>(s_e + pos * (SGR_FWD_ENTRY_BYTES / 4));
#if defined(SGR_BLEND_CXX) && ((SGR_BLEND_CXX) & 1)
            e[0] = make_float4(v0.x, v0.y, v0.z, v0.w);   // (analysis build: the raw conic)
            e[1] = make_float4(v1.x, v1.y, v2.x, v2.y);
#else
            if (EXACT) {  // the raw conic, the two halvings applied (exact): SGR_FWD_BODY_X
#ifdef SGR_FWD_PK
                e[0] = make_float4(v0.x, v0.y, -0.5f * v0.z, -0.5f * v1.x);   // (-0.5 cx, -0.5 cz) as a pair, cy behind them
                e[1] = make_float4(v0.w, v1.y, v2.x, v2.y);
#else
                e[0] = make_float4(v0.x, v0.y, -0.5f * v0.z, v0.w);
                e[1] = make_float4(-0.5f * v1.x, v1.y, v2.x, v2.y);
#endif
            } else {
                e[0] = make_float4(v0.x, v0.y, -0.5f * LOG2E * v0.z, -LOG2E * v0.w);
                e[1] = make_float4(-0.5f * LOG2E * v1.x, v1.y, v2.x, v2.y);
            }
#endif
            *reinterpret_cast<float2*>(e + 2) = make_float2(v2.z, __uint_as_float((uint32_t)(base + lane + 1)));
        }
        if (lane == 0) my_mask[4 * (size_t)n_batches] = m;
        n_batches++;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n > 0) {
            const int n0 = n;
#ifdef SGR_BLEND_CXX
            fwd_walk_cxx(s_e, n, live, pixfx, pixfy, T, C0, C1, C2, last_contributor);
#else
            fwd_walk<EXACT>(lds0, n, live, pixfx, pixfy, T, C0, C1, C2, last_contributor);
#endif
            if (live == 0ull) {  // every pixel finished at compacted entry n0 - n: its list position is the furthest examined
                walked = __float_as_uint(s_e[(n0 - n) * (SGR_FWD_ENTRY_BYTES / 4) + 9]);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // the hint was too short: pixels of this block are still accumulating where the written prefix ends
    if (total < total_all && live != 0ull && lane == 0) {
        if (repair_flag) {
            // (round 5) ... which costs THIS TILE a second pass, not the whole forward: the first of its four blocks to notice puts
            // the tile on the repair list; the list-write pass and the blend run once more for the listed tiles (capi.hip)
            if (atomicExch(&repair_flag[tile], 0xFFFFFFFFu) == 0u) repair_list[atomicAdd(&header[SGR_HDR_REPAIR], 1u)] = (uint32_t)tile;
        } else {
            atomicOr(&header[SGR_HDR_HINT_MISS], 1u);
        }
    }
    if (inside) {
        const size_t pix_id = (size_t)W * py + px;
        const size_t HW = (size_t)H * W;
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
#if defined(SGR_BLEND_CXX) && ((SGR_BLEND_CXX) & 8)
        out_color[pix_id] = __fadd_rn(C0, __fmul_rn(T, bg[0]));
        out_color[HW + pix_id] = __fadd_rn(C1, __fmul_rn(T, bg[1]));
        out_color[2 * HW + pix_id] = __fadd_rn(C2, __fmul_rn(T, bg[2]));
#else
        out_color[pix_id] = C0 + T * bg[0];
        out_color[HW + pix_id] = C1 + T * bg[1];
        out_color[2 * HW + pix_id] = C2 + T * bg[2];
#endif
    }
    // deepest contributor / furthest examined position of the TILE (zeroed by the launcher): maximum over its four blocks
    uint32_t mc = inside ? last_contributor : 0u;
    for (int o = 32; o > 0; o >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, o));
    if (lane == 0) {
        atomicMax(&tile_maxc[tile], mc);
        atomicMax(&tile_walked[tile], __ballot(inside) ? walked : 0u);
        blk_nb[4 * tile + sub] = mc ? (uint32_t)n_batches : 0u;  // batches with a mask (0: nothing contributed to the block)
    }
}

#define SGR_FWD_PARAMS                                                                                                         \
    int W, int H, int gx, int T_tiles, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,               \
        const GeomRec *__restrict__ rec, const float *__restrict__ bg, float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, \
        uint32_t *__restrict__ tile_maxc, uint32_t *__restrict__ tile_walked, float *__restrict__ out_color,                         \
        unsigned long long *__restrict__ blk_mask, uint32_t *__restrict__ blk_nb, uint32_t *__restrict__ header, uint32_t list_cap
#define SGR_FWD_ARGS W, H, gx, T_tiles, tile_start, point_list, rec, bg, final_T, n_contrib, tile_maxc, tile_walked, out_color, blk_mask, blk_nb, header, list_cap
__global__ void __launch_bounds__(64) k_blend_fwd_w(SGR_FWD_PARAMS, const uint32_t* __restrict__ tile_need, const uint32_t* __restrict__ launch_order,
                                                    uint32_t* __restrict__ repair_flag, uint32_t* __restrict__ repair_list, uint32_t deep_min)
{
    blend_fwd_body<false, false>(SGR_FWD_ARGS, tile_need, launch_order, repair_flag, repair_list, deep_min);
}
// (the exact-alpha variant: SGR_FWD_BODY_X)
__global__ void __launch_bounds__(64) k_blend_fwd_wx(SGR_FWD_PARAMS, const uint32_t* __restrict__ tile_need, const uint32_t* __restrict__ launch_order,
                                                     uint32_t* __restrict__ repair_flag, uint32_t* __restrict__ repair_list, uint32_t deep_min)
{
    blend_fwd_body<false, true>(SGR_FWD_ARGS, tile_need, launch_order, repair_flag, repair_list, deep_min);
}

// the repair pass of the walk hint (a kernel name of its own, so that a trace tells the gated, usually empty launch from the blend)
__global__ void __launch_bounds__(64) k_blend_fwd_repair(SGR_FWD_PARAMS, const uint32_t* __restrict__ repair_list)
{
    blend_fwd_body<true, false>(SGR_FWD_ARGS, nullptr, repair_list, nullptr, nullptr);
}
__global__ void __launch_bounds__(64) k_blend_fwd_repairx(SGR_FWD_PARAMS, const uint32_t* __restrict__ repair_list)
{
    blend_fwd_body<true, true>(SGR_FWD_ARGS, nullptr, repair_list, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------
// Long lists (round 6): EIGHT waves per 8 x 8 block.
//
// A block whose (hinted) list runs to thousands of entries -- the silhouette tiles of BASELINE config 4's flat, mesh-bound splats walk
// 4 500 -- is a serial chain for its one wave while the rest of the chip has long finished.  Splitting the LIST between waves and
// composing (C1 + T1 C2, T1 T2) was built in round 5 and was slower: pixels stop all along such a list, nearly every run had to be
// walked twice.  What parallelises without that problem is the EXPENSIVE part of an entry, alpha (cull, gather, power, exp: 25 of
// the walk's 33 instructions), which does not depend on the state of the pixel; what stays sequential is the cheap part, the
// transmittance chain T <- T (1 - alpha) with its stop rule and the colour sums (same operations in the same order as the one-wave
// walk, so the result is bit-identical to it -- and with exact alpha to the reference):
//   phase A  every wave takes one 64-entry batch of the next eight: ids and records gathered (lane = entry), the exact block cull,
//            the survivor mask for the backward, then per survivor (wave-uniform, broadcast with v_readlane) one alpha per pixel lane
//            into the wave's LDS panel (0 = "skipped": power > 0 or alpha < 1/255);
//   phase B  wave 0 walks the eight panels in list order: per survivor a load, the three transmittance operations, the stop test,
//            three FMAs.
// The blocks are listed by k_deep_list (tiles whose hinted length exceeds deep_min); k_blend_fwd_w(x) skips exactly those, and the
// two kernels run side by side on two streams (capi.hip).  136 KB of LDS: one workgroup per CU, which is all a handful of blocks need.
#define DEEP_WAVES 8
#define DEEP_ROWS 64
// Phase B of k_blend_fwd_deep as a hand-scheduled loop (the compiler's version of it -- a dependent LDS read, a 64-bit shift and a
// ballot per row -- ran at ~350 cycles per row, which made the eight-wave kernel no faster than the one-wave walk it replaces):
// the tail of SGR_FWD_BODY(_X) behind the alpha it reads from the panel, rows double-buffered like the entries of fwd_walk.
//   aaddr: LDS byte address of this lane's alpha in row 0 (row stride 256 B); eaddr: of row 0's {r, g, b, position} (16 B per row)
//   n rows (> 0); on return n = rows not yet started when every lane had finished (0: walked to the end), live updated
#define SGR_DEEP_TAIL(EXACT_OPS, A, R, G, BL, POS)                                                  \
    "s_mov_b64 %[live], exec\n"                                                                     \
    "v_cmp_neq_f32 vcc, 0, " A "\n"     /* 0 = skipped in phase A (power > 0 or alpha < 1/255) */   \
    "s_and_b64 exec, exec, vcc\n"                                                                   \
    EXACT_OPS(A)                                                                                    \
    "v_cmp_gt_f32 vcc, 0x38d1b717, v60\n" /* test_T < 0.0001: this lane is finished */              \
    "s_andn2_b64 %[live], %[live], vcc\n"                                                           \
    "s_andn2_b64 exec, exec, vcc\n"                                                                 \
    "v_mov_b32 %[T], v60\n"                                                                         \
    "v_fmac_f32 %[C0], " R ", v61\n"                                                                \
    "v_fmac_f32 %[C1], " G ", v61\n"                                                                \
    "v_fmac_f32 %[C2], " BL ", v61\n"                                                               \
    "v_mov_b32 %[last], " POS "\n"                                                                  \
    "s_mov_b64 exec, %[live]\n"
#define SGR_DEEP_T_EXACT(A) "v_sub_f32 v60, 1.0, " A "\n" "v_mul_f32 v61, " A ", %[T]\n" "v_mul_f32 v60, %[T], v60\n"
#define SGR_DEEP_T_FAST(A) "v_mul_f32 v61, " A ", %[T]\n" "v_sub_f32 v60, %[T], v61\n"
#define SGR_DEEP_WALK_ASM(OPS)                                                                      \
        "s_mov_b64 %[full], exec\n"                                                                 \
        "s_and_b64 exec, exec, %[live]\n"                                                           \
        "s_waitcnt lgkmcnt(0)\n"                                                                    \
        "ds_read_b32 v40, %[aaddr]\n"                                                               \
        "ds_read_b128 v[44:47], %[eaddr]\n"                                                         \
        "1:\n"                                                                                      \
        "ds_read_b32 v50, %[aaddr] offset:256\n"                                                    \
        "ds_read_b128 v[54:57], %[eaddr] offset:16\n"                                               \
        "s_waitcnt lgkmcnt(2)\n"                                                                    \
        SGR_DEEP_TAIL(OPS, "v40", "v44", "v45", "v46", "v47")                                       \
        "s_cbranch_execz 3f\n"                                                                      \
        "s_add_i32 %[n], %[n], -1\n"                                                                \
        "s_cmp_eq_u32 %[n], 0\n"                                                                    \
        "s_cbranch_scc1 3f\n"                                                                       \
        "ds_read_b32 v40, %[aaddr] offset:512\n"                                                    \
        "ds_read_b128 v[44:47], %[eaddr] offset:32\n"                                               \
        "v_add_u32 %[aaddr], 512, %[aaddr]\n"                                                       \
        "v_add_u32 %[eaddr], 32, %[eaddr]\n"                                                        \
        "s_waitcnt lgkmcnt(2)\n"                                                                    \
        SGR_DEEP_TAIL(OPS, "v50", "v54", "v55", "v56", "v57")                                       \
        "s_cbranch_execz 3f\n"                                                                      \
        "s_add_i32 %[n], %[n], -1\n"                                                                \
        "s_cmp_eq_u32 %[n], 0\n"                                                                    \
        "s_cbranch_scc0 1b\n"                                                                       \
        "3:\n"                                                                                      \
        "s_waitcnt lgkmcnt(0)\n"                                                                    \
        "s_mov_b64 exec, %[full]\n"
#define SGR_DEEP_WALK_OPERANDS                                                                      \
        : [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [last] "+v"(last), [aaddr] "+v"(aaddr), [eaddr] "+v"(eaddr), \
          [n] "+s"(n), [live] "+s"(live), [full] "=&s"(full)                                        \
        :                                                                                           \
        : "v40", "v44", "v45", "v46", "v47", "v50", "v54", "v55", "v56", "v57", "v60", "v61", "vcc", "scc", "memory"
template <bool EXACT>
__device__ __forceinline__ void deep_walk(uint32_t aaddr, uint32_t eaddr, int& n, unsigned long long& live, float& T, float& C0, float& C1,
                                          float& C2, uint32_t& last)
{
    unsigned long long full;
    if constexpr (EXACT) asm volatile(SGR_DEEP_WALK_ASM(SGR_DEEP_T_EXACT) SGR_DEEP_WALK_OPERANDS);
    else asm volatile(SGR_DEEP_WALK_ASM(SGR_DEEP_T_FAST) SGR_DEEP_WALK_OPERANDS);
}

template <bool EXACT>
__device__ __forceinline__ float deep_alpha(float x, float y, float cx, float cy, float cz, float op, float px, float py)
{
#pragma clang fp contract(off)
    const float dx = x - px, dy = y - py;
    if (EXACT) {   // SGR_FWD_BODY_X, operation for operation
        const float t1 = ((-0.5f * cx) * dx) * dx, t2 = ((-0.5f * cz) * dy) * dy;
        const float power = (t1 + t2) - (cy * dx) * dy;
        if (power > 0.f) return 0.f;
        const float alpha = fminf(0.99f, op * expf(power));
        return alpha < 1.0f / 255.0f ? 0.f : alpha;
    } else {       // SGR_FWD_BODY
        const float A = -0.5f * LOG2E * cx, B = -LOG2E * cy, CZ = -0.5f * LOG2E * cz;
        const float t = __builtin_fmaf(A, dx, B * dy);
        const float p2 = __builtin_fmaf(dx, t, (CZ * dy) * dy);
        if (p2 > 0.f) return 0.f;
        const float alpha = fminf(0.99f, op * __builtin_amdgcn_exp2f(p2));
        return alpha < 1.0f / 255.0f ? 0.f : alpha;
    }
}

__global__ void __launch_bounds__(256) k_deep_list(int T, const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_need,
                                                   uint32_t deep_min, uint32_t* __restrict__ header, uint32_t list_cap,
                                                   uint32_t* __restrict__ deep_list)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T || header[SGR_HDR_R] > list_cap || header[4 + SGR_B2_HDR_OVERFLOW]) return;
    const uint32_t total = min(tile_start[t + 1] - tile_start[t], tile_need[t]);
    if (total > deep_min) deep_list[atomicAdd(&header[SGR_HDR_DEEP], 1u)] = (uint32_t)t;
}

template <bool EXACT>
__global__ void __launch_bounds__(64 * DEEP_WAVES)
k_blend_fwd_deep(SGR_FWD_PARAMS, const uint32_t* __restrict__ tile_need, const uint32_t* __restrict__ deep_list,
                 uint32_t* __restrict__ repair_flag, uint32_t* __restrict__ repair_list)
{
    extern __shared__ __attribute__((aligned(16))) float s_deep[];
    float* s_alpha = s_deep;                                                              // [DEEP_WAVES][DEEP_ROWS][64]
    float4* s_ent = reinterpret_cast<float4*>(s_deep + DEEP_WAVES * DEEP_ROWS * 64);      // [DEEP_WAVES][DEEP_ROWS] {r, g, b, position}
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_ent + DEEP_WAVES * DEEP_ROWS);        // [DEEP_WAVES] survivors of the wave's batch, [8] stop
    if (header[SGR_HDR_R] > list_cap || header[4 + SGR_B2_HDR_OVERFLOW]) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_deep = header[SGR_HDR_DEEP];
    for (uint32_t slot = blockIdx.x; slot < 4u * n_deep; slot += gridDim.x) {
        const int tile = (int)min(deep_list[slot >> 2], (uint32_t)(T_tiles - 1)), sub = (int)(slot & 3u);
        const int tx = tile % gx, ty = tile / gx;
        const int bx0 = tx * SGR_TILE_X + 8 * (sub & 1), by0 = ty * SGR_TILE_Y + 8 * (sub >> 1);
        const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
        const bool inside = px < W && py < H;
        const float pixfx = (float)px, pixfy = (float)py;
        const uint32_t r0 = tile_start[tile];
        const int total_all = (int)(tile_start[tile + 1] - r0);
        const int total = (int)min((uint32_t)total_all, tile_need[tile]);
        const int n_b = (total + 63) / 64;
        unsigned long long* my_mask = blk_mask + 4 * ((size_t)(r0 >> 6) + (size_t)tile) + sub;
        // the pixel state lives in wave 0
        float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
        uint32_t last_contributor = 0, walked = (uint32_t)total;
        unsigned long long live = __ballot(inside);
        int batches_seen = 0;
        for (int seg = 0; seg < n_b; seg += DEEP_WAVES) {
            // ---------------- phase A: one batch per wave
            const int b = seg + wave;
            uint32_t cnt = 0;
            if (b < n_b) {   // (wave-uniform)
                const uint32_t id = point_list[r0 + (uint32_t)min(64 * b + lane, total - 1)];
                const float4* rp = reinterpret_cast<const float4*>(rec + id);
                const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
                const bool hit = (64 * b + lane < total) && block_hit(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, (float)bx0, (float)by0);
                unsigned long long m = __ballot(hit);
                if (lane == 0) my_mask[4 * (size_t)b] = m;
                float* arow = s_alpha + ((size_t)wave * DEEP_ROWS) * 64 + lane;
                while (m) {
                    const int j = (int)__builtin_ctzll(m);
                    m &= m - 1;
#define RL(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j))
                    const float ex = RL(v0.x), ey = RL(v0.y), ecx = RL(v0.z), ecy = RL(v0.w), ecz = RL(v1.x), eop = RL(v1.y);
                    arow[(size_t)cnt * 64] = deep_alpha<EXACT>(ex, ey, ecx, ecy, ecz, eop, pixfx, pixfy);
                    if (lane == 0) s_ent[wave * DEEP_ROWS + cnt] = make_float4(RL(v2.x), RL(v2.y), RL(v2.z), __uint_as_float((uint32_t)(64 * b + j + 1)));
#undef RL
                    cnt++;
                }
            }
            if (lane == 0) s_cnt[wave] = cnt;
            __syncthreads();
            // ---------------- phase B: wave 0, in list order (deep_walk)
            if (wave == 0) {
                bool stop = false;
                for (int w = 0; w < DEEP_WAVES && !stop; w++) {
                    if (seg + w >= n_b) break;
                    batches_seen = seg + w + 1;
                    int n = __builtin_amdgcn_readfirstlane((int)s_cnt[w]);
                    if (n == 0) continue;
                    const int n0 = n;
                    const uint32_t a0 = (uint32_t)(uintptr_t)(s_alpha + ((size_t)w * DEEP_ROWS) * 64 + lane);
                    const uint32_t e0 = (uint32_t)(uintptr_t)(s_ent + w * DEEP_ROWS);
                    deep_walk<EXACT>(a0, e0, n, live, T, C0, C1, C2, last_contributor);
                    if (live == 0ull) { walked = __float_as_uint(s_ent[w * DEEP_ROWS + (n0 - n)].w); stop = true; }
                }
                if (lane == 0) s_cnt[DEEP_WAVES] = stop ? 1u : 0u;
            }
            __syncthreads();
            if (s_cnt[DEEP_WAVES]) break;
        }
        if (wave == 0) {
            if (total < total_all && live != 0ull && lane == 0) {   // the hint was too short (see blend_fwd_body)
                if (repair_flag) {
                    if (atomicExch(&repair_flag[tile], 0xFFFFFFFFu) == 0u) repair_list[atomicAdd(&header[SGR_HDR_REPAIR], 1u)] = (uint32_t)tile;
                } else {
                    atomicOr(&header[SGR_HDR_HINT_MISS], 1u);
                }
            }
            if (inside) {
                const size_t pix_id = (size_t)W * py + px;
                const size_t HW = (size_t)H * W;
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                out_color[pix_id] = C0 + T * bg[0];
                out_color[HW + pix_id] = C1 + T * bg[1];
                out_color[2 * HW + pix_id] = C2 + T * bg[2];
            }
            uint32_t mc = inside ? last_contributor : 0u;
            for (int o = 32; o > 0; o >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, o));
            if (lane == 0) {
                atomicMax(&tile_maxc[tile], mc);
                atomicMax(&tile_walked[tile], __ballot(inside) ? walked : 0u);
                // batches with a mask: every batch of the segments phase A touched (the backward drops what lies behind the deepest contributor)
                const int masked = min(n_b, ((batches_seen + DEEP_WAVES - 1) / DEEP_WAVES) * DEEP_WAVES);
                blk_nb[4 * tile + sub] = mc ? (uint32_t)masked : 0u;
            }
        }
        __syncthreads();   // (the panels are rewritten by the next block of this workgroup)
    }
}

// ---------------------------------------------------------------------------------------------
// Backward blend.
//
// Two lane mappings alternate over groups of BW_SUB = 8 Gaussians of the block's list (walked back to front; 16 until round 3):
//   phase A  lane = pixel.  The per-pixel recurrence of backward.cu:486-534 (T /= 1-alpha, accum_rec, dL_dalpha) runs
//            sequentially over the group's Gaussians; for each pair the lane stores just Z = G * dL_dalpha and Wt = alpha * T into
//            a wave-private LDS panel zw[g][pixel].  Hand-scheduled like the forward walk: 27.5 VALU per entry, the
//            reference's tests as EXEC masks.
//   phase B  lane = (Gaussian g, pixel row q).  Each lane streams its 8 pixels out of the panel and accumulates, in registers,
//            the colour sums  sum Wt*dL_dpix  and the raw moments of Z about the block origin; one 8-lane reduction per group
//            (a DPP row rotate, then two values per v_permlane16/32_swap).
// All gradient terms of a pair are linear in {Wt*g_c, Z, Z*dx, Z*dy, Z*dx^2, Z*dx*dy, Z*dy^2} with per-Gaussian
// coefficients (backward.cu:538-554), so only those nine sums leave the block; the coefficients are applied once per
// Gaussian by the fused backward-preprocess kernel.  A (block, Gaussian) pair is met exactly once, so the nine sums go
// straight to global memory: they are handed to nine neighbouring lanes through LDS, and one atomic instruction then
// carries whole 36-byte records -- ONE request per pair into acc[P][16] (64-byte records).  (Nine separate atomic
// instructions per group cost 5x the whole rest of the kernel.)
// Which list entries to take comes from the forward: it leaves one 64-bit mask per (64-entry batch, block) -- the lanes that
// survived the block's exact cull -- so the backward neither culls again nor gathers records it will not use; the
// survivors of several batches are collected in a small LDS queue so that phase B always sees full groups of BW_SUB.
#define BW_SUB 8            // Gaussians per group (panel rows); phase B's lane roles are written for 8
#define BW_QCAP (BW_SUB + 64)  // queue entries: at most BW_SUB - 1 left over + 64 new
#define BW_ZW_STRIDE 65     // float2 units: conflict-free for the phase-A writes and the phase-B reads
#define BW_ENTRY_DW 12      // x, y, A, B | C, opacity, r, g | b, list position (1-based), id, -

// The ten registers of an entry are reused in place as it is evaluated (no further temporaries):
//   X -> dx -> Z        Y -> dy -> Wt       A -> G         B -> t -> 1/(1-alpha)       CZ -> power -> dL_dalpha     OP -> alpha
//   R -> colour . g
// (X, Y must be an even-aligned register pair: they leave as the (Z, Wt) panel entry.)
// accum_rec (backward.cu:514-516) is advanced at the END of the entry that produced (alpha, colour) instead of at the start of
// the next contributing one: the difference colour.g - accum_rec.g it needs is the one dL_dalpha uses anyway, and no copy of
// alpha or of the colour term has to be carried.  Same operations on the same values: bit-identical sums.
// (Measured, same box: the walk is bound by the latency of this dependent chain at the 3 waves per SIMD the LDS footprint allows,
// not by instruction issue -- removing 2.5 of 30 instructions that sit beside the chain changed the kernel by < 1 %, zeroing the
// panel row with a second LDS write instead of the two v_mov cost 1.5 % there and 5 % at the 5 waves per SIMD of the 8-row groups.)
#define SGR_BWD_HEAD(X, Y, A, B, CZ, OP, POS)                                                     \
    "v_sub_f32 " X ", " X ", %[px]\n"                                                             \
    "v_sub_f32 " Y ", " Y ", %[py]\n"                                                             \
    "v_mul_f32 " B ", " B ", " Y "\n"                                                             \
    "v_fmac_f32 " B ", " A ", " X "\n"                                                            \
    "v_mul_f32 " CZ ", " CZ ", " Y "\n"                                                           \
    "v_mul_f32 " CZ ", " CZ ", " Y "\n"                                                           \
    "v_fmac_f32 " CZ ", " X ", " B "\n"  /* log2(e) * power */                                    \
    "v_exp_f32 " A ", " CZ "\n"          /* G */                                                  \
    "v_mov_b32 " X ", 0\n"                                                                        \
    "v_mov_b32 " Y ", 0\n"                                                                        \
    "v_cmp_nlt_f32 %[m0], 0, " CZ "\n"   /* !(power > 0) */                                       \
    "v_cmp_le_u32 %[m1], " POS ", %[lastc]\n" /* at or before this pixel's last contributor */
// the exact-alpha head (see SGR_FWD_BODY_X: backward.cu:492-499 operation for operation; A = -0.5 cx, B = cy, CZ = -0.5 cz on entry,
// A = G and CZ = power on exit, like the head above)
#define SGR_BWD_HEAD_X(X, Y, A, B, CZ, OP, POS)                                                   \
    "v_sub_f32 " X ", " X ", %[px]\n"                                                             \
    "v_sub_f32 " Y ", " Y ", %[py]\n"                                                             \
    "v_mul_f32 " A ", " A ", " X "\n"                                                             \
    "v_mul_f32 " CZ ", " CZ ", " Y "\n"                                                           \
    "v_mul_f32 " A ", " A ", " X "\n"                                                             \
    "v_mul_f32 " CZ ", " CZ ", " Y "\n"                                                           \
    "v_mul_f32 " B ", " B ", " X "\n"                                                             \
    "v_add_f32 " A ", " A ", " CZ "\n"                                                            \
    "v_mul_f32 " B ", " B ", " Y "\n"                                                             \
    "v_sub_f32 " CZ ", " A ", " B "\n"   /* power */                                              \
    "v_mul_f32 " A ", 0x3fb8aa3b, " CZ "\n"                                                       \
    "v_rndne_f32 " B ", " A "\n"                                                                  \
    "v_fma_f32 " X ", " CZ ", %[chi], -" A "\n"                                                   \
    "v_fmac_f32 " X ", 0x32a5705f, " CZ "\n"                                                      \
    "v_sub_f32 " A ", " A ", " B "\n"                                                             \
    "v_add_f32 " A ", " A ", " X "\n"                                                             \
    "v_exp_f32 " A ", " A "\n"                                                                    \
    "v_cvt_i32_f32 " B ", " B "\n"                                                                \
    "v_mov_b32 " X ", 0\n"                                                                        \
    "v_mov_b32 " Y ", 0\n"                                                                        \
    "v_cmp_nlt_f32 %[m0], 0, " CZ "\n"   /* !(power > 0) */                                       \
    "v_cmp_le_u32 %[m1], " POS ", %[lastc]\n" /* at or before this pixel's last contributor */    \
    "v_ldexp_f32 " A ", " A ", " B "\n"  /* G = expf(power) */
#define SGR_BWD_BODY(HEAD, X, Y, A, B, CZ, OP, R, G_, BL, POS, XY, OFF)                           \
    HEAD(X, Y, A, B, CZ, OP, POS)                                                                 \
    "v_mul_f32 " OP ", " OP ", " A "\n"                                                           \
    "v_min_f32 " OP ", 0x3f7d70a4, " OP "\n" /* alpha */                                          \
    "s_and_b64 %[m0], %[m0], %[m1]\n"                                                             \
    "v_cmp_ngt_f32 vcc, 0x3b808081, " OP "\n" /* !(alpha < 1/255) */                              \
    "s_and_b64 %[m0], %[m0], %[inside]\n"                                                         \
    "s_and_b64 exec, %[m0], vcc\n"                                                                \
    "v_sub_f32 " B ", 1.0, " OP "\n"                                                              \
    "v_rcp_f32 " B ", " B "\n"           /* 1 / (1 - alpha) */                                    \
    "v_mul_f32 " R ", " R ", %[g0]\n"                                                             \
    "v_fmac_f32 " R ", " G_ ", %[g1]\n"                                                           \
    "v_fmac_f32 " R ", " BL ", %[g2]\n"  /* colour . g */                                         \
    "v_mul_f32 %[T], %[T], " B "\n"                                                               \
    "v_mul_f32 " Y ", " OP ", %[T]\n"    /* Wt = alpha * T */                                     \
    "v_sub_f32 " CZ ", " R ", %[acc]\n"  /* (colour - accum_rec) . g */                           \
    "v_fmac_f32 %[acc], " OP ", " CZ "\n" /* accum_rec . g as the next contributing entry sees it */ \
    "v_mul_f32 " CZ ", " CZ ", %[T]\n"                                                            \
    "v_fmac_f32 " CZ ", %[ntb], " B "\n" /* dL_dalpha (backward.cu:523-529) */                    \
    "v_mul_f32 " X ", " A ", " CZ "\n"   /* Z = G * dL_dalpha */                                  \
    "s_mov_b64 exec, %[full]\n"                                                                   \
    "ds_write_b64 %[waddr], " XY OFF "\n"

// rows (1..BW_SUB) queue entries starting at LDS address e_addr -> panel rows 0..rows-1 at w_addr (+ 8 * lane already added)
#ifndef SGR_BLEND_CXX
#define SGR_BWD_WALK_ASM(HEAD)                                                                                                  \
        "s_mov_b64 %[full], exec\n"                                                                                             \
        "s_waitcnt lgkmcnt(0)\n"                                                                                                \
        "ds_read_b128 v[64:67], %[eaddr]\n"                                                                                     \
        "ds_read_b128 v[68:71], %[eaddr] offset:16\n"                                                                           \
        "ds_read_b64 v[72:73], %[eaddr] offset:32\n"                                                                            \
        "1:\n"                                                                                                                  \
        "ds_read_b128 v[74:77], %[eaddr] offset:48\n"                                                                           \
        "ds_read_b128 v[78:81], %[eaddr] offset:64\n"                                                                           \
        "ds_read_b64 v[82:83], %[eaddr] offset:80\n"                                                                            \
        "s_waitcnt lgkmcnt(3)\n"                                                                                                \
        SGR_BWD_BODY(HEAD, "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v[64:65]", "")                \
        "s_add_i32 %[n], %[n], -1\n"                                                                                            \
        "s_cmp_eq_u32 %[n], 0\n"                                                                                                \
        "s_cbranch_scc1 3f\n"                                                                                                   \
        "ds_read_b128 v[64:67], %[eaddr] offset:96\n"                                                                           \
        "ds_read_b128 v[68:71], %[eaddr] offset:112\n"                                                                          \
        "ds_read_b64 v[72:73], %[eaddr] offset:128\n"                                                                           \
        "v_add_u32 %[eaddr], 96, %[eaddr]\n"                                                                                    \
        "s_waitcnt lgkmcnt(4)\n"  /* the panel write of the previous entry may still be counted */                              \
        SGR_BWD_BODY(HEAD, "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v[74:75]", " offset:520")      \
        "v_add_u32 %[waddr], 1040, %[waddr]\n"                                                                                  \
        "s_add_i32 %[n], %[n], -1\n"                                                                                            \
        "s_cmp_eq_u32 %[n], 0\n"                                                                                                \
        "s_cbranch_scc0 1b\n"                                                                                                   \
        "3:\n"                                                                                                                  \
        "s_waitcnt lgkmcnt(0)\n"
#define SGR_BWD_WALK_OPERANDS                                                                                                   \
        : [T] "+v"(T), [acc] "+v"(acc_g), [eaddr] "+v"(e_addr), [waddr] "+v"(w_addr), [n] "+s"(rows), [full] "=&s"(full),        \
          [m0] "=&s"(m0), [m1] "=&s"(m1)                                                                                        \
        : [px] "v"(pixfx), [py] "v"(pixfy), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [ntb] "v"(ntb), [lastc] "v"(lastc),        \
          [inside] "s"(inside_mask), [chi] "s"(LOG2E)                                                                           \
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",              \
          "v79", "v80", "v81", "v82", "v83", "vcc", "scc", "memory"
template <bool EXACT>
__device__ __forceinline__ void bwd_phase_a(uint32_t e_addr, uint32_t w_addr, int rows, unsigned long long inside_mask, float pixfx,
                                            float pixfy, float g0, float g1, float g2, float ntb, uint32_t lastc, float& T,
                                            float& acc_g)
{
    unsigned long long full, m0, m1;
    if constexpr (EXACT) asm volatile(SGR_BWD_WALK_ASM(SGR_BWD_HEAD_X) SGR_BWD_WALK_OPERANDS);
    else asm volatile(SGR_BWD_WALK_ASM(SGR_BWD_HEAD) SGR_BWD_WALK_OPERANDS);
}
#endif

#ifdef SGR_BLEND_CXX
struct XBwd { float accum_rec[3] = {0.f, 0.f, 0.f}, last_alpha = 0.f, last_color[3] = {0.f, 0.f, 0.f}; };
#pragma clang fp contract(off)
__device__ __forceinline__ void bwd_phase_a_cxx(const float* q, float2* zw, int rows, unsigned long long inside_mask, float pixfx, float pixfy,
                                                float g0, float g1, float g2, float ntb, uint32_t lastc, float& T, float& acc_g, XBwd& st,
                                                const float* bg, float T_final)
{
    const int lane = threadIdx.x;
    const bool inside = (inside_mask >> lane) & 1ull;
    for (int r = 0; r < rows; r++) {
        const float* e = q + r * BW_ENTRY_DW;
        const float dx = e[0] - pixfx, dy = e[1] - pixfy;
        float ps;
        const float G = x_power_G(e, dx, dy, ps);
        const float alpha = fminf(0.99f, e[5] * G);
        float Z = 0.f, Wt = 0.f;
        if (!(ps > 0.f) && __float_as_uint(e[9]) <= lastc && inside && !(alpha < 1.0f / 255.0f)) {
#if SGR_X(16)
            T = T / (1.f - alpha);
            const float B = 1.f / (1.f - alpha);
#else
            const float B = __builtin_amdgcn_rcpf(1.f - alpha);
            T = T * B;
#endif
            Wt = alpha * T;
            float dLda;
#if SGR_X(32)
            const float c[3] = {e[6], e[7], e[8]}, g[3] = {g0, g1, g2};
            dLda = 0.f;
            for (int ch = 0; ch < 3; ch++) {
                st.accum_rec[ch] = st.last_alpha * st.last_color[ch] + (1.f - st.last_alpha) * st.accum_rec[ch];
                st.last_color[ch] = c[ch];
                dLda += (c[ch] - st.accum_rec[ch]) * g[ch];
            }
            dLda *= T;
            st.last_alpha = alpha;
            float bg_dot = 0.f;
            for (int i = 0; i < 3; i++) bg_dot += bg[i] * g[i];
            dLda += (-T_final / (1.f - alpha)) * bg_dot;
            (void)B; (void)ntb;
#else
            const float cg = __builtin_fmaf(e[8], g2, __builtin_fmaf(e[7], g1, e[6] * g0));
            float d = cg - acc_g;
            acc_g = __builtin_fmaf(alpha, d, acc_g);
            d = d * T;
            dLda = __builtin_fmaf(ntb, B, d);
#endif
            Z = G * dLda;
        }
        zw[r * BW_ZW_STRIDE] = make_float2(Z, Wt);
    }
}
#pragma clang fp contract(fast)
#endif

template <bool EXACT>
__device__ __forceinline__ void
blend_bwd_body(int W, int H, int gx, int T_tiles, const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ point_list,
              const char* __restrict__ binning, const uint32_t* __restrict__ blk_nb, const GeomRec* __restrict__ rec,
              const float* __restrict__ bg, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
              const float* __restrict__ dL_dpix, float* __restrict__ acc, const uint32_t* __restrict__ tile_order,
              const uint32_t* __restrict__ header, uint32_t list_cap)
{
    __shared__ __attribute__((aligned(16))) float s_q[(BW_QCAP + 1) * BW_ENTRY_DW];  // (+1: phase A's look-ahead)
    if (SGR_FORWARD_INVALID(header, list_cap)) return;  // the forward was a no-op (blk_nb, masks, lists are not there)
    __shared__ float2 s_zw[BW_SUB * BW_ZW_STRIDE];
    const int wg = blockIdx.x;
    int slot, sub;
    sgr_slot_of_workgroup(wg, slot, sub);
    if (slot >= T_tiles) return;
    const int tile = tile_order ? (int)tile_order[slot] : slot;  // deepest tiles first (k_tile_order)
    const int nb = (int)blk_nb[4 * tile + sub];  // batches the forward walked for this block (0: nothing contributed)
    if (nb == 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int bx0 = tx * SGR_TILE_X + 8 * (sub & 1), by0 = ty * SGR_TILE_Y + 8 * (sub >> 1);
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const unsigned long long inside_mask = __ballot(inside);
    const float pixfx = (float)px, pixfy = (float)py;
    const uint32_t r0 = tile_start[tile];
    const int total = (int)(tile_start[tile + 1] - r0);
    // (the survivor masks sit behind the point list as the FORWARD laid it out: its capacity is in the header, not the caller's R)
    const unsigned long long* blk_mask = reinterpret_cast<const unsigned long long*>(binning + SGR_BIN_MASK_OFFSET(header[SGR_HDR_LAYOUT_CAP]));
    const unsigned long long* my_mask = blk_mask + 4 * ((size_t)(r0 >> 6) + (size_t)tile) + sub;
    const size_t pix_id = (size_t)W * py + px;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_Ts[pix_id] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
    // entries behind the block's deepest contributor are inactive for every pixel: they are dropped at staging
    uint32_t blk_lc = last_contributor;
    for (int o = 32; o > 0; o >>= 1) blk_lc = max(blk_lc, (uint32_t)__shfl_xor((int)blk_lc, o));
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) { g0 = dL_dpix[pix_id]; g1 = dL_dpix[HW + pix_id]; g2 = dL_dpix[2 * HW + pix_id]; }
    const float ntb = -T_final * (bg[0] * g0 + bg[1] * g1 + bg[2] * g2);
    float acc_g = 0.f;

    // phase-B role: panel row bg_ (Gaussian), pixel row bq of the block (pixels 8 bq .. 8 bq + 7)
    const int bg_ = lane & 7, bq = lane >> 3;
    constexpr int BPIX = 8;
    float rg0[BPIX], rg1[BPIX], rg2[BPIX];
    {
        float* gp = reinterpret_cast<float*>(s_zw);
        gp[lane] = g0; gp[64 + lane] = g1; gp[128 + lane] = g2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < BPIX; i++) { rg0[i] = gp[BPIX * bq + i]; rg1[i] = gp[64 + BPIX * bq + i]; rg2[i] = gp[128 + BPIX * bq + i]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t q_lds = (uint32_t)(uintptr_t)s_q;
    const uint32_t zw_lds = (uint32_t)(uintptr_t)s_zw + 8u * (uint32_t)lane;

#ifdef SGR_BLEND_CXX
    XBwd xst;
#endif
    int qn = 0;  // entries waiting in the queue (they sit at its front)
    // the queued entries in groups of BW_SUB (all of them at the end, full groups only before), the rest moves to the front
    auto drain = [&](const bool last_batch) {
        int qs = 0;
        while (qn - qs >= BW_SUB || (last_batch && qn > qs)) {
            const int rows = min(BW_SUB, qn - qs);
#ifdef SGR_BLEND_CXX
            bwd_phase_a_cxx(s_q + qs * BW_ENTRY_DW, s_zw + lane, rows, inside_mask, pixfx, pixfy, g0, g1, g2, ntb, last_contributor, T, acc_g,
                            xst, bg, T_final);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#else
            bwd_phase_a<EXACT>(q_lds + (uint32_t)qs * (BW_ENTRY_DW * 4), zw_lds, rows, inside_mask, pixfx, pixfy, g0, g1, g2, ntb,
                               last_contributor, T, acc_g);
#endif
            // ---------------- phase B: lane = (panel row bg_, pixel rows 2 bq and 2 bq + 1)
            if (bg_ < rows) {
                const float* e = s_q + (qs + bg_) * BW_ENTRY_DW;
                const float2* row = s_zw + bg_ * BW_ZW_STRIDE + BPIX * bq;
                // Moments of Z about the PIXEL NEAREST THE GAUSSIAN'S CENTRE (xc, yc), not about the block origin: with the origin up
                // to eight pixels from the centre, the second moments of a splat a pixel wide were differences of numbers a hundred
                // times their size (x^2 S0 - 2 x Sx + Sxx with x ~ 8 against dx^2 ~ 0.3) and lost two digits -- what BASELINE config 4's
                // flat, sub-pixel Gaussians exposed (scale gradients 6e-5 off the reference, whose own float-atomic spread there is
                // 5e-8).  xb - xc is exact, and every product below is of the size of the moment it contributes to.
                const float xb = e[0] - (float)bx0, yb = e[1] - (float)by0;
                const float xc = rintf(xb), yc = rintf(yb);
#if defined(SGR_BLEND_CXX) && ((SGR_BLEND_CXX) & 64)
                double s0 = 0., sx = 0., sxx = 0., k0 = 0., k1 = 0., k2 = 0.;   // (analysis build)
#else
                float s0 = 0.f, sx = 0.f, sxx = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f;
#endif
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float2 v = row[i];
                    const float t = (float)i - xc;
#if defined(SGR_BLEND_CXX) && ((SGR_BLEND_CXX) & 64)
                    const double u = (double)v.x * t;
#else
                    const float u = v.x * t;
#endif
                    s0 += v.x; sx += u; sxx += u * t;
                    k0 += v.y * rg0[i]; k1 += v.y * rg1[i]; k2 += v.y * rg2[i];
                }
                // this lane's share of the moments (its pixels all have y = bq), shifted from (xc, yc) to the Gaussian's centre,
                // d = centre - pixel = (xb - x, yb - y) -- the shift is linear in the moments, so it is applied to the shares and the
                // shifted shares are summed over the eight lanes of the Gaussian
#if defined(SGR_BLEND_CXX) && ((SGR_BLEND_CXX) & 64)
                typedef double mom_t;
#else
                typedef float mom_t;
#endif
                const mom_t y0 = (float)bq - yc;
                const mom_t sy = y0 * s0, sxy = y0 * sx, syy = y0 * sy;
                const mom_t xr = xb - xc, yr = yb - yc;
                const mom_t dxs = xr * s0 - sx, dys = yr * s0 - sy;
                const mom_t dxx = xr * (xr * s0 - 2.f * sx) + sxx;
                const mom_t dyy = yr * (yr * s0 - 2.f * sy) + syy;
                const mom_t dxy = xr * dys - yr * sx + sxy;  // xr yr S0 - xr Sy - yr Sx + Sxy
                float o[9] = {(float)k0, (float)k1, (float)k2, (float)s0, (float)dxs, (float)dys, (float)dxx, (float)dxy, (float)dyy};
#pragma unroll
                for (int v = 0; v < 9; v++) o[v] += pair_in_row(o[v]);  // lanes bq and bq ^ 1
                // ... and over the four 16-lane rows, two values per lane swap: row r of q0 ends up with the total of o[r], row r
                // of q1 with that of o[4 + r], row 0 of q2 with that of o[8]
                const float q0 = pairsum32(pairsum16(o[0], o[1]), pairsum16(o[2], o[3]));
                const float q1 = pairsum32(pairsum16(o[4], o[5]), pairsum16(o[6], o[7]));
                const float q2 = pairsum32(pairsum16(o[8], 0.f), 0.f);
                // the nine sums of a pair are handed to nine neighbouring lanes through LDS (the panel is free now), so that one
                // atomic instruction carries whole 36-byte records (four Gaussians at a time) and the memory system sees ONE
                // request per (block, Gaussian) pair instead of nine
                if ((lane & 8) == 0) {
                    float* tb = reinterpret_cast<float*>(s_zw) + bg_ * 16;
                    const int r = lane >> 4;
                    tb[r] = q0;
                    tb[4 + r] = q1;
                    if (r == 0) { tb[8] = q2; tb[9] = e[10]; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                const float* tbl = reinterpret_cast<const float*>(s_zw);
                const int c = lane & 15;
#pragma unroll
                for (int pass = 0; pass < BW_SUB / 4; pass++) {
                    const int g = 4 * pass + (lane >> 4);
                    if (g < rows && c < 9) {
                        const float val = tbl[g * 16 + c];
                        if (val != 0.f) atomicAdd(acc + (size_t)__float_as_uint(tbl[g * 16 + 9]) * SGR_ACC_STRIDE + c, val);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            qs += rows;
        }
        const int rem = qn - qs;
        if (qs > 0 && rem > 0) {
            float4 t0v, t1v, t2v;
            if (lane < rem) {
                const float4* src = reinterpret_cast<const float4*>(s_q + (qs + lane) * BW_ENTRY_DW);
                t0v = src[0]; t1v = src[1]; t2v = src[2];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < rem) {
                float4* dstq = reinterpret_cast<float4*>(s_q + lane * BW_ENTRY_DW);
                dstq[0] = t0v; dstq[1] = t1v; dstq[2] = t2v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        qn = rem;
    };
    // The forward's batches back to front: batch b holds list entries 64 b .. 64 b + 63, of which the lanes in the forward's
    // mask survived the block's cull; inside a batch the walk goes from lane 63 down.  Software pipeline: the ids of batch
    // b - 1 and the records of batch b are ISSUED, the groups already queued are processed while those loads travel, and only
    // then are the records staged.  (Plain compiler-scheduled loads, issued and consumed inside one iteration, unconditional
    // with clamped addresses: see k_blend_fwd_w.  Lanes outside the mask re-read record 0 of the list: one cached line.)
    // (round 5: masks and ids run TWO batches ahead, as in the forward: with one batch of lead a tile whose batches hold few
    // survivors paid two dependent round trips per batch, ids then records)
    unsigned long long m_next = my_mask[4 * (size_t)(nb - 1)];
    uint32_t id_next = point_list[r0 + (uint32_t)min(64 * (nb - 1) + lane, total - 1)];
    unsigned long long m_next2 = my_mask[4 * (size_t)max(nb - 2, 0)];
    uint32_t id_next2 = point_list[r0 + (uint32_t)min(64 * max(nb - 2, 0) + lane, total - 1)];
    for (int b = nb - 1; b >= 0; b--) {
        const unsigned long long m_all = m_next;
        const uint32_t pos = (uint32_t)(64 * b + lane + 1);  // 1-based list position of this lane's entry
        const bool take = ((m_all >> lane) & 1ull) && pos <= blk_lc;
        const uint32_t id_cur = take ? id_next : point_list[r0];
        const float4* rp = reinterpret_cast<const float4*>(rec + id_cur);
        const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2];
        m_next = m_next2;
        id_next = id_next2;
        m_next2 = my_mask[4 * (size_t)max(b - 2, 0)];
        id_next2 = point_list[r0 + (uint32_t)min(64 * max(b - 2, 0) + lane, total - 1)];
        drain(false);
        const unsigned long long m = __ballot(take);
        if (take) {
            // back to front: rank = taken lanes ABOVE this one
            const uint32_t slot = (uint32_t)qn + (uint32_t)__popcll(lane == 63 ? 0ull : (m >> (lane + 1)));
            float4* e = reinterpret_cast<float4*>(s_q + slot * BW_ENTRY_DW);
#if defined(SGR_BLEND_CXX) && ((SGR_BLEND_CXX) & 1)
            e[0] = make_float4(v0.x, v0.y, v0.z, v0.w);
            e[1] = make_float4(v1.x, v1.y, v2.x, v2.y);
#else
            if (EXACT) {
                e[0] = make_float4(v0.x, v0.y, -0.5f * v0.z, v0.w);
                e[1] = make_float4(-0.5f * v1.x, v1.y, v2.x, v2.y);
            } else {
                e[0] = make_float4(v0.x, v0.y, -0.5f * LOG2E * v0.z, -LOG2E * v0.w);
                e[1] = make_float4(-0.5f * LOG2E * v1.x, v1.y, v2.x, v2.y);
            }
#endif
            e[2] = make_float4(v2.z, __uint_as_float(pos), __uint_as_float(id_cur), 0.f);
        }
        qn += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    drain(true);
}

#define SGR_BWD_PARAMS                                                                                                              \
    int W, int H, int gx, int T_tiles, const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,                    \
        const char *__restrict__ binning, const uint32_t *__restrict__ blk_nb, const GeomRec *__restrict__ rec, const float *__restrict__ bg, \
        const float *__restrict__ final_Ts, const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,                      \
        float *__restrict__ acc, const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ header, uint32_t list_cap
#define SGR_BWD_ARGS W, H, gx, T_tiles, tile_start, point_list, binning, blk_nb, rec, bg, final_Ts, n_contrib, dL_dpix, acc, tile_order, header, list_cap
#ifdef SGR_BWD_WAVES   // (A/B build option: force the waves per SIMD the register allocation aims at)
#define SGR_BWD_OCC __attribute__((amdgpu_waves_per_eu(SGR_BWD_WAVES, SGR_BWD_WAVES)))
#else
#define SGR_BWD_OCC
#endif
__global__ void __launch_bounds__(64) SGR_BWD_OCC k_blend_bwd_w(SGR_BWD_PARAMS) { blend_bwd_body<false>(SGR_BWD_ARGS); }
__global__ void __launch_bounds__(64) SGR_BWD_OCC k_blend_bwd_wx(SGR_BWD_PARAMS) { blend_bwd_body<true>(SGR_BWD_ARGS); }  // exact alpha (SGR_BWD_HEAD_X)

// Launch order of the backward: tiles by how deep the forward walked them (tile_maxc), deepest first, so that the waves still
// running when the grid drains are the short ones.  (Workgroups start in index order; with ~2.5 dispatch rounds of waves whose
// lifetimes spread over an order of magnitude, raster order leaves a quarter of the chip idle at the end.)  One workgroup:
// the job of tile_order.h (behind a forward that keeps the order it also writes the walk hint and the host's header copy;
// in the train step the same job rides in the loss kernel instead, csrc/loss.hip).
__global__ void __launch_bounds__(1024) k_tile_order(SgrTileOrderJob job) { sgr_tile_order_block<1024>(job); }

// walk hint for the next visit of this camera: what the tile walked now, plus a margin, plus one batch
__global__ void __launch_bounds__(256) k_make_hint(int T, const uint32_t* __restrict__ tile_walked, const uint32_t* __restrict__ header,
                                                   uint32_t list_cap, float margin, uint32_t* __restrict__ need_out,
                                                   uint32_t* __restrict__ header_host)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    // the second header copy for the host (word 3, the hint-miss flag, is final now that the blend kernel is done)
    if (header_host && t < 8) header_host[8 + t] = header[t];
    if (!need_out || t >= T || SGR_FORWARD_INVALID(header, list_cap)) return;  // (an invalid forward leaves the previous hint in place)
    const uint32_t w = tile_walked[t];
    need_out[t] = w + (uint32_t)((float)w * margin) + 64u;
}

}  // namespace

void sgr_launch_blend_fwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const GeomRec* rec, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc,
                          uint32_t* tile_walked, float* out_color, unsigned long long* blk_mask, uint32_t* blk_nb,
                          uint32_t* header, uint32_t list_cap, const uint32_t* tile_need, const uint32_t* launch_order, hipStream_t s,
                          uint32_t* repair_flag, uint32_t* repair_list, int exact, uint32_t deep_min)
{
    const int T = gx * gy;  // (tile_maxc and tile_walked were zeroed by the tile scan: the blocks of a tile combine with atomicMax)
    hipLaunchKernelGGL(exact ? k_blend_fwd_wx : k_blend_fwd_w, dim3(sgr_blend_grid(T)), dim3(64), 0, s, W, H, gx, T, tile_start, point_list, rec, bg, final_T,
                       n_contrib, tile_maxc, tile_walked, out_color, blk_mask, blk_nb, header, list_cap, tile_need, launch_order,
                       tile_need ? repair_flag : nullptr, repair_list, tile_need ? deep_min : 0u);
}

// the list of tiles whose hinted list is longer than deep_min (on the caller's stream, before the fork: as the first kernel of the
// side stream it waited 80 us for a slot next to the one-wave kernel)
void sgr_launch_deep_list(int gx, int gy, const uint32_t* tile_start, const uint32_t* tile_need, uint32_t deep_min, uint32_t* header,
                          uint32_t list_cap, uint32_t* deep_list, hipStream_t s)
{
    const int T = gx * gy;
    hipLaunchKernelGGL(k_deep_list, dim3((T + 255) / 256), dim3(256), 0, s, T, tile_start, tile_need, deep_min, header, list_cap, deep_list);
}

// the blocks of tiles whose hinted list is longer than deep_min, eight waves per block (the kernel above skips exactly those)
void sgr_launch_blend_fwd_deep(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list, const GeomRec* rec,
                               const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc, uint32_t* tile_walked,
                               float* out_color, unsigned long long* blk_mask, uint32_t* blk_nb, uint32_t* header, uint32_t list_cap,
                               const uint32_t* tile_need, uint32_t* deep_list, uint32_t deep_min, hipStream_t s, uint32_t* repair_flag,
                               uint32_t* repair_list, int exact)
{
    const int T = gx * gy;
    const size_t lds = (size_t)DEEP_WAVES * DEEP_ROWS * 64 * 4 + (size_t)DEEP_WAVES * DEEP_ROWS * 16 + 64;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_blend_fwd_deep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_blend_fwd_deep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    if (exact)
        hipLaunchKernelGGL(k_blend_fwd_deep<true>, dim3(256), dim3(64 * DEEP_WAVES), lds, s, W, H, gx, T, tile_start, point_list, rec, bg, final_T,
                           n_contrib, tile_maxc, tile_walked, out_color, blk_mask, blk_nb, header, list_cap, tile_need, deep_list, repair_flag,
                           repair_list);
    else
        hipLaunchKernelGGL(k_blend_fwd_deep<false>, dim3(256), dim3(64 * DEEP_WAVES), lds, s, W, H, gx, T, tile_start, point_list, rec, bg, final_T,
                           n_contrib, tile_maxc, tile_walked, out_color, blk_mask, blk_nb, header, list_cap, tile_need, deep_list, repair_flag,
                           repair_list);
}

void sgr_launch_blend_fwd_repair(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                                 const GeomRec* rec, const float* bg, float* final_T, uint32_t* n_contrib, uint32_t* tile_maxc,
                                 uint32_t* tile_walked, float* out_color, unsigned long long* blk_mask, uint32_t* blk_nb,
                                 uint32_t* header, uint32_t list_cap, const uint32_t* repair_list, hipStream_t s, int exact)
{
    const int T = gx * gy;
    const int cover = T < SGR_REPAIR_TILES ? T : SGR_REPAIR_TILES;
    // (T_tiles stays the tile count: it clamps the list's entries; the slots beyond the listed tiles leave at once)
    hipLaunchKernelGGL(exact ? k_blend_fwd_repairx : k_blend_fwd_repair, dim3(sgr_blend_grid(cover)), dim3(64), 0, s, W, H, gx, T, tile_start, point_list, rec, bg,
                       final_T, n_contrib, tile_maxc, tile_walked, out_color, blk_mask, blk_nb, header, list_cap, repair_list);
}

// behind the blend: this view's launch order (for its backward: order_scratch; for the camera's next forward: order_out), the walk
// hint for the camera's next visit and the second header copy
void sgr_launch_blend_fwd_post(int gx, int gy, const uint32_t* tile_maxc, const uint32_t* tile_walked, uint32_t* header, uint32_t list_cap,
                               uint32_t* tile_need_out, float hint_margin, uint32_t* header_host_dev, uint32_t* order_scratch,
                               uint32_t* order_out, hipStream_t s)
{
    const int T = gx * gy;
    if (order_out) {
        SgrTileOrderJob job = {T, list_cap, tile_maxc, tile_need_out ? tile_walked : nullptr, header, order_scratch, order_out, tile_need_out,
                               header_host_dev, hint_margin > 0.f ? hint_margin : 0.25f, gx, gy};
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, job);
        return;
    }
    if (tile_need_out || header_host_dev)
        hipLaunchKernelGGL(k_make_hint, dim3(tile_need_out ? (T + 255) / 256 : 1), dim3(256), 0, s, T, tile_walked, header, list_cap,
                           hint_margin > 0.f ? hint_margin : 0.25f, tile_need_out, header_host_dev);
}

void sgr_launch_blend_bwd(int W, int H, int gx, int gy, const uint32_t* tile_start, const uint32_t* point_list,
                          const char* binning, const uint32_t* blk_nb, const GeomRec* rec, const float* bg,
                          const float* final_T, const uint32_t* n_contrib, const float* dL_dpix, float* acc,
                          const uint32_t* tile_maxc, const uint32_t* header, uint32_t list_cap, uint32_t* tile_order, int order_ready,
                          hipStream_t s, int exact)
{
    const int T = gx * gy;
    if (order_ready) {}                           // (the forward sorted: sgr_forward_opts.tile_order_out)
    else if (4 * T < 8192) tile_order = nullptr;  // (fewer waves than the chip holds at once: nothing to order)
    else {
        SgrTileOrderJob job = {T, list_cap, tile_maxc, nullptr, header, tile_order, nullptr, nullptr, nullptr, 0.f, gx, gy};
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, job);
    }
    hipLaunchKernelGGL(exact ? k_blend_bwd_wx : k_blend_bwd_w, dim3(sgr_blend_grid(T)), dim3(64), 0, s, W, H, gx, T, tile_start, point_list, binning, blk_nb, rec,
                       bg, final_T, n_contrib, dL_dpix, acc, tile_order, header, list_cap);
}
