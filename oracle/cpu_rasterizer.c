/*
 * oracle/cpu_rasterizer.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C (gcc) CPU restatement of the reference's differentiable tile rasterizer
 * (INRIA diff-gaussian-rasterization as vendored by Anttwo/SuGaR).  It is the parity
 * oracle for the HIP path in sugar_amd/csrc and the `cpu_baseline` leg of bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it; the
 * product never routes through it.
 *
 * Every function cites the reference lines it follows.  Paths are relative to
 *   DGR = /root/reference/gaussian_splatting/submodules/diff-gaussian-rasterization
 *
 * Arithmetic contract (what "bit-exact" means between this file and the HIP kernels):
 *   - float32 everywhere the reference uses float, double in ndc2Pix (DGR/cuda_rasterizer/auxiliary.h:41-44);
 *   - every +,-,*,/,sqrt is an individually rounded IEEE-754 operation in the written
 *     (glm / C left-to-right) order: this file MUST be compiled with -ffp-contract=off and
 *     without -ffast-math (see oracle/Makefile); the HIP preprocess kernels are compiled the
 *     same way, so radii / tile rectangles / depth keys / sorted lists agree bit for bit;
 *   - the blend stage uses expf(); its HIP counterpart uses the hardware exp2 path, so the
 *     blend outputs are compared with a tolerance (tests/), not bit-exactly.
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or CPU rasterizer
 * (SURVEY.md section 4 / 8c).  This oracle is pinned by (i) golden vectors generated from the
 * reference's own Python helpers (eval_sh, build_covariance, getProjectionMatrix) under
 * tests/golden/, (ii) an independent PyTorch-autograd restatement (oracle/torch_cpu_rasterizer.py)
 * plus float64 finite differences for the analytic backward, and (iii) on the GPU box, the
 * reference's own .cu sources compiled by hipcc into oracle/_ref (see oracle/ref_build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* DGR/cuda_rasterizer/config.h:16 */
#define BLOCK_Y 16 /* DGR/cuda_rasterizer/config.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NCH 3 /* NUM_CHANNELS, DGR/cuda_rasterizer/config.h:15 */

/* DGR/cuda_rasterizer/auxiliary.h:21-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct { float x, y, z; } f3;

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* float -> int conversion with the GPU's semantics (saturating, NaN -> 0); a C cast is UB
 * out of range and x86 cvttss2si returns INT_MIN for both overflow directions. */
static inline int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

/* DGR/cuda_rasterizer/auxiliary.h:41-44 : the 1.0 / 0.5 literals make this double arithmetic */
static inline float ndc2Pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

/* DGR/cuda_rasterizer/auxiliary.h:46-56 */
static inline void getRect(float px, float py, int max_radius, int gx, int gy,
                           int* minx, int* miny, int* maxx, int* maxy)
{
    const float r = (float)max_radius;
    *minx = imin(gx, imax(0, f2i_sat((px - r) / (float)BLOCK_X)));
    *miny = imin(gy, imax(0, f2i_sat((py - r) / (float)BLOCK_Y)));
    *maxx = imin(gx, imax(0, f2i_sat((px + r + (float)BLOCK_X - 1.0f) / (float)BLOCK_X)));
    *maxy = imin(gy, imax(0, f2i_sat((py + r + (float)BLOCK_Y - 1.0f) / (float)BLOCK_Y)));
}

/* DGR/cuda_rasterizer/auxiliary.h:58-66 */
static inline f3 transformPoint4x3(f3 p, const float* m)
{
    f3 t;
    t.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    t.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    t.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return t;
}

/* DGR/cuda_rasterizer/auxiliary.h:68-77 */
static inline void transformPoint4x4(f3 p, const float* m, float out[4])
{
    out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}

/* DGR/cuda_rasterizer/auxiliary.h:89-97 */
static inline f3 transformVec4x3Transpose(f3 p, const float* m)
{
    f3 t;
    t.x = m[0] * p.x + m[1] * p.y + m[2] * p.z;
    t.y = m[4] * p.x + m[5] * p.y + m[6] * p.z;
    t.z = m[8] * p.x + m[9] * p.y + m[10] * p.z;
    return t;
}

/* DGR/cuda_rasterizer/auxiliary.h:107-117 */
static inline f3 dnormvdv(f3 v, f3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    f3 o;
    o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return o;
}

/* DGR/cuda_rasterizer/auxiliary.h:139-164 (prefiltered trap omitted: it aborts the process) */
static inline int in_frustum(int idx, const float* orig_points, const float* viewmatrix, f3* p_view)
{
    f3 p = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
    *p_view = transformPoint4x3(p, viewmatrix);
    if (p_view->z <= 0.2f) return 0;
    return 1;
}

/* DGR/cuda_rasterizer/rasterizer_impl.cu:54-66, 141-153 */
void orc_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present)
{
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        f3 pv;
        present[i] = (uint8_t)in_frustum(i, means3D, viewmatrix, &pv);
    }
}

/* DGR/cuda_rasterizer/forward.cu:20-71.  glm::vec3 arithmetic is component-wise, scalars are
 * combined first (SH_C1 * y is a float product, then scales the vec3). */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped, float out[3])
{
    f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
    /* glm::length = sqrt(dot); glm::dot(vec3) = (x*x + y*y) + z*z */
    float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float x = dir.x, y = dir.y, z = dir.z;
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[3 * (k) + c]
        float r = SH_C0 * SH(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                    SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                    SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        clamped[3 * idx + c] = (uint8_t)(r < 0);
        out[c] = fmaxf_(r, 0.0f);
    }
}

/* glm quaternion->matrix as written in DGR/cuda_rasterizer/forward.cu:134-138 and backward.cu:287-291.
 * R[c][k] = glm column c, row k (so R is the transpose of the usual rotation matrix). */
static inline void quat_to_glmR(const float* rot, float R[3][3])
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* DGR/cuda_rasterizer/forward.cu:118-152.  M = S*R (glm) -> M[c][k] = s_k * R[c][k];
 * Sigma = transpose(M)*M -> Sigma[c][r] = M[r][0]*M[c][0] + M[r][1]*M[c][1] + M[r][2]*M[c][2]. */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float R[3][3], M[3][3];
    quat_to_glmR(rot, R);
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) M[c][k] = s[k] * R[c][k];
#define SIG(c, r) (M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2])
    cov3D[0] = SIG(0, 0); cov3D[1] = SIG(0, 1); cov3D[2] = SIG(0, 2);
    cov3D[3] = SIG(1, 1); cov3D[4] = SIG(1, 2); cov3D[5] = SIG(2, 2);
#undef SIG
}

/* The glm matrix T = W*J of DGR/cuda_rasterizer/forward.cu:89-99 (also backward.cu:178-192).
 * T[c][r] (glm column c, row r): T[0][r] = v[4r]*J00 + v[4r+2]*J02, T[1][r] = v[4r+1]*J11 + v[4r+2]*J12,
 * T[2][r] = 0.  Returns the clamped t as well. */
static inline void compute_T(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                             const float* v, float T[2][3], f3* t_out, float* txtz_out, float* tytz_out)
{
    f3 t = transformPoint4x3(mean, v);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf_(limx, fmaxf_(-limx, txtz)) * t.z;
    t.y = fminf_(limy, fmaxf_(-limy, tytz)) * t.z;
    float J00 = focal_x / t.z, J02 = -(focal_x * t.x) / (t.z * t.z);
    float J11 = focal_y / t.z, J12 = -(focal_y * t.y) / (t.z * t.z);
    for (int r = 0; r < 3; r++) {
        T[0][r] = v[4 * r + 0] * J00 + v[4 * r + 2] * J02;
        T[1][r] = v[4 * r + 1] * J11 + v[4 * r + 2] * J12;
    }
    *t_out = t; *txtz_out = txtz; *tytz_out = tytz;
}

/* cov = transpose(T) * transpose(Vrk) * T, DGR/cuda_rasterizer/forward.cu:101-112.
 * A = Tt*Vrk: A[c][r] = T[r][0]*V[c][0] + T[r][1]*V[c][1] + T[r][2]*V[c][2] (r < 2 needed);
 * cov[c][r] = A[0][r]*T[c][0] + A[1][r]*T[c][1] + A[2][r]*T[c][2]. */
static inline void cov2d_from_T(const float T[2][3], const float* c3, float* a, float* b, float* c)
{
    float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float A[3][2];
    for (int cc = 0; cc < 3; cc++)
        for (int r = 0; r < 2; r++) A[cc][r] = T[r][0] * V[cc][0] + T[r][1] * V[cc][1] + T[r][2] * V[cc][2];
    *a = A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2];
    *b = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
    *c = A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2];
}

/* DGR/cuda_rasterizer/forward.cu:155-256 (kernel K2) with focal from rasterizer_impl.cu:222-223.
 * Arrays follow GeometryState (rasterizer_impl.h:33-48): depths[P], clamped[3P], radii[P],
 * means2D[2P], cov3Ds[6P], conic_opacity[4P], rgb[3P], tiles_touched[P].
 * Null pointers mean "input absent" exactly as in the reference (forward.cu:205,241). */
void orc_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                    const float* rotations, const float* opacities, const float* shs, uint8_t* clamped,
                    const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy,
                    int* radii, float* means2D, float* depths, float* cov3Ds, float* rgb, float* conic_opacity,
                    uint32_t* tiles_touched)
{
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        f3 p_view;
        if (!in_frustum(idx, means3D, viewmatrix, &p_view)) continue;
        f3 p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj_x = p_hom[0] * p_w, p_proj_y = p_hom[1] * p_w;
        const float* cov3D;
        if (cov3D_precomp) {
            cov3D = cov3D_precomp + (size_t)idx * 6;
        } else {
            computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, cov3Ds + (size_t)idx * 6);
            cov3D = cov3Ds + (size_t)idx * 6;
        }
        float T[2][3], txtz, tytz; f3 t;
        compute_T(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, T, &t, &txtz, &tytz);
        float cx, cy, cz;
        cov2d_from_T(T, cov3D, &cx, &cy, &cz);
        cx += 0.3f; cz += 0.3f; /* forward.cu:110-111 */
        float det = (cx * cz - cy * cy);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};
        float mid = 0.5f * (cx + cz);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        float pix_x = ndc2Pix(p_proj_x, W), pix_y = ndc2Pix(p_proj_y, H);
        int minx, miny, maxx, maxy;
        getRect(pix_x, pix_y, f2i_sat(my_radius), gx, gy, &minx, &miny, &maxx, &maxy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;
        if (!colors_precomp) {
            float c[3];
            computeColorFromSH(idx, D, M, means3D, cam_pos, shs, clamped, c);
            rgb[idx * NCH + 0] = c[0]; rgb[idx * NCH + 1] = c[1]; rgb[idx * NCH + 2] = c[2];
        }
        depths[idx] = p_view.z;
        radii[idx] = f2i_sat(my_radius);
        means2D[2 * idx] = pix_x; means2D[2 * idx + 1] = pix_y;
        conic_opacity[4 * idx + 0] = conic[0]; conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2]; conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((maxy - miny) * (maxx - minx));
    }
}

/* Inclusive prefix sum of tiles_touched (cub::DeviceScan::InclusiveSum at
 * DGR/cuda_rasterizer/rasterizer_impl.cu:277); returns num_rendered (:280-281). */
int64_t orc_scan(int P, const uint32_t* tiles_touched, uint32_t* point_offsets)
{
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
    return P > 0 ? (int64_t)acc : 0;
}

/* DGR/cuda_rasterizer/rasterizer_impl.cu:35-50 */
uint32_t orc_getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* Stable LSD radix sort of (u64 key, u32 value) pairs on key bits [0, end_bit): the semantics of
 * cub::DeviceRadixSort::SortPairs as called at DGR/cuda_rasterizer/rasterizer_impl.cu:303-308. */
static void radix_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp, size_t n, int end_bit)
{
    uint64_t *ka = keys, *kb = keys_tmp; uint32_t *va = vals, *vb = vals_tmp;
    for (int shift = 0; shift < end_bit; shift += 8) {
        size_t hist[257]; memset(hist, 0, sizeof(hist));
        int nb = end_bit - shift < 8 ? end_bit - shift : 8;
        uint64_t mask = ((uint64_t)1 << nb) - 1;
        for (size_t i = 0; i < n; i++) hist[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (size_t i = 0; i < n; i++) { size_t dst = hist[(ka[i] >> shift) & mask]++; kb[dst] = ka[i]; vb[dst] = va[i]; }
        uint64_t* tk = ka; ka = kb; kb = tk; uint32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != keys) { memcpy(keys, ka, n * sizeof(uint64_t)); memcpy(vals, va, n * sizeof(uint32_t)); }
}

/* K4 duplicateWithKeys (DGR/cuda_rasterizer/rasterizer_impl.cu:70-111), K5 sort (:300-308),
 * memset + K6 identifyTileRanges (:116-138, :310-317).  Caller allocates keys[R], point_list[R],
 * ranges[2*T] (T = tiles). */
void orc_bin(int P, int W, int H, const int* radii, const float* means2D, const float* depths,
             const uint32_t* point_offsets, int64_t R, uint64_t* point_list_keys, uint32_t* point_list,
             uint32_t* ranges)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            int minx, miny, maxx, maxy;
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &minx, &miny, &maxx, &maxy);
            uint32_t dbits; memcpy(&dbits, &depths[idx], 4);
            for (int y = miny; y < maxy; y++)
                for (int x = minx; x < maxx; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    point_list_keys[off] = key;
                    point_list[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    if (R > 0) {
        uint64_t* kt = (uint64_t*)malloc((size_t)R * sizeof(uint64_t));
        uint32_t* vt = (uint32_t*)malloc((size_t)R * sizeof(uint32_t));
        int bit = (int)orc_getHigherMsb((uint32_t)(gx * gy));
        radix_sort_pairs(point_list_keys, point_list, kt, vt, (size_t)R, 32 + bit);
        free(kt); free(vt);
    }
    memset(ranges, 0, (size_t)gx * gy * 2 * sizeof(uint32_t));
    for (int64_t i = 0; i < R; i++) {
        uint32_t currtile = (uint32_t)(point_list_keys[i] >> 32);
        if (i == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(point_list_keys[i - 1] >> 32);
            if (currtile != prevtile) { ranges[2 * prevtile + 1] = (uint32_t)i; ranges[2 * currtile] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
    }
}

/* K7 renderCUDA forward, DGR/cuda_rasterizer/forward.cu:261-374.  One pixel at a time: the per-pixel
 * result does not depend on the 256-wide batching (a done pixel ignores everything after it).
 * Also returns the number of list entries fetched by the block (rounds*BLOCK_SIZE semantics of :306-311)
 * in walked[tile] when walked != NULL: the prefix length a block actually stages, used for R_f in the
 * roofline accounting. */
void orc_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                        const float* features, const float* conic_opacity, float* final_T, uint32_t* n_contrib,
                        const float* bg_color, float* out_color, uint32_t* walked)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        uint32_t maxpos = 0; /* furthest 1-based list position any pixel of the tile looked at */
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const int pix_id = W * py + px;
                const float pixfx = (float)px, pixfy = (float)py;
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float C[NCH] = {0, 0, 0};
                for (uint32_t i = r0; i < r1; i++) {
                    contributor++;
                    const uint32_t id = point_list[i];
                    float dx = means2D[2 * id] - pixfx, dy = means2D[2 * id + 1] - pixfy;
                    const float* con_o = conic_opacity + 4 * (size_t)id;
                    float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > 0.0f) continue;
                    float alpha = fminf_(0.99f, con_o[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done = true; nothing after it is looked at */
                    for (int ch = 0; ch < NCH; ch++) C[ch] += features[id * NCH + ch] * alpha * T;
                    T = test_T;
                    last_contributor = contributor;
                }
                if (contributor > maxpos) maxpos = contributor;
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < NCH; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg_color[ch];
            }
        if (walked) walked[tile] = maxpos;
    }
}

/* K8 renderCUDA backward, DGR/cuda_rasterizer/backward.cu:399-557.  Every per-(pixel, Gaussian) term is formed in float
 * exactly as the reference forms it.  The reference then adds the terms with float atomicAdd in a nondeterministic order,
 * so its result is one draw from a cloud of roundings (relative spread ~1e-4 for Gaussians hundreds of pixels wide, whose
 * terms cancel); the oracle's expected value is the centre of that cloud: the same float terms summed in double (order
 * no longer matters at the 1e-16 level, so the oracle is deterministic when threaded) and rounded to float once.
 * Outputs must be zero-initialised by the caller (DGR/rasterize_points.cu:151-159):
 * dL_dmean2D[P*3], dL_dconic2D[P*4], dL_dopacity[P], dL_dcolors[P*3]. */
void orc_render_backward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg_color,
                         const float* means2D, const float* conic_opacity, const float* colors,
                         const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels,
                         float* dL_dmean2D, float* dL_dconic2D, float* dL_dopacity, float* dL_dcolors)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const float ddelx_dx = (float)(0.5 * W); /* backward.cu:460-461 */
    const float ddely_dy = (float)(0.5 * H);
    /* double accumulators, 9 per Gaussian: {mean2D x, y, conic xx, xy, yy, opacity, colour r, g, b} */
    uint32_t R = 0; /* (an empty tile's range is {0, 0}: rasterizer_impl.cu:116-138 writes only where a tile starts or ends) */
    for (int t = 0; t < gx * gy; t++) if (ranges[2 * t + 1] > R) R = ranges[2 * t + 1];
    uint32_t max_id = 0;
    for (uint32_t i = 0; i < R; i++) if (point_list[i] > max_id) max_id = point_list[i];
    double* acc = (double*)calloc((size_t)(max_id + 1) * 9, sizeof(double));
    if (!acc) return;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const int pix_id = W * py + px;
                const float pixfx = (float)px, pixfy = (float)py;
                const float T_final = final_Ts[pix_id];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = n_contrib[pix_id];
                float accum_rec[NCH] = {0, 0, 0}, dL_dpixel[NCH], last_color[NCH] = {0, 0, 0};
                for (int i = 0; i < NCH; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
                float last_alpha = 0;
                for (uint32_t k = 0; k < r1 - r0; k++) {
                    const uint32_t id = point_list[r1 - k - 1];
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    float dx = means2D[2 * id] - pixfx, dy = means2D[2 * id + 1] - pixfy;
                    const float* con_o = conic_opacity + 4 * (size_t)id;
                    const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = fminf_(0.99f, con_o[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < NCH; ch++) {
                        const float c = colors[id * NCH + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
#pragma omp atomic
                        acc[9 * (size_t)id + 6 + ch] += (double)(dchannel_dcolor * dL_dchannel);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < NCH; i++) bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = con_o[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                    const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
                    double* a = acc + 9 * (size_t)id;
#pragma omp atomic
                    a[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
#pragma omp atomic
                    a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
#pragma omp atomic
                    a[2] += (double)(-0.5f * gdx * dx * dL_dG);
#pragma omp atomic
                    a[3] += (double)(-0.5f * gdx * dy * dL_dG);
#pragma omp atomic
                    a[4] += (double)(-0.5f * gdy * dy * dL_dG);
#pragma omp atomic
                    a[5] += (double)(G * dL_dalpha);
                }
            }
    }
    for (size_t id = 0; id <= max_id; id++) {
        const double* a = acc + 9 * id;
        dL_dmean2D[3 * id + 0] += (float)a[0];
        dL_dmean2D[3 * id + 1] += (float)a[1];
        dL_dconic2D[4 * id + 0] += (float)a[2];
        dL_dconic2D[4 * id + 1] += (float)a[3];
        dL_dconic2D[4 * id + 3] += (float)a[4];
        dL_dopacity[id] += (float)a[5];
        for (int ch = 0; ch < NCH; ch++) dL_dcolors[id * NCH + ch] += (float)a[6 + ch];
    }
    free(acc);
}

/* Backward of SH -> RGB, DGR/cuda_rasterizer/backward.cu:20-139 */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                   const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                   float* dL_dmeans, float* dL_dshs)
{
    f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f3 dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
    float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
    float dL_ddir[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[3 * (k) + c]
#define DSH(k) dL_dsh[3 * (k) + c]
        float dL_dRGB = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0.0f : 1.0f);
        float dRGBdx = 0, dRGBdy = 0, dRGBdz = 0;
        DSH(0) = SH_C0 * dL_dRGB;
        if (deg > 0) {
            DSH(1) = (-SH_C1 * y) * dL_dRGB; DSH(2) = (SH_C1 * z) * dL_dRGB; DSH(3) = (-SH_C1 * x) * dL_dRGB;
            dRGBdx = -SH_C1 * SH(3); dRGBdy = -SH_C1 * SH(1); dRGBdz = SH_C1 * SH(2);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                DSH(4) = (SH_C2[0] * xy) * dL_dRGB; DSH(5) = (SH_C2[1] * yz) * dL_dRGB;
                DSH(6) = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB; DSH(7) = (SH_C2[3] * xz) * dL_dRGB;
                DSH(8) = (SH_C2[4] * (xx - yy)) * dL_dRGB;
                dRGBdx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                dRGBdy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                dRGBdz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
                if (deg > 2) {
                    DSH(9) = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB; DSH(10) = (SH_C3[1] * xy * z) * dL_dRGB;
                    DSH(11) = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
                    DSH(12) = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
                    DSH(13) = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
                    DSH(14) = (SH_C3[5] * z * (xx - yy)) * dL_dRGB; DSH(15) = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
                    dRGBdx += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                               SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                               SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    dRGBdy += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                               SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                               SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    dRGBdz += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                               SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                               SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
#undef SH
#undef DSH
        /* glm::dot(dRGBdx, dL_dRGB) sums the three channels x,y,z in order (backward.cu:130) */
        dL_ddir[0] += dRGBdx * dL_dRGB; dL_ddir[1] += dRGBdy * dL_dRGB; dL_ddir[2] += dRGBdz * dL_dRGB;
    }
    f3 dd = {dL_ddir[0], dL_ddir[1], dL_ddir[2]};
    f3 dm = dnormvdv(dir_orig, dd);
    dL_dmeans[3 * idx] += dm.x; dL_dmeans[3 * idx + 1] += dm.y; dL_dmeans[3 * idx + 2] += dm.z;
}

/* K9 computeCov2DCUDA, DGR/cuda_rasterizer/backward.cu:144-274 */
static void computeCov2D_bwd(int idx, const float* means, const float* cov3Ds, float h_x, float h_y, float tan_fovx,
                             float tan_fovy, const float* view_matrix, const float* dL_dconics, float* dL_dmeans,
                             float* dL_dcov)
{
    const float* cov3D = cov3Ds + 6 * (size_t)idx;
    f3 mean = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    float g0 = dL_dconics[4 * idx], g1 = dL_dconics[4 * idx + 1], g3 = dL_dconics[4 * idx + 3];
    float T[2][3], txtz, tytz; f3 t;
    compute_T(mean, h_x, h_y, tan_fovx, tan_fovy, view_matrix, T, &t, &txtz, &tytz);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    float a, b, c;
    cov2d_from_T(T, cov3D, &a, &b, &c);
    a += 0.3f; c += 0.3f;
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    const float* v = view_matrix;
    float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * g0 + 2 * b * c * g1 + (denom - a * c) * g3);
        dL_dc = denom2inv * (-a * a * g3 + 2 * a * b * g1 + (denom - a * c) * g0);
        dL_db = denom2inv * 2 * (b * c * g0 - (denom + 2 * b * b) * g1 + a * b * g3);
        dL_dcov[6 * idx + 0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
        dL_dcov[6 * idx + 3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
        dL_dcov[6 * idx + 5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
        dL_dcov[6 * idx + 1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
        dL_dcov[6 * idx + 2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
        dL_dcov[6 * idx + 4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
    }
    float dL_dT00 = 2 * (T[0][0] * V[0][0] + T[0][1] * V[0][1] + T[0][2] * V[0][2]) * dL_da + (T[1][0] * V[0][0] + T[1][1] * V[0][1] + T[1][2] * V[0][2]) * dL_db;
    float dL_dT01 = 2 * (T[0][0] * V[1][0] + T[0][1] * V[1][1] + T[0][2] * V[1][2]) * dL_da + (T[1][0] * V[1][0] + T[1][1] * V[1][1] + T[1][2] * V[1][2]) * dL_db;
    float dL_dT02 = 2 * (T[0][0] * V[2][0] + T[0][1] * V[2][1] + T[0][2] * V[2][2]) * dL_da + (T[1][0] * V[2][0] + T[1][1] * V[2][1] + T[1][2] * V[2][2]) * dL_db;
    float dL_dT10 = 2 * (T[1][0] * V[0][0] + T[1][1] * V[0][1] + T[1][2] * V[0][2]) * dL_dc + (T[0][0] * V[0][0] + T[0][1] * V[0][1] + T[0][2] * V[0][2]) * dL_db;
    float dL_dT11 = 2 * (T[1][0] * V[1][0] + T[1][1] * V[1][1] + T[1][2] * V[1][2]) * dL_dc + (T[0][0] * V[1][0] + T[0][1] * V[1][1] + T[0][2] * V[1][2]) * dL_db;
    float dL_dT12 = 2 * (T[1][0] * V[2][0] + T[1][1] * V[2][1] + T[1][2] * V[2][2]) * dL_dc + (T[0][0] * V[2][0] + T[0][1] * V[2][1] + T[0][2] * V[2][2]) * dL_db;
    /* glm W[c][r]: W[0]=(v0,v4,v8), W[1]=(v1,v5,v9), W[2]=(v2,v6,v10)  (backward.cu:182-185,252-255) */
    float dL_dJ00 = v[0] * dL_dT00 + v[4] * dL_dT01 + v[8] * dL_dT02;
    float dL_dJ02 = v[2] * dL_dT00 + v[6] * dL_dT01 + v[10] * dL_dT02;
    float dL_dJ11 = v[1] * dL_dT10 + v[5] * dL_dT11 + v[9] * dL_dT12;
    float dL_dJ12 = v[2] * dL_dT10 + v[6] * dL_dT11 + v[10] * dL_dT12;
    float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    f3 d = {dL_dtx, dL_dty, dL_dtz};
    f3 dm = transformVec4x3Transpose(d, view_matrix);
    dL_dmeans[3 * idx] = dm.x; dL_dmeans[3 * idx + 1] = dm.y; dL_dmeans[3 * idx + 2] = dm.z; /* assignment, :273 */
}

/* Backward of scale/rotation -> cov3D, DGR/cuda_rasterizer/backward.cu:278-341 */
static void computeCov3D_bwd(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                             float* dL_dscales, float* dL_drots)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float R[3][3], M[3][3];
    quat_to_glmR(rot, R);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) M[c][k] = s[k] * R[c][k];
    const float* d = dL_dcov3Ds + 6 * (size_t)idx;
    float Dm[3][3] = {{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}};
    /* dL_dM = (2.0f * M) * dL_dSigma : dL_dM[c][r] = X[0][r]*D[c][0] + X[1][r]*D[c][1] + X[2][r]*D[c][2] */
    float X[3][3], dL_dM[3][3];
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) X[c][k] = 2.0f * M[c][k];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) dL_dM[c][rr] = X[0][rr] * Dm[c][0] + X[1][rr] * Dm[c][1] + X[2][rr] * Dm[c][2];
    /* Rt[c][r] = R[r][c], dL_dMt[c][r] = dL_dM[r][c] */
    float dL_dMt[3][3];
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) dL_dMt[c][rr] = dL_dM[rr][c];
    for (int k = 0; k < 3; k++)
        dL_dscales[3 * idx + k] = R[0][k] * dL_dMt[k][0] + R[1][k] * dL_dMt[k][1] + R[2][k] * dL_dMt[k][2];
    for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) dL_dMt[k][rr] *= s[k];
    float q0 = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
    float q1 = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
    float q2 = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) + 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
    float q3 = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) + 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
    dL_drots[4 * idx] = q0; dL_drots[4 * idx + 1] = q1; dL_drots[4 * idx + 2] = q2; dL_drots[4 * idx + 3] = q3;
}

/* BACKWARD::preprocess = K9 then K10, DGR/cuda_rasterizer/backward.cu:559-622, kernel K10 :346-396.
 * cov3D_ptr is cov3D_precomp if given, else the forward's cov3Ds (rasterizer_impl.cu:411).
 * Outputs zero-initialised by the caller; rows of culled Gaussians (radii <= 0) stay untouched. */
void orc_preprocess_backward(int P, int D, int M, const float* means3D, const int* radii, const float* shs,
                             const uint8_t* clamped, const float* scales, const float* rotations, float scale_modifier,
                             const float* cov3D_ptr, const float* viewmatrix, const float* projmatrix, int W, int H,
                             float tan_fovx, float tan_fovy, const float* campos, const float* dL_dmean2D,
                             const float* dL_dconic, float* dL_dmean3D, float* dL_dcolor, float* dL_dcov3D,
                             float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const float* proj = projmatrix;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        computeCov2D_bwd(idx, means3D, cov3D_ptr, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, dL_dconic, dL_dmean3D, dL_dcov3D);
        f3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
        float dx = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
        float dy = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
        float dz = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        dL_dmean3D[3 * idx] += dx; dL_dmean3D[3 * idx + 1] += dy; dL_dmean3D[3 * idx + 2] += dz;
        if (shs) computeColorFromSH_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmean3D, dL_dsh);
        if (scales) computeCov3D_bwd(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov3D, dL_dscale, dL_drot);
    }
}

/* ---- simple-knn: distCUDA2 = mean squared distance to the 3 nearest other points --------------
 * KNN = /root/reference/gaussian_splatting/submodules/simple-knn.  The reference reaches the exact
 * 3-NN through Morton boxes + AABB pruning (KNN/simple_knn.cu:119-183); the result it defines is the
 * exact value below (KNN/simple_knn.cu:182: mean of the three best squared distances; updateKBest
 * :131-145 keeps the 3 smallest, initial best = FLT_MAX).  O(P^2) brute force, small P only. */
void orc_dist2(int P, const float* points, float* meanDists)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = points[3 * j] - points[3 * i], dy = points[3 * j + 1] - points[3 * i + 1], dz = points[3 * j + 2] - points[3 * i + 2];
            float dist = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++) /* updateKBest<3>, KNN/simple_knn.cu:131-145 */
                if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
        }
        meanDists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
