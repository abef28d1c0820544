/*
 * oracle/mesh_rasterizer.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C CPU restatement of the triangle-mesh z-buffer SuGaR's level-set sampler reads through
 *     pytorch3d.renderer.MeshRasterizer   ->   fragments.zbuf / fragments.pix_to_face
 * (/root/reference/sugar_scene/sugar_model.py:1880-1893, 1927-1928, 1966; sugar_extractors/coarse_mesh.py:216-225).
 * It is the parity oracle of sugar_amd/csrc/mesh_raster.hip.  Only tests/ may load it.
 *
 * PARITY UNPINNED: the algorithm lives in a third-party dependency that is ABSENT from /root/reference and from this image:
 * pytorch3d 0.7.4 (pinned by the reference's environment.yml:161).  What follows restates its PUBLISHED naive rasterizer
 *     pytorch3d/csrc/rasterize_meshes/rasterize_meshes.cu   RasterizeMeshesNaiveCudaKernel, CheckPixelInsideFace
 *     pytorch3d/csrc/rasterize_meshes/rasterize_meshes_cpu.cpp (the same per-pixel rule on the CPU)
 *     pytorch3d/csrc/utils/geometry_utils.cuh               EdgeFunctionForward, BarycentricCoordsForward,
 *                                                           BarycentricPerspectiveCorrectionForward, BarycentricClipForward,
 *                                                           PointLineDistanceForward, PointTriangleDistanceForward
 *     pytorch3d/csrc/rasterize_meshes/rasterization_utils.cuh  PixToNonSquareNdc
 * from the public sources as the builder knows them; nothing here could be checked against pytorch3d itself.  What the tests
 * pin instead: geometric properties (the face named for a pixel contains the pixel centre; its z is the perspective-correct
 * depth of the face's plane there; the K entries are the K nearest covering faces in ascending z), consistency with the
 * Gaussian rasterizer's camera (tests/test_mesh_oracle.py), and the reference's own sampler run on top of it
 * (tests/golden/make_sugar_meshdepth.py).
 *
 * (The coarse-to-fine CUDA path pytorch3d takes on a GPU for bin_size != 0 evaluates the same CheckPixelInsideFace over the
 *  faces of a bin in ascending face order, so its result equals the naive one unless a bin overflows max_faces_per_bin, in
 *  which case pytorch3d drops faces with a warning.  That overflow is not restated.)
 *
 * Arithmetic contract with the HIP kernel: float32, every operation individually rounded in the order written here
 * (-ffp-contract=off, oracle/Makefile); the HIP translation unit is compiled the same way, so pix_to_face, zbuf, the
 * barycentric coordinates and the distances agree BIT FOR BIT.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define K_EPSILON 1e-8f /* kEpsilon, geometry_utils.cuh */
#define MAX_K 64

/* rasterization_utils.cuh: NonSquareNdcRange / PixToNonSquareNdc -- NDC coordinate of the centre of pixel i along an axis of
 * S1 pixels when the other axis has S2 (the SHORTER axis spans [-1, 1], the longer one [-S1/S2, S1/S2]) */
static float pix_to_non_square_ndc(int i, int S1, int S2)
{
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S1;
}
float orc_mesh_pix_to_ndc(int i, int S1, int S2) { return pix_to_non_square_ndc(i, S1, S2); }

/* geometry_utils.cuh: EdgeFunctionForward(p, v0, v1) */
static float edge_function(float px, float py, float v0x, float v0y, float v1x, float v1y)
{
    return (px - v0x) * (v1y - v0y) - (py - v0y) * (v1x - v0x);
}

/* geometry_utils.cuh: PointLineDistanceForward -- squared distance of p to the segment (a, b) */
static float point_line_distance(float px, float py, float ax, float ay, float bx, float by)
{
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    if (l2 <= K_EPSILON) return (px - bx) * (px - bx) + (py - by) * (py - by);
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = qx - px, dy = qy - py;
    return dx * dx + dy * dy;
}

/* geometry_utils.cuh: PointTriangleDistanceForward -- minimum over the three edges */
static float point_triangle_distance(float px, float py, const float* v)
{
    const float e01 = point_line_distance(px, py, v[0], v[1], v[3], v[4]);
    const float e02 = point_line_distance(px, py, v[0], v[1], v[6], v[7]);
    const float e12 = point_line_distance(px, py, v[3], v[4], v[6], v[7]);
    return fminf(fminf(e01, e02), e12);
}

typedef struct { float z; int64_t idx; float dist; float b[3]; } Pix;

/* rasterize_meshes.cu: CheckPixelInsideFace.  v = the face's nine floats {x0,y0,z0, x1,y1,z1, x2,y2,z2}: x, y in NDC
 * (+x left, +y up), z = view-space depth. */
static void check_pixel_inside_face(const float* v, const int64_t* neighbor_idx, int64_t face_idx, int* q_size, float* q_max_z,
                                    int* q_max_idx, Pix* q, float blur_radius, float px, float py, int K, int perspective_correct,
                                    int clip_barycentric_coords, int cull_backfaces)
{
    const float v0x = v[0], v0y = v[1], v0z = v[2], v1x = v[3], v1y = v[4], v1z = v[5], v2x = v[6], v2y = v[7], v2z = v[8];
    /* CheckPointOutsideBoundingBox (with GetFaceBoundingBox): blur in NDC, faces with a vertex at z < kEpsilon never render */
    const float zmax = fmaxf(fmaxf(v0z, v1z), v2z);
    const float sb = sqrtf(blur_radius);
    const float xmin = fminf(fminf(v0x, v1x), v2x) - sb, xmax = fmaxf(fmaxf(v0x, v1x), v2x) + sb;
    const float ymin = fminf(fminf(v0y, v1y), v2y) - sb, ymax = fmaxf(fmaxf(v0y, v1y), v2y) + sb;
    const float zmin = fminf(fminf(v0z, v1z), v2z);
    const int z_invalid = zmin < K_EPSILON;
    const int outside_bbox = (px > xmax || px < xmin || py > ymax || py < ymin || z_invalid);
    const float face_area = edge_function(v0x, v0y, v1x, v1y, v2x, v2y);
    const int back_face = face_area < 0.0f;
    const int zero_face_area = (face_area <= K_EPSILON && face_area >= -1.0f * K_EPSILON);
    if (zmax < 0 || (cull_backfaces && back_face) || outside_bbox || zero_face_area) return;

    /* BarycentricCoordsForward */
    const float area = edge_function(v2x, v2y, v0x, v0y, v1x, v1y) + K_EPSILON;
    const float w0 = edge_function(px, py, v1x, v1y, v2x, v2y) / area;
    const float w1 = edge_function(px, py, v2x, v2y, v0x, v0y) / area;
    const float w2 = edge_function(px, py, v0x, v0y, v1x, v1y) / area;
    float b0 = w0, b1 = w1, b2 = w2;
    if (perspective_correct) { /* BarycentricPerspectiveCorrectionForward */
        const float w0_top = w0 * v1z * v2z;
        const float w1_top = v0z * w1 * v2z;
        const float w2_top = v0z * v1z * w2;
        const float denom = fmaxf(w0_top + w1_top + w2_top, K_EPSILON);
        b0 = w0_top / denom; b1 = w1_top / denom; b2 = w2_top / denom;
    }
    float c0 = b0, c1 = b1, c2 = b2;
    if (clip_barycentric_coords) { /* BarycentricClipForward */
        c0 = fmaxf(0.0f, fminf(1.0f, b0)); c1 = fmaxf(0.0f, fminf(1.0f, b1)); c2 = fmaxf(0.0f, fminf(1.0f, b2));
        const float s = fmaxf(c0 + c1 + c2, 1e-5f);
        c0 /= s; c1 /= s; c2 /= s;
    }
    const float pz = c0 * v0z + c1 * v1z + c2 * v2z;
    if (pz < 0) return; /* behind the image plane */
    const float dist = point_triangle_distance(px, py, v);
    const int inside = b0 > 0.0f && b1 > 0.0f && b2 > 0.0f; /* the UNCLIPPED coordinates decide */
    const float signed_dist = inside ? -dist : dist;
    if (!inside && dist >= blur_radius) return;

    /* the two halves of a face split by the near-plane clip: keep the closer (in the image plane) of the pair */
    const int64_t nb = neighbor_idx ? neighbor_idx[face_idx] : -1;
    int nb_top_k = -1;
    if (nb != -1)
        for (int i = 0; i < *q_size; i++)
            if (q[i].idx == nb) { nb_top_k = i; break; }
    const Pix cand = {pz, face_idx, signed_dist, {c0, c1, c2}};
    if (nb_top_k != -1) {
        if (dist < fabsf(q[nb_top_k].dist)) {
            q[nb_top_k] = cand;
            if (pz > *q_max_z) { *q_max_z = pz; *q_max_idx = nb_top_k; }
        }
    } else if (*q_size < K) {
        q[*q_size] = cand;
        if (pz > *q_max_z) { *q_max_z = pz; *q_max_idx = *q_size; }
        (*q_size)++;
    } else if (pz < *q_max_z) {
        q[*q_max_idx] = cand;
        *q_max_z = pz;
        for (int i = 0; i < K; i++)
            if (q[i].z > *q_max_z) { *q_max_z = q[i].z; *q_max_idx = i; }
    }
}

/* RasterizeMeshesNaive for ONE mesh.  face_verts [F,3,3]; outputs [H,W,K] (bary [H,W,K,3]), unfilled slots -1.
 * Output pixel (row r, column c) looks along NDC (x, y) = (PixToNonSquareNdc(W-1-c, W, H), PixToNonSquareNdc(H-1-r, H, W)):
 * the "reverse ordering of X and Y axes" of the kernel.
 * Speed only: faces are bucketed by the pixel rows their (blurred) bounding box can touch, conservatively; a pixel still
 * visits its candidates in ascending face order and applies the full rule, so the result is the naive one. */
int orc_rasterize_meshes_naive(const float* face_verts, int64_t F, const int64_t* neighbor_idx, int H, int W, float blur_radius,
                               int K, int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                               int64_t* pix_to_face, float* zbuf, float* bary, float* dists)
{
    if (K < 1 || K > MAX_K || H < 1 || W < 1 || F < 0) return -1;
    /* rows: yi = H-1-r has NDC y(yi), increasing in yi */
    int64_t* row_cnt = (int64_t*)calloc((size_t)H + 1, sizeof(int64_t));
    int* y_lo = (int*)malloc(sizeof(int) * (size_t)(F > 0 ? F : 1));
    int* y_hi = (int*)malloc(sizeof(int) * (size_t)(F > 0 ? F : 1));
    const float sb = sqrtf(blur_radius);
    const float y_first = pix_to_non_square_ndc(0, H, W), y_last = pix_to_non_square_ndc(H - 1, H, W);
    const float dy = H > 1 ? (y_last - y_first) / (float)(H - 1) : 1.0f;
    for (int64_t f = 0; f < F; f++) {
        const float* v = face_verts + 9 * f;
        const float ymin = fminf(fminf(v[1], v[4]), v[7]) - sb, ymax = fmaxf(fmaxf(v[1], v[4]), v[7]) + sb;
        int lo = 0, hi = -1;
        if (ymin == ymin && ymax == ymax && ymax >= y_first - 1.0f && ymin <= y_last + 1.0f) { /* NaN: never inside */
            double a = floor(((double)ymin - (double)y_first) / (double)dy) - 2.0, b = ceil(((double)ymax - (double)y_first) / (double)dy) + 2.0;
            if (a < 0) a = 0;
            if (b > H - 1) b = H - 1;
            lo = (int)a; hi = (int)b;
        }
        y_lo[f] = lo; y_hi[f] = hi;
        for (int y = lo; y <= hi; y++) row_cnt[y + 1]++;
    }
    for (int y = 0; y < H; y++) row_cnt[y + 1] += row_cnt[y];
    int64_t* row_faces = (int64_t*)malloc(sizeof(int64_t) * (size_t)(row_cnt[H] > 0 ? row_cnt[H] : 1));
    int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)H);
    memcpy(cur, row_cnt, sizeof(int64_t) * (size_t)H);
    for (int64_t f = 0; f < F; f++)
        for (int y = y_lo[f]; y <= y_hi[f]; y++) row_faces[cur[y]++] = f; /* ascending f per row */

#pragma omp parallel for schedule(dynamic, 4)
    for (int r = 0; r < H; r++) {
        const int yi = H - 1 - r;
        const float yf = pix_to_non_square_ndc(yi, H, W);
        for (int c = 0; c < W; c++) {
            const int xi = W - 1 - c;
            const float xf = pix_to_non_square_ndc(xi, W, H);
            Pix q[MAX_K];
            int q_size = 0, q_max_idx = -1;
            float q_max_z = -1000.0f;
            for (int64_t j = row_cnt[yi]; j < row_cnt[yi + 1]; j++) {
                const int64_t f = row_faces[j];
                check_pixel_inside_face(face_verts + 9 * f, neighbor_idx, f, &q_size, &q_max_z, &q_max_idx, q, blur_radius, xf, yf, K,
                                        perspective_correct, clip_barycentric_coords, cull_backfaces);
            }
            /* BubbleSort by z (stable) */
            for (int i = 0; i < q_size - 1; i++)
                for (int j = 0; j < q_size - i - 1; j++)
                    if (q[j + 1].z < q[j].z) { Pix t = q[j]; q[j] = q[j + 1]; q[j + 1] = t; }
            const size_t base = ((size_t)r * W + c) * (size_t)K;
            for (int k = 0; k < K; k++) {
                const int have = k < q_size;
                pix_to_face[base + k] = have ? q[k].idx : -1;
                zbuf[base + k] = have ? q[k].z : -1.0f;
                if (dists) dists[base + k] = have ? q[k].dist : -1.0f;
                if (bary) for (int i = 0; i < 3; i++) bary[(base + k) * 3 + i] = have ? q[k].b[i] : -1.0f;
            }
        }
    }
    free(row_cnt); free(y_lo); free(y_hi); free(row_faces); free(cur);
    return 0;
}
