/*
 * onepiece_oracle.c -- CPU restatement (plain C99) of the OnePiece TSDF-fusion + ICP hot path.
 * TEST INFRASTRUCTURE ONLY; see onepiece_oracle.h for the rules and the parity status
 * ("parity unpinned" against the compiled reference; pinned against Eigen/Sophus and nanoflann golden vectors
 * and the SURVEY.md reference-run statistics).
 *
 * Build: gcc -O3 -msse4.2 -ffp-contract=off -fopenmp (the reference's flags, CMakeLists.txt:149-150,
 * plus -ffp-contract=off so no fused multiply-add can appear: the reference is built for SSE4.2,
 * which has none).  Every float expression below keeps the reference's operand order and
 * intermediate rounding; citations are file:line under /root/reference/src unless stated.
 */
#include "onepiece_oracle.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CUBE 8
#define NVOX 512

/* Threads for the fusion path.  The reference integrates serially (CubeHandler.cpp:205-208), so the default is 1
 * and that is what bench.py's cpu_baseline times.  Every unit the loops below hand to a thread is independent in
 * the reference too (a pixel of ComputeBounding, a candidate block of PrepareCubes, a selected block of
 * IntegrateImage -- blocks are exclusively owned), and min/max/integer sums do not depend on the order, so the
 * results are bit-identical for any thread count; tests raise it so that full-size sequences finish in seconds. */
static int g_fusion_threads = 1;
void orc_set_fusion_threads(int n) { g_fusion_threads = n < 1 ? 1 : n; }

/* ------------------------------------------------------------------------------------------ */
/* Eigen 3.3.7 fixed-size semantics used all over the reference                                */
/* ------------------------------------------------------------------------------------------ */

/* 3-term reduction: Eigen's redux_novec_unroller splits [0,3) as {0} + {1,2}
 * (3rdparty/Eigen/Eigen/src/Core/Redux.h, HalfLength = Length/2). */
static inline float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
static inline float dot3(const float *a, const float *b) {
    return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]);
}

/* Matrix4f * Vector4f(x,y,z,1): coefficient-based packet product accumulating column by column,
 * ((c0*x + c1*y) + c2*z) + c3*1 (3rdparty/Eigen/Eigen/src/Core/ProductEvaluators.h:626-632). */
static inline void mat4_mul_p1(const float *M, float x, float y, float z, float out[4]) {
    for (int r = 0; r < 4; ++r)
        out[r] = ((M[r * 4 + 0] * x + M[r * 4 + 1] * y) + M[r * 4 + 2] * z) + M[r * 4 + 3] * 1.0f;
}

/* ---- 4x4 float inverse, SSE cofactor kernel restated lane by lane -------------------------- */
typedef struct { float v[4]; } q4;
static inline q4 q_shuf(q4 a, q4 b, int imm) { /* _mm_shuffle_ps */
    q4 r = {{a.v[imm & 3], a.v[(imm >> 2) & 3], b.v[(imm >> 4) & 3], b.v[(imm >> 6) & 3]}};
    return r;
}
static inline q4 q_movelh(q4 a, q4 b) { q4 r = {{a.v[0], a.v[1], b.v[0], b.v[1]}}; return r; }
static inline q4 q_movehl(q4 a, q4 b) { q4 r = {{b.v[2], b.v[3], a.v[2], a.v[3]}}; return r; }
static inline q4 q_mul(q4 a, q4 b) { q4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] * b.v[i]; return r; }
static inline q4 q_add(q4 a, q4 b) { q4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
static inline q4 q_sub(q4 a, q4 b) { q4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
static inline q4 q_sub_ss(q4 a, q4 b) { a.v[0] = a.v[0] - b.v[0]; return a; }
static inline q4 q_add_ss(q4 a, q4 b) { a.v[0] = a.v[0] + b.v[0]; return a; }
static inline q4 q_mul_ss(q4 a, q4 b) { a.v[0] = a.v[0] * b.v[0]; return a; }
static inline q4 q_bcast0(q4 a) { q4 r = {{a.v[0], a.v[0], a.v[0], a.v[0]}}; return r; }

void orc_mat4_inverse(const float m[16], float out[16]) {
    /* Column-major registers L1..L4 = columns (the reference's Matrix4f is column-major and
     * StorageOrdersMatch is true; Inverse_SSE.h:50-73).  m is row-major here: col c = m[r*4+c]. */
    q4 L1 = {{m[0], m[4], m[8], m[12]}}, L2 = {{m[1], m[5], m[9], m[13]}};
    q4 L3 = {{m[2], m[6], m[10], m[14]}}, L4 = {{m[3], m[7], m[11], m[15]}};
    q4 A = q_movelh(L1, L2), B = q_movehl(L2, L1), C = q_movelh(L3, L4), D = q_movehl(L4, L3);
    q4 AB, DC, dA, dB, dC, dD, d, iA, iB, iC, iD, d1, d2, det, rd;
    /* Inverse_SSE.h:82-87 */
    AB = q_mul(q_shuf(A, A, 0x0F), B);
    AB = q_sub(AB, q_mul(q_shuf(A, A, 0xA5), q_shuf(B, B, 0x4E)));
    DC = q_mul(q_shuf(D, D, 0x0F), C);
    DC = q_sub(DC, q_mul(q_shuf(D, D, 0xA5), q_shuf(C, C, 0x4E)));
    /* :89-101 sub-determinants */
    dA = q_mul(q_shuf(A, A, 0x5F), A); dA = q_sub_ss(dA, q_movehl(dA, dA));
    dB = q_mul(q_shuf(B, B, 0x5F), B); dB = q_sub_ss(dB, q_movehl(dB, dB));
    dC = q_mul(q_shuf(C, C, 0x5F), C); dC = q_sub_ss(dC, q_movehl(dC, dC));
    dD = q_mul(q_shuf(D, D, 0x5F), D); dD = q_sub_ss(dD, q_movehl(dD, dD));
    /* :103-111 */
    d = q_mul(q_shuf(DC, DC, 0xD8), AB);
    iD = q_mul(q_shuf(C, C, 0xA0), q_movelh(AB, AB));
    iD = q_add(iD, q_mul(q_shuf(C, C, 0xF5), q_movehl(AB, AB)));
    iA = q_mul(q_shuf(B, B, 0xA0), q_movelh(DC, DC));
    iA = q_add(iA, q_mul(q_shuf(B, B, 0xF5), q_movehl(DC, DC)));
    /* :113-117 */
    d = q_add(d, q_movehl(d, d));
    d = q_add_ss(d, q_shuf(d, d, 1));
    d1 = q_mul_ss(dA, dD);
    d2 = q_mul_ss(dB, dC);
    /* :119-123 */
    iD = q_sub(q_mul(D, q_bcast0(dA)), iD);
    iA = q_sub(q_mul(A, q_bcast0(dD)), iA);
    /* :125-127: det, true division for the reciprocal */
    det = q_sub_ss(q_add_ss(d1, d2), d);
    rd = det; rd.v[0] = 1.0f / det.v[0];
    /* :133-138 */
    iB = q_mul(D, q_shuf(AB, AB, 0x33));
    iB = q_sub(iB, q_mul(q_shuf(D, D, 0xB1), q_shuf(AB, AB, 0x66)));
    iC = q_mul(A, q_shuf(DC, DC, 0x33));
    iC = q_sub(iC, q_mul(q_shuf(A, A, 0xB1), q_shuf(DC, DC, 0x66)));
    /* :140-141 sign mask (+,-,-,+) */
    rd = q_bcast0(rd); rd.v[1] = -rd.v[1]; rd.v[2] = -rd.v[2];
    /* :143-147 */
    iB = q_sub(q_mul(C, q_bcast0(dB)), iB);
    iC = q_sub(q_mul(B, q_bcast0(dC)), iC);
    /* :149-153 */
    iA = q_mul(rd, iA); iB = q_mul(rd, iB); iC = q_mul(rd, iC); iD = q_mul(rd, iD);
    /* :155-160 result columns */
    q4 c0 = q_shuf(iA, iB, 0x77), c1 = q_shuf(iA, iB, 0x22);
    q4 c2 = q_shuf(iC, iD, 0x77), c3 = q_shuf(iC, iD, 0x22);
    for (int r = 0; r < 4; ++r) {
        out[r * 4 + 0] = c0.v[r]; out[r * 4 + 1] = c1.v[r];
        out[r * 4 + 2] = c2.v[r]; out[r * 4 + 3] = c3.v[r];
    }
}

/* Geometry/Geometry.h:101-112: ints are converted to size_t (sign-extended), 64-bit wrapping
 * multiply, precedence (x*p1) ^ (y*p2) ^ (z*p3). */
uint64_t orc_hash(int x, int y, int z) {
    return ((uint64_t)(int64_t)x * 73856093ULL) ^ ((uint64_t)(int64_t)y * 19349663ULL) ^
           ((uint64_t)(int64_t)z * 83492791ULL);
}

/* ------------------------------------------------------------------------------------------ */
/* Frustum                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Geometry/Geometry.cpp:165-171 */
static void get_plane(const float *p1, const float *p2, const float *p3, float *plane) {
    float a[3], b[3], n[3];
    for (int i = 0; i < 3; ++i) { a[i] = p2[i] - p1[i]; b[i] = p3[i] - p1[i]; }
    n[0] = a[1] * b[2] - a[2] * b[1];
    n[1] = a[2] * b[0] - a[0] * b[2];
    n[2] = a[0] * b[1] - a[1] * b[0];
    float z = sum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
    if (z > 0.0f) { float s = sqrtf(z); n[0] = n[0] / s; n[1] = n[1] / s; n[2] = n[2] / s; }
    double d = -dot3(p1, n);
    plane[0] = n[0]; plane[1] = n[1]; plane[2] = n[2]; plane[3] = (float)d;
}

void orc_frustum_planes(const orc_camera *cam, const float pose[16], float far_d, float near_d,
                        float planes[24]) {
    /* Frustum.cpp:7-25.  atan2/tan are the double C functions (unqualified calls with only
     * <cmath> in scope resolve to ::atan2(double,double)); results are rounded to float. */
    float fx = cam->fx, fy = cam->fy, cy = cam->cy;
    float height = (float)cam->height, width = (float)cam->width;
    float right[3] = {pose[0], pose[4], pose[8]};
    float up[3] = {-pose[1], -pose[5], -pose[9]};
    float fwd[3] = {pose[2], pose[6], pose[10]};
    float pos[3] = {pose[3], pose[7], pose[11]};
    float aspect = (fy * width) / (fx * height);
    float fov = (float)(atan2((double)cy, (double)fy) + atan2((double)(height - cy), (double)fy));
    /* Frustum.cpp:26-46 */
    float angle_tangent = (float)tan((double)(fov / 2));
    float height_far = angle_tangent * far_d, width_far = height_far * aspect;
    float height_near = angle_tangent * near_d, width_near = height_near * aspect;
    float fc[3], nc[3], ftl[3], ftr[3], fbl[3], fbr[3], ntl[3], ntr[3], nbl[3], nbr[3];
    for (int i = 0; i < 3; ++i) {
        fc[i] = pos[i] + fwd[i] * far_d;
        ftl[i] = (fc[i] + up[i] * height_far) - right[i] * width_far;
        ftr[i] = (fc[i] + up[i] * height_far) + right[i] * width_far;
        fbl[i] = (fc[i] - up[i] * height_far) - right[i] * width_far;
        fbr[i] = (fc[i] - up[i] * height_far) + right[i] * width_far;
        nc[i] = pos[i] + fwd[i] * near_d;
        ntl[i] = (nc[i] + up[i] * height_near) - right[i] * width_near;
        ntr[i] = (nc[i] + up[i] * height_near) + right[i] * width_near;
        nbl[i] = (nc[i] - up[i] * height_near) - right[i] * width_near;
        nbr[i] = (nc[i] - up[i] * height_near) + right[i] * width_near;
    }
    get_plane(ntl, ftl, ntr, planes + 0);   /* top    */
    get_plane(ftl, ntl, fbl, planes + 4);   /* left   */
    get_plane(ntr, ftr, nbr, planes + 8);   /* right  */
    get_plane(nbr, fbl, nbl, planes + 12);  /* bottom */
    get_plane(nbl, ntl, nbr, planes + 16);  /* near   */
    get_plane(ftr, ftl, fbr, planes + 20);  /* far    */
}

/* Frustum.h:74-103, including the early "distance == 0 -> true". */
static int frustum_contains(const float planes[24], const float *p) {
    for (int k = 0; k < 6; ++k) {
        float distance = dot3(planes + 4 * k, p) + planes[4 * k + 3];
        if (distance < 0) return 0;
        if (distance == 0) return 1;
    }
    return 1;
}

static inline float depth_at(const void *depth, int is_u16, float depth_scale, size_t idx) {
    /* Integrator.cpp:26-29 */
    if (!is_u16) return ((const float *)depth)[idx];
    return ((const unsigned short *)depth)[idx] / depth_scale;
}

size_t orc_compute_bounding(const orc_camera *cam, const void *depth, int is_u16,
                            const float pose[16], float far_d, float near_d, float max_pos[3],
                            float min_pos[3]) {
    float planes[24];
    orc_frustum_planes(cam, pose, far_d, near_d, planes);
    for (int i = 0; i < 3; ++i) { max_pos[i] = -FLT_MAX; min_pos[i] = FLT_MAX; }
    float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
    size_t inside = 0;
    float mx0 = -FLT_MAX, mx1 = -FLT_MAX, mx2 = -FLT_MAX, mn0 = FLT_MAX, mn1 = FLT_MAX, mn2 = FLT_MAX;
#pragma omp parallel for num_threads(g_fusion_threads) schedule(static) reduction(+ : inside) reduction(max : mx0, mx1, mx2) reduction(min : mn0, mn1, mn2)
    for (int i = 0; i < cam->height; ++i)
        for (int j = 0; j < cam->width; ++j) {
            float z = depth_at(depth, is_u16, cam->depth_scale, (size_t)i * cam->width + j);
            if (!(z > 0)) continue;
            /* PointCloud.cpp:90-93 */
            float x = (j - cx) * z / fx;
            float y = (i - cy) * z / fy;
            /* Geometry.cpp:19-27 */
            float q[4], p[3];
            mat4_mul_p1(pose, x, y, z, q);
            p[0] = q[0] / q[3]; p[1] = q[1] / q[3]; p[2] = q[2] / q[3];
            if (frustum_contains(planes, p)) {
                ++inside;
                /* std::max(p, max) / std::min: NaN never replaces the running value */
                mx0 = p[0] > mx0 ? p[0] : mx0; mx1 = p[1] > mx1 ? p[1] : mx1; mx2 = p[2] > mx2 ? p[2] : mx2;
                mn0 = p[0] < mn0 ? p[0] : mn0; mn1 = p[1] < mn1 ? p[1] : mn1; mn2 = p[2] < mn2 ? p[2] : mn2;
            }
        }
    max_pos[0] = mx0; max_pos[1] = mx1; max_pos[2] = mx2;
    min_pos[0] = mn0; min_pos[1] = mn1; min_pos[2] = mn2;
    return inside;
}

/* ------------------------------------------------------------------------------------------ */
/* Projection / GetSDF                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* Integrator.cpp:20-21,61-62: (((f*X)/Z) + 0.5) + c with the 0.5 literal promoting to double, then
 * truncation toward zero.  Non-finite / out-of-int-range values are UB in C++; x86 cvttsd2si
 * yields INT_MIN, which fails the u >= 0 test, so they are rejected here too. */
static inline int project(float f, float X, float Z, float c) {
    double t = (double)((f * X) / Z) + 0.5 + (double)c;
    if (!(t > -2147483649.0 && t < 2147483648.0)) return INT_MIN;
    return (int)t;
}

float orc_get_sdf(const orc_camera *cam, const float p[3], const float pose_inv[16],
                  const void *depth, int is_u16) {
    float q[4];
    mat4_mul_p1(pose_inv, p[0], p[1], p[2], q);
    int u = project(cam->fx, q[0], q[2], cam->cx);
    int v = project(cam->fy, q[1], q[2], cam->cy);
    if (v < 0 || v >= cam->height || u < 0 || u >= cam->width) return 999;
    float d = depth_at(depth, is_u16, cam->depth_scale, (size_t)v * cam->width + u);
    if (d <= 0) return 999;
    return d - q[2];
}

void orc_cube_id(float res, const float p[3], int id[3]) {
    /* VoxelCube.h:63-74 */
    for (int i = 0; i < 3; ++i) {
        int pb = (int)floorf(p[i] / res);
        id[i] = (int)floor((pb + 0.0) / CUBE);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Volume = hash map of 8^3 blocks (std::unordered_map<CubeID,VoxelCube>, CubeHandler.h:22)    */
/* ------------------------------------------------------------------------------------------ */
struct orc_volume {
    orc_camera cam;
    float res, trunc, far_d, near_d;
    float offset[NVOX][3]; /* VoxelCentroidOffSet, VoxelCube.h:48-61 */
    size_t n, cap;         /* blocks */
    int32_t *keys;         /* n x 3 */
    float *vox;            /* n x 512 x 5 */
    size_t tcap;           /* hash table capacity (power of two) */
    int64_t *table;        /* block index or -1 */
};

static void vol_init_offsets(orc_volume *v) {
    float half = v->res / 2;
    for (size_t x = 0; x < CUBE; ++x)
        for (size_t y = 0; y < CUBE; ++y)
            for (size_t z = 0; z < CUBE; ++z) {
                float *o = v->offset[x + y * CUBE + z * CUBE * CUBE];
                o[0] = x * v->res + half; o[1] = y * v->res + half; o[2] = z * v->res + half;
            }
}

orc_volume *orc_volume_create(const orc_camera *cam, float voxel_res, float trunc, float far_d,
                              float near_d) {
    orc_volume *v = (orc_volume *)calloc(1, sizeof(*v));
    v->cam = *cam; v->res = voxel_res; v->trunc = trunc; v->far_d = far_d; v->near_d = near_d;
    vol_init_offsets(v);
    v->tcap = 1 << 16;
    v->table = (int64_t *)malloc(v->tcap * sizeof(int64_t));
    memset(v->table, 0xff, v->tcap * sizeof(int64_t));
    return v;
}
void orc_volume_clear(orc_volume *v) {
    v->n = 0;
    memset(v->table, 0xff, v->tcap * sizeof(int64_t));
}
void orc_volume_destroy(orc_volume *v) {
    if (!v) return;
    free(v->keys); free(v->vox); free(v->table); free(v);
}
size_t orc_volume_block_count(const orc_volume *v) { return v->n; }

static int64_t vol_find(const orc_volume *v, int x, int y, int z) {
    size_t mask = v->tcap - 1, h = (size_t)orc_hash(x, y, z) & mask;
    for (;;) {
        int64_t b = v->table[h];
        if (b < 0) return -1;
        const int32_t *k = v->keys + 3 * b;
        if (k[0] == x && k[1] == y && k[2] == z) return b;
        h = (h + 1) & mask;
    }
}
static void vol_table_put(orc_volume *v, int64_t b) {
    const int32_t *k = v->keys + 3 * b;
    size_t mask = v->tcap - 1, h = (size_t)orc_hash(k[0], k[1], k[2]) & mask;
    while (v->table[h] >= 0) h = (h + 1) & mask;
    v->table[h] = b;
}
/* VoxelCube(id): 512 default TSDFVoxel {sdf 999, w 0, color -1} (TSDFVoxel.h:79-81). */
static int64_t vol_add(orc_volume *v, int x, int y, int z) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->keys = (int32_t *)realloc(v->keys, v->cap * 3 * sizeof(int32_t));
        v->vox = (float *)realloc(v->vox, v->cap * NVOX * 5 * sizeof(float));
    }
    int64_t b = (int64_t)v->n++;
    v->keys[3 * b] = x; v->keys[3 * b + 1] = y; v->keys[3 * b + 2] = z;
    float *p = v->vox + (size_t)b * NVOX * 5;
    for (int i = 0; i < NVOX; ++i, p += 5) { p[0] = 999; p[1] = 0; p[2] = p[3] = p[4] = -1; }
    if (v->n * 2 > v->tcap) {
        v->tcap *= 2;
        v->table = (int64_t *)realloc(v->table, v->tcap * sizeof(int64_t));
        memset(v->table, 0xff, v->tcap * sizeof(int64_t));
        for (size_t i = 0; i < v->n; ++i) vol_table_put(v, (int64_t)i);
    } else
        vol_table_put(v, b);
    return b;
}

static const int kCornerVoxel[8] = {0, 7, 56, 63, 448, 455, 504, 511}; /* CubeHandler.cpp:158-162 */

size_t orc_volume_prepare_cubes(orc_volume *v, const void *depth, int is_u16, const float pose[16],
                                int32_t *ids, size_t cap, size_t *n_candidates) {
    float maxp[3], minp[3], pose_inv[16];
    size_t inside = orc_compute_bounding(&v->cam, depth, is_u16, pose, v->far_d, v->near_d, maxp, minp);
    if (n_candidates) *n_candidates = 0;
    if (inside == 0) return 0; /* the reference would loop over an undefined range; defined as empty */
    int maxid[3], minid[3];
    orc_cube_id(v->res, maxp, maxid);
    orc_cube_id(v->res, minp, minid);
    orc_mat4_inverse(pose, pose_inv); /* Integrator.cpp:18 (recomputed per call there) */
    float cube_res = v->res * CUBE;   /* CubeHandler.cpp:164 */
    size_t n = 0;
    /* the 8-corner test of every candidate is independent (CubeHandler.cpp:166-180): evaluated first, possibly on
     * several threads; allocation and the id list then follow in the reference's i, j, k loop order (:181-190) */
    const long long ni = (long long)maxid[0] - minid[0] + 3, nj = (long long)maxid[1] - minid[1] + 3,
                    nk = (long long)maxid[2] - minid[2] + 3;
    const long long ncand = ni * nj * nk;
    unsigned char *sel = (unsigned char *)malloc((size_t)(ncand > 0 ? ncand : 1));
#pragma omp parallel for num_threads(g_fusion_threads) schedule(static)
    for (long long q = 0; q < ncand; ++q) {
        const int i = minid[0] - 1 + (int)(q / (nj * nk)), j = minid[1] - 1 + (int)((q / nk) % nj), k = minid[2] - 1 + (int)(q % nk);
        float min_sdf = FLT_MAX;
        for (int c = 0; c < 8; ++c) {
            const float *o = v->offset[kCornerVoxel[c]];
            float p[3] = {i * cube_res + o[0], j * cube_res + o[1], k * cube_res + o[2]};
            float sdf = orc_get_sdf(&v->cam, p, pose_inv, depth, is_u16);
            if (min_sdf > fabsf(sdf)) min_sdf = fabsf(sdf);
        }
        sel[q] = min_sdf < v->trunc;
    }
    for (long long q = 0; q < ncand; ++q) {
        if (!sel[q]) continue;
        const int i = minid[0] - 1 + (int)(q / (nj * nk)), j = minid[1] - 1 + (int)((q / nk) % nj), k = minid[2] - 1 + (int)(q % nk);
        if (vol_find(v, i, j, k) < 0) vol_add(v, i, j, k);
        if (ids && n < cap) { ids[3 * n] = i; ids[3 * n + 1] = j; ids[3 * n + 2] = k; }
        ++n;
    }
    free(sel);
    if (n_candidates) *n_candidates = (size_t)(ncand > 0 ? ncand : 0);
    return n;
}

/* Integrator.cpp:36-94 for one block; returns the number of voxels updated. */
static uint64_t integrate_block(orc_volume *v, int64_t b, const void *depth, int is_u16,
                                const uint8_t *rgb, const float pose_inv[16]) {
    const orc_camera *cam = &v->cam;
    const int32_t *id = v->keys + 3 * b;
    float *vox = v->vox + (size_t)b * NVOX * 5;
    /* VoxelCube.h:75-80: Point3(id) * CUBE_SIZE * VoxelResolution, left to right */
    float start[3] = {((float)id[0] * (float)CUBE) * v->res, ((float)id[1] * (float)CUBE) * v->res,
                      ((float)id[2] * (float)CUBE) * v->res};
    uint64_t updated = 0;
    for (int vid = 0; vid < NVOX; ++vid) {
        const float *o = v->offset[vid];
        float q[4];
        mat4_mul_p1(pose_inv, start[0] + o[0], start[1] + o[1], start[2] + o[2], q);
        int u = project(cam->fx, q[0], q[2], cam->cx);
        int w = project(cam->fy, q[1], q[2], cam->cy);
        if (w < 0 || w >= cam->height || u < 0 || u >= cam->width) continue;
        size_t pix = (size_t)w * cam->width + u;
        float d = depth_at(depth, is_u16, cam->depth_scale, pix);
        if (d <= 0) continue;
        float new_sdf = d - q[2];
        if (fabsf(new_sdf) < v->trunc) {
            ++updated;
            float c[3] = {rgb[3 * pix] / 255.0f, rgb[3 * pix + 1] / 255.0f, rgb[3 * pix + 2] / 255.0f};
            float *t = vox + 5 * vid;
            int valid = !(t[0] >= 1 || t[1] <= 0); /* TSDFVoxel.h:75-78 */
            if (valid) {
                /* TSDFVoxel.h:24-39 with other = (new_sdf, 1.0, c) */
                float wsum = t[1] + 1.0f;
                if (wsum != 0) {
                    float s = (t[1] * t[0] + 1.0f * new_sdf) / wsum;
                    float c0 = (t[1] * t[2] + 1.0f * c[0]) / wsum;
                    float c1 = (t[1] * t[3] + 1.0f * c[1]) / wsum;
                    float c2 = (t[1] * t[4] + 1.0f * c[2]) / wsum;
                    t[0] = s; t[2] = c0; t[3] = c1; t[4] = c2;
                } else { t[0] = 999; t[2] = t[3] = t[4] = -1; }
                t[1] = wsum;
            } else {
                t[0] = new_sdf; t[1] = 1.0f; t[2] = c[0]; t[3] = c[1]; t[4] = c[2];
            }
        }
    }
    return updated;
}

size_t orc_volume_integrate(orc_volume *v, const void *depth, int is_u16, const uint8_t *rgb,
                            const float pose[16], uint64_t *n_visited, uint64_t *n_updated) {
    size_t cap = 1 << 16, n;
    int32_t *ids = (int32_t *)malloc(cap * 3 * sizeof(int32_t));
    n = orc_volume_prepare_cubes(v, depth, is_u16, pose, ids, cap, NULL);
    if (n > cap) {
        /* re-run selection with a big enough list; blocks are already allocated so this is pure */
        cap = n; ids = (int32_t *)realloc(ids, cap * 3 * sizeof(int32_t));
        n = orc_volume_prepare_cubes(v, depth, is_u16, pose, ids, cap, NULL);
    }
    float pose_inv[16];
    orc_mat4_inverse(pose, pose_inv); /* Integrator.cpp:48 */
    uint64_t upd = 0;
    /* CubeHandler.cpp:205-208: serial in the reference; every selected block is owned by exactly one iteration */
#pragma omp parallel for num_threads(g_fusion_threads) schedule(dynamic, 32) reduction(+ : upd)
    for (size_t i = 0; i < n; ++i) {
        int64_t b = vol_find(v, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
        upd += integrate_block(v, b, depth, is_u16, rgb, pose_inv);
    }
    free(ids);
    if (n_visited) *n_visited = (uint64_t)n * NVOX;
    if (n_updated) *n_updated = upd;
    return n;
}

size_t orc_volume_export(const orc_volume *v, int32_t *keys, float *voxels, size_t cap) {
    size_t n = v->n < cap ? v->n : cap;
    if (keys) memcpy(keys, v->keys, n * 3 * sizeof(int32_t));
    if (voxels) memcpy(voxels, v->vox, n * NVOX * 5 * sizeof(float));
    return v->n;
}

void orc_volume_import(orc_volume *v, const int32_t *keys, const float *voxels, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        int64_t b = vol_find(v, keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]);
        if (b < 0) b = vol_add(v, keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]);
        memcpy(v->vox + (size_t)b * NVOX * 5, voxels + i * NVOX * 5, NVOX * 5 * sizeof(float));
    }
}

/* TSDFVoxel::operator+ (TSDFVoxel.h:24-39), general weights. */
static void voxel_add(float *a, const float *b) {
    if (a[1] == 0) { memcpy(a, b, 5 * sizeof(float)); return; }
    if (b[1] == 0) return;
    float w = a[1] + b[1];
    if (w != 0) {
        float s = (a[1] * a[0] + b[1] * b[0]) / w;
        float c0 = (a[1] * a[2] + b[1] * b[2]) / w;
        float c1 = (a[1] * a[3] + b[1] * b[3]) / w;
        float c2 = (a[1] * a[4] + b[1] * b[4]) / w;
        a[0] = s; a[2] = c0; a[3] = c1; a[4] = c2;
    } else { a[0] = 999; a[2] = a[3] = a[4] = -1; }
    a[1] = w;
}

int orc_volume_merge(orc_volume *dst, const orc_volume *src) {
    if (dst->res != src->res) return 1; /* CubeHandler.h:147-151 */
    for (size_t i = 0; i < src->n; ++i) {
        const int32_t *k = src->keys + 3 * i;
        const float *sv = src->vox + i * NVOX * 5;
        int64_t b = vol_find(dst, k[0], k[1], k[2]);
        if (b < 0) {
            b = vol_add(dst, k[0], k[1], k[2]);
            memcpy(dst->vox + (size_t)b * NVOX * 5, sv, NVOX * 5 * sizeof(float));
        } else {
            float *dv = dst->vox + (size_t)b * NVOX * 5;
            for (int j = 0; j < NVOX; ++j) voxel_add(dv + 5 * j, sv + 5 * j);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Volume resampling, point cloud and .map files (CubeHandler.h:40-128,199-338, CubeHandler.cpp:45-69) */
/* ------------------------------------------------------------------------------------------ */
static inline int floor_div8(int v) { return (int)floor((v + 0.0) / CUBE); } /* VoxelCube.h:63-67 */

static void set_default_voxel(float *t) { t[0] = 999; t[1] = 0; t[2] = t[3] = t[4] = -1; }

/* voxel of global integer voxel coordinate p in volume v (default voxel when the block is absent):
 * GetCubeID(Point3i) + GetVoxelID(Point3i) (VoxelCube.h:63-67,81-86) + cube_map.find */
static void fetch_voxel(const orc_volume *v, const int p[3], float out[5]) {
    int c[3] = {floor_div8(p[0]), floor_div8(p[1]), floor_div8(p[2])};
    int64_t b = vol_find(v, c[0], c[1], c[2]);
    if (b < 0) { set_default_voxel(out); return; }
    int vid = (p[0] - c[0] * CUBE) + (p[1] - c[1] * CUBE) * CUBE + (p[2] - c[2] * CUBE) * CUBE * CUBE;
    memcpy(out, v->vox + ((size_t)b * NVOX + vid) * 5, 5 * sizeof(float));
}

/* TSDFVoxel::operator*(float) (TSDFVoxel.h:56-67) */
static void voxel_scale(const float *a, float wgt, float *r) {
    if (wgt == 0 || a[1] == 0) { set_default_voxel(r); return; }
    r[1] = a[1] * wgt; r[0] = a[0] * wgt; r[2] = a[2] * wgt; r[3] = a[3] * wgt; r[4] = a[4] * wgt;
}
/* TSDFVoxel::add (TSDFVoxel.h:40-51) */
static void voxel_add_direct(const float *a, const float *b, float *r) {
    if (a[1] == 0) { memcpy(r, b, 5 * sizeof(float)); return; }
    if (b[1] == 0) { memcpy(r, a, 5 * sizeof(float)); return; }
    r[1] = a[1] + b[1]; r[0] = a[0] + b[0]; r[2] = a[2] + b[2]; r[3] = a[3] + b[3]; r[4] = a[4] + b[4];
}
/* one lerp stage of ReadVoxelInterpolate (VoxelCube.cpp:17-20 etc.):
 * ((a * (1 - t)).add(b * t)) / ((1 - t) * (a.weight != 0) + t * (b.weight != 0)), default if both empty */
static void interp_stage(const float *a, const float *b, float t, float *r) {
    if (!(a[1] != 0 || b[1] != 0)) { set_default_voxel(r); return; }
    float A[5], B[5], S[5];
    voxel_scale(a, 1 - t, A);
    voxel_scale(b, t, B);
    voxel_add_direct(A, B, S);
    float d = (1 - t) * (float)(a[1] != 0) + t * (float)(b[1] != 0);
    voxel_scale(S, 1 / d, r); /* operator/(w) = operator*(1 / w) (TSDFVoxel.h:68-71) */
}
/* VoxelCube.cpp:6-50 */
static void read_voxel_interpolate(const int n0[3], float vox8[8][5], const float pos[3], float res, float *out) {
    float xw = (pos[0] - n0[0] * res) / res, yw = (pos[1] - n0[1] * res) / res, zw = (pos[2] - n0[2] * res) / res;
    float r1[5], r2[5], r3[5], r4[5], z1[5], z2[5];
    interp_stage(vox8[0], vox8[1], xw, r1);
    interp_stage(vox8[2], vox8[3], xw, r2);
    interp_stage(r1, r2, yw, z1);
    interp_stage(vox8[4], vox8[5], xw, r3);
    interp_stage(vox8[6], vox8[7], xw, r4);
    interp_stage(r3, r4, yw, z2);
    interp_stage(z1, z2, zw, out);
}

/* CubeHandler::TransformNearest (CubeHandler.h:299-338) / Transform (:242-298).
 * Quirk kept: TransformNearest never copies c_para into the result, so the result volume has the
 * DEFAULT resolution 0.01 and its allocation pass (AddTransformedCubeNearest, :225-241, executed on
 * the result object) uses that resolution, while the fill pass uses the source's (this->c_para). */
orc_volume *orc_volume_transform(const orc_volume *src, const float T[16], int nearest) {
    float alloc_res = nearest ? 0.01f : src->res;
    orc_volume *dst = orc_volume_create(&src->cam, alloc_res, src->trunc, src->far_d, src->near_d);
    float Tinv[16];
    orc_mat4_inverse(T, Tinv);
    float half_a = alloc_res / 2;
    /* pass 1: allocation, with the RESULT's c_para */
    for (size_t b = 0; b < src->n; ++b) {
        const int32_t *id = src->keys + 3 * b;
        float start[3] = {((float)id[0] * (float)CUBE) * alloc_res, ((float)id[1] * (float)CUBE) * alloc_res,
                          ((float)id[2] * (float)CUBE) * alloc_res};
        for (int vid = 0; vid < NVOX; ++vid) {
            const float *o = dst->offset[vid];
            float q[4], np[3];
            mat4_mul_p1(T, start[0] + o[0], start[1] + o[1], start[2] + o[2], q);
            for (int c = 0; c < 3; ++c) np[c] = nearest ? q[c] / q[3] : q[c] / q[3] - half_a;
            int p0[3] = {(int)floorf(np[0] / alloc_res), (int)floorf(np[1] / alloc_res), (int)floorf(np[2] / alloc_res)};
            for (int k = 0; k < (nearest ? 1 : 8); ++k) {
                int p[3] = {p0[0] + (k & 1), p0[1] + ((k >> 1) & 1), p0[2] + ((k >> 2) & 1)};
                int c[3] = {floor_div8(p[0]), floor_div8(p[1]), floor_div8(p[2])};
                if (vol_find(dst, c[0], c[1], c[2]) < 0) vol_add(dst, c[0], c[1], c[2]);
            }
        }
    }
    /* pass 2: fill, with the SOURCE's c_para (this->c_para in the reference) */
    float half_s = src->res / 2;
    for (size_t b = 0; b < dst->n; ++b) {
        const int32_t *id = dst->keys + 3 * b;
        float start[3] = {((float)id[0] * (float)CUBE) * src->res, ((float)id[1] * (float)CUBE) * src->res,
                          ((float)id[2] * (float)CUBE) * src->res};
        for (int vid = 0; vid < NVOX; ++vid) {
            const float *o = src->offset[vid];
            float q[4], np[3], r[5];
            mat4_mul_p1(Tinv, start[0] + o[0], start[1] + o[1], start[2] + o[2], q);
            for (int c = 0; c < 3; ++c) np[c] = nearest ? q[c] / q[3] : q[c] / q[3] - half_s;
            int p0[3] = {(int)floorf(np[0] / src->res), (int)floorf(np[1] / src->res), (int)floorf(np[2] / src->res)};
            if (nearest) {
                fetch_voxel(src, p0, r);
            } else {
                float v8[8][5];
                for (int k = 0; k < 8; ++k) {
                    int p[3] = {p0[0] + (k & 1), p0[1] + ((k >> 1) & 1), p0[2] + ((k >> 2) & 1)};
                    fetch_voxel(src, p, v8[k]);
                }
                read_voxel_interpolate(p0, v8, np, src->res, r);
            }
            voxel_add(dst->vox + ((size_t)b * NVOX + vid) * 5, r); /* v_cube.voxels[voxel_id] += result */
        }
    }
    return dst;
}

float orc_volume_resolution(const orc_volume *v) { return v->res; }

/* CubeHandler::GetPointCloud (CubeHandler.cpp:45-69): blocks in insertion order here (the
 * reference's map order is unspecified), voxels in the reference's x,y,z loop nest order. */
size_t orc_volume_point_cloud(const orc_volume *v, float *xyz, float *colors, size_t cap) {
    size_t n = 0;
    float cube_res = CUBE * v->res; /* VoxelCube.h:150 */
    for (size_t b = 0; b < v->n; ++b) {
        const int32_t *id = v->keys + 3 * b;
        for (int x = 0; x < CUBE; ++x)
            for (int y = 0; y < CUBE; ++y)
                for (int z = 0; z < CUBE; ++z) {
                    int vid = x + y * CUBE + z * CUBE * CUBE;
                    const float *t = v->vox + ((size_t)b * NVOX + vid) * 5;
                    if (t[1] != 0 && fabsf(t[0]) < v->trunc) {
                        if (n < cap) {
                            float f = fabsf(t[0]) / v->trunc;
                            for (int c = 0; c < 3; ++c) {
                                xyz[3 * n + c] = id[c] * cube_res + v->offset[vid][c];
                                colors[3 * n + c] = f;
                            }
                        }
                        ++n;
                    }
                }
    }
    return n;
}

#include <stdio.h>
/* CubeHandler::WriteToFile (CubeHandler.h:113-128) + VoxelCube::WriteToBuffer (VoxelCube.h:128-148) */
int orc_volume_write_file(const orc_volume *v, const char *path) {
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    unsigned int size = (unsigned int)v->n;
    fwrite(&size, 4, 1, f); /* the count's raw bits live in a float slot */
    for (size_t b = 0; b < v->n; ++b) {
        float hdr[3] = {(float)v->keys[3 * b], (float)v->keys[3 * b + 1], (float)v->keys[3 * b + 2]};
        fwrite(hdr, 4, 3, f);
        for (int i = 0; i < NVOX; ++i) {
            const float *t = v->vox + ((size_t)b * NVOX + i) * 5;
            if (fabsf(t[0]) < 1 && t[1] != 0) {
                float rec[6] = {(float)i, t[0], t[1], t[2], t[3], t[4]};
                fwrite(rec, 4, 6, f);
            }
        }
        float end = -2.0f;
        fwrite(&end, 4, 1, f);
    }
    fclose(f);
    return 0;
}

/* CubeHandler::ReadFromFile (CubeHandler.h:40-69) / ReadFromFileFloat (:73-109) */
int orc_volume_read_file(orc_volume *v, const char *path, int legacy_float) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    size_t nf = (size_t)len / 4;
    float *buf = (float *)malloc((nf + 1) * 4);
    if (fread(buf, 4, nf, f) != nf) { fclose(f); free(buf); return 1; }
    fclose(f);
    orc_volume_clear(v);
    unsigned int count;
    size_t ptr;
    if (legacy_float) { count = (unsigned int)buf[1]; ptr = 2; }
    else { memcpy(&count, buf, 4); ptr = 1; }
    for (unsigned int c = 0; c < count; ++c) {
        int x = (int)buf[ptr], y = (int)buf[ptr + 1], z = (int)buf[ptr + 2];
        ptr += 3;
        int64_t b = vol_find(v, x, y, z);
        if (b < 0) b = vol_add(v, x, y, z);
        float *vox = v->vox + (size_t)b * NVOX * 5;
        for (int i = 0; i < NVOX; ++i) set_default_voxel(vox + 5 * i); /* cube_map[id] = VoxelCube(id) */
        if (!legacy_float) { /* VoxelCube.h:153-166 */
            while (buf[ptr] != -2.0f) {
                int i = (int)buf[ptr++];
                float *t = vox + 5 * i;
                t[0] = buf[ptr++]; t[1] = buf[ptr++]; t[2] = buf[ptr++]; t[3] = buf[ptr++]; t[4] = buf[ptr++];
            }
            ptr++;
        } else { /* VoxelCube.h:168-193 */
            ptr++;
            while (buf[ptr] != -2.0f) {
                int i = (int)buf[ptr++];
                vox[5 * i] = buf[ptr++]; vox[5 * i + 1] = buf[ptr++];
            }
            ptr++;
            size_t cnt = (size_t)buf[ptr++];
            for (size_t k = 0; k < cnt; ++k) {
                int i = (int)buf[ptr++];
                float *t = vox + 5 * i;
                t[2] = (float)(buf[ptr++] / 255.0); t[3] = (float)(buf[ptr++] / 255.0); t[4] = (float)(buf[ptr++] / 255.0);
                float cw = buf[ptr++];
                t[2] = t[2] / cw; t[3] = t[3] / cw; t[4] = t[4] / cw;
            }
        }
    }
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Ray casting.  NO REFERENCE COUNTERPART (the reference has no raycast although         */
/* north_star names one).  This is the CPU restatement of the definition in include/onepiece_hip.h, */
/* validated against the analytic synthetic scene, not against OnePiece.                            */
/* ------------------------------------------------------------------------------------------ */
/* Definition (include/onepiece_hip.h, onepiece_amd/csrc/raycast.hip):                                                 */
/*   ray of pixel (u, v): origin = pose translation, direction d = R ((u - cx) / fx, (v - cy) / fy, 1);                */
/*   lattice t_k = near + k * res (t_k <= far), p_k = origin + t_k d;                                                  */
/*   s_k = trilinear sdf over the 8 voxel centres around p_k (g = p * (1 / res) - 0.5), valid when all 8 have w > 0;   */
/*   hit = the smallest k >= 1 with s_(k-1) valid and > 0 and s_k valid and <= 0 (and a positive depth):               */
/*   depth = t_(k-1) + (t_k - t_(k-1)) * s_(k-1) / (s_(k-1) - s_k); 0 = no hit.                                        */
/* This walks EVERY lattice point of every ray in order (the HIP path marches block-major and takes a minimum).        */
/* trilinear sdf / colour at world point p; needs all 8 surrounding voxel centres observed (w > 0) */
static int sample_trilinear(const orc_volume *v, float inv_res, const float p[3], float *sdf, float col[3]) {
    float g[3], f[3];
    int i0[3];
    for (int c = 0; c < 3; ++c) {
        g[c] = p[c] * inv_res - 0.5f;
        float fl = floorf(g[c]);
        i0[c] = (int)fl;
        f[c] = g[c] - fl;
    }
    /* (the base voxel is corner 0: if its block is absent the sample is invalid whatever the other corners hold) */
    if (vol_find(v, i0[0] >> 3, i0[1] >> 3, i0[2] >> 3) < 0) return 0;
    float acc = 0, ac[3] = {0, 0, 0};
    for (int k = 0; k < 8; ++k) {
        int q[3] = {i0[0] + (k & 1), i0[1] + ((k >> 1) & 1), i0[2] + ((k >> 2) & 1)};
        float t[5];
        fetch_voxel(v, q, t);
        if (!(t[1] > 0)) return 0;
        float wx = (k & 1) ? f[0] : 1.0f - f[0], wy = (k & 2) ? f[1] : 1.0f - f[1], wz = (k & 4) ? f[2] : 1.0f - f[2];
        float w = (wx * wy) * wz;
        acc += w * t[0];
        if (col) for (int c = 0; c < 3; ++c) ac[c] += w * t[2 + c];
    }
    *sdf = acc;
    if (col) { col[0] = ac[0]; col[1] = ac[1]; col[2] = ac[2]; }
    return 1;
}

void orc_volume_raycast(const orc_volume *v, const orc_camera *cam, const float pose[16], float *depth_out, float *normals_out,
                        float *colors_out) {
    const float inv_res = 1.0f / v->res;
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < cam->height; ++py)
        for (int px = 0; px < cam->width; ++px) {
            const size_t pix = (size_t)py * cam->width + px;
            const float dcx = ((float)px - cam->cx) / cam->fx, dcy = ((float)py - cam->cy) / cam->fy;
            float dir[3], org[3] = {pose[3], pose[7], pose[11]};
            for (int r = 0; r < 3; ++r) dir[r] = (pose[r * 4] * dcx + pose[r * 4 + 1] * dcy) + pose[r * 4 + 2];
            float t_prev = 0, s_prev = 0, hit = 0;
            int have_prev = 0;
            for (int k = 0;; ++k) {
                const float t = v->near_d + (float)k * v->res;
                if (!(t <= v->far_d) || k > (1 << 24)) break;
                float p[3] = {org[0] + t * dir[0], org[1] + t * dir[1], org[2] + t * dir[2]}, sdf;
                if (sample_trilinear(v, inv_res, p, &sdf, NULL)) {
                    if (have_prev && s_prev > 0 && sdf <= 0) {
                        const float depth = t_prev + (t - t_prev) * (s_prev / (s_prev - sdf));
                        if (depth > 0) { hit = depth; break; }
                    }
                    have_prev = 1; s_prev = sdf; t_prev = t;
                } else
                    have_prev = 0;
            }
            depth_out[pix] = hit;
            float n[3] = {0, 0, 0}, c[3] = {0, 0, 0};
            if (hit > 0) {
                float p[3] = {org[0] + hit * dir[0], org[1] + hit * dir[1], org[2] + hit * dir[2]}, s0, h = 0.5f * v->res;
                if (!sample_trilinear(v, inv_res, p, &s0, c)) { c[0] = c[1] = c[2] = 0; }
                int ok = 1;
                for (int a = 0; a < 3 && ok; ++a) {
                    float pp[3] = {p[0], p[1], p[2]}, pm[3] = {p[0], p[1], p[2]}, sp = 0, sm = 0;
                    pp[a] += h; pm[a] -= h;
                    ok = sample_trilinear(v, inv_res, pp, &sp, NULL) && sample_trilinear(v, inv_res, pm, &sm, NULL);
                    n[a] = sp - sm;
                }
                float l2 = sum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
                if (ok && l2 > 0) { float l = sqrtf(l2); n[0] /= l; n[1] /= l; n[2] /= l; } else { n[0] = n[1] = n[2] = 0; }
            }
            if (normals_out) { normals_out[3 * pix] = n[0]; normals_out[3 * pix + 1] = n[1]; normals_out[3 * pix + 2] = n[2]; }
            if (colors_out) { colors_out[3 * pix] = c[0]; colors_out[3 * pix + 1] = c[1]; colors_out[3 * pix + 2] = c[2]; }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Registration                                                                                */
/* ------------------------------------------------------------------------------------------ */

size_t orc_load_from_depth(const orc_camera *cam, const void *depth, int is_u16, float *xyz) {
    size_t cnt = 0;
    for (int i = 0; i < cam->height; ++i)
        for (int j = 0; j < cam->width; ++j) {
            float z = depth_at(depth, is_u16, cam->depth_scale, (size_t)i * cam->width + j);
            if (z > 0) {
                xyz[3 * cnt] = (j - cam->cx) * z / cam->fx;
                xyz[3 * cnt + 1] = (i - cam->cy) * z / cam->fy;
                xyz[3 * cnt + 2] = z;
                ++cnt;
            }
        }
    return cnt;
}

/* Sophus SE3::exp (3rdparty/Sophus/sophus/se3.hpp:468-489, so3.hpp:388-414): x[0:3] = upsilon,
 * x[3:6] = omega.  Evaluated in double with the closed forms, rounded to float at the end
 * (the reference evaluates in float through a unit quaternion; agreement is ~1e-7). */
void orc_se3_exp(const float x[6], float T[16]) {
    double w[3] = {x[3], x[4], x[5]}, u[3] = {x[0], x[1], x[2]};
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double a, b, c; /* R = I + a W + b W^2 ; V = I + b W + c W^2 */
    if (th < 1e-5) {
        a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0;
    } else {
        a = sin(th) / th; b = (1.0 - cos(th)) / th2; c = (th - sin(th)) / (th2 * th);
    }
    double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += W[i * 3 + k] * W[k * 3 + j];
            W2[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i) {
        double t = 0;
        for (int j = 0; j < 3; ++j) {
            double I = i == j ? 1.0 : 0.0;
            T[i * 4 + j] = (float)(I + a * W[i * 3 + j] + b * W2[i * 3 + j]);
            t += (I + b * W[i * 3 + j] + c * W2[i * 3 + j]) * u[j];
        }
        T[i * 4 + 3] = (float)t;
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

/* Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (double). A is destroyed,
 * V columns are eigenvectors, diag(A) the eigenvalues. */
static void jacobi_sym(double *A, double *V, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

/* x = JacobiSVD(JTJ).solve(-JTr) (ICP.cpp:137-138).  JTJ is symmetric PSD, so its SVD is its
 * eigen-decomposition; Eigen's solve() drops singular values <= eps*diagSize*max (float eps). */
void orc_solve6(const float JTJ[36], const float JTr[6], float x[6]) {
    double A[36], V[36], y[6];
    for (int i = 0; i < 36; ++i) A[i] = JTJ[i];
    for (int i = 0; i < 6; ++i) /* symmetrise against float accumulation noise */
        for (int j = i + 1; j < 6; ++j) A[i * 6 + j] = A[j * 6 + i] = 0.5 * (A[i * 6 + j] + A[j * 6 + i]);
    jacobi_sym(A, V, 6);
    double smax = 0;
    for (int i = 0; i < 6; ++i) if (fabs(A[i * 7]) > smax) smax = fabs(A[i * 7]);
    double thr = smax * 6.0 * (double)FLT_EPSILON;
    for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int i = 0; i < 6; ++i) s += V[i * 6 + k] * (-(double)JTr[i]);
        y[k] = fabs(A[k * 7]) > thr ? s / A[k * 7] : 0.0;
    }
    for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += V[i * 6 + k] * y[k];
        x[i] = (float)s;
    }
}

/* Geometry.cpp:107-151 (Kabsch).  Sums in float in the reference's order; the 3x3 SVD is
 * obtained from the eigen-decomposition of W^T W in double (R = V U^T with det fix). */
void orc_kabsch(const float *pairs, size_t n, float T[16]) {
    float ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0}, W[9] = {0};
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) { ms[c] += pairs[6 * i + c]; mt[c] += pairs[6 * i + 3 + c]; }
    for (int c = 0; c < 3; ++c) { ms[c] /= (float)n; mt[c] /= (float)n; }
    for (size_t i = 0; i < n; ++i) {
        float a[3], b[3];
        for (int c = 0; c < 3; ++c) { a[c] = pairs[6 * i + c] - ms[c]; b[c] = pairs[6 * i + 3 + c] - mt[c]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) W[r * 3 + c] += a[r] * b[c];
    }
    /* W = U S V^T.  R = V U^T (det-fixed) is the orthogonal polar factor of W^T:
     * W^T W = V S^2 V^T; U = W V S^-1. */
    double A[9], V[9], S[3], U[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)W[k * 3 + i] * (double)W[k * 3 + j];
            A[i * 3 + j] = s;
        }
    jacobi_sym(A, V, 3);
    /* sort singular values descending (JacobiSVD order) */
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (A[ord[j] * 4] > A[ord[i] * 4]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Vs[9];
    for (int k = 0; k < 3; ++k) {
        S[k] = sqrt(A[ord[k] * 4] > 0 ? A[ord[k] * 4] : 0);
        for (int i = 0; i < 3; ++i) Vs[i * 3 + k] = V[i * 3 + ord[k]];
    }
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) {
            double s = 0;
            for (int j = 0; j < 3; ++j) s += (double)W[i * 3 + j] * Vs[j * 3 + k];
            U[i * 3 + k] = S[k] > 1e-300 ? s / S[k] : 0;
        }
    /* if the smallest singular value is ~0, complete U's last column by a cross product */
    if (S[2] <= 1e-12 * (S[0] > 0 ? S[0] : 1)) {
        U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
        U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
        U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    }
    double R[9];
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += Vs[i * 3 + k] * U[j * 3 + k];
                R[i * 3 + j] = s;
            }
        double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                     R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (det >= 0) break;
        for (int i = 0; i < 3; ++i) Vs[i * 3 + 2] = -Vs[i * 3 + 2]; /* Geometry.cpp:139-144 */
    }
    memset(T, 0, 16 * sizeof(float));
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) { T[i * 4 + j] = (float)R[i * 3 + j]; s += R[i * 3 + j] * ms[j]; }
        T[i * 4 + 3] = (float)(mt[i] - s);
    }
    T[15] = 1;
}

/* Diagnostic switch (NOT reference behaviour): accumulate JTJ/JTr in double instead of the
 * reference's float, to measure how much of a pose difference is the reference's own rounding. */
static int g_accumulate_double = 0;
void orc_set_accumulate_double(int on) { g_accumulate_double = on; }

/* ICP.cpp:108-144 */
void orc_p2plane_step(const float *src, const float *tgt, const float *tgt_n,
                      const int32_t *inliers, size_t n, float T[16], float JTJ[36], float JTr[6]) {
    float jtj[36] = {0}, jtr[6] = {0};
    double djtj[36] = {0}, djtr[6] = {0};
    for (size_t i = 0; i < n; ++i) {
        const float *s = src + 3 * inliers[2 * i], *t = tgt + 3 * inliers[2 * i + 1];
        const float *nn = tgt_n + 3 * inliers[2 * i + 1];
        double r = (double)(dot3(nn, s) - dot3(nn, t)); /* float dots, float difference -> double */
        float row[6] = {nn[0], nn[1], nn[2], s[1] * nn[2] - s[2] * nn[1], s[2] * nn[0] - s[0] * nn[2],
                        s[0] * nn[1] - s[1] * nn[0]};
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) { jtj[a * 6 + b] += row[a] * row[b]; djtj[a * 6 + b] += (double)(row[a] * row[b]); }
            jtr[a] += (float)r * row[a];
            djtr[a] += (double)((float)r * row[a]);
        }
    }
    if (g_accumulate_double) {
        for (int a = 0; a < 36; ++a) jtj[a] = (float)djtj[a];
        for (int a = 0; a < 6; ++a) jtr[a] = (float)djtr[a];
    }
    float x[6];
    orc_solve6(jtj, jtr, x);
    orc_se3_exp(x, T);
    if (JTJ) memcpy(JTJ, jtj, sizeof(jtj));
    if (JTr) memcpy(JTr, jtr, sizeof(jtr));
}

/* ---- nearest neighbours: a restatement of nanoflann 1.3.2 as geometry::KDTree drives it ------------------------------
 * (third-party dependency of the reference, vendored under 3rdparty/nanoflann/include/nanoflann.hpp; KDTree.h:62-98 builds a
 * KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, cloud>, cloud, 3> with leaf size 10, KDTree.h:171-190 / :230-255 call
 * knnSearch with the default SearchParams, i.e. eps = 0).  The restatement follows the published algorithm step for step --
 * bounding boxes, middle split with its spread test, the three-way plane split and its permutation of the index array,
 * near-child-first descent with the per-dimension lower bound, and the result set that keeps the FIRST of equally distant
 * candidates -- because the answer on exactly equidistant candidates (duplicated points, lattices) is decided by nothing else:
 * tests/golden/nanoflann_golden.json (generated with the real header) holds such cases, and the indices must agree there too.
 * All arithmetic is float, as in the instantiation above (ElementType = DistanceType = float). */
typedef struct { int child1, child2; size_t left, right; int divfeat; float divlow, divhigh; } nf_node;
typedef struct { const float *pts; size_t *vind; nf_node *nodes; int n_nodes; size_t n; float root_lo[3], root_hi[3]; } nf_tree;

static void nf_min_max(const nf_tree *t, const size_t *ind, size_t count, int element, float *mn, float *mx) { /* computeMinMax */
    *mn = *mx = t->pts[3 * ind[0] + element];
    for (size_t i = 1; i < count; ++i) {
        float v = t->pts[3 * ind[i] + element];
        if (v < *mn) *mn = v;
        if (v > *mx) *mx = v;
    }
}
static void nf_plane_split(const nf_tree *t, size_t *ind, size_t count, int cutfeat, float cutval, size_t *lim1, size_t *lim2) { /* planeSplit */
    size_t left = 0, right = count - 1;
    for (;;) {
        while (left <= right && t->pts[3 * ind[left] + cutfeat] < cutval) ++left;
        while (right && left <= right && t->pts[3 * ind[right] + cutfeat] >= cutval) --right;
        if (left > right || !right) break;
        size_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim1 = left;
    right = count - 1;
    for (;;) {
        while (left <= right && t->pts[3 * ind[left] + cutfeat] <= cutval) ++left;
        while (right && left <= right && t->pts[3 * ind[right] + cutfeat] > cutval) --right;
        if (left > right || !right) break;
        size_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim2 = left;
}
static void nf_middle_split(const nf_tree *t, size_t *ind, size_t count, size_t *index, int *cutfeat, float *cutval, const float lo[3], const float hi[3]) { /* middleSplit_ */
    const float eps = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int i = 1; i < 3; ++i) { float span = hi[i] - lo[i]; if (span > max_span) max_span = span; }
    float max_spread = -1;
    *cutfeat = 0;
    for (int i = 0; i < 3; ++i) {
        float span = hi[i] - lo[i];
        if (span > (1 - eps) * max_span) {
            float mn, mx;
            nf_min_max(t, ind, count, i, &mn, &mx);
            float spread = mx - mn;
            if (spread > max_spread) { *cutfeat = i; max_spread = spread; }
        }
    }
    float split_val = (lo[*cutfeat] + hi[*cutfeat]) / 2;
    float mn, mx;
    nf_min_max(t, ind, count, *cutfeat, &mn, &mx);
    if (split_val < mn) *cutval = mn;
    else if (split_val > mx) *cutval = mx;
    else *cutval = split_val;
    size_t lim1, lim2;
    nf_plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
    if (lim1 > count / 2) *index = lim1;
    else if (lim2 < count / 2) *index = lim2;
    else *index = count / 2;
}
static int nf_divide(nf_tree *t, size_t left, size_t right, float lo[3], float hi[3]) { /* divideTree; lo/hi in: the cell, out: the box of the points */
    int id = t->n_nodes++;
    if (right - left <= 10) { /* leaf_max_size: KDTree(int _max_leaf = 10) */
        t->nodes[id].child1 = t->nodes[id].child2 = -1;
        t->nodes[id].left = left; t->nodes[id].right = right;
        for (int i = 0; i < 3; ++i) lo[i] = hi[i] = t->pts[3 * t->vind[left] + i];
        for (size_t k = left + 1; k < right; ++k)
            for (int i = 0; i < 3; ++i) {
                float v = t->pts[3 * t->vind[k] + i];
                if (lo[i] > v) lo[i] = v;
                if (hi[i] < v) hi[i] = v;
            }
        return id;
    }
    size_t idx; int cutfeat; float cutval;
    nf_middle_split(t, t->vind + left, right - left, &idx, &cutfeat, &cutval, lo, hi);
    float llo[3], lhi[3], rlo[3], rhi[3];
    memcpy(llo, lo, sizeof(llo)); memcpy(lhi, hi, sizeof(lhi)); memcpy(rlo, lo, sizeof(rlo)); memcpy(rhi, hi, sizeof(rhi));
    lhi[cutfeat] = cutval;
    int c1 = nf_divide(t, left, left + idx, llo, lhi);
    rlo[cutfeat] = cutval;
    int c2 = nf_divide(t, left + idx, right, rlo, rhi);
    nf_node *nd = &t->nodes[id]; /* (the array does not move: allocated for 2 n nodes up front) */
    nd->child1 = c1; nd->child2 = c2; nd->divfeat = cutfeat; nd->left = left; nd->right = right;
    nd->divlow = lhi[cutfeat]; nd->divhigh = rlo[cutfeat];
    for (int i = 0; i < 3; ++i) { lo[i] = llo[i] < rlo[i] ? llo[i] : rlo[i]; hi[i] = lhi[i] > rhi[i] ? lhi[i] : rhi[i]; }
    return id;
}
static void nf_build(nf_tree *t, const float *pts, size_t n) { /* buildIndex */
    t->pts = pts; t->n = n; t->n_nodes = 0;
    t->vind = (size_t *)malloc((n + 1) * sizeof(size_t));
    t->nodes = (nf_node *)malloc((2 * n + 2) * sizeof(nf_node));
    for (size_t i = 0; i < n; ++i) t->vind[i] = i;
    if (!n) return;
    for (int i = 0; i < 3; ++i) t->root_lo[i] = t->root_hi[i] = pts[i];   /* computeBoundingBox */
    for (size_t k = 1; k < n; ++k)
        for (int i = 0; i < 3; ++i) {
            if (pts[3 * k + i] < t->root_lo[i]) t->root_lo[i] = pts[3 * k + i];
            if (pts[3 * k + i] > t->root_hi[i]) t->root_hi[i] = pts[3 * k + i];
        }
    nf_divide(t, 0, n, t->root_lo, t->root_hi);
}
static void nf_free(nf_tree *t) { free(t->vind); free(t->nodes); }

/* KNNResultSet: ascending distances; an equally distant newcomer goes BEHIND the entries already there (and is dropped when the set is full) */
typedef struct { float d[64]; int i[64]; int n, k; } knn_set;
static void knn_init(knn_set *s, int k) { s->n = 0; s->k = k; s->d[k - 1] = FLT_MAX; }
static void knn_add(knn_set *s, float dist, int index) { /* addPoint */
    int i;
    for (i = s->n; i > 0; --i) {
        if (s->d[i - 1] > dist) { if (i < s->k) { s->d[i] = s->d[i - 1]; s->i[i] = s->i[i - 1]; } }
        else break;
    }
    if (i < s->k) { s->d[i] = dist; s->i[i] = index; }
    if (s->n < s->k) s->n++;
}
static void nf_search_level(const nf_tree *t, int id, const float *q, float mindistsq, float dists[3], knn_set *s) { /* searchLevel, epsError = 1 */
    const nf_node *nd = &t->nodes[id];
    if (nd->child1 < 0) {
        float worst = s->d[s->k - 1]; /* read once per leaf, as nanoflann does */
        for (size_t i = nd->left; i < nd->right; ++i) {
            const float *p = t->pts + 3 * t->vind[i];
            float dist = 0; /* L2_Simple_Adaptor::evalMetric: result += diff * diff over the dimensions */
            for (int c = 0; c < 3; ++c) { float diff = q[c] - p[c]; dist += diff * diff; }
            if (dist < worst) knn_add(s, dist, (int)t->vind[i]);
        }
        return;
    }
    int idx = nd->divfeat;
    float val = q[idx], diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int best, other; float cut;
    if ((diff1 + diff2) < 0) { best = nd->child1; other = nd->child2; cut = (val - nd->divhigh) * (val - nd->divhigh); }
    else { best = nd->child2; other = nd->child1; cut = (val - nd->divlow) * (val - nd->divlow); }
    nf_search_level(t, best, q, mindistsq, dists, s);
    float dst = dists[idx];
    mindistsq = mindistsq + cut - dst;
    dists[idx] = cut;
    if (mindistsq * 1.0f <= s->d[s->k - 1]) nf_search_level(t, other, q, mindistsq, dists, s);
    dists[idx] = dst;
}
/* knnSearch -> number found; s holds them in nanoflann's order */
static int nf_knn(const nf_tree *t, const float *q, int k, knn_set *s) { /* findNeighbors */
    knn_init(s, k);
    if (!t->n) return 0;
    float dists[3] = {0, 0, 0}, distsq = 0;
    for (int i = 0; i < 3; ++i) { /* computeInitialDistances */
        if (q[i] < t->root_lo[i]) { dists[i] = (q[i] - t->root_lo[i]) * (q[i] - t->root_lo[i]); distsq += dists[i]; }
        if (q[i] > t->root_hi[i]) { dists[i] = (q[i] - t->root_hi[i]) * (q[i] - t->root_hi[i]); distsq += dists[i]; }
    }
    nf_search_level(t, 0, q, distsq, dists, s);
    return s->n;
}

/* geometry::FitPlane (Geometry.cpp:172-218): float sums in neighbour order, W / n, normal = the
 * singular vector of the smallest singular value (JacobiSVD U.col(2); sign is whatever the SVD
 * yields in the reference -- undetermined here), normalised. */
static void fit_plane_normal(const float *pts, const int *idx, int n, float normal[3]) {
    if (n < 3) { normal[0] = normal[1] = normal[2] = 0; return; }
    float sum[3] = {0, 0, 0}, mean[3], W[9] = {0};
    for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) sum[c] += pts[3 * idx[i] + c];
    for (int c = 0; c < 3; ++c) mean[c] = sum[c] / (float)n;
    for (int i = 0; i < n; ++i) {
        float d[3];
        for (int c = 0; c < 3; ++c) d[c] = pts[3 * idx[i] + c] - mean[c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) W[r * 3 + c] += d[r] * d[c];
    }
    double A[9], V[9];
    for (int k = 0; k < 9; ++k) A[k] = (double)(W[k] / (float)n);
    for (int r = 0; r < 3; ++r) for (int c = r + 1; c < 3; ++c) A[r * 3 + c] = A[c * 3 + r] = 0.5 * (A[r * 3 + c] + A[c * 3 + r]);
    jacobi_sym(A, V, 3);
    int m = 0;
    if (A[4] < A[m * 4]) m = 1;
    if (A[8] < A[m * 4]) m = 2;
    float v[3] = {(float)V[0 * 3 + m], (float)V[1 * 3 + m], (float)V[2 * 3 + m]};
    float z = sum3(v[0] * v[0], v[1] * v[1], v[2] * v[2]);
    if (z > 0) { float l = sqrtf(z); v[0] /= l; v[1] /= l; v[2] /= l; }
    normal[0] = v[0]; normal[1] = v[1]; normal[2] = v[2];
}

/* PointCloud::EstimateNormals(radius, knn) (PointCloud.cpp:102-144; KnnRadiusSearch KDTree.h:230-255:
 * k nearest, then the prefix whose SQUARED distance is <= radius). */
void orc_estimate_normals(const float *pts, size_t n, float radius, int knn, float *normals) {
    if (knn > 64) knn = 64;
    if (knn < 1) { memset(normals, 0, n * 3 * sizeof(float)); return; }
    nf_tree t;
    nf_build(&t, pts, n);
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < (long)n; ++i) {
        knn_set s;
        nf_knn(&t, pts + 3 * i, knn, &s);
        int used = 0;
        while (used < s.n && !(s.d[used] > radius)) ++used;
        fit_plane_normal(pts, s.i, used, normals + 3 * i);
    }
    nf_free(&t);
}

/* The search above, exposed for tests/test_oracle_golden.py's comparison with the real nanoflann (tests/golden/nanoflann_golden.json):
 * k == 1 is ICP's search, k > 1 EstimateNormals'.  idx / d2 hold k entries per query (-1 / -1.0f beyond `found`). */
void orc_knn_search(const float *tgt, size_t n_tgt, const float *queries, size_t n_q, int k, int32_t *idx, float *d2, int32_t *found) {
    if (k > 64) k = 64;
    if (k < 1) k = 1;
    nf_tree t;
    nf_build(&t, tgt, n_tgt);
    for (size_t i = 0; i < n_q; ++i) {
        for (int j = 0; j < k; ++j) { idx[i * k + j] = -1; d2[i * k + j] = -1.0f; }
        knn_set s;
        found[i] = nf_knn(&t, queries + 3 * i, k, &s);
        for (int j = 0; j < s.n; ++j) { idx[i * k + j] = s.i[j]; d2[i * k + j] = s.d[j]; }
    }
    nf_free(&t);
}

/* ICP.cpp:9-30 */
static double count_inliers(const float *src, const float *tgt, const int *corr, size_t n,
                            const float T[16], double threshold, int32_t *inliers, size_t *n_inl) {
    double sum_error = 0, thr2 = threshold * threshold;
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (corr[i] == -1) continue;
        const float *s = src + 3 * i, *t = tgt + 3 * corr[i];
        float d[3];
        for (int r = 0; r < 3; ++r) /* (R*s + t) - target, Matrix3f*Vector3f coefficient = p0+(p1+p2) */
            d[r] = (sum3(T[r * 4] * s[0], T[r * 4 + 1] * s[1], T[r * 4 + 2] * s[2]) + T[r * 4 + 3]) - t[r];
        double e = (double)sum3(d[0] * d[0], d[1] * d[1], d[2] * d[2]);
        if (e < thr2) {
            inliers[2 * m] = (int32_t)i; inliers[2 * m + 1] = corr[i]; ++m;
            sum_error += e;
        }
    }
    *n_inl = m;
    return sqrt(sum_error / (double)m);
}

static void mat4_mul(const float *A, const float *B, float *C) { /* Matrix4f*Matrix4f, column accumulate */
    float out[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            out[r * 4 + c] = ((A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c]) + A[r * 4 + 2] * B[8 + c]) +
                             A[r * 4 + 3] * B[12 + c];
    memcpy(C, out, sizeof(out));
}

int orc_icp(int point_to_plane, const float *src, size_t n_src, const float *tgt, size_t n_tgt,
            const float *tgt_normals, const float init_T[16], int max_iter, double threshold,
            orc_icp_result *res, int32_t *inlier_pairs, int32_t *per_iter_inliers,
            float *per_iter_T) {
    if (point_to_plane && !tgt_normals) return 1; /* ICP.cpp:159-163 */
    nf_tree t;
    nf_build(&t, tgt, n_tgt);
    float start_T[16];
    memcpy(start_T, init_T, sizeof(start_T));
    int *corr = (int *)malloc(n_src * sizeof(int));
    float *tp = (float *)malloc(n_src * 3 * sizeof(float));
    int32_t *inl = (int32_t *)malloc(n_src * 2 * sizeof(int32_t));
    float *pairs = (float *)malloc(n_src * 6 * sizeof(float));
    size_t n_inl = 0;
    for (size_t i = 0; i < n_src; ++i) corr[i] = -1; /* ICP.cpp:177: what the final CountInliers sees when max_iteration is 0 */
    for (int it = 0; it < max_iter; ++it) {
        for (size_t i = 0; i < n_src; ++i) { /* ICP.cpp:182-183, Geometry.cpp:19-27 */
            float q[4];
            mat4_mul_p1(start_T, src[3 * i], src[3 * i + 1], src[3 * i + 2], q);
            tp[3 * i] = q[0] / q[3]; tp[3 * i + 1] = q[1] / q[3]; tp[3 * i + 2] = q[2] / q[3];
        }
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)n_src; ++i) { /* ICP.cpp:184-192 */
            knn_set s1; /* indices.size() > 0 ? indices[0] : -1 (ICP.cpp:69-72, :189-192) */
            corr[i] = nf_knn(&t, tp + 3 * i, 1, &s1) > 0 ? s1.i[0] : -1;
        }
        count_inliers(src, tgt, corr, n_src, start_T, threshold, inl, &n_inl); /* ICP.cpp:193 */
        float tmp_T[16];
        if (point_to_plane) {
            orc_p2plane_step(tp, tgt, tgt_normals, inl, n_inl, tmp_T, NULL, NULL); /* :195 */
        } else {
            for (size_t i = 0; i < n_inl; ++i) { /* ICP.cpp:73-78 */
                memcpy(pairs + 6 * i, tp + 3 * inl[2 * i], 12);
                memcpy(pairs + 6 * i + 3, tgt + 3 * inl[2 * i + 1], 12);
            }
            orc_kabsch(pairs, n_inl, tmp_T);
        }
        mat4_mul(tmp_T, start_T, start_T); /* :198 */
        if (per_iter_inliers) per_iter_inliers[it] = (int32_t)n_inl;
        if (per_iter_T) memcpy(per_iter_T + 16 * it, start_T, sizeof(start_T));
    }
    /* ICP.cpp:206-221.  corr is the last iteration's NN set (not recomputed). */
    res->rmse = count_inliers(src, tgt, corr, n_src, start_T, threshold, inl, &n_inl);
    res->n_inliers = n_inl;
    res->iterations = max_iter;
    memcpy(res->last_T, start_T, sizeof(start_T));
    for (size_t i = 0; i < n_inl; ++i) {
        memcpy(pairs + 6 * i, src + 3 * inl[2 * i], 12);
        memcpy(pairs + 6 * i + 3, tgt + 3 * inl[2 * i + 1], 12);
    }
    orc_kabsch(pairs, n_inl, res->T);
    if (inlier_pairs) memcpy(inlier_pairs, inl, n_inl * 2 * sizeof(int32_t));
    free(corr); free(tp); free(inl); free(pairs); nf_free(&t);
    return 0;
}

/* ======================= dense RGB-D tracker (Odometry/) ================================== */

/* Eigen 3.3.7 LU/InverseImpl.h:126-170 (compute_inverse<Matrix3f>): cofactors of column 0,
 * det = 3-term redux a0+(a1+a2), every entry = cofactor * (1/det).  Row-major in/out. */
void orc_mat3_inverse(const float m[9], float out[9]) {
#define M3(i, j) m[((i) % 3) * 3 + ((j) % 3)]
#define COF(i, j) (M3((i) + 1, (j) + 1) * M3((i) + 2, (j) + 2) - M3((i) + 1, (j) + 2) * M3((i) + 2, (j) + 1))
    float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
    float det = sum3(c0 * m[0], c1 * m[3], c2 * m[6]);
    float invdet = 1.0f / det;
    out[0] = c0 * invdet; out[1] = c1 * invdet; out[2] = c2 * invdet;
    out[3] = COF(0, 1) * invdet; out[4] = COF(1, 1) * invdet; out[5] = COF(2, 1) * invdet;
    out[6] = COF(0, 2) * invdet; out[7] = COF(1, 2) * invdet; out[8] = COF(2, 2) * invdet;
#undef COF
#undef M3
}

static void mat3_mul(const float *A, const float *B, float *C) { /* Matrix3f * Matrix3f, 3-term redux */
    float o[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[r * 3 + c] = sum3(A[r * 3] * B[c], A[r * 3 + 1] * B[3 + c], A[r * 3 + 2] * B[6 + c]);
    memcpy(C, o, sizeof(o));
}

/* DenseOdometryFunction.cpp:82-87: K, Kt = K*t, K_inv = K.inverse(), KRK_inv = K*R*K_inv. */
void orc_track_projection(const float cam[4], const float T[16], float K_inv[9], float KRK_inv[9], float Kt[3]) {
    float K[9] = {cam[0], 0, cam[2], 0, cam[1], cam[3], 0, 0, 1};
    float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]}, t[3] = {T[3], T[7], T[11]};
    float KR[9];
    for (int r = 0; r < 3; ++r) Kt[r] = sum3(K[r * 3] * t[0], K[r * 3 + 1] * t[1], K[r * 3 + 2] * t[2]);
    orc_mat3_inverse(K, K_inv);
    mat3_mul(K, R, KR);
    mat3_mul(KR, K_inv, KRK_inv);
}

/* DenseOdometryFunction.cpp:95-101: uv = d_s * KRK_inv * Point3(j,i,1.0) + Kt (the scalar*matrix
 * is evaluated first), u_t = (int)(uv0/uv2 + 0.5) with the division in float and the +0.5 in
 * double.  A non-finite quotient (x86 cvttsd2si -> INT_MIN) is reported as INT32_MIN. */
void orc_track_project_pixel(const float KRK_inv[9], const float Kt[3], float d_s, int j, int i, float uv[3], int ut[2]) {
    float fj = (float)j, fi = (float)i;
    for (int r = 0; r < 3; ++r)
        uv[r] = sum3((d_s * KRK_inv[r * 3]) * fj, (d_s * KRK_inv[r * 3 + 1]) * fi, (d_s * KRK_inv[r * 3 + 2]) * 1.0f) + Kt[r];
    for (int c = 0; c < 2; ++c) {
        double q = (double)(uv[c] / uv[2]) + 0.5;
        ut[c] = (q > -2147483649.0 && q < 2147483648.0) ? (int)q : INT32_MIN; /* NaN fails both */
    }
}

/* DenseOdometryFunction.cpp:72-128 incl. AddElementToCorrespondenceMap (:9-27): the "z-buffer"
 * READS wraping_depth at the TARGET pixel but WRITES at the SOURCE pixel; restated as written.
 * corr: 4 int32 per correspondence {v_s, u_s, v_t, u_t} in raster order; returns the count. */
size_t orc_pixel_correspondences(const orc_track_level *L, const float T[16], int32_t *corr) {
    int W = L->width, H = L->height;
    float cam[4] = {L->fx, L->fy, L->cx, L->cy}, K_inv[9], KRK[9], Kt[3];
    orc_track_projection(cam, T, K_inv, KRK, Kt);
    float *wd = (float *)malloc((size_t)W * H * sizeof(float));
    int32_t *wm = (int32_t *)malloc((size_t)W * H * 2 * sizeof(int32_t));
    for (size_t k = 0; k < (size_t)W * H; ++k) { wd[k] = -1.0f; wm[2 * k] = wm[2 * k + 1] = -1; }
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float d_s = L->source_depth[(size_t)i * W + j];
            if (isnan(d_s)) continue;
            float uv[3]; int ut[2];
            orc_track_project_pixel(KRK, Kt, d_s, j, i, uv, ut);
            float td = uv[2];
            int u_t = ut[0], v_t = ut[1];
            if (!(u_t >= 0 && u_t < W && v_t >= 0 && v_t < H)) continue;
            float d_t = L->target_depth[(size_t)v_t * W + u_t];
            if (isnan(d_t) || !((double)fabsf(d_t - td) < 0.05)) continue; /* MAX_DIFF_DEPTH */
            float existing = wd[(size_t)v_t * W + u_t];
            if (existing == -1.0f || existing > td) {
                wm[2 * ((size_t)i * W + j)] = v_t; wm[2 * ((size_t)i * W + j) + 1] = u_t;
                wd[(size_t)i * W + j] = td;
            }
        }
    size_t n = 0;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j)
            if (wd[(size_t)i * W + j] != -1.0f) {
                if (corr) { corr[4 * n] = i; corr[4 * n + 1] = j; corr[4 * n + 2] = wm[2 * ((size_t)i * W + j)]; corr[4 * n + 3] = wm[2 * ((size_t)i * W + j) + 1]; }
                ++n;
            }
    free(wd); free(wm);
    return n;
}

/* Geometry/Geometry.cpp:72-106 (TransformToMatXYZ) for one pixel. */
static void pixel_xyz(const orc_track_level *L, const float *depth, int v, int u, float p[3]) {
    float z = depth[(size_t)v * L->width + u];
    if (z > 0) { p[0] = (u - L->cx) * z / L->fx; p[1] = (v - L->cy) * z / L->fy; p[2] = z; }
    else p[0] = p[1] = p[2] = -1.0f;
}

/* DenseOdometryFunction.cpp:146-296: Jacobian rows + residuals of one correspondence.
 * term: 0 hybrid (2 rows), 1 photo, 2 depth.  Returns the row count. */
static int track_rows(const orc_track_level *L, const float T[16], const int32_t *c, int term, float J[2][6], float r[2]) {
    const float sqrt_dep = (float)sqrt(0.5), sqrt_img = (float)sqrt(1.0 - 0.5); /* LAMBDA_HYBRID_DEPTH */
    int W = L->width, v_s = c[0], u_s = c[1], v_t = c[2], u_t = c[3];
    size_t is = (size_t)v_s * W + u_s, it = (size_t)v_t * W + u_t;
    float p[3], q[3];
    pixel_xyz(L, L->source_depth, v_s, u_s, p);
    for (int k = 0; k < 3; ++k) q[k] = sum3(T[k * 4] * p[0], T[k * 4 + 1] * p[1], T[k * 4 + 2] * p[2]) + T[k * 4 + 3];
    float invz = (float)(1. / (double)q[2]);
    float c0 = 0, c1 = 0, c2 = 0, d0 = 0, d1 = 0, d2 = 0, diff_photo = 0, diff_geo = 0;
    if (term != 2) {
        diff_photo = L->target_color[it] - L->source_color[is];
        float dIdx = (float)(0.125 * (double)L->target_color_dx[it]), dIdy = (float)(0.125 * (double)L->target_color_dy[it]);
        c0 = dIdx * L->fx * invz; c1 = dIdy * L->fy * invz;
        c2 = -(c0 * q[0] + c1 * q[1]) * invz;
    }
    if (term != 1) {
        float dDdx = (float)(0.125 * (double)L->target_depth_dx[it]), dDdy = (float)(0.125 * (double)L->target_depth_dy[it]);
        if (isnan(dDdx)) dDdx = 0;
        if (isnan(dDdy)) dDdy = 0;
        diff_geo = L->target_depth[it] - q[2];
        d0 = dDdx * L->fx * invz; d1 = dDdy * L->fy * invz;
        d2 = -(d0 * q[0] + d1 * q[1]) * invz;
    }
    float jp[6] = {c0, c1, c2, -q[2] * c1 + q[1] * c2, q[2] * c0 - q[0] * c2, -q[1] * c0 + q[0] * c1};
    float jd[6] = {d0, d1, d2 - 1.0f, (-q[2] * d1 + q[1] * d2) - q[1], (q[2] * d0 - q[0] * d2) + q[0], (-q[1] * d0 + q[0] * d1)};
    if (term == 0) {
        for (int k = 0; k < 6; ++k) { J[0][k] = sqrt_img * jp[k]; J[1][k] = sqrt_dep * jd[k]; }
        r[0] = sqrt_img * diff_photo; r[1] = sqrt_dep * diff_geo;
        return 2;
    }
    memcpy(J[0], term == 1 ? jp : jd, sizeof(jp));
    r[0] = term == 1 ? diff_photo : diff_geo;
    return 1;
}

/* DenseOdometryFunction.cpp:297-381: float, sequential, `JTJ.noalias() += J*J^T; JTr += J*r;
 * r2 += r*r` per Jacobian row. */
typedef struct { float jtj[36], jtr[6], rs; double djtj[36], djtr[6], drs; } gn_acc;
static inline void gn_add_row(gn_acc *s, const float J[6], float r) {
    for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) { s->jtj[a * 6 + b] += J[a] * J[b]; s->djtj[a * 6 + b] += (double)(J[a] * J[b]); }
        s->jtr[a] += J[a] * r;
        s->djtr[a] += (double)(J[a] * r);
    }
    s->rs += r * r; s->drs += (double)(r * r);
}
static void gn_finish(const gn_acc *s, float JTJ[36], float JTr[6], float *r2) {
    if (g_accumulate_double) {
        for (int a = 0; a < 36; ++a) JTJ[a] = (float)s->djtj[a];
        for (int a = 0; a < 6; ++a) JTr[a] = (float)s->djtr[a];
        if (r2) *r2 = (float)s->drs;
    } else {
        memcpy(JTJ, s->jtj, sizeof(s->jtj)); memcpy(JTr, s->jtr, sizeof(s->jtr));
        if (r2) *r2 = s->rs;
    }
}
/* The accumulation alone, over caller-supplied rows (pins the order against Eigen's, tests/golden). */
void orc_track_accumulate_rows(const float *J, const float *r, size_t n, float JTJ[36], float JTr[6], float *r2) {
    gn_acc s; memset(&s, 0, sizeof(s));
    for (size_t i = 0; i < n; ++i) gn_add_row(&s, J + 6 * i, r[i]);
    gn_finish(&s, JTJ, JTr, r2);
}
void orc_track_normal_equations(const orc_track_level *L, const float T[16], const int32_t *corr, size_t n, int term,
                                float JTJ[36], float JTr[6], float *r2) {
    gn_acc s; memset(&s, 0, sizeof(s));
    for (size_t i = 0; i < n; ++i) {
        float J[2][6], r[2];
        int rows = track_rows(L, T, corr + 4 * i, term, J, r);
        for (int k = 0; k < rows; ++k) gn_add_row(&s, J[k], r[k]);
    }
    gn_finish(&s, JTJ, JTr, r2);
}

/* x = JTJ.ldlt().solve(-JTr) (DenseOdometryFunction.cpp:404).  Restated as a symmetric-pivoted
 * LDL^T in double from the float inputs (Eigen factorises in float; agreement is float rounding
 * times the condition number, pinned at 1e-4 in tests/test_oracle_golden.py).  Zero pivots give
 * zero components, as Eigen's pseudo-inverse of D does. */
void orc_ldlt_solve6(const float JTJ[36], const float JTr[6], float x[6]) {
    double A[36], b[6], y[6];
    int perm[6];
    for (int i = 0; i < 6; ++i) { perm[i] = i; b[i] = -(double)JTr[i]; for (int j = 0; j < 6; ++j) A[i * 6 + j] = 0.5 * ((double)JTJ[i * 6 + j] + (double)JTJ[j * 6 + i]); }
    for (int k = 0; k < 6; ++k) {
        int p = k;
        for (int i = k + 1; i < 6; ++i) if (fabs(A[i * 7]) > fabs(A[p * 7])) p = i;
        if (p != k) {
            for (int j = 0; j < 6; ++j) { double t = A[k * 6 + j]; A[k * 6 + j] = A[p * 6 + j]; A[p * 6 + j] = t; }
            for (int j = 0; j < 6; ++j) { double t = A[j * 6 + k]; A[j * 6 + k] = A[j * 6 + p]; A[j * 6 + p] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
            double tb = b[k]; b[k] = b[p]; b[p] = tb;
        }
        double d = A[k * 7];
        if (d == 0) continue;
        for (int i = k + 1; i < 6; ++i) {
            double l = A[i * 6 + k] / d;
            for (int j = k + 1; j < 6; ++j) A[i * 6 + j] -= l * A[k * 6 + j];
            A[i * 6 + k] = l;
        }
    }
    for (int i = 0; i < 6; ++i) { double s = b[i]; for (int j = 0; j < i; ++j) s -= A[i * 6 + j] * y[j]; y[i] = s; }
    for (int i = 0; i < 6; ++i) y[i] = A[i * 7] != 0 ? y[i] / A[i * 7] : 0.0;
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < 6; ++j) s -= A[j * 6 + i] * y[j]; y[i] = s; }
    for (int i = 0; i < 6; ++i) x[perm[i]] = (float)y[i];
}

/* DoSingleIteration* (DenseOdometryFunction.cpp:382-475): correspondences at T, normal equations,
 * delta, T <- Se3ToSE3(delta) * T.  Returns the correspondence count. */
size_t orc_track_iteration(const orc_track_level *L, int term, float T[16], int32_t *corr) {
    size_t n = orc_pixel_correspondences(L, T, corr);
    float JTJ[36], JTr[6], x[6], D[16];
    orc_track_normal_equations(L, T, corr, n, term, JTJ, JTr, NULL);
    orc_ldlt_solve6(JTJ, JTr, x);
    orc_se3_exp(x, D);
    mat4_mul(D, T, T);
    return n;
}

/* Odometry::MultiScaleComputing (Odometry.cpp:621-687) + the rmse line of DenseTracking (:606,
 * Geometry.cpp:48-61).  levels[0] is full resolution; iters[i] = iter_count_per_level[i]
 * (Odometry.h:170).  The inlier ratios divide by the FULL-resolution width*height at every level,
 * as the reference does.  pixel_corr: room for 4*w0*h0 int32; per_iter_count / per_iter_T: room for
 * sum(iters) entries / x16 floats; may be NULL. */
int orc_dense_track(const orc_track_level *levels, int n_levels, const int *iters, int full_w, int full_h, int term,
                    const float init_T[16], orc_track_result *res, int32_t *pixel_corr, int32_t *per_iter_count,
                    float *per_iter_T) {
    float T[16];
    memcpy(T, init_T, sizeof(T));
    size_t cap = 0;
    for (int l = 0; l < n_levels; ++l) { size_t s = (size_t)levels[l].width * levels[l].height; if (s > cap) cap = s; }
    int32_t *corr = (int32_t *)malloc((cap ? cap : 1) * 4 * sizeof(int32_t));
    size_t n = 0;
    int total = 0;
    for (int l = n_levels - 1; l >= 0; --l)
        for (int j = 0; j != iters[l]; ++j) {
            n = orc_track_iteration(&levels[l], term, T, corr);
            if (per_iter_count) per_iter_count[total] = (int32_t)n;
            if (per_iter_T) memcpy(per_iter_T + 16 * total, T, sizeof(T));
            ++total;
            if ((float)n / (full_h * full_w) > 0.9) break; /* MAX_INLIER_RATIO_DENSE (double compare) */
        }
    /* correspondence_set: source xyz and TARGET xyz both read at the SOURCE pixel of LEVEL 0
     * (Odometry.cpp:676-683) whatever level `correspondences` came from. */
    double sum_error = 0.0;
    const orc_track_level *L0 = &levels[0];
    for (size_t i = 0; i < n; ++i) {
        float p[3], q[3], tp[4];
        pixel_xyz(L0, L0->source_depth, corr[4 * i], corr[4 * i + 1], p);
        pixel_xyz(L0, L0->target_depth, corr[4 * i], corr[4 * i + 1], q);
        mat4_mul_p1(T, p[0], p[1], p[2], tp);
        float e[3] = {tp[0] / tp[3] - q[0], tp[1] / tp[3] - q[1], tp[2] / tp[3] - q[2]};
        sum_error += (double)sum3(e[0] * e[0], e[1] * e[1], e[2] * e[2]);
    }
    memcpy(res->T, T, sizeof(T));
    res->n_correspondences = n;
    res->rmse = sqrt(sum_error / (double)n);
    res->tracking_success = (float)n / (full_h * full_w) >= 0.3; /* MIN_INLIER_RATIO_DENSE */
    res->iterations = total;
    if (pixel_corr) memcpy(pixel_corr, corr, n * 4 * sizeof(int32_t));
    free(corr);
    return 0;
}

/* ======================= tracker image preparation ========================================== */
/* Odometry::InitializeRGBDDenseTracking / CreateImagePyramid (Odometry.cpp:436-449,609-620) delegate to
 * OpenCV (cvtColor, GaussianBlur, pyrDown, Sobel), which the reference does not vendor: NOTHING in
 * this block is pinned to OpenCV's arithmetic.  It restates the DEFINITIONS the product uses for that
 * stage (op_tracker_dense_tracking): OpenCV's published kernels and border rule (BORDER_REFLECT_101),
 * evaluated in float, horizontal pass then vertical pass, taps accumulated left to right / top to
 * bottom.  Only ConvertDepthTo32FNaN, the /255 intensity scaling and NormalizeIntensity are reference
 * code (DenseOdometryFunction.cpp:28-71,129-145). */
static inline int reflect101(int i, int n) { /* n >= 2 for radius-2 kernels on >= 3 px images */
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    if (i < 0) i = 0; /* degenerate tiny images */
    return i;
}

/* cvtColor(CV_RGB2GRAY) on 8-bit data (channel 0 weighted as R, as the reference's call does on
 * imread's BGR data), then / 255.0 (DenseOdometryFunction.cpp:58-71). */
void orc_prep_intensity(const uint8_t *rgb, int w, int h, float *out) {
    for (size_t k = 0; k < (size_t)w * h; ++k) {
        int g = (rgb[3 * k] * 4899 + rgb[3 * k + 1] * 9617 + rgb[3 * k + 2] * 1868 + 8192) >> 14;
        out[k] = (float)(unsigned char)g / 255.0f; /* gray.at<uchar>(i,j)/scale with float scale = 255 */
    }
}

/* DenseOdometryFunction.cpp:28-56 */
void orc_prep_depth_nan(const void *depth, int is_u16, float depth_scale, int w, int h, float *out) {
    for (size_t k = 0; k < (size_t)w * h; ++k) {
        if (is_u16) {
            unsigned short d = ((const unsigned short *)depth)[k];
            out[k] = ((double)d > 0.5 * (double)depth_scale && (double)d < 4 * (double)depth_scale) ? (float)d / depth_scale : NAN;
        } else {
            float d = ((const float *)depth)[k];
            out[k] = ((double)d > 0.5 && d < 4) ? d : NAN;
        }
    }
}

static void sep_filter(const float *in, int w, int h, const float *kx, int nx, const float *ky, int ny, float *out) {
    float *tmp = (float *)malloc((size_t)w * h * sizeof(float));
    int rx = nx / 2, ry = ny / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = kx[0] * in[(size_t)y * w + reflect101(x - rx, w)];
            for (int k = 1; k < nx; ++k) s = s + kx[k] * in[(size_t)y * w + reflect101(x - rx + k, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = ky[0] * tmp[(size_t)reflect101(y - ry, h) * w + x];
            for (int k = 1; k < ny; ++k) s = s + ky[k] * tmp[(size_t)reflect101(y - ry + k, h) * w + x];
            out[(size_t)y * w + x] = s;
        }
    free(tmp);
}

/* GaussianBlur(3x3, sigma 0) -> [1 2 1]/4 (ImageProcessing.cpp:43-46) */
void orc_prep_blur3(const float *in, int w, int h, float *out) {
    const float k[3] = {0.25f, 0.5f, 0.25f};
    sep_filter(in, w, h, k, 3, k, 3, out);
}
/* pyrDown to (w/2, h/2): [1 4 6 4 1]/16 then even samples (ImageProcessing.cpp:6-20) */
void orc_prep_pyrdown(const float *in, int w, int h, float *out) {
    const float k[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    float *full = (float *)malloc((size_t)w * h * sizeof(float));
    sep_filter(in, w, h, k, 5, k, 5, full);
    int w2 = w / 2, h2 = h / 2;
    for (int y = 0; y < h2; ++y)
        for (int x = 0; x < w2; ++x) out[(size_t)y * w2 + x] = full[(size_t)(2 * y) * w + 2 * x];
    free(full);
}
/* Sobel 3x3 (ImageProcessing.cpp:25-34): axis 0 = d/dx, 1 = d/dy */
void orc_prep_sobel(const float *in, int w, int h, int axis, float *out) {
    const float d[3] = {-1.0f, 0.0f, 1.0f}, s[3] = {1.0f, 2.0f, 1.0f};
    if (axis == 0) sep_filter(in, w, h, d, 3, s, 3, out);
    else sep_filter(in, w, h, s, 3, d, 3, out);
}

/* tool::ConvertDepthTo32F + tool::BilateralFilter (Tool/ImageProcessing.cpp:68-91, 64-67) =
 * cv::bilateralFilter(src, dst, d, sigma_color, sigma_space) on CV_32FC1.  OpenCV is not vendored: this restates the
 * DOCUMENTED definition (the one include/onepiece_hip.h gives for op_bilateral_filter_depth), parity unpinned. */
static int reflect101_any(int i, int n) { /* cv::borderInterpolate(BORDER_REFLECT_101) for any offset */
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
void orc_bilateral_filter(const void *depth, int is_u16, float depth_scale, int w, int h, int d, float sigma_color,
                          float sigma_space, float *out) {
    if (!(sigma_color > 0)) sigma_color = 1.0f;
    if (!(sigma_space > 0)) sigma_space = 1.0f;
    int radius = d <= 0 ? (int)lrint((double)sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    const float gc = (float)(-0.5 / ((double)sigma_color * sigma_color)), gs = (float)(-0.5 / ((double)sigma_space * sigma_space));
    float *src = (float *)malloc((size_t)w * h * sizeof(float));
    for (size_t k = 0; k < (size_t)w * h; ++k) /* ImageProcessing.cpp:76-88 */
        src[k] = is_u16 ? (float)((const uint16_t *)depth)[k] / depth_scale : ((const float *)depth)[k];
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float v0 = src[(size_t)y * w + x];
            float sum = 0.0f, wsum = 0.0f;
            for (int i = -radius; i <= radius; ++i)
                for (int j = -radius; j <= radius; ++j) {
                    if (i * i + j * j > radius * radius) continue;
                    const float v = src[(size_t)reflect101_any(y + i, h) * w + reflect101_any(x + j, w)];
                    const float wgt = expf((float)(i * i + j * j) * gs) * expf((v - v0) * (v - v0) * gc);
                    sum += v * wgt;
                    wsum += wgt;
                }
            out[(size_t)y * w + x] = sum / wsum;
        }
    free(src);
}

/* DenseOdometryFunction.cpp:129-145 (float sequential means, LinearTransform with a float scale). */
void orc_normalize_intensity(float *source, float *target, int w, int h, const int32_t *corr, size_t n) {
    float mean_s = 0.0f, mean_t = 0.0f;
    for (size_t r = 0; r < n; ++r) {
        mean_s += source[(size_t)corr[4 * r] * w + corr[4 * r + 1]];
        mean_t += target[(size_t)corr[4 * r + 2] * w + corr[4 * r + 3]];
    }
    mean_s /= (float)n; mean_t /= (float)n;
    const float ss = (float)(0.5 / (double)mean_s), st = (float)(0.5 / (double)mean_t);
    for (size_t k = 0; k < (size_t)w * h; ++k) { source[k] = source[k] * ss + 0.0f; target[k] = target[k] * st + 0.0f; }
}

/* Odometry::DenseTracking, cv::Mat overload (Odometry.cpp:463-524), with the preparation above.
 * pyr_out (optional): receives, for frame f (0 source, 1 target), kind k (0 colour, 1 depth, 2 colour_dx,
 * 3 colour_dy, 4 depth_dx, 5 depth_dy) and level l, a malloc'ed image at pyr_out[(f*6+k)*n_levels+l]
 * (caller frees). */
int orc_dense_tracking(const orc_camera *cam, int n_levels, const int *iters, const uint8_t *src_rgb, const uint8_t *tgt_rgb,
                       const void *src_depth, const void *tgt_depth, int is_u16, int term, const float init_T[16],
                       orc_track_result *res, int32_t *pixel_corr, float **pyr_out) {
    int w = cam->width, h = cam->height;
    size_t np = (size_t)w * h;
    float *img[2][6][8];
    memset(img, 0, sizeof(img));
    const uint8_t *rgb[2] = {src_rgb, tgt_rgb};
    const void *dep[2] = {src_depth, tgt_depth};
    float *raw = (float *)malloc(np * sizeof(float));
    for (int f = 0; f < 2; ++f) {
        img[f][0][0] = (float *)malloc(np * sizeof(float)); img[f][1][0] = (float *)malloc(np * sizeof(float));
        orc_prep_intensity(rgb[f], w, h, raw); orc_prep_blur3(raw, w, h, img[f][0][0]);
        orc_prep_depth_nan(dep[f], is_u16, cam->depth_scale, w, h, raw); orc_prep_blur3(raw, w, h, img[f][1][0]);
    }
    free(raw);
    /* identity-pose correspondences on the full-resolution depth, then NormalizeIntensity (:543-544) */
    orc_track_level L0;
    memset(&L0, 0, sizeof(L0));
    L0.width = w; L0.height = h; L0.fx = cam->fx; L0.fy = cam->fy; L0.cx = cam->cx; L0.cy = cam->cy;
    L0.source_depth = img[0][1][0]; L0.target_depth = img[1][1][0];
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int32_t *corr = (int32_t *)malloc(np * 4 * sizeof(int32_t));
    size_t nc = orc_pixel_correspondences(&L0, I4, corr);
    orc_normalize_intensity(img[0][0][0], img[1][0][0], w, h, corr, nc);
    free(corr);
    orc_track_level lv[8];
    int lw = w, lh = h;
    float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
    for (int l = 0; l < n_levels; ++l) {
        if (l > 0) {
            for (int f = 0; f < 2; ++f)
                for (int k = 0; k < 2; ++k) {
                    img[f][k][l] = (float *)malloc((size_t)(lw / 2) * (lh / 2) * sizeof(float));
                    orc_prep_pyrdown(img[f][k][l - 1], lw, lh, img[f][k][l]);
                }
            lw /= 2; lh /= 2; fx /= 2; fy /= 2; cx /= 2; cy /= 2; /* Camera.h:38-42 */
        }
        for (int f = 0; f < 2; ++f)
            for (int k = 0; k < 2; ++k)
                for (int a = 0; a < 2; ++a) {
                    img[f][2 + 2 * k + a][l] = (float *)malloc((size_t)lw * lh * sizeof(float));
                    orc_prep_sobel(img[f][k][l], lw, lh, a, img[f][2 + 2 * k + a][l]);
                }
        lv[l].width = lw; lv[l].height = lh; lv[l].fx = fx; lv[l].fy = fy; lv[l].cx = cx; lv[l].cy = cy;
        lv[l].source_color = img[0][0][l]; lv[l].source_depth = img[0][1][l];
        lv[l].target_color = img[1][0][l]; lv[l].target_depth = img[1][1][l];
        lv[l].target_color_dx = img[1][2][l]; lv[l].target_color_dy = img[1][3][l];
        lv[l].target_depth_dx = img[1][4][l]; lv[l].target_depth_dy = img[1][5][l];
    }
    int rc = orc_dense_track(lv, n_levels, iters, w, h, term, init_T, res, pixel_corr, NULL, NULL);
    for (int f = 0; f < 2; ++f)
        for (int k = 0; k < 6; ++k)
            for (int l = 0; l < n_levels; ++l) {
                if (pyr_out) pyr_out[(f * 6 + k) * n_levels + l] = img[f][k][l];
                else free(img[f][k][l]);
            }
    return rc;
}
void orc_free(void *p) { free(p); }

/* ======================= mesh extraction (Integration/MarchingCube.cpp, CubeHandler.cpp:9-114) == */
/* CubeHandler::ExtractTriangleMesh = GenerateMeshByCube (CubeHandler.cpp:70-114) for every block +
 * MarchingCube (MarchingCube.cpp:31-74).  The 256x16 triangle table (MarchingCubePredefined.h:17-274) and
 * the 12x2 edge table (:275-289) are INPUTS: they are data of the reference's header, supplied by whoever
 * calls (the reference-side shim passes its own arrays); nothing here depends on their content beyond
 * "rows are -1 terminated edge triples".  Vertices are emitted three per triangle, unshared, in the order
 * MarchingCube() pushes them; blocks in insertion order, voxels in the reference's x, y, z loop order.
 * only_block: NULL or the one CubeID to mesh (GenerateMeshByCube alone).  Returns the vertex count. */
size_t orc_volume_extract_mesh(const orc_volume *v, const int32_t *tri_table, const int32_t *edge_pairs, const int32_t *only_block,
                               float *points, float *colors, size_t cap_vertices) {
    static const int cxo[8] = {0, 1, 1, 0, 0, 1, 1, 0}, cyo[8] = {0, 0, 1, 1, 0, 0, 1, 1}, czo[8] = {0, 0, 0, 0, 1, 1, 1, 1}; /* VoxelCube.h:45-47 */
    size_t n = 0;
    float cube_res = CUBE * v->res; /* VoxelCube.h:149-153 */
    for (size_t b = 0; b < v->n; ++b) {
        const int32_t *id = v->keys + 3 * b;
        if (only_block && (id[0] != only_block[0] || id[1] != only_block[1] || id[2] != only_block[2])) continue;
        for (int x = 0; x < CUBE; ++x)
            for (int y = 0; y < CUBE; ++y)
                for (int z = 0; z < CUBE; ++z) {
                    int ox = x == CUBE - 1, oy = y == CUBE - 1, oz = z == CUBE - 1; /* NeighborCubeIDOffset[index] */
                    float corner[8][3], csdf[8], ccol[8][3];
                    int ok = 1;
                    for (int i = 0; i < 8 && ok; ++i) {
                        int nx = id[0] + (cxo[i] & ox), ny = id[1] + (cyo[i] & oy), nz = id[2] + (czo[i] & oz);
                        int vx = (x + cxo[i]) % CUBE, vy = (y + cyo[i]) % CUBE, vz = (z + czo[i]) % CUBE;
                        int vid = vx + vy * CUBE + vz * CUBE * CUBE;
                        int64_t nb = vol_find(v, nx, ny, nz);
                        if (nb < 0) { ok = 0; break; }
                        const float *t = v->vox + ((size_t)nb * NVOX + vid) * 5;
                        corner[i][0] = nx * cube_res + v->offset[vid][0];
                        corner[i][1] = ny * cube_res + v->offset[vid][1];
                        corner[i][2] = nz * cube_res + v->offset[vid][2];
                        csdf[i] = t[0]; ccol[i][0] = t[2]; ccol[i][1] = t[3]; ccol[i][2] = t[4];
                        if (t[0] >= 1 || t[1] <= 0) ok = 0; /* !IsValid (TSDFVoxel.h:75-78) */
                    }
                    if (!ok) continue;
                    int ci = 0;
                    for (int i = 0; i < 8; ++i) ci |= (csdf[i] > 0 ? 1 << i : 0); /* DetermineCase */
                    const int32_t *E = tri_table + 16 * ci;
                    for (int i = 0; i < 16; i += 3) {
                        if (E[i] == -1) break;
                        for (int j = 0; j < 3; ++j) {
                            int e = E[i + j], a = edge_pairs[2 * e], c = edge_pairs[2 * e + 1];
                            if (n < cap_vertices) {
                                float sdf_diff = csdf[c] - csdf[a];            /* InterpolateEdgeVetex */
                                float t = csdf[a] / sdf_diff;
                                for (int k = 0; k < 3; ++k) {
                                    points[3 * n + k] = corner[a][k] - t * (corner[c][k] - corner[a][k]);
                                    colors[3 * n + k] = (ccol[a][k] + ccol[c][k]) / 2;
                                }
                            }
                            ++n;
                        }
                    }
                }
    }
    return n;
}
