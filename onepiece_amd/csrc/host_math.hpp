// host_math.hpp -- host-side (CPU) geometry that sits on the hot path between kernels.
//
// These are the few scalar computations the reference performs once per frame / once per ICP
// iteration with Eigen + Sophus on the host (pose inverse, frustum planes, 6x6 solve, SE3 exp,
// Kabsch).  north_star keeps them on the host; they are written here without Eigen so that the
// C-ABI library is self-contained, and in the operation order of the reference's Eigen 3.3.7 /
// -msse4.2 build wherever bit-exactness matters (pose inverse and frustum planes feed block
// selection, which must be bit-exact).  The translation unit is compiled with -ffp-contract=off.
//
// Reference citations are file:line under /root/reference/src (or 3rdparty/...).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

#include "px_round.hpp" // OP_HD: __host__ __device__ under hipcc (the ICP update step also runs on the device)

namespace op_host {

// ---- Eigen fixed-size evaluation orders (3rdparty/Eigen/Eigen/src/Core/Redux.h,
//      ProductEvaluators.h:626-632): 3-term sums associate as a0 + (a1 + a2); a 4x4 * (x,y,z,1)
//      product accumulates column by column.
inline float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
inline float dot3(const float* a, const float* b) { return sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }

// ---- 4x4 inverse: the SSE cofactor kernel of Eigen 3.3.7 (LU/arch/Inverse_SSE.h:35-165), written
//      as explicit lane arithmetic.  Lane vectors are std::array-like PODs.
struct V4 { float a, b, c, d; };
inline V4 mul(V4 x, V4 y) { return {x.a * y.a, x.b * y.b, x.c * y.c, x.d * y.d}; }
inline V4 add(V4 x, V4 y) { return {x.a + y.a, x.b + y.b, x.c + y.c, x.d + y.d}; }
inline V4 sub(V4 x, V4 y) { return {x.a - y.a, x.b - y.b, x.c - y.c, x.d - y.d}; }
inline V4 splat(float s) { return {s, s, s, s}; }

inline void mat4_inverse(const float m[16], float out[16]) {
    // 2x2 sub-blocks as the kernel holds them for a column-major source (Inverse_SSE.h:68-73):
    // A = (m00,m10,m01,m11)  B = (m20,m30,m21,m31)  C = (m02,m12,m03,m13)  D = (m22,m32,m23,m33)
    const V4 A{m[0], m[4], m[1], m[5]}, B{m[8], m[12], m[9], m[13]};
    const V4 C{m[2], m[6], m[3], m[7]}, D{m[10], m[14], m[11], m[15]};
    // AB = A# * B, DC = D# * C (:82-87)
    V4 AB = sub(mul(V4{A.d, A.d, A.a, A.a}, B), mul(V4{A.b, A.b, A.c, A.c}, V4{B.c, B.d, B.a, B.b}));
    V4 DC = sub(mul(V4{D.d, D.d, D.a, D.a}, C), mul(V4{D.b, D.b, D.c, D.c}, V4{C.c, C.d, C.a, C.b}));
    // 2x2 determinants (:89-101): lane0 of (x.d*x.a) - (x.b*x.c)
    const float dA = A.d * A.a - A.b * A.c, dB = B.d * B.a - B.b * B.c;
    const float dC = C.d * C.a - C.b * C.c, dD = D.d * D.a - D.b * D.c;
    // d = trace(AB*DC) (:103-104,113-115)
    V4 d4 = mul(V4{DC.a, DC.c, DC.b, DC.d}, AB);
    const float tr = (d4.a + d4.c) + (d4.b + d4.d);
    // iD = C*A#*B, iA = B*D#*C (:106-111)
    V4 iD = add(mul(V4{C.a, C.a, C.c, C.c}, V4{AB.a, AB.b, AB.a, AB.b}),
                mul(V4{C.b, C.b, C.d, C.d}, V4{AB.c, AB.d, AB.c, AB.d}));
    V4 iA = add(mul(V4{B.a, B.a, B.c, B.c}, V4{DC.a, DC.b, DC.a, DC.b}),
                mul(V4{B.b, B.b, B.d, B.d}, V4{DC.c, DC.d, DC.c, DC.d}));
    const float d1 = dA * dD, d2 = dB * dC; // (:116-117)
    iD = sub(mul(D, splat(dA)), iD);        // (:119-120)
    iA = sub(mul(A, splat(dD)), iA);        // (:122-123)
    const float det = (d1 + d2) - tr;       // (:125-126)
    const float rd = 1.0f / det;            // _mm_div_ss(1, det) (:127)
    // iB = D*(A#B)#, iC = A*(D#C)# (:133-138)
    V4 iB = sub(mul(D, V4{AB.d, AB.a, AB.d, AB.a}), mul(V4{D.b, D.a, D.d, D.c}, V4{AB.c, AB.b, AB.c, AB.b}));
    V4 iC = sub(mul(A, V4{DC.d, DC.a, DC.d, DC.a}), mul(V4{A.b, A.a, A.d, A.c}, V4{DC.c, DC.b, DC.c, DC.b}));
    const V4 rds{rd, -rd, -rd, rd};          // sign mask PNNP (:140-141)
    iB = sub(mul(C, splat(dB)), iB);         // (:143-144)
    iC = sub(mul(B, splat(dC)), iC);         // (:146-147)
    iA = mul(rds, iA); iB = mul(rds, iB); iC = mul(rds, iC); iD = mul(rds, iD); // (:149-153)
    // result columns (:155-160): col0 = (iA.d,iA.b,iB.d,iB.b) col1 = (iA.c,iA.a,iB.c,iB.a) etc.
    const float c0[4] = {iA.d, iA.b, iB.d, iB.b}, c1[4] = {iA.c, iA.a, iB.c, iB.a};
    const float c2[4] = {iC.d, iC.b, iD.d, iD.b}, c3[4] = {iC.c, iC.a, iD.c, iD.a};
    for (int r = 0; r < 4; ++r) {
        out[r * 4 + 0] = c0[r]; out[r * 4 + 1] = c1[r]; out[r * 4 + 2] = c2[r]; out[r * 4 + 3] = c3[r];
    }
}

// ---- Frustum (Integration/Frustum.cpp:7-46; GetPlane Geometry/Geometry.cpp:165-171).
inline void plane_from_points(const float* p1, const float* p2, const float* p3, float* plane) {
    float e1[3], e2[3], n[3];
    for (int i = 0; i < 3; ++i) { e1[i] = p2[i] - p1[i]; e2[i] = p3[i] - p1[i]; }
    n[0] = e1[1] * e2[2] - e1[2] * e2[1];
    n[1] = e1[2] * e2[0] - e1[0] * e2[2];
    n[2] = e1[0] * e2[1] - e1[1] * e2[0];
    const float len2 = sum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
    if (len2 > 0.0f) { const float len = std::sqrt(len2); n[0] /= len; n[1] /= len; n[2] /= len; }
    const double d = -dot3(p1, n);
    plane[0] = n[0]; plane[1] = n[1]; plane[2] = n[2]; plane[3] = static_cast<float>(d);
}

struct CameraPOD { float fx, fy, cx, cy; int32_t width, height; float depth_scale; };

// Frustum::ComputeFromVectors (Integration/Frustum.cpp:25-94).  planes: top, left, right, bottom, near, far -- the order
// Frustum::ContainPoint tests them in (Integration/Frustum.h:74-103); corners (optional) in the order of the reference's
// public `corners` member (:61-68): far top-left, far top-right, far bottom-left, far bottom-right, near bottom-right,
// near top-left, near top-right, near bottom-left.
inline void frustum_from_vectors(const float fwd[3], const float pos[3], const float right[3], const float up[3], float far_d, float near_d,
                                 float fov, float aspect, float planes[24], float* corners24) {
    // tan: the reference calls the unqualified C function on a float -> double version.
    const float tangent = static_cast<float>(std::tan(static_cast<double>(fov / 2)));
    const float hf = tangent * far_d, wf = hf * aspect, hn = tangent * near_d, wn = hn * aspect;
    float corner[8][3]; // ftl ftr fbl fbr ntl ntr nbl nbr
    for (int i = 0; i < 3; ++i) {
        const float fc = pos[i] + fwd[i] * far_d, nc = pos[i] + fwd[i] * near_d;
        corner[0][i] = (fc + up[i] * hf) - right[i] * wf;
        corner[1][i] = (fc + up[i] * hf) + right[i] * wf;
        corner[2][i] = (fc - up[i] * hf) - right[i] * wf;
        corner[3][i] = (fc - up[i] * hf) + right[i] * wf;
        corner[4][i] = (nc + up[i] * hn) - right[i] * wn;
        corner[5][i] = (nc + up[i] * hn) + right[i] * wn;
        corner[6][i] = (nc - up[i] * hn) - right[i] * wn;
        corner[7][i] = (nc - up[i] * hn) + right[i] * wn;
    }
    const float *ftl = corner[0], *ftr = corner[1], *fbl = corner[2], *fbr = corner[3];
    const float *ntl = corner[4], *ntr = corner[5], *nbl = corner[6], *nbr = corner[7];
    plane_from_points(ntl, ftl, ntr, planes + 0);  // top
    plane_from_points(ftl, ntl, fbl, planes + 4);  // left
    plane_from_points(ntr, ftr, nbr, planes + 8);  // right
    plane_from_points(nbr, fbl, nbl, planes + 12); // bottom
    plane_from_points(nbl, ntl, nbr, planes + 16); // near
    plane_from_points(ftr, ftl, fbr, planes + 20); // far
    if (corners24) {
        const float* order[8] = {ftl, ftr, fbl, fbr, nbr, ntl, ntr, nbl};
        for (int k = 0; k < 8; ++k)
            for (int i = 0; i < 3; ++i) corners24[3 * k + i] = order[k][i];
    }
}

// Frustum::ComputeFromCamera (Integration/Frustum.cpp:7-24)
inline void frustum_planes(const CameraPOD& cam, const float pose[16], float far_d, float near_d, float planes[24], float* corners24 = nullptr) {
    const float height = static_cast<float>(cam.height), width = static_cast<float>(cam.width);
    const float right[3] = {pose[0], pose[4], pose[8]};
    const float up[3] = {-pose[1], -pose[5], -pose[9]};
    const float fwd[3] = {pose[2], pose[6], pose[10]};
    const float pos[3] = {pose[3], pose[7], pose[11]};
    const float aspect = (cam.fy * width) / (cam.fx * height);
    // atan2: the reference calls the unqualified C function on floats -> double version.
    const float fov = static_cast<float>(std::atan2(static_cast<double>(cam.cy), static_cast<double>(cam.fy)) +
                                         std::atan2(static_cast<double>(height - cam.cy), static_cast<double>(cam.fy)));
    frustum_from_vectors(fwd, pos, right, up, far_d, near_d, fov, aspect, planes, corners24);
}

// ---- SE3 exponential, Sophus convention x = (upsilon, omega) (3rdparty/Sophus/sophus/se3.hpp:468-489).
OP_HD void se3_exp(const float x[6], float T[16]) {
    const double w[3] = {x[3], x[4], x[5]}, u[3] = {x[0], x[1], x[2]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double a, b, c; // R = I + a W + b W^2, V = I + b W + c W^2
    if (th < 1e-5) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0; }
    else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; c = (th - sin(th)) / (th2 * th); }
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    double W2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[i][j] = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
    for (int i = 0; i < 3; ++i) {
        double t = 0;
        for (int j = 0; j < 3; ++j) {
            const double I = i == j ? 1.0 : 0.0;
            T[i * 4 + j] = static_cast<float>(I + a * W[i][j] + b * W2[i][j]);
            t += (I + b * W[i][j] + c * W2[i][j]) * u[j];
        }
        T[i * 4 + 3] = static_cast<float>(t);
    }
    T[12] = T[13] = T[14] = 0.0f; T[15] = 1.0f;
}

// ---- symmetric eigen-decomposition (cyclic Jacobi), N <= 6, double.
// FULL = true sweeps until the off-diagonal mass underflows (the textbook stopping rule the CPU restatement uses;
// validation mode of the ICP path) instead of stopping at the limit of double.
template <int N, bool FULL = false>
OP_HD void sym_eig(double A[N][N], double V[N][N]) {
    // every inner loop is fully unrolled so that, on the device, A and V are indexed with compile-time
    // constants and stay in registers (dynamic indexing would put them in scratch memory)
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0, diag = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diag += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
        }
        // converged: off-diagonal mass below 1e-17 of the diagonal's (the limit of double) or exactly zero
        if (off < 1e-300 || (!FULL && off <= 1e-34 * diag)) break;
#pragma unroll
        for (int p = 0; p < N; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double cs = 1 / sqrt(t * t + 1), sn = t * cs;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double kp = A[k][p], kq = A[k][q];
                    A[k][p] = cs * kp - sn * kq; A[k][q] = sn * kp + cs * kq;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double pk = A[p][k], qk = A[q][k];
                    A[p][k] = cs * pk - sn * qk; A[q][k] = sn * pk + cs * qk;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double kp = V[k][p], kq = V[k][q];
                    V[k][p] = cs * kp - sn * kq; V[k][q] = sn * kp + cs * kq;
                }
            }
        }
    }
}

// x = JacobiSVD(JTJ).solve(-JTr) (Registration/ICP.cpp:137-138): minimum-norm least squares with
// Eigen's default rank threshold (singular values <= eps_float * 6 * max are dropped).
template <bool FULL = false>
OP_HD void solve6_psd(const double JTJ[36], const double JTr[6], float x[6]) {
    double A[6][6], V[6][6], y[6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) A[i][j] = 0.5 * (JTJ[i * 6 + j] + JTJ[j * 6 + i]);
    sym_eig<6, FULL>(A, V);
    double smax = 0;
    for (int i = 0; i < 6; ++i) smax = fabs(A[i][i]) > smax ? fabs(A[i][i]) : smax;
    const double thr = smax * 6.0 * static_cast<double>(FLT_EPSILON);
    for (int k = 0; k < 6; ++k) {
        double s = 0;
        for (int i = 0; i < 6; ++i) s += V[i][k] * (-JTr[i]);
        y[k] = fabs(A[k][k]) > thr ? s / A[k][k] : 0.0;
    }
    for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += V[i][k] * y[k];
        x[i] = static_cast<float>(s);
    }
}

// Kabsch, the part after the sums (Geometry/Geometry.cpp:134-150): W = U S V^T (from the eigen-decomposition of
// W^T W), R = V U^T (det-fixed), t = mt - R ms.
template <bool FULL = false>
OP_HD void kabsch_finish(const double ms[3], const double mt[3], const double W[3][3], float T[16]) {
    double A[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = W[0][i] * W[0][j] + W[1][i] * W[1][j] + W[2][i] * W[2][j];
    sym_eig<3, FULL>(A, V);
    // eigenvalues in descending order with their eigenvectors (columns): three predicated exchanges, all
    // indices static
    double ev[3] = {A[0][0], A[1][1], A[2][2]};
    double Vs[3][3], S[3], U[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Vs[i][j] = V[i][j];
    }
#define OP_KSWAP(a, b)                                                                               \
    {                                                                                                \
        const bool sw = ev[b] > ev[a];                                                               \
        const double te = ev[a]; ev[a] = sw ? ev[b] : ev[a]; ev[b] = sw ? te : ev[b];                \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                              \
            const double tv = Vs[i][a]; Vs[i][a] = sw ? Vs[i][b] : Vs[i][a]; Vs[i][b] = sw ? tv : Vs[i][b]; \
        }                                                                                            \
    }
    OP_KSWAP(0, 1) OP_KSWAP(0, 2) OP_KSWAP(1, 2)
#undef OP_KSWAP
#pragma unroll
    for (int k = 0; k < 3; ++k) S[k] = sqrt(ev[k] > 0 ? ev[k] : 0);
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) {
            const double s = W[i][0] * Vs[0][k] + W[i][1] * Vs[1][k] + W[i][2] * Vs[2][k];
            U[i][k] = S[k] > 1e-300 ? s / S[k] : 0.0;
        }
    if (S[2] <= 1e-12 * (S[0] > 0 ? S[0] : 1.0)) { // rank-deficient: complete U by a cross product
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
    double R[3][3];
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = Vs[i][0] * U[j][0] + Vs[i][1] * U[j][1] + Vs[i][2] * U[j][2];
        const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) -
                           R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                           R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
        if (det >= 0) break;
        for (int i = 0; i < 3; ++i) Vs[i][2] = -Vs[i][2]; // Geometry.cpp:139-144
    }
    for (int i = 0; i < 16; ++i) T[i] = 0.0f;
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) { T[i * 4 + j] = static_cast<float>(R[i][j]); s += R[i][j] * ms[j]; }
        T[i * 4 + 3] = static_cast<float>(mt[i] - s);
    }
    T[15] = 1.0f;
}

// Kabsch from sufficient statistics reduced in fp64 (the order-free variant): n, sum s, sum t, sum s t^T.
// W = sum (s - ms)(t - mt)^T = sum s t^T - n ms mt^T.
OP_HD void kabsch_from_sums(double n, const double ss[3], const double st[3], const double sst[9], float T[16]) {
    double ms[3], mt[3], W[3][3];
    for (int i = 0; i < 3; ++i) { ms[i] = ss[i] / n; mt[i] = st[i] / n; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W[i][j] = sst[i * 3 + j] - n * ms[i] * mt[j];
    kabsch_finish(ms, mt, W, T);
}

// geometry::EstimateRigidTransformation in the REFERENCE'S ORDER (Geometry/Geometry.cpp:117-133): the two means and
// then the 3x3 sum of centred outer products are accumulated pair by pair in float32, exactly as the reference's
// loops do (Eigen evaluates `mean += p` and `W += a * b^T` component-wise: one rounded product, one rounded add per
// entry; no FMA in its -msse4.2 build).  Over 3e5 near-planar pairs that rounding noise is ~1e-3 of the result, and it
// is part of what RegistrationResult::T is -- so this is what op_icp_run returns by default.  pairs: n x 6 floats
// (source xyz, target xyz) in correspondence_set order (ascending source index).
//
// KabschReferenceOrder does the two sequential passes incrementally (rows may arrive in chunks while the first pass
// runs) and, on x86, four components per SSE instruction.  Every component still sees exactly the reference's
// operations in the reference's order -- one rounded add per row and sum, one rounded multiply and one rounded add per
// row and W entry (mulps/addps, never fused) -- so the result is bit-identical to the scalar loops; what the vector
// form buys is that a row costs one add latency per pass instead of ~5 cycles of scalar issue.
struct KabschReferenceOrder {
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
    __m128 sum_a = _mm_setzero_ps(), sum_b = _mm_setzero_ps(); // (s0,s1,s2,t0), (s2,t0,t1,t2): lanes 0,1 of sum_b are duplicates
    // first pass (:122-127): rows [first, first + count) of n x 6 floats
    void add_rows(const float* pairs, size_t count) {
        __m128 a = sum_a, b = sum_b;
        for (size_t i = 0; i < count; ++i) {
            a = _mm_add_ps(a, _mm_loadu_ps(pairs + 6 * i));
            b = _mm_add_ps(b, _mm_loadu_ps(pairs + 6 * i + 2));
        }
        sum_a = a; sum_b = b;
    }
    void sums(float ms[3], float mt[3]) const {
        float a[4], b[4];
        _mm_storeu_ps(a, sum_a); _mm_storeu_ps(b, sum_b);
        ms[0] = a[0]; ms[1] = a[1]; ms[2] = a[2]; mt[0] = a[3]; mt[1] = b[2]; mt[2] = b[3];
    }
    // second pass (:130-133)
    static void cross(const float* pairs, size_t n, const float ms[3], const float mt[3], float W[9]) {
        const __m128 ma = _mm_setr_ps(ms[0], ms[1], ms[2], mt[0]), mb = _mm_setr_ps(ms[2], mt[0], mt[1], mt[2]);
        __m128 w0 = _mm_setzero_ps(), w1 = _mm_setzero_ps(), w2 = _mm_setzero_ps(); // rows of W in lanes 1..3 (lane 0 unused)
        for (size_t i = 0; i < n; ++i) {
            const __m128 a = _mm_sub_ps(_mm_loadu_ps(pairs + 6 * i), ma);     // (a0, a1, a2, b0)
            const __m128 b = _mm_sub_ps(_mm_loadu_ps(pairs + 6 * i + 2), mb); // (a2, b0, b1, b2)
            w0 = _mm_add_ps(w0, _mm_mul_ps(_mm_shuffle_ps(a, a, _MM_SHUFFLE(0, 0, 0, 0)), b));
            w1 = _mm_add_ps(w1, _mm_mul_ps(_mm_shuffle_ps(a, a, _MM_SHUFFLE(1, 1, 1, 1)), b));
            w2 = _mm_add_ps(w2, _mm_mul_ps(_mm_shuffle_ps(a, a, _MM_SHUFFLE(2, 2, 2, 2)), b));
        }
        float r[3][4];
        _mm_storeu_ps(r[0], w0); _mm_storeu_ps(r[1], w1); _mm_storeu_ps(r[2], w2);
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c) W[k * 3 + c] = r[k][1 + c];
    }
#else
    float acc[6] = {0, 0, 0, 0, 0, 0};
    void add_rows(const float* pairs, size_t count) {
        for (size_t i = 0; i < count; ++i)
            for (int c = 0; c < 6; ++c) acc[c] += pairs[6 * i + c];
    }
    void sums(float ms[3], float mt[3]) const { for (int c = 0; c < 3; ++c) { ms[c] = acc[c]; mt[c] = acc[3 + c]; } }
    static void cross(const float* pairs, size_t n, const float ms[3], const float mt[3], float W[9]) {
        for (int k = 0; k < 9; ++k) W[k] = 0.0f;
        for (size_t i = 0; i < n; ++i) {
            float a[3], b[3];
            for (int c = 0; c < 3; ++c) { a[c] = pairs[6 * i + c] - ms[c]; b[c] = pairs[6 * i + 3 + c] - mt[c]; }
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) W[r * 3 + c] += a[r] * b[c];
        }
    }
#endif
    // the fit, once add_rows has seen all n rows (rows n-1's loads read pairs[6n-4 .. 6n-1]: inside the array)
    template <bool FULL>
    void finish(const float* pairs, size_t n, float T[16]) const {
        float ms[3], mt[3], W[9];
        sums(ms, mt);
        for (int c = 0; c < 3; ++c) { ms[c] /= static_cast<float>(n); mt[c] /= static_cast<float>(n); } // :128-129
        cross(pairs, n, ms, mt, W);
        double dms[3], dmt[3], dW[3][3];
        for (int i = 0; i < 3; ++i) {
            dms[i] = ms[i]; dmt[i] = mt[i];
            for (int j = 0; j < 3; ++j) dW[i][j] = W[i * 3 + j];
        }
        kabsch_finish<FULL>(dms, dmt, dW, T);
    }
};

template <bool FULL = false>
inline void kabsch_reference_order(const float* pairs, size_t n, float T[16]) {
    KabschReferenceOrder k;
    k.add_rows(pairs, n);
    k.finish<FULL>(pairs, n, T);
}

// registration::EstimateRigidTransformationPointToPlane's accumulation in the REFERENCE'S ORDER (ICP.cpp:121-136):
// JTJ (all 36 entries) and JTr summed row by row in float32; tmp_r is the float difference of two float dots, widened
// to double and narrowed again by Eigen's scalar promotion before it multiplies the row.  rows: n x 9 floats
// (transformed source point, target point, target normal) in inlier order.  Validation mode of op_icp_run.
inline void plane_sums_reference_order(const float* rows, size_t n, double JTJ[36], double JTr[6]) {
    float jtj[36], jtr[6];
    for (int k = 0; k < 36; ++k) jtj[k] = 0.0f;
    for (int k = 0; k < 6; ++k) jtr[k] = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float *s = rows + 9 * i, *t = s + 3, *nn = s + 6;
        const double r = static_cast<double>(dot3(nn, s) - dot3(nn, t));
        const float row[6] = {nn[0], nn[1], nn[2], s[1] * nn[2] - s[2] * nn[1], s[2] * nn[0] - s[0] * nn[2], s[0] * nn[1] - s[1] * nn[0]};
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) jtj[a * 6 + b] += row[a] * row[b];
            jtr[a] += static_cast<float>(r) * row[a];
        }
    }
    for (int k = 0; k < 36; ++k) JTJ[k] = jtj[k];
    for (int k = 0; k < 6; ++k) JTr[k] = jtr[k];
}

// Matrix4f * Matrix4f (start_T = tmp_T * start_T, ICP.cpp:198), column-accumulating product.
OP_HD void mat4_mul(const float* A, const float* B, float* C) {
    float out[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            out[r * 4 + c] = ((A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c]) + A[r * 4 + 2] * B[8 + c]) + A[r * 4 + 3] * B[12 + c];
    for (int i = 0; i < 16; ++i) C[i] = out[i];
}

// ---- dense tracker helpers (Odometry/DenseOdometryFunction.cpp) -------------------------------
OP_HD float hm_sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); } // Eigen 3-term redux order

// Matrix3f::inverse() as Eigen 3.3.7 evaluates it (LU/InverseImpl.h:126-170): cofactors of column 0,
// det = c0*m00 + (c1*m10 + c2*m20), every entry = cofactor * (1/det).  Row-major.
OP_HD void mat3_inverse(const float m[9], float out[9]) {
#define OPM3(i, j) m[(((i)) % 3) * 3 + (((j)) % 3)]
#define OPCOF(i, j) (OPM3((i) + 1, (j) + 1) * OPM3((i) + 2, (j) + 2) - OPM3((i) + 1, (j) + 2) * OPM3((i) + 2, (j) + 1))
    const float c0 = OPCOF(0, 0), c1 = OPCOF(1, 0), c2 = OPCOF(2, 0);
    const float det = hm_sum3(c0 * m[0], c1 * m[3], c2 * m[6]);
    const float invdet = 1.0f / det;
    out[0] = c0 * invdet; out[1] = c1 * invdet; out[2] = c2 * invdet;
    out[3] = OPCOF(0, 1) * invdet; out[4] = OPCOF(1, 1) * invdet; out[5] = OPCOF(2, 1) * invdet;
    out[6] = OPCOF(0, 2) * invdet; out[7] = OPCOF(1, 2) * invdet; out[8] = OPCOF(2, 2) * invdet;
#undef OPCOF
#undef OPM3
}

OP_HD void mat3_mul(const float* A, const float* B, float* C) {
    float o[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[r * 3 + c] = hm_sum3(A[r * 3] * B[c], A[r * 3 + 1] * B[3 + c], A[r * 3 + 2] * B[6 + c]);
    for (int i = 0; i < 9; ++i) C[i] = o[i];
}

// DenseOdometryFunction.cpp:82-87: Kt = K*t, KRK_inv = K*R*K.inverse() (float, Eigen order).
OP_HD void track_projection(float fx, float fy, float cx, float cy, const float T[16], float KRK_inv[9], float Kt[3]) {
    const float K[9] = {fx, 0.0f, cx, 0.0f, fy, cy, 0.0f, 0.0f, 1.0f};
    const float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    float K_inv[9], KR[9];
    for (int r = 0; r < 3; ++r) Kt[r] = hm_sum3(K[r * 3] * T[3], K[r * 3 + 1] * T[7], K[r * 3 + 2] * T[11]);
    mat3_inverse(K, K_inv);
    mat3_mul(K, R, KR);
    mat3_mul(KR, K_inv, KRK_inv);
}

// x = JTJ.ldlt().solve(-JTr) (DenseOdometryFunction.cpp:404): symmetric-pivoted LDL^T in double;
// zero pivots give zero components (Eigen's pseudo-inverse of D).  Every index below is a
// compile-time constant after unrolling (the pivot swap is a chain of predicated exchanges), so on
// the device the 6x6 system lives in registers instead of scratch memory.
OP_HD void ldlt_swap(double& a, double& b, bool c) { const double t = a; a = c ? b : a; b = c ? t : b; }
// Fast path for the usual case: JTJ is symmetric positive definite, where LDL^T needs no pivoting to
// be backward stable (it is Cholesky); ~200 dependent flops in registers.  Returns false -- and the
// caller takes the pivoted route below -- as soon as a pivot is not safely positive.
OP_HD bool ldlt_solve6_spd(const double JTJ[36], const double JTr[6], float x[6]) {
    double A[6][6], y[6];
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) A[i][j] = 0.5 * (JTJ[i * 6 + j] + JTJ[j * 6 + i]);
        dmax = A[i][i] > dmax ? A[i][i] : dmax;
    }
    const double floor_ = dmax * 1e-9;
    bool ok = dmax > 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double d = A[k][k];
        ok = ok && d > floor_;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double l = A[i][k] * inv;
#pragma unroll
            for (int j = k + 1; j <= i; ++j) A[i][j] -= l * A[j][k];   // lower triangle only; A[j][k] still holds d*L(j,k)
        }
#pragma unroll
        for (int i = k + 1; i < 6; ++i) A[i][k] *= inv;
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = -JTr[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= A[i][j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = y[i] / A[i][i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= A[j][i] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = static_cast<float>(y[i]);
    return true;
}

OP_HD void ldlt_solve6(const double JTJ[36], const double JTr[6], float x[6]) {
    if (ldlt_solve6_spd(JTJ, JTr, x)) return;
    double A[6][6], b[6], y[6];
    int perm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        perm[i] = i; b[i] = -JTr[i];
#pragma unroll
        for (int j = 0; j < 6; ++j) A[i][j] = 0.5 * (JTJ[i * 6 + j] + JTJ[j * 6 + i]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) { const double v = fabs(A[i][i]); if (v > best) { best = v; p = i; } }
#pragma unroll
        for (int q = k + 1; q < 6; ++q) { // exchange rows/columns k <-> q iff q is the pivot
            const bool c = q == p;
#pragma unroll
            for (int j = 0; j < 6; ++j) ldlt_swap(A[k][j], A[q][j], c);
#pragma unroll
            for (int j = 0; j < 6; ++j) ldlt_swap(A[j][k], A[j][q], c);
            ldlt_swap(b[k], b[q], c);
            const int t = perm[k]; perm[k] = c ? perm[q] : perm[k]; perm[q] = c ? t : perm[q];
        }
        const double d = A[k][k];
        if (d != 0) {
#pragma unroll
            for (int i = k + 1; i < 6; ++i) {
                const double l = A[i][k] / d;
#pragma unroll
                for (int j = k + 1; j < 6; ++j) A[i][j] -= l * A[k][j];
                A[i][k] = l;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= A[i][j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = A[i][i] != 0 ? y[i] / A[i][i] : 0.0;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= A[j][i] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (perm[i] == j) x[j] = static_cast<float>(y[i]);
}

// ---- validation mode of the dense tracker (OP_TRACK_SUMS_REFERENCE_F32) ---------------------------------------------
// The reference's accumulation (DenseOdometryFunction.cpp:297-381): for every correspondence in raster order of the source
// pixel, and for each of its Jacobian rows, `JTJ.noalias() += J * J^T; JTr += J * r` in float32 -- one rounded product and
// one rounded add per entry, sequentially.  rows: 14 floats per SOURCE PIXEL ({J0[6], r0, J1[6], r1}), pair_t[s] >= 0 marks
// the accepted pixels, rows_per_pixel = 2 for the hybrid term.
inline void track_sums_reference_order(const float* rows, const int* pair_t, size_t npix, int rows_per_pixel, float JTJ[36], float JTr[6], size_t* n_pairs) {
    for (int k = 0; k < 36; ++k) JTJ[k] = 0.0f;
    for (int k = 0; k < 6; ++k) JTr[k] = 0.0f;
    size_t n = 0;
    for (size_t s = 0; s < npix; ++s) {
        if (pair_t[s] < 0) continue;
        ++n;
        for (int m = 0; m < rows_per_pixel; ++m) {
            const float* J = rows + 14 * s + 7 * m;
            const float r = J[6];
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) JTJ[a * 6 + b] += J[a] * J[b];
                JTr[a] += J[a] * r;
            }
        }
    }
    *n_pairs = n;
}

// x = JTJ.ldlt().solve(-JTr) (DenseOdometryFunction.cpp:404) from the float32 sums: symmetric-pivoted LDL^T in double, the
// textbook loop (largest remaining diagonal first, zero pivots give zero components as Eigen's pseudo-inverse of D does).
// The validation mode uses this form -- one fixed sequence of operations on the host -- rather than the register-resident
// variants above, so that identical sums give an identical step.
inline void ldlt_solve6_float_sums(const float JTJ[36], const float JTr[6], float x[6]) {
    double A[36], b[6], y[6];
    int perm[6];
    for (int i = 0; i < 6; ++i) {
        perm[i] = i; b[i] = -static_cast<double>(JTr[i]);
        for (int j = 0; j < 6; ++j) A[i * 6 + j] = 0.5 * (static_cast<double>(JTJ[i * 6 + j]) + static_cast<double>(JTJ[j * 6 + i]));
    }
    for (int k = 0; k < 6; ++k) {
        int p = k;
        for (int i = k + 1; i < 6; ++i) if (fabs(A[i * 7]) > fabs(A[p * 7])) p = i;
        if (p != k) {
            for (int j = 0; j < 6; ++j) { const double t = A[k * 6 + j]; A[k * 6 + j] = A[p * 6 + j]; A[p * 6 + j] = t; }
            for (int j = 0; j < 6; ++j) { const double t = A[j * 6 + k]; A[j * 6 + k] = A[j * 6 + p]; A[j * 6 + p] = t; }
            const int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
            const double tb = b[k]; b[k] = b[p]; b[p] = tb;
        }
        const double d = A[k * 7];
        if (d == 0) continue;
        for (int i = k + 1; i < 6; ++i) {
            const double l = A[i * 6 + k] / d;
            for (int j = k + 1; j < 6; ++j) A[i * 6 + j] -= l * A[k * 6 + j];
            A[i * 6 + k] = l;
        }
    }
    for (int i = 0; i < 6; ++i) { double acc = b[i]; for (int j = 0; j < i; ++j) acc -= A[i * 6 + j] * y[j]; y[i] = acc; }
    for (int i = 0; i < 6; ++i) y[i] = A[i * 7] != 0 ? y[i] / A[i * 7] : 0.0;
    for (int i = 5; i >= 0; --i) { double acc = y[i]; for (int j = i + 1; j < 6; ++j) acc -= A[j * 6 + i] * y[j]; y[i] = acc; }
    for (int i = 0; i < 6; ++i) x[perm[i]] = static_cast<float>(y[i]);
}

inline uint64_t hash_key(int32_t x, int32_t y, int32_t z) { // Geometry/Geometry.h:101-112
    return (static_cast<uint64_t>(static_cast<int64_t>(x)) * 73856093ULL) ^
           (static_cast<uint64_t>(static_cast<int64_t>(y)) * 19349663ULL) ^
           (static_cast<uint64_t>(static_cast<int64_t>(z)) * 83492791ULL);
}

} // namespace op_host
