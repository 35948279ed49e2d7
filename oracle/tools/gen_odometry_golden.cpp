// gen_odometry_golden.cpp -- generates tests/golden/odometry_golden.json.
//
// Build-container only (needs /root/reference/3rdparty/{Eigen,Sophus}); the JSON is the committed
// fixture.  It pins the third-party (Eigen 3.3.7 / Sophus) arithmetic that the dense RGB-D tracker
// reaches through the expression shapes used on that path: Matrix3f::inverse(), K*R*K_inv,
// d * KRK_inv * Point3(j,i,1.0) + Kt and the (int)(x / z + 0.5) rounding, the rank-1
// `JTJ.noalias() += J * J.transpose()` accumulation, `JTJ.ldlt().solve(-JTr)` and
// `SE3::exp(delta).matrix() * T`.  Floats are recorded as uint32 bit patterns.  It contains no
// OnePiece source.  Build + run: oracle/tools/gen_golden.sh
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/LU>
#include <Eigen/Cholesky>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "sophus/se3.hpp"

using Eigen::Matrix3f; using Eigen::Matrix4f; using Eigen::Vector3f;
typedef Eigen::Matrix<float, 6, 6> Matrix6f;
typedef Eigen::Matrix<float, 6, 1> Vector6f;

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static FILE* out;
static void arr_bits(const char* name, const float* p, int n, bool comma = true) {
    fprintf(out, "\"%s\": [", name);
    for (int i = 0; i < n; ++i) fprintf(out, "%s%u", i ? ", " : "", bits(p[i]));
    fprintf(out, "]%s", comma ? ", " : "");
}
static void arr_int(const char* name, const int* p, int n, bool comma = true) {
    fprintf(out, "\"%s\": [", name);
    for (int i = 0; i < n; ++i) fprintf(out, "%s%d", i ? ", " : "", p[i]);
    fprintf(out, "]%s", comma ? ", " : "");
}
static void rm3(const Matrix3f& M, float* p) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) p[r * 3 + c] = M(r, c); }
static void rm4(const Matrix4f& M, float* p) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) p[r * 4 + c] = M(r, c); }

int main(int argc, char** argv) {
    out = fopen(argc > 1 ? argv[1] : "odometry_golden.json", "w");
    std::mt19937 g(20260928);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    fprintf(out, "{\n\"generator\": \"oracle/tools/gen_odometry_golden.cpp against /root/reference/3rdparty Eigen 3.3.7 + Sophus, g++ -O3 -msse4.2\",\n");

    // ---- 1. projective association arithmetic
    fprintf(out, "\"projective\": [\n");
    const int NC = 16, NS = 48;
    for (int k = 0; k < NC; ++k) {
        float sc = (k % 3 == 0) ? 1.f : (k % 3 == 1 ? 0.5f : 0.25f);   // pyramid level intrinsics
        float fx = sc * (514.817f + 10 * u(g)), fy = sc * (515.375f + 10 * u(g));
        float cx = sc * (318.771f + 4 * u(g)), cy = sc * (238.447f + 4 * u(g));
        int W = (int)(640 * sc), H = (int)(480 * sc);
        Matrix3f K; K << fx, 0, cx, 0, fy, cy, 0, 0, 1;
        Matrix4f T = Matrix4f::Identity();
        if (k) {
            Vector3f ax(u(g), u(g), u(g)); ax.normalize();
            T.block<3, 3>(0, 0) = Eigen::AngleAxisf(0.06f * u(g), ax).toRotationMatrix();
            T.block<3, 1>(0, 3) = Vector3f(0.08f * u(g), 0.08f * u(g), 0.08f * u(g));
        }
        Matrix3f R = T.block<3, 3>(0, 0);
        Vector3f t = T.block<3, 1>(0, 3);
        Vector3f Kt = K * t;
        Matrix3f K_inv = K.inverse();
        Matrix3f KRK_inv = K * R * K_inv;
        float a[16], ki[9], krk[9], cam[4] = {fx, fy, cx, cy};
        rm4(T, a); rm3(K_inv, ki); rm3(KRK_inv, krk);
        std::vector<float> D(NS), UV(3 * NS);
        std::vector<int> JI(2 * NS), UVT(2 * NS);
        for (int s = 0; s < NS; ++s) {
            int j = (int)(g() % W), i = (int)(g() % H);
            if (s < 4) { j = (s & 1) ? W - 1 : 0; i = (s & 2) ? H - 1 : 0; }
            float d_s = 0.5f + 3.5f * (0.5f + 0.5f * u(g));
            Vector3f uv_in_s = d_s * KRK_inv * Vector3f(j, i, 1.0) + Kt;
            float transformed_d_s = uv_in_s(2);
            int u_t = (int)(uv_in_s(0) / transformed_d_s + 0.5);
            int v_t = (int)(uv_in_s(1) / transformed_d_s + 0.5);
            D[s] = d_s; JI[2 * s] = j; JI[2 * s + 1] = i;
            for (int c = 0; c < 3; ++c) UV[3 * s + c] = uv_in_s(c);
            UVT[2 * s] = u_t; UVT[2 * s + 1] = v_t;
        }
        fprintf(out, "  {"); arr_bits("cam", cam, 4); arr_bits("T", a, 16); arr_bits("K_inv", ki, 9); arr_bits("KRK_inv", krk, 9);
        arr_bits("Kt", Kt.data(), 3); arr_bits("d", D.data(), NS); arr_int("ji", JI.data(), 2 * NS);
        arr_bits("uv", UV.data(), 3 * NS); arr_int("uvt", UVT.data(), 2 * NS, false);
        fprintf(out, "}%s\n", k < NC - 1 ? "," : "");
    }
    fprintf(out, "],\n");

    // ---- 2. Gauss-Newton normal equations: rank-1 float accumulation, LDLT solve, exp, left-multiply
    fprintf(out, "\"gauss_newton\": [\n");
    const int NG = 8;
    for (int k = 0; k < NG; ++k) {
        int n = 40 + 40 * k;
        std::vector<float> J(6 * n), Rr(n);
        Matrix6f JTJ; JTJ.setZero(); Vector6f JTr; JTr.setZero();
        float r2 = 0.0;
        for (int i = 0; i < n; ++i) {
            Vector6f row;
            // shape of a photometric/depth row: small translational part, larger rotational part
            row << 0.6f * u(g), 0.6f * u(g), 0.3f * u(g) - (i & 1), 1.5f * u(g), 1.5f * u(g), 0.8f * u(g);
            float r = 0.02f * u(g);
            for (int c = 0; c < 6; ++c) J[6 * i + c] = row(c);
            Rr[i] = r;
            JTJ.noalias() += row * row.transpose();
            JTr.noalias() += row * r;
            r2 += r * r;
        }
        Vector6f delta = JTJ.ldlt().solve(-JTr);
        Matrix4f T = Matrix4f::Identity();
        Vector3f ax(u(g), u(g), u(g)); ax.normalize();
        T.block<3, 3>(0, 0) = Eigen::AngleAxisf(0.2f * u(g), ax).toRotationMatrix();
        T.block<3, 1>(0, 3) = Vector3f(0.3f * u(g), 0.3f * u(g), 0.3f * u(g));
        Matrix4f E = Sophus::SE3Group<float>::exp(delta).matrix();
        Matrix4f Tn = E * T;
        float jtj[36], a[16], b[16];
        for (int r_ = 0; r_ < 6; ++r_) for (int c = 0; c < 6; ++c) jtj[r_ * 6 + c] = JTJ(r_, c);
        rm4(T, a); rm4(Tn, b);
        fprintf(out, "  {"); arr_bits("J", J.data(), 6 * n); arr_bits("r", Rr.data(), n); arr_bits("JTJ", jtj, 36);
        arr_bits("JTr", JTr.data(), 6); arr_bits("r2", &r2, 1); arr_bits("delta", delta.data(), 6);
        arr_bits("T", a, 16); arr_bits("T_new", b, 16, false);
        fprintf(out, "}%s\n", k < NG - 1 ? "," : "");
    }
    fprintf(out, "]\n}\n");
    fclose(out);
    return 0;
}
