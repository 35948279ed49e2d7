// gen_eigen_golden.cpp -- generates tests/golden/eigen_golden.json.
//
// Runs ONLY in the build container (needs /root/reference/3rdparty/{Eigen,Sophus}); the JSON it
// writes is the committed fixture.  It exercises the reference's vendored third-party arithmetic
// (Eigen 3.3.7, Sophus) through the same kinds of expressions the hot path uses, compiled with the
// reference's flags (-O3 -msse4.2), and records inputs + outputs (floats as uint32 bit patterns so
// the comparison can be bit-exact where that is the bar).  It contains no OnePiece source.
//
// Build + run: see oracle/tools/gen_golden.sh
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/SVD>
#include <Eigen/LU>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "sophus/se3.hpp"

using Eigen::Matrix3f; using Eigen::Matrix4f; using Eigen::Vector3f; using Eigen::Vector4f;
typedef Eigen::Matrix<float, 6, 6> Matrix6f;
typedef Eigen::Matrix<float, 6, 1> Vector6f;

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static FILE* out;
static void arr_bits(const char* name, const float* p, int n, bool comma = true) {
    fprintf(out, "\"%s\": [", name);
    for (int i = 0; i < n; ++i) fprintf(out, "%s%u", i ? ", " : "", bits(p[i]));
    fprintf(out, "]%s", comma ? ", " : "");
}
static void rowmajor(const Matrix4f& M, float* p) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) p[r * 4 + c] = M(r, c); }

int main(int argc, char** argv) {
    out = fopen(argc > 1 ? argv[1] : "eigen_golden.json", "w");
    std::mt19937 g(20240928);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    fprintf(out, "{\n\"generator\": \"oracle/tools/gen_eigen_golden.cpp against /root/reference/3rdparty Eigen 3.3.7 + Sophus, g++ -O3 -msse4.2\",\n");

    // ---- 1. Matrix4f::inverse() (SSE path) on rigid poses and general matrices
    fprintf(out, "\"inverse\": [\n");
    for (int k = 0; k < 48; ++k) {
        Matrix4f M = Matrix4f::Identity();
        if (k % 3 != 2) {
            Vector3f ax(u(g), u(g), u(g)); ax.normalize();
            M.block<3, 3>(0, 0) = Eigen::AngleAxisf(3.0f * u(g), ax).toRotationMatrix();
            M.block<3, 1>(0, 3) = Vector3f(4 * u(g), 4 * u(g), 4 * u(g));
        } else {
            for (int i = 0; i < 16; ++i) M(i) = 2 * u(g);
        }
        Matrix4f Mi = M.inverse();
        float a[16], b[16]; rowmajor(M, a); rowmajor(Mi, b);
        fprintf(out, "  {"); arr_bits("m", a, 16); arr_bits("inv", b, 16, false); fprintf(out, "}%s\n", k < 47 ? "," : "");
    }
    fprintf(out, "],\n");

    // ---- 2. fixed-size products / reductions: T*(x,y,z,1), head<3>/w, plane.head<3>().dot(p)+d,
    //         Matrix3f*Vector3f + t, squaredNorm, cross/normalize plane construction
    fprintf(out, "\"products\": [\n");
    for (int k = 0; k < 64; ++k) {
        Matrix4f T; for (int i = 0; i < 16; ++i) T(i) = 2 * u(g);
        if (k % 2) { T.row(3) << 0, 0, 0, 1; }
        Vector3f p(3 * u(g), 3 * u(g), 3 * u(g));
        Vector4f q = T * Vector4f(p(0), p(1), p(2), 1.0);
        Vector3f h = q.head<3>() / q(3);
        Vector4f plane(u(g), u(g), u(g), u(g));
        float dist = plane.head<3>().dot(p) + plane(3);
        Vector3f rs = T.block<3, 3>(0, 0) * p + T.block<3, 1>(0, 3) - h;
        float sn = rs.squaredNorm();
        Vector3f p2(u(g), u(g), u(g)), p3(u(g), u(g), u(g));
        Vector3f nrm = (p2 - p).cross(p3 - p); nrm.normalize();
        double dd = -p.dot(nrm);
        Vector4f pl(nrm(0), nrm(1), nrm(2), dd);
        float a[16]; rowmajor(T, a);
        fprintf(out, "  {"); arr_bits("T", a, 16); arr_bits("p", p.data(), 3); arr_bits("q", q.data(), 4); arr_bits("h", h.data(), 3);
        arr_bits("plane", plane.data(), 4); arr_bits("dist", &dist, 1); arr_bits("rs", rs.data(), 3); arr_bits("sqnorm", &sn, 1);
        arr_bits("p2", p2.data(), 3); arr_bits("p3", p3.data(), 3); arr_bits("get_plane", pl.data(), 4, false);
        fprintf(out, "}%s\n", k < 63 ? "," : "");
    }
    fprintf(out, "],\n");

    // ---- 3. running weighted mean with Vector3f colour (TSDF voxel update arithmetic)
    fprintf(out, "\"voxel_update\": [\n");
    for (int k = 0; k < 64; ++k) {
        float sdf = 0.1f * u(g), w = (float)(1 + (g() % 200)), nsdf = 0.1f * u(g);
        Vector3f col(0.5f + 0.5f * u(g), 0.5f + 0.5f * u(g), 0.5f + 0.5f * u(g));
        unsigned char b[3] = {(unsigned char)(g() % 256), (unsigned char)(g() % 256), (unsigned char)(g() % 256)};
        Vector3f ncol = Vector3f(b[0], b[1], b[2]) / 255.0;
        float ow = 1.0;
        float rw = w + ow;
        float rs = (w * sdf + ow * nsdf) / rw;
        Vector3f rc = (w * col + ow * ncol) / rw;
        float in[5] = {sdf, w, col(0), col(1), col(2)}, nw[4] = {nsdf, ncol(0), ncol(1), ncol(2)}, res[5] = {rs, rw, rc(0), rc(1), rc(2)};
        float bf[3] = {(float)b[0], (float)b[1], (float)b[2]};
        fprintf(out, "  {"); arr_bits("old", in, 5); arr_bits("bytes", bf, 3); arr_bits("new", nw, 4); arr_bits("result", res, 5, false);
        fprintf(out, "}%s\n", k < 63 ? "," : "");
    }
    fprintf(out, "],\n");

    // ---- 4. Sophus SE3::exp
    fprintf(out, "\"se3_exp\": [\n");
    for (int k = 0; k < 32; ++k) {
        Vector6f x; float s = k < 8 ? 1e-6f : (k < 20 ? 0.02f : 1.5f);
        for (int i = 0; i < 6; ++i) x(i) = s * u(g);
        if (k == 0) x.setZero();
        Matrix4f M = Sophus::SE3Group<float>::exp(x).matrix();
        float a[16]; rowmajor(M, a);
        fprintf(out, "  {"); arr_bits("x", x.data(), 6); arr_bits("T", a, 16, false); fprintf(out, "}%s\n", k < 31 ? "," : "");
    }
    fprintf(out, "],\n");

    // ---- 5. point-to-plane normal equations (float accumulation) + JacobiSVD solve + exp
    fprintf(out, "\"p2plane\": [\n");
    for (int k = 0; k < 6; ++k) {
        int n = 40 + 30 * k;
        std::vector<float> S(3 * n), Tg(3 * n), N(3 * n);
        Matrix6f JTJ = Matrix6f::Zero(); Vector6f JTr = Vector6f::Zero();
        for (int i = 0; i < n; ++i) {
            Vector3f t(2 * u(g), 2 * u(g), 1.5f + u(g)), nn(u(g), u(g), u(g)); nn.normalize();
            Vector3f s = t + 0.004f * Vector3f(u(g), u(g), u(g));
            for (int c = 0; c < 3; ++c) { S[3 * i + c] = s(c); Tg[3 * i + c] = t(c); N[3 * i + c] = nn(c); }
            Vector6f row;
            double r = (nn.transpose() * s - nn.transpose() * t)(0);
            row.block<3, 1>(0, 0) = nn;
            row.block<3, 1>(3, 0) = s.cross(nn);
            JTJ.noalias() += row * row.transpose();
            JTr.noalias() += r * row;
        }
        Eigen::JacobiSVD<Eigen::MatrixXf> svd(JTJ, Eigen::ComputeThinU | Eigen::ComputeThinV);
        Vector6f x = svd.solve(-JTr);
        Matrix4f M = Sophus::SE3Group<float>::exp(x).matrix();
        float a[16]; rowmajor(M, a);
        float jtj[36]; for (int r2 = 0; r2 < 6; ++r2) for (int c = 0; c < 6; ++c) jtj[r2 * 6 + c] = JTJ(r2, c);
        fprintf(out, "  {"); arr_bits("src", S.data(), 3 * n); arr_bits("tgt", Tg.data(), 3 * n); arr_bits("nrm", N.data(), 3 * n);
        arr_bits("JTJ", jtj, 36); arr_bits("JTr", JTr.data(), 6); arr_bits("x", x.data(), 6); arr_bits("T", a, 16, false);
        fprintf(out, "}%s\n", k < 5 ? "," : "");
    }
    fprintf(out, "],\n");

    // ---- 6. Kabsch via JacobiSVD (rigid fit of correspondences)
    fprintf(out, "\"kabsch\": [\n");
    for (int k = 0; k < 6; ++k) {
        int n = 30 + 50 * k;
        Vector3f ax(u(g), u(g), u(g)); ax.normalize();
        Matrix3f Rt = Eigen::AngleAxisf(0.3f * u(g), ax).toRotationMatrix();
        Vector3f tt(0.2f * u(g), 0.2f * u(g), 0.2f * u(g));
        std::vector<float> S(3 * n), Tg(3 * n);
        std::vector<Vector3f> sv(n), tv(n);
        Vector3f ms = Vector3f::Zero(), mt = Vector3f::Zero();
        for (int i = 0; i < n; ++i) {
            sv[i] = Vector3f(2 * u(g), 2 * u(g), 2 * u(g));
            tv[i] = Rt * sv[i] + tt + 0.002f * Vector3f(u(g), u(g), u(g));
            for (int c = 0; c < 3; ++c) { S[3 * i + c] = sv[i](c); Tg[3 * i + c] = tv[i](c); }
            ms += sv[i]; mt += tv[i];
        }
        ms /= n; mt /= n;
        Matrix3f W = Matrix3f::Zero();
        for (int i = 0; i < n; ++i) W += (sv[i] - ms) * (tv[i] - mt).transpose();
        Eigen::JacobiSVD<Eigen::MatrixXf> svd(W, Eigen::ComputeThinU | Eigen::ComputeThinV);
        Matrix3f UT = svd.matrixU().transpose(), V = svd.matrixV();
        Matrix3f R = V * UT;
        if (R.determinant() < 0) { V.col(2) = -V.col(2); R = V * UT; }
        Matrix4f T = Matrix4f::Zero();
        T.block<3, 3>(0, 0) = R; T.block<3, 1>(0, 3) = mt - R * ms; T(3, 3) = 1;
        float a[16]; rowmajor(T, a);
        fprintf(out, "  {"); arr_bits("src", S.data(), 3 * n); arr_bits("tgt", Tg.data(), 3 * n); arr_bits("T", a, 16, false);
        fprintf(out, "}%s\n", k < 5 ? "," : "");
    }
    fprintf(out, "]\n}\n");
    fclose(out);
    return 0;
}
