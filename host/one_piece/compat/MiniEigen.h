// MiniEigen.h -- fixed-size matrix look-alikes for building libone_piece_hip_host WITHOUT Eigen.
//
// The reference's public types are Eigen typedefs (Geometry/Geometry.h:34-71).  When this library is built inside the
// reference tree (or anywhere Eigen is on the include path: -DONEPIECE_HAVE_EIGEN) Geometry/Geometry.h uses the real
// Eigen types and this file is not included.  On a machine without Eigen (the GPU box of this repository's tests)
// the same headers are compiled against the small column-major matrix below, which offers the members the hot-path
// surface and its callers touch: operator()(r,c) / (i), data(), Zero(), Identity(), setZero(), +, -, scalar *, /,
// matrix products, transpose(), inverse() for 4x4 (through the C-ABI's op_mat4_inverse, i.e. Eigen's own order),
// normalize() / normalized(), read-only head<N>() / tail<N>() / block<P,Q>(r,c) / col(c) / row(r) (copies, which is all
// the drivers of the path need: `T.block<3,3>(0,0)`, `plane.head<3>().dot(p)`), stream output.  Storage order, size and alignment of the 3-vectors match Eigen's (12-byte xyz), so
// std::vector<Point3> is the same contiguous float array either way.
#pragma once
#include <cmath>
#include <cstddef>
#include <ostream>
#include <vector>

#include "onepiece_hip.h"

namespace one_piece {
namespace compat {

template <class T, int R, int C>
struct Mat {
    T v[R * C]; // column-major, like Eigen's default

    Mat() { for (int i = 0; i < R * C; ++i) v[i] = T(0); }
    Mat(T x, T y) { static_assert(R * C == 2, "2-vector constructor"); v[0] = x; v[1] = y; }
    Mat(T x, T y, T z) { static_assert(R * C == 3, "3-vector constructor"); v[0] = x; v[1] = y; v[2] = z; }
    Mat(T x, T y, T z, T w) { static_assert(R * C == 4, "4-vector constructor"); v[0] = x; v[1] = y; v[2] = z; v[3] = w; }

    T& operator()(int r, int c) { return v[c * R + r]; }
    const T& operator()(int r, int c) const { return v[c * R + r]; }
    T& operator()(int i) { return v[i]; }
    const T& operator()(int i) const { return v[i]; }
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
    T* data() { return v; }
    const T* data() const { return v; }
    static int rows() { return R; }
    static int cols() { return C; }

    static Mat Zero() { return Mat(); }
    static Mat Identity() { Mat m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
    void setZero() { for (int i = 0; i < R * C; ++i) v[i] = T(0); }
    void setIdentity() { *this = Identity(); }

    Mat operator+(const Mat& o) const { Mat m; for (int i = 0; i < R * C; ++i) m.v[i] = v[i] + o.v[i]; return m; }
    Mat operator-(const Mat& o) const { Mat m; for (int i = 0; i < R * C; ++i) m.v[i] = v[i] - o.v[i]; return m; }
    Mat operator-() const { Mat m; for (int i = 0; i < R * C; ++i) m.v[i] = -v[i]; return m; }
    Mat operator*(T s) const { Mat m; for (int i = 0; i < R * C; ++i) m.v[i] = v[i] * s; return m; }
    Mat operator/(T s) const { Mat m; for (int i = 0; i < R * C; ++i) m.v[i] = v[i] / s; return m; }
    Mat& operator+=(const Mat& o) { for (int i = 0; i < R * C; ++i) v[i] += o.v[i]; return *this; }
    Mat& operator-=(const Mat& o) { for (int i = 0; i < R * C; ++i) v[i] -= o.v[i]; return *this; }
    Mat& operator*=(T s) { for (int i = 0; i < R * C; ++i) v[i] *= s; return *this; }
    Mat& operator/=(T s) { for (int i = 0; i < R * C; ++i) v[i] /= s; return *this; }
    bool operator==(const Mat& o) const { for (int i = 0; i < R * C; ++i) if (!(v[i] == o.v[i])) return false; return true; }
    bool operator!=(const Mat& o) const { return !(*this == o); }

    Mat<T, C, R> transpose() const { Mat<T, C, R> m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(c, r) = (*this)(r, c); return m; }
    T dot(const Mat& o) const { T s = T(0); for (int i = 0; i < R * C; ++i) s += v[i] * o.v[i]; return s; }
    T squaredNorm() const { return dot(*this); }
    T norm() const { return std::sqrt(squaredNorm()); }
    // Eigen's normalize(): divides by the norm when it is positive
    void normalize() { const T n = norm(); if (n > T(0)) for (int i = 0; i < R * C; ++i) v[i] /= n; }
    Mat normalized() const { Mat m = *this; m.normalize(); return m; }
    template <int N> Mat<T, N, 1> head() const { static_assert(C == 1 && N <= R, "head<N>() of a column vector"); Mat<T, N, 1> m; for (int i = 0; i < N; ++i) m(i) = v[i]; return m; }
    template <int N> Mat<T, N, 1> tail() const { static_assert(C == 1 && N <= R, "tail<N>() of a column vector"); Mat<T, N, 1> m; for (int i = 0; i < N; ++i) m(i) = v[R - N + i]; return m; }
    template <int P, int Q> Mat<T, P, Q> block(int r0, int c0) const { Mat<T, P, Q> m; for (int r = 0; r < P; ++r) for (int c = 0; c < Q; ++c) m(r, c) = (*this)(r0 + r, c0 + c); return m; }
    Mat<T, R, 1> col(int c) const { Mat<T, R, 1> m; for (int r = 0; r < R; ++r) m(r) = (*this)(r, c); return m; }
    Mat<T, 1, C> row(int r) const { Mat<T, 1, C> m; for (int c = 0; c < C; ++c) m(0, c) = (*this)(r, c); return m; }
    Mat cross(const Mat& o) const {
        static_assert(R * C == 3, "cross product of 3-vectors");
        return Mat(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
    }
    // 4x4 float only: Eigen 3.3.7's SSE cofactor kernel, restated inside the library (op_mat4_inverse)
    Mat inverse() const {
        static_assert(R == 4 && C == 4, "inverse() is provided for 4x4 matrices");
        float in[16], out[16];
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) in[r * 4 + c] = static_cast<float>((*this)(r, c));
        op_mat4_inverse(in, out);
        Mat m;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m(r, c) = static_cast<T>(out[r * 4 + c]);
        return m;
    }
};

template <class T, int R, int C>
inline Mat<T, R, C> operator*(T s, const Mat<T, R, C>& m) { return m * s; }

// Run-time sized column vector (the reference's geometry::VectorX = Eigen::Matrix<scalar, Dynamic, 1>, Geometry/Geometry.h:45): what a 33-bin
// FPFH feature is stored in (Registration/3DFeature.h).  The members the feature / registration code and its callers touch.
template <class T>
struct VecX {
    std::vector<T> v;
    VecX() {}
    explicit VecX(int n) : v(static_cast<size_t>(n), T(0)) {}
    void resize(int n) { v.resize(static_cast<size_t>(n)); }
    void setZero() { for (size_t i = 0; i < v.size(); ++i) v[i] = T(0); }
    int rows() const { return static_cast<int>(v.size()); }
    int cols() const { return 1; }
    int size() const { return static_cast<int>(v.size()); }
    T& operator()(int i) { return v[static_cast<size_t>(i)]; }
    const T& operator()(int i) const { return v[static_cast<size_t>(i)]; }
    T& operator[](int i) { return v[static_cast<size_t>(i)]; }
    const T& operator[](int i) const { return v[static_cast<size_t>(i)]; }
    T* data() { return v.data(); }
    const T* data() const { return v.data(); }
    VecX& operator+=(const VecX& o) { for (size_t i = 0; i < v.size() && i < o.v.size(); ++i) v[i] += o.v[i]; return *this; }
    VecX operator*(T s) const { VecX m = *this; for (size_t i = 0; i < m.v.size(); ++i) m.v[i] *= s; return m; }
    T sum() const { T s = T(0); for (size_t i = 0; i < v.size(); ++i) s += v[i]; return s; }
};
template <class T>
inline std::ostream& operator<<(std::ostream& os, const VecX<T>& m) {
    for (int i = 0; i < m.rows(); ++i) os << (i ? "\n" : "") << m(i);
    return os;
}

// matrix product, accumulated column by column as Eigen's coefficient-based product does
template <class T, int R, int K, int C>
inline Mat<T, R, C> operator*(const Mat<T, R, K>& a, const Mat<T, K, C>& b) {
    Mat<T, R, C> m;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            T s = a(r, 0) * b(0, c);
            for (int k = 1; k < K; ++k) s = s + a(r, k) * b(k, c);
            m(r, c) = s;
        }
    return m;
}

template <class T, int R, int C>
inline std::ostream& operator<<(std::ostream& os, const Mat<T, R, C>& m) {
    for (int r = 0; r < R; ++r) {
        for (int c = 0; c < C; ++c) os << (c ? " " : "") << m(r, c);
        if (r + 1 < R) os << "\n";
    }
    return os;
}

} // namespace compat
} // namespace one_piece
