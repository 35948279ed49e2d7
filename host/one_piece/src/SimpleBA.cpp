// SimpleBA.cpp -- optimization::SimpleBA (Optimization/SimpleBA.h): pose-only Gauss-Newton over submap correspondences.  Host C++ on
// example/DenseFusion's path (DenseSlam.cpp:121-125); restates src/Optimization/SimpleBA.cpp:19-155 for this library's types.
#include "Optimization/SimpleBA.h"

#include <cmath>

namespace one_piece {
namespace optimization {

namespace {

// J^T J and -J^T r of one pair for the two poses it ties (SimpleBA.cpp:41-75): r = a - b with a = pose_s * p, b = pose_t * q;
// d r / d xi_s = [I | -skew(a)], d r / d xi_t = [-I | skew(b)] for left-multiplied increments xi = (translation, rotation)
struct PairBlocks {
    double ss[36], tt[36], st[36], ts[36], rs[6], rt[6];
    PairBlocks() { for (int i = 0; i < 36; ++i) ss[i] = tt[i] = st[i] = ts[i] = 0; for (int i = 0; i < 6; ++i) rs[i] = rt[i] = 0; }
};

inline void Rows(const geometry::Point3& a, double sign, double J[3][6]) { // sign * [I | -skew(a)]
    const double x = a(0), y = a(1), z = a(2);
    const double S[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { J[r][c] = sign * (r == c ? 1.0 : 0.0); J[r][3 + c] = -sign * S[r][c]; }
}

PairBlocks Accumulate(const Correspondence& c, const geometry::SE3List& poses) {
    PairBlocks B;
    const geometry::SE3& Ps = poses[static_cast<size_t>(c.source_id)];
    const geometry::SE3& Pt = poses[static_cast<size_t>(c.target_id)];
    for (size_t i = 0; i != c.correspondence_set.size(); ++i) {
        const geometry::Point3& p = c.correspondence_set[i].first;
        const geometry::Point3& q = c.correspondence_set[i].second;
        geometry::Point3 a, b;
        for (int r = 0; r < 3; ++r) {
            a(r) = (Ps(r, 0) * p(0) + Ps(r, 1) * p(1) + Ps(r, 2) * p(2)) + Ps(r, 3);
            b(r) = (Pt(r, 0) * q(0) + Pt(r, 1) * q(1) + Pt(r, 2) * q(2)) + Pt(r, 3);
        }
        const double res[3] = {static_cast<double>(a(0)) - b(0), static_cast<double>(a(1)) - b(1), static_cast<double>(a(2)) - b(2)};
        double Js[3][6], Jt[3][6];
        Rows(a, 1.0, Js);
        Rows(b, -1.0, Jt);
        for (int u = 0; u < 6; ++u) {
            for (int v = 0; v < 6; ++v)
                for (int r = 0; r < 3; ++r) {
                    B.ss[u * 6 + v] += Js[r][u] * Js[r][v];
                    B.tt[u * 6 + v] += Jt[r][u] * Jt[r][v];
                    B.st[u * 6 + v] += Js[r][u] * Jt[r][v];
                    B.ts[u * 6 + v] += Jt[r][u] * Js[r][v];
                }
            for (int r = 0; r < 3; ++r) { B.rs[u] -= Js[r][u] * res[r]; B.rt[u] -= Jt[r][u] * res[r]; }
        }
    }
    return B;
}

// A x = b for a symmetric A (n x n, row-major) by LDL^T without pivoting, as SimplicialLDLT factorises (SimpleBA.cpp:135-138).  A zero pivot
// (a pose no correspondence touches) leaves its unknown at zero instead of dividing.
void SolveLDLT(std::vector<double>& A, std::vector<double>& b, int n) {
    std::vector<double> D(static_cast<size_t>(n), 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[static_cast<size_t>(j) * n + j];
        for (int k = 0; k < j; ++k) d -= A[static_cast<size_t>(j) * n + k] * A[static_cast<size_t>(j) * n + k] * D[static_cast<size_t>(k)];
        D[static_cast<size_t>(j)] = d;
        for (int i = j + 1; i < n; ++i) {
            double l = A[static_cast<size_t>(i) * n + j];
            for (int k = 0; k < j; ++k) l -= A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(j) * n + k] * D[static_cast<size_t>(k)];
            A[static_cast<size_t>(i) * n + j] = d != 0.0 ? l / d : 0.0; // L below the diagonal
        }
    }
    for (int i = 0; i < n; ++i) // L y = b
        for (int k = 0; k < i; ++k) b[static_cast<size_t>(i)] -= A[static_cast<size_t>(i) * n + k] * b[static_cast<size_t>(k)];
    for (int i = 0; i < n; ++i) b[static_cast<size_t>(i)] = D[static_cast<size_t>(i)] != 0.0 ? b[static_cast<size_t>(i)] / D[static_cast<size_t>(i)] : 0.0;
    for (int i = n - 1; i >= 0; --i) // L^T x = y
        for (int k = i + 1; k < n; ++k) b[static_cast<size_t>(i)] -= A[static_cast<size_t>(k) * n + i] * b[static_cast<size_t>(k)];
}

} // namespace

void SimpleBA(const std::vector<Correspondence>& correspondences, geometry::SE3List& camera_poses, int max_iteration) {
    if (camera_poses.size() < 3) {
        std::cout << BLUE << "[INFO]::[SimpleBA]::Too few optimization variables, No need to optimize." << RESET << std::endl;
        return;
    }
    if (correspondences.size() < camera_poses.size() - 1) {
        std::cout << RED << "[ERROR]::[SimpleBA]::There are unconnected components." << RESET << std::endl;
        return;
    }
    const int n_poses = static_cast<int>(camera_poses.size()), n = 6 * (n_poses - 1); // the first pose stays where it is
    for (int iter = 0; iter != max_iteration; ++iter) {
        std::vector<double> A(static_cast<size_t>(n) * n, 0.0), g(static_cast<size_t>(n), 0.0);
        auto add = [&](int row_pose, int col_pose, const double* M) {
            for (int u = 0; u < 6; ++u)
                for (int v = 0; v < 6; ++v) A[static_cast<size_t>((row_pose - 1) * 6 + u) * n + (col_pose - 1) * 6 + v] += M[u * 6 + v];
        };
        for (size_t i = 0; i != correspondences.size(); ++i) {
            const int s = correspondences[i].source_id, t = correspondences[i].target_id;
            if (s < 0 || t < 0 || s >= n_poses || t >= n_poses) continue;
            const PairBlocks B = Accumulate(correspondences[i], camera_poses);
            if (s != 0) {
                add(s, s, B.ss);
                if (t != 0) { add(s, t, B.st); add(t, s, B.ts); }
                for (int u = 0; u < 6; ++u) g[static_cast<size_t>((s - 1) * 6 + u)] += B.rs[u];
            }
            if (t != 0) { // (the reference indexes block t - 1 unconditionally: its callers always pass source < target)
                add(t, t, B.tt);
                for (int u = 0; u < 6; ++u) g[static_cast<size_t>((t - 1) * 6 + u)] += B.rt[u];
            }
        }
        SolveLDLT(A, g, n);
        for (int i = 1; i < n_poses; ++i) {
            geometry::Se3 delta;
            for (int u = 0; u < 6; ++u) delta(u) = static_cast<float>(g[static_cast<size_t>((i - 1) * 6 + u)]);
            camera_poses[static_cast<size_t>(i)] = geometry::Se3ToSE3(delta) * camera_poses[static_cast<size_t>(i)];
        }
    }
}

} // namespace optimization
} // namespace one_piece
