// Geometry.cpp -- the free functions of Geometry/Geometry.h the hot path calls.
#include "Geometry/Geometry.h"

#include <cmath>

#include "Bridge.h"

namespace one_piece {
namespace geometry {

Matrix4 Se3ToSE3(const Vector6& input) {
    float x[6], T[16];
    for (int i = 0; i < 6; ++i) x[i] = input(i);
    op_se3_exp(x, T);
    return bridge::FromRowMajor(T);
}

// 4x4 * (p, 1) accumulated column by column, then divided by w (the reference's TransformPoints, Geometry.cpp:19-27)
Point3 TransformPoint(const Matrix4& T, const Point3& p) {
    float q[4];
    for (int r = 0; r < 4; ++r) q[r] = ((T(r, 0) * p(0) + T(r, 1) * p(1)) + T(r, 2) * p(2)) + T(r, 3) * 1.0f;
    return Point3(q[0] / q[3], q[1] / q[3], q[2] / q[3]);
}
void TransformPoints(const Matrix4& T, Point3List& points) {
    for (size_t i = 0; i < points.size(); ++i) points[i] = TransformPoint(T, points[i]);
}
void TransformNormals(const Matrix4& T, Point3List& normals) {
    for (size_t i = 0; i < normals.size(); ++i) {
        const Point3 n = normals[i];
        float q[3];
        for (int r = 0; r < 3; ++r) q[r] = ((T(r, 0) * n(0) + T(r, 1) * n(1)) + T(r, 2) * n(2)) + T(r, 3) * 0.0f;
        normals[i] = Point3(q[0], q[1], q[2]);
    }
}

Plane GetPlane(const Point3& p1, const Point3& p2, const Point3& p3) {
    // through the library's own host routine, so that Frustum planes built here and inside op_volume_integrate agree to the bit
    const float fwd[3] = {0, 0, 0};
    (void)fwd;
    const Point3 e1 = p2 - p1, e2 = p3 - p1;
    Point3 n(e1(1) * e2(2) - e1(2) * e2(1), e1(2) * e2(0) - e1(0) * e2(2), e1(0) * e2(1) - e1(1) * e2(0));
    const float len2 = n(0) * n(0) + (n(1) * n(1) + n(2) * n(2));
    if (len2 > 0.0f) { const float len = std::sqrt(len2); n(0) /= len; n(1) /= len; n(2) /= len; }
    const double d = -static_cast<double>(p1(0) * n(0) + (p1(1) * n(1) + p1(2) * n(2)));
    return Plane(n(0), n(1), n(2), static_cast<float>(d));
}

double ComputeReprojectionError3D(const PointCorrespondenceSet& correspondence_set, const SE3& camera_pose) {
    double sum_error = 0.0;
    for (size_t j = 0; j != correspondence_set.size(); ++j) {
        const Point3 d = TransformPoint(camera_pose, correspondence_set[j].first) - correspondence_set[j].second;
        sum_error += d(0) * d(0) + (d(1) * d(1) + d(2) * d(2));
    }
    return std::sqrt(sum_error / correspondence_set.size());
}

TransformationMatrix EstimateRigidTransformation(const PointCorrespondenceSet& correspondence_set) {
    std::vector<float> pairs(6 * correspondence_set.size());
    for (size_t i = 0; i < correspondence_set.size(); ++i)
        for (int k = 0; k < 3; ++k) {
            pairs[6 * i + k] = correspondence_set[i].first(k);
            pairs[6 * i + 3 + k] = correspondence_set[i].second(k);
        }
    float T[16] = {0};
    if (bridge::Failed(op_estimate_rigid_transformation(pairs.data(), correspondence_set.size(), OP_MEM_HOST, bridge::Device(), T),
                       "EstimateRigidTransformation"))
        return TransformationMatrix::Zero();
    return bridge::FromRowMajor(T);
}

} // namespace geometry
} // namespace one_piece
