// Frustum.cpp -- integration::Frustum over the library's host-side frustum routine.
#include "Integration/Frustum.h"

#include <cmath>

#include "Bridge.h"

namespace one_piece {
namespace integration {

void Frustum::Adopt(const float planes[24], const float c[24]) {
    geometry::Plane* dst[6] = {&top_plane, &left_plane, &right_plane, &bottom_plane, &near_plane, &far_plane};
    for (int k = 0; k < 6; ++k) *dst[k] = geometry::Plane(planes[4 * k], planes[4 * k + 1], planes[4 * k + 2], planes[4 * k + 3]);
    for (int k = 0; k < 8; ++k) corners[k] = geometry::Point3(c[3 * k], c[3 * k + 1], c[3 * k + 2]);
    // edges as pairs of corner indices: far face, near face, connecting edges (Frustum.cpp:58-93)
    static const int edge[12][2] = {{0, 1}, {3, 2}, {1, 3}, {2, 0}, {4, 7}, {6, 5}, {5, 7}, {6, 4}, {0, 5}, {1, 6}, {2, 7}, {3, 4}};
    for (int k = 0; k < 12; ++k) { lines[k].first = corners[edge[k][0]]; lines[k].second = corners[edge[k][1]]; }
}

void Frustum::ComputeFromCamera(const camera::PinholeCamera& camera, const geometry::TransformationMatrix& T, float far_dist, float near_dist) {
    float pose[16], planes[24], c[24];
    bridge::RowMajor(T, pose);
    const op_camera cam = camera.Pod();
    if (bridge::Failed(op_frustum_from_camera(&cam, pose, far_dist, near_dist, planes, c), "Frustum::ComputeFromCamera")) return;
    Adopt(planes, c);
}

void Frustum::ComputeFromVectors(const geometry::Point3& forward, const geometry::Point3& position, const geometry::Point3& right,
                                 const geometry::Point3& up, float far_dist, float near_dist, float fov, float aspect) {
    float planes[24], c[24];
    if (bridge::Failed(op_frustum_from_vectors(forward.data(), position.data(), right.data(), up.data(), far_dist, near_dist, fov, aspect, planes, c),
                       "Frustum::ComputeFromVectors"))
        return;
    Adopt(planes, c);
}

bool Frustum::ContainPoint(const geometry::Point3& p) {
    const geometry::Plane* order[6] = {&top_plane, &left_plane, &right_plane, &bottom_plane, &near_plane, &far_plane};
    for (int k = 0; k < 6; ++k) {
        const geometry::Plane& q = *order[k];
        const float distance = (q(0) * p(0) + (q(1) * p(1) + q(2) * p(2))) + q(3);
        if (distance < 0) return false;
        if (distance == 0) return true;
    }
    return true;
}

std::shared_ptr<geometry::PointCloud> Frustum::GetPointCloud() const {
    std::shared_ptr<geometry::PointCloud> pcd = std::make_shared<geometry::PointCloud>();
    const int point_num = 1000;
    pcd->points.reserve(12 * point_num);
    pcd->colors.reserve(12 * point_num);
    for (int i = 0; i != 12; ++i) {
        geometry::Point3 diff = lines[i].first - lines[i].second;
        diff.normalize();
        // parameter step along `diff` that reaches lines[i].second after point_num steps (first component that is not ~0)
        float step = 0;
        for (int j = 0; j != 3; ++j)
            if (std::fabs(diff(j)) >= 0.000001) { step = (lines[i].second(j) - lines[i].first(j)) / diff(j); break; }
        step /= point_num;
        float t = 0;
        for (int j = 0; j != point_num; ++j) {
            pcd->points.push_back(geometry::Point3(lines[i].first(0) + diff(0) * t, lines[i].first(1) + diff(1) * t, lines[i].first(2) + diff(2) * t));
            pcd->colors.push_back(diff);
            t += step;
        }
    }
    return pcd;
}

} // namespace integration
} // namespace one_piece
