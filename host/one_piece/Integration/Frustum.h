// Integration/Frustum.h -- one_piece::integration::Frustum (reference: src/Integration/Frustum.h:10-105, Frustum.cpp:7-94):
// the view frustum CubeHandler::ComputeBounding clips the back-projected depth points with, and that the fusion
// drivers build directly (example/ImageIntegration.cpp:30-35).  Same members, same public data: six planes
// (normal, d), corners[8] and lines[12] in the reference's order.  The arithmetic is the library's host routine
// (op_frustum_from_camera / op_frustum_from_vectors, include/onepiece_hip.h) -- the very planes op_volume_integrate
// hands to its kernels, in the reference's float order (fov through double atan2 / tan).
#pragma once
#include <memory>
#include <utility>

#include "Camera/Camera.h"
#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"

namespace one_piece {
namespace integration {

class Frustum {
  public:
    void ComputeFromCamera(const camera::PinholeCamera& camera, const geometry::TransformationMatrix& T, float far_dist, float near_dist);
    void ComputeFromVectors(const geometry::Point3& forward, const geometry::Point3& position, const geometry::Point3& right,
                            const geometry::Point3& up, float far_dist, float near_dist, float fov, float aspect);
    geometry::Plane GetFarPlane() const { return far_plane; }
    geometry::Plane GetNearPlane() const { return near_plane; }
    geometry::Plane GetTopPlane() const { return top_plane; }
    geometry::Plane GetBottomPlane() const { return bottom_plane; }
    geometry::Plane GetLeftPlane() const { return left_plane; }
    geometry::Plane GetRightPlane() const { return right_plane; }
    // 1000 points along each of the 12 edges, coloured with the edge direction (Frustum.h:41-72)
    std::shared_ptr<geometry::PointCloud> GetPointCloud() const;
    // top, left, right, bottom, near, far in this order; a point exactly ON a plane counts as inside at once (:74-103)
    bool ContainPoint(const geometry::Point3& p);

    geometry::Point3 corners[8];
    std::pair<geometry::Point3, geometry::Point3> lines[12];
    geometry::Plane top_plane;
    geometry::Plane left_plane;
    geometry::Plane right_plane;
    geometry::Plane bottom_plane;
    geometry::Plane near_plane;
    geometry::Plane far_plane;

  private:
    void Adopt(const float planes[24], const float corner_xyz[24]);
};

} // namespace integration
} // namespace one_piece
