// Optimization/Correspondence.h -- optimization::Correspondence (reference: src/Optimization/Correspondence.h:11-70): the point pairs that tie
// two poses together, with the public members example/DenseFusion fills (DenseSlam.cpp:88-113).
#pragma once
#include <cmath>

#include "Camera/Camera.h"
#include "Geometry/Geometry.h"

namespace one_piece {
namespace optimization {

class Correspondence {
  public:
    Correspondence() = default;
    Correspondence(int sid, int tid, const geometry::PointCorrespondenceSet& cs = geometry::PointCorrespondenceSet()) {
        source_id = sid;
        target_id = tid;
        correspondence_set = cs;
    }
    // mean pixel distance of the pairs' projections (Correspondence.h:23-40)
    void CalculateAverageDisparity(const camera::PinholeCamera& camera) {
        if (correspondence_set.size() <= 0) return;
        const geometry::Matrix3 K = camera.ToCameraMatrix();
        float sum_disparity = 0;
        for (size_t i = 0; i < correspondence_set.size(); ++i) {
            const geometry::Point3 a = K * correspondence_set[i].first, b = K * correspondence_set[i].second;
            const float du = a(0) / a(2) - b(0) / b(2), dv = a(1) / a(2) - b(1) / b(2);
            sum_disparity += std::sqrt(du * du + dv * dv);
        }
        average_disparity = sum_disparity / correspondence_set.size();
    }
    // root mean squared distance of the pairs under the two poses / under one source-to-target pose (Correspondence.h:41-66)
    float ComputeReprojectionError3D(const geometry::SE3List& camera_poses) const {
        float sum_error = 0.0;
        for (size_t j = 0; j != correspondence_set.size(); ++j)
            sum_error += (geometry::TransformPoint(camera_poses[source_id], correspondence_set[j].first) -
                          geometry::TransformPoint(camera_poses[target_id], correspondence_set[j].second)).squaredNorm();
        return std::sqrt(sum_error / correspondence_set.size());
    }
    float ComputeReprojectionError3D(const geometry::SE3& camera_pose) const {
        float sum_error = 0.0;
        for (size_t j = 0; j != correspondence_set.size(); ++j)
            sum_error += (geometry::TransformPoint(camera_pose, correspondence_set[j].first) - correspondence_set[j].second).squaredNorm();
        return std::sqrt(sum_error / correspondence_set.size());
    }
    int source_id = -1;
    int target_id = -1;
    float average_disparity = 1e6;
    geometry::PointCorrespondenceSet correspondence_set;
};

} // namespace optimization
} // namespace one_piece
