// Registration/ICP.h -- registration::PointToPoint / PointToPlane / EstimateRigidTransformationPointToPlane with the
// reference's signatures and defaults (src/Registration/ICP.h:13-26), computed on an MI355X through op_icp_*.
#pragma once
#include <memory>

#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"
#include "Registration/RegistrationResult.h"

namespace one_piece {
namespace registration {

class ICPParameter {
  public:
    int max_iteration = 30;  // iterations of the loop (there is no convergence test)
    double threshold = 0.2;  // maximum correspondence distance
    double scaling = 1.0;
};

geometry::TransformationMatrix EstimateRigidTransformationPointToPlane(const geometry::Point3List& source, const geometry::Point3List& target,
                                                                       const geometry::Point3List& target_normal, const geometry::FMatchSet& inliers);
std::shared_ptr<RegistrationResult> PointToPoint(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                                 const geometry::TransformationMatrix& init_T = geometry::TransformationMatrix::Identity(),
                                                 const ICPParameter& icp_para = ICPParameter());
std::shared_ptr<RegistrationResult> PointToPlane(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                                 const geometry::TransformationMatrix& init_T = geometry::TransformationMatrix::Identity(),
                                                 const ICPParameter& icp_para = ICPParameter());

} // namespace registration
} // namespace one_piece
