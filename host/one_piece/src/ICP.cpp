// ICP.cpp -- registration::PointToPoint / PointToPlane / EstimateRigidTransformationPointToPlane over op_icp_*.
#include "Registration/ICP.h"

#include "Bridge.h"

#include <utility>

namespace one_piece {
namespace registration {

namespace {
std::shared_ptr<RegistrationResult> Run(int mode, const geometry::PointCloud& _source, const geometry::PointCloud& _target,
                                        const geometry::TransformationMatrix& init_T, const ICPParameter& icp_para) {
    RegistrationResult result;
    // ICP.cpp:35-43 copies both clouds and scales the copies; the copies are only made here when there is something to scale
    geometry::Point3List scaled_source, scaled_target;
    if (icp_para.scaling != 1) {
        scaled_source = _source.points; scaled_target = _target.points;
        for (size_t i = 0; i < scaled_source.size(); ++i) scaled_source[i] = scaled_source[i] * static_cast<float>(icp_para.scaling);
        for (size_t i = 0; i < scaled_target.size(); ++i) scaled_target[i] = scaled_target[i] * static_cast<float>(icp_para.scaling);
    }
    const geometry::Point3List& source = icp_para.scaling != 1 ? scaled_source : _source.points;
    const geometry::Point3List& target = icp_para.scaling != 1 ? scaled_target : _target.points;
    float T0[16];
    bridge::RowMajor(init_T, T0);
    op_icp_result r;
    std::vector<int32_t> pairs(2 * source.size() + 2);
    const float* nrm = mode == OP_ICP_POINT_TO_PLANE ? bridge::Floats(_target.normals) : nullptr;
    if (bridge::Failed(op_icp_register(mode, bridge::Floats(source), source.size(), bridge::Floats(target), nrm, target.size(), T0, icp_para.max_iteration,
                                       icp_para.threshold, bridge::Device(), &r, pairs.data(), source.size()),
                       mode == OP_ICP_POINT_TO_PLANE ? "ICPPointToPlane" : "ICPPointToPoint"))
        return std::make_shared<RegistrationResult>(result);
    result.T = bridge::FromRowMajor(r.T);
    if (icp_para.scaling != 1) // ICP.cpp:207-221: the clouds are un-scaled before the final Kabsch -- same R, translation / scaling
        for (int k = 0; k < 3; ++k) result.T(k, 3) = result.T(k, 3) / static_cast<float>(icp_para.scaling);
    result.rmse = r.rmse;
    result.correspondence_set_index.reserve(r.n_inliers);
    result.correspondence_set.reserve(r.n_inliers);
    for (uint64_t k = 0; k < r.n_inliers; ++k) {
        const int s = pairs[2 * k], t = pairs[2 * k + 1];
        result.correspondence_set_index.push_back(std::make_pair(s, t));
        result.correspondence_set.push_back(std::make_pair(_source.points[s], _target.points[t]));
    }
    return std::make_shared<RegistrationResult>(std::move(result));
}
} // namespace

std::shared_ptr<RegistrationResult> PointToPoint(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                                 const geometry::TransformationMatrix& init_T, const ICPParameter& icp_para) {
    return Run(OP_ICP_POINT_TO_POINT, source, target, init_T, icp_para);
}

std::shared_ptr<RegistrationResult> PointToPlane(const geometry::PointCloud& source, const geometry::PointCloud& target,
                                                 const geometry::TransformationMatrix& init_T, const ICPParameter& icp_para) {
    if (!target.HasNormals() || icp_para.scaling != 1) { // ICP.cpp:159-163
        std::cout << RED << "[ERROR]::[ICPPointToPlane]::target point cloud need to have normals." << RESET << std::endl;
        return std::make_shared<RegistrationResult>(RegistrationResult());
    }
    return Run(OP_ICP_POINT_TO_PLANE, source, target, init_T, icp_para);
}

geometry::TransformationMatrix EstimateRigidTransformationPointToPlane(const geometry::Point3List& source, const geometry::Point3List& target,
                                                                       const geometry::Point3List& target_normal, const geometry::FMatchSet& inliers) {
    std::vector<int32_t> ids(2 * inliers.size());
    for (size_t i = 0; i < inliers.size(); ++i) { ids[2 * i] = inliers[i].first; ids[2 * i + 1] = inliers[i].second; }
    float T[16];
    if (bridge::Failed(op_estimate_rigid_point_to_plane(bridge::Floats(source), source.size(), bridge::Floats(target), bridge::Floats(target_normal),
                                                        target.size(), ids.data(), inliers.size(), OP_MEM_HOST, bridge::Device(), T),
                       "EstimateRigidTransformationPointToPlane"))
        return geometry::TransformationMatrix::Identity();
    return bridge::FromRowMajor(T);
}

} // namespace registration
} // namespace one_piece
