// Integration/Integrator.h -- integration::Integrator (reference: src/Integration/Integrator.h:10-27, Integrator.cpp:8-94).
// Callers read and set `truncation` directly.  The per-voxel update is the k_integrate kernel: CubeHandler::IntegrateImage
// runs it for every selected cube of a frame; the reference's public per-cube member IntegrateImage(depth, rgb, pose,
// camera, voxel_cube, c_para) is kept for host code that drives single cubes itself -- it round-trips that one cube
// through the GPU (op_volume_integrate_cubes), so use CubeHandler for throughput.  GetSDF is the scalar probe of
// Integrator.cpp:8-35 (op_get_sdf, host arithmetic in the reference's order; the kernels evaluate it per candidate corner).
#pragma once
#include "Camera/Camera.h"
#include "Geometry/Geometry.h"
#include "Integration/VoxelCube.h"

namespace one_piece {
namespace integration {

class Integrator {
  public:
    Integrator() = default;
    void IntegrateImage(const cv::Mat& depth, const cv::Mat& rgb, const geometry::TransformationMatrix& pose, const camera::PinholeCamera& camera,
                        VoxelCube& voxel_cube, const CubePara& c_para);
    float GetSDF(const geometry::Point3& point, const camera::PinholeCamera& camera, const geometry::TransformationMatrix& pose, const cv::Mat& depth);
    void SetTruncation(float _trunc) { truncation = _trunc; }
    float truncation = 0.1;
    // constant weight kept for source compatibility; the update uses the literal 1.0 (Integrator.cpp:77)
    float Weight = 1.0 / (2 * truncation);
};

} // namespace integration
} // namespace one_piece
