// Integrator.cpp -- integration::Integrator's public members over the C-ABI.
#include "Integration/Integrator.h"

#include <vector>

#include "Bridge.h"

namespace one_piece {
namespace integration {

float Integrator::GetSDF(const geometry::Point3& point, const camera::PinholeCamera& camera, const geometry::TransformationMatrix& pose, const cv::Mat& depth) {
    float p[16], sdf = 999;
    bridge::RowMajor(pose, p);
    const op_camera cam = camera.Pod();
    if (bridge::Failed(op_get_sdf(&cam, point.data(), p, nullptr, depth.data, bridge::DepthFormat(depth), &sdf), "Integrator::GetSDF")) return 999;
    return sdf;
}

// One cube through the GPU: a scratch device volume with the caller's camera / resolution / truncation receives the cube,
// the frame is fused into exactly that cube (no PrepareCubes selection, as in the reference's member), the cube comes back.
void Integrator::IntegrateImage(const cv::Mat& depth, const cv::Mat& rgb, const geometry::TransformationMatrix& pose, const camera::PinholeCamera& camera,
                                VoxelCube& voxel_cube, const CubePara& c_para) {
    const op_camera cam = camera.Pod();
    op_volume* vol = nullptr;
    if (bridge::Failed(op_volume_create(&cam, c_para.VoxelResolution, truncation, 5.0f, 0.5f, bridge::Device(), 64, &vol), "Integrator::IntegrateImage")) return;
    const int32_t key[3] = {voxel_cube.cube_id(0), voxel_cube.cube_id(1), voxel_cube.cube_id(2)};
    std::vector<float> aos(512 * 5);
    for (int v = 0; v < 512; ++v) {
        const TSDFVoxel& t = voxel_cube.voxels[v];
        aos[5 * v] = t.sdf; aos[5 * v + 1] = t.weight; aos[5 * v + 2] = t.color(0); aos[5 * v + 3] = t.color(1); aos[5 * v + 4] = t.color(2);
    }
    float p[16];
    bridge::RowMajor(pose, p);
    size_t n = 0;
    int32_t back[3];
    const bool ok = !bridge::Failed(op_volume_upload(vol, key, aos.data(), 1), "Integrator::IntegrateImage") &&
                    !bridge::Failed(op_volume_integrate_cubes(vol, depth.data, bridge::DepthFormat(depth), rgb.data, OP_MEM_HOST, p, nullptr, key, 1), "Integrator::IntegrateImage") &&
                    !bridge::Failed(op_volume_download(vol, back, aos.data(), 1, &n), "Integrator::IntegrateImage");
    if (ok && n == 1)
        for (int v = 0; v < 512; ++v) {
            TSDFVoxel& t = voxel_cube.voxels[v];
            t.sdf = aos[5 * v]; t.weight = aos[5 * v + 1];
            t.color = geometry::Point3(aos[5 * v + 2], aos[5 * v + 3], aos[5 * v + 4]);
        }
    op_volume_destroy(vol);
}

} // namespace integration
} // namespace one_piece
