// Compile-only check of host/onepiece_hip_shim.hpp against REAL Eigen types (the reference's
// geometry typedefs, Geometry/Geometry.h:34-75), available only in the build container.
#include <Eigen/Core>
#include <Eigen/StdVector>
#include <unordered_map>
#include <vector>
#include "onepiece_hip_shim.hpp"
typedef Eigen::Matrix4f TransformationMatrix;
typedef Eigen::Vector3f Point3;
typedef Eigen::Vector3i CubeID;
typedef std::vector<Point3, Eigen::aligned_allocator<Point3>> Point3List;
struct Hasher { size_t operator()(const CubeID& k) const { return (size_t)op_hash_key(k(0), k(1), k(2)); } };
struct TSDFVoxel { float sdf = 999, weight = 0; Point3 color = Point3(-1, -1, -1); TSDFVoxel() = default; TSDFVoxel(float s, float w, const Point3& c) : sdf(s), weight(w), color(c) {} };
struct VoxelCube { std::vector<TSDFVoxel> voxels; CubeID cube_id; VoxelCube() : voxels(512) {} VoxelCube(const CubeID& id) : voxels(512), cube_id(id) {} };
typedef std::unordered_map<CubeID, VoxelCube, Hasher> CubeMap;
struct Mat { unsigned char* data; int flags; int depth() const { return flags & 7; } };
struct RegistrationResult { TransformationMatrix T; std::vector<std::pair<int, int>> correspondence_set_index; std::vector<std::pair<Point3, Point3>> correspondence_set; double rmse; };
int main() {
    namespace sh = one_piece::hip_shim;
    op_volume* vol = nullptr; Mat d{nullptr, 5}, c{nullptr, 16};
    TransformationMatrix pose = TransformationMatrix::Identity();
    std::vector<CubeID> list; CubeMap map; RegistrationResult res; Point3List a, b, n;
    if (vol) {
        sh::IntegrateImage(vol, d, c, pose, TransformationMatrix(pose.inverse()));
        sh::PrepareCubes(vol, d, pose, TransformationMatrix(pose.inverse()), list);
        sh::DownloadInto<CubeMap, CubeID, VoxelCube, TSDFVoxel, Point3>(vol, map);
        sh::UploadFrom(vol, map);
    }
    sh::RunICP(OP_ICP_POINT_TO_PLANE, a, b, &n, pose, 0, 0.01, 0, res);
    return 0;
}
