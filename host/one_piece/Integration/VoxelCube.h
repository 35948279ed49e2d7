// Integration/VoxelCube.h -- block geometry (CubePara) and the host-side 8x8x8 block (VoxelCube) of the voxel hash
// (reference: src/Integration/VoxelCube.h:17-197).  In-block voxel index = x + 8 y + 64 z everywhere (device planes,
// downloads, the .map stream).  The float expressions of CubePara follow the reference's operand order because block
// selection on the device is bit-exact against them.
#pragma once
#include <cmath>
#include <cstddef>
#include <iostream>
#include <vector>

#include "Integration/TSDFVoxel.h"
#include "Tool/ConsoleColor.h"

#define CUBE_SIZE 8

namespace one_piece {
namespace integration {

typedef geometry::Point3i CubeID;
typedef geometry::VoxelGridHasher CubeHasher;

class CubePara {
  public:
#ifdef ONEPIECE_HAVE_EIGEN
    typedef Eigen::Matrix<int, 3, 8> CornerTable;
#else
    typedef compat::Mat<int, 3, 8> CornerTable;
#endif
    geometry::Point3iList NeighborCubeIDOffset; // the 8 blocks a voxel's 2x2x2 neighbourhood can reach: bit k of i -> axis k
    CornerTable CornerXYZOffset;                // cube corners in marching-cubes order (bottom ring, then top ring)
    geometry::Point3List VoxelCentroidOffSet;   // centre of voxel (x,y,z) relative to the block origin, index x + 8y + 64z
    float VoxelResolution = 0.01;               // metres

    CubePara() { InitializeVoxelCube(); }
    void SetVoxelResolution(float resolution) {
        VoxelResolution = resolution;
        std::cout << "Set Voxel resolution: " << VoxelResolution << std::endl;
        InitializeVoxelCube();
    }
    void InitializeVoxelCube() {
        NeighborCubeIDOffset.resize(8);
        for (int i = 0; i < 8; ++i) NeighborCubeIDOffset[i] = geometry::Point3i(i & 1, (i >> 1) & 1, (i >> 2) & 1);
        static const int ring_x[4] = {0, 1, 1, 0}, ring_y[4] = {0, 0, 1, 1};
        for (int c = 0; c < 8; ++c) { CornerXYZOffset(0, c) = ring_x[c & 3]; CornerXYZOffset(1, c) = ring_y[c & 3]; CornerXYZOffset(2, c) = c >> 2; }
        VoxelCentroidOffSet.resize(CUBE_SIZE * CUBE_SIZE * CUBE_SIZE);
        const float half = VoxelResolution / 2;
        for (size_t vid = 0; vid < VoxelCentroidOffSet.size(); ++vid) {
            const size_t x = vid % CUBE_SIZE, y = (vid / CUBE_SIZE) % CUBE_SIZE, z = vid / (CUBE_SIZE * CUBE_SIZE);
            VoxelCentroidOffSet[vid] = geometry::Point3(x * VoxelResolution + half, y * VoxelResolution + half, z * VoxelResolution + half);
        }
    }
    // voxel grid coordinate -> block id: floor division by 8 evaluated in double (:63-67)
    CubeID GetCubeID(const geometry::Point3i& point) const {
        return CubeID(static_cast<int>(std::floor((point(0) + 0.0) / CUBE_SIZE)), static_cast<int>(std::floor((point(1) + 0.0) / CUBE_SIZE)),
                      static_cast<int>(std::floor((point(2) + 0.0) / CUBE_SIZE)));
    }
    // world point -> block id: float division by the resolution, floor, then the integer rule above (:68-74)
    CubeID GetCubeID(const geometry::Point3& point) const { return GetCubeID(ToVoxelGrid(point)); }
    // block origin = ((id * 8) * resolution) evaluated left to right in float, plus the centroid offset (:75-80)
    geometry::Point3 GetGlobalPoint(const CubeID& cube_id, int voxel_id) const {
        const geometry::Point3 start = geometry::Point3(cube_id(0), cube_id(1), cube_id(2)) * CUBE_SIZE * VoxelResolution;
        return start + VoxelCentroidOffSet[voxel_id];
    }
    int GetVoxelID(const geometry::Point3i& point) const {
        const CubeID block = GetCubeID(point);
        const int ox = point(0) - block(0) * CUBE_SIZE, oy = point(1) - block(1) * CUBE_SIZE, oz = point(2) - block(2) * CUBE_SIZE;
        return ox + oy * CUBE_SIZE + oz * CUBE_SIZE * CUBE_SIZE;
    }
    int GetVoxelID(const geometry::Point3& point) const { return GetVoxelID(ToVoxelGrid(point)); }

  private:
    geometry::Point3i ToVoxelGrid(const geometry::Point3& p) const {
        return geometry::Point3i(static_cast<int>(std::floor(p(0) / VoxelResolution)), static_cast<int>(std::floor(p(1) / VoxelResolution)),
                                 static_cast<int>(std::floor(p(2) / VoxelResolution)));
    }
};

class VoxelCube {
  public:
    std::vector<TSDFVoxel> voxels;
    CubeID cube_id;

    VoxelCube() : voxels(CUBE_SIZE * CUBE_SIZE * CUBE_SIZE) {}
    VoxelCube(const CubeID& id) : voxels(CUBE_SIZE * CUBE_SIZE * CUBE_SIZE), cube_id(id) {}

    void IntegrateWithOtherCube(const VoxelCube& other) {
        if (cube_id != other.cube_id) {
            std::cout << YELLOW << "[Integration]::[WARNING]::Integrate two cubes which do not have the same cube_id." << RESET << std::endl;
            return;
        }
        for (size_t i = 0; i < voxels.size(); ++i) voxels[i] += other.voxels[i];
    }
    float GetSDF(int voxel_id) const { return voxels[voxel_id].sdf; }
    TSDFVoxel GetVoxel(int voxel_id) const { return voxels[voxel_id]; }
    geometry::Point3 GetOrigin(const CubePara& c_para) const {
        const float cube_resolution = CUBE_SIZE * c_para.VoxelResolution;
        return geometry::Point3(cube_id(0) * cube_resolution, cube_id(1) * cube_resolution, cube_id(2) * cube_resolution);
    }
    // One block of the .map float stream (:128-193): id as three floats, then {index, sdf, weight, c0, c1, c2} for every
    // voxel with |sdf| < 1 and weight != 0, terminated by -2.  (Whole files go through op_volume_write_file /
    // op_volume_read_file; these two serve host code that assembles streams itself.)
    void WriteToBuffer(std::vector<float>& buffer) const {
        for (int k = 0; k < 3; ++k) buffer.push_back(static_cast<float>(cube_id(k)));
        for (size_t i = 0; i < voxels.size(); ++i) {
            const TSDFVoxel& t = voxels[i];
            if (!(std::fabs(t.sdf) < 1) || t.weight == 0) continue;
            const float rec[6] = {static_cast<float>(i), t.sdf, t.weight, t.color(0), t.color(1), t.color(2)};
            buffer.insert(buffer.end(), rec, rec + 6);
        }
        buffer.push_back(-2.0f);
    }
    void ReadFromBuffer(const std::vector<float>& buffer, size_t& ptr) {
        for (; buffer[ptr] != -2.0f; ptr += 6) {
            TSDFVoxel& t = voxels[static_cast<int>(buffer[ptr])];
            t.sdf = buffer[ptr + 1]; t.weight = buffer[ptr + 2];
            t.color = geometry::Point3(buffer[ptr + 3], buffer[ptr + 4], buffer[ptr + 5]);
        }
        ++ptr;
    }
    // One block of the older float stream CubeHandler::ReadFromFileFloat reads (:168-193): a size word, {index, sdf,
    // weight}* -2, then a count and that many {index, r, g, b (0..255), colour weight} records (colour = rgb / 255 / weight)
    void ReadFromBufferFloat(const std::vector<float>& buffer, size_t& ptr) {
        ++ptr; // size
        for (; buffer[ptr] != -2.0f; ptr += 3) {
            TSDFVoxel& t = voxels[static_cast<int>(buffer[ptr])];
            t.sdf = buffer[ptr + 1]; t.weight = buffer[ptr + 2];
        }
        ++ptr;
        const size_t count = static_cast<size_t>(buffer[ptr++]);
        for (size_t c = 0; c < count; ++c, ptr += 5) {
            TSDFVoxel& t = voxels[static_cast<int>(buffer[ptr])];
            // the reference divides a float by the double literal 255.0 and stores a float, then divides by the weight
            t.color = geometry::Point3(static_cast<float>(buffer[ptr + 1] / 255.0), static_cast<float>(buffer[ptr + 2] / 255.0), static_cast<float>(buffer[ptr + 3] / 255.0));
            t.color /= buffer[ptr + 4];
        }
    }
};

} // namespace integration
} // namespace one_piece
