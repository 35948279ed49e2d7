// Geometry/PointCloud.h -- geometry::PointCloud with the reference's public surface (src/Geometry/PointCloud.h:12-62).
// On the hot path and on the GPU through the C-ABI: LoadFromDepth / LoadFromRGBD (PointCloud.cpp:17-100), EstimateNormals
// (:102-144).  Host-side conveniences of the drivers (no parity claim beyond the documented behaviour): LoadFromXYZ,
// LoadFromPLY / LoadFromOBJ / LoadFromFile, WriteToPLY / WriteToOBJ, DownSample (voxel-grid mean), MergePCD, Transform.
// Inside the reference tree (-DONEPIECE_IN_REFERENCE_TREE) this header steps aside for the
// reference's own declaration of the same class, so that there is exactly one geometry::PointCloud in the program.
#pragma once
#ifdef ONEPIECE_IN_REFERENCE_TREE
#include_next "Geometry/PointCloud.h"
#else
#include <memory>
#include <string>

#include "Camera/Camera.h"
#include "Geometry/Geometry.h"

namespace one_piece {
namespace geometry {

class RGBDFrame;
class PointCloud {
  public:
    PointCloud() = default;
    size_t GetSize() const { return points.size(); }
    bool HasColors() const { return colors.size() == points.size() && colors.size() > 0; }
    bool HasNormals() const { return normals.size() == points.size() && normals.size() > 0; }
    void LoadFromRGBD(const cv::Mat& rgb, const cv::Mat& depth, const camera::PinholeCamera& camera);
    void LoadFromRGBD(const RGBDFrame& rgbd, const camera::PinholeCamera& camera);
    void LoadFromDepth(const cv::Mat& depth, const camera::PinholeCamera& camera);
    // keeps the points with z > 0, row by row (PointCloud.cpp:190-204)
    void LoadFromXYZ(const ImageXYZ& xyz);
    bool LoadFromPLY(const std::string& filename);
    bool LoadFromOBJ(const std::string& filename);
    // by extension: .ply or .obj; anything else is a warning and false (PointCloud.cpp:217-233)
    bool LoadFromFile(const std::string& filename);
    bool WriteToOBJ(const std::string& filename);
    void EstimateNormals(float radius = 0.1, int knn = 30);
    void Transform(const TransformationMatrix& T);
    // one point per occupied grid cell of edge grid_len (floor(p / grid_len)): the mean of the cell's points, colours and
    // normals, cells in order of first appearance (PointCloud.cpp:145-189)
    std::shared_ptr<PointCloud> DownSample(float grid_len) const;
    bool WriteToPLY(const std::string& fileName) const;
    // appends another cloud; refuses (message) when colours / normals would no longer match the points (PointCloud.cpp:49-67)
    void MergePCD(const PointCloud& another_pcd);
    void Reset() { points.clear(); normals.clear(); colors.clear(); }

    geometry::Point3List points;
    geometry::Point3List normals;
    geometry::Point3List colors;
};

typedef std::shared_ptr<PointCloud> PointCloudPtr;

} // namespace geometry
} // namespace one_piece
#endif // ONEPIECE_IN_REFERENCE_TREE
