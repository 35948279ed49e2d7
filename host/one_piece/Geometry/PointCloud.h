// Geometry/PointCloud.h -- geometry::PointCloud, the part the ICP / fusion path and its drivers use
// (reference: src/Geometry/PointCloud.h:12-62; LoadFromDepth / LoadFromRGBD PointCloud.cpp:17-100, EstimateNormals
// :102-144, Transform, WriteToPLY).  The loaders and EstimateNormals run on the GPU through the C-ABI.
#pragma once
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
    void EstimateNormals(float radius = 0.1, int knn = 30);
    void Transform(const TransformationMatrix& T);
    bool WriteToPLY(const std::string& fileName) const;
    void Reset() { points.clear(); normals.clear(); colors.clear(); }

    geometry::Point3List points;
    geometry::Point3List normals;
    geometry::Point3List colors;
};

typedef std::shared_ptr<PointCloud> PointCloudPtr;

} // namespace geometry
} // namespace one_piece
