// Geometry/RGBDFrame.h -- geometry::RGBDFrame as CubeHandler::IntegrateImage(const RGBDFrame&, pose) reads it
// (reference: src/Geometry/RGBDFrame.h:11-24: rgb, depth, frame_id).
#pragma once
#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"

namespace one_piece {
namespace geometry {

class RGBDFrame {
  public:
    RGBDFrame() = default;
    RGBDFrame(const cv::Mat& _rgb, const cv::Mat& _depth, int id = -1) : rgb(_rgb), depth(_depth), frame_id(id) {}
    cv::Mat rgb;
    cv::Mat depth;
    int frame_id = -1;
    cv::Mat depth32f; // refined depth (tool::ConvertDepthTo32F)
    bool tracking_success = false;
    void Release() { rgb.release(); depth.release(); depth32f.release(); }
};

} // namespace geometry
} // namespace one_piece
