// Geometry/RGBDFrame.h -- geometry::RGBDFrame as CubeHandler::IntegrateImage(const RGBDFrame&, pose) reads it
// (reference: src/Geometry/RGBDFrame.h:11-24: rgb, depth, frame_id).
#pragma once
#include <memory>

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
    void Release() { rgb.release(); depth.release(); depth32f.release(); on_device.reset(); }
    // Device copies of rgb and depth, made by the first GPU consumer that asks for them (odometry::Odometry::DenseTrackingEnqueue,
    // CubeHandler::IntegrateImage(const RGBDFrame&, ...) afterwards) and shared by the copies of this frame -- the counterpart of the prepared
    // images the reference caches inside its frames (src/Geometry/RGBDFrame.h:26-40).  A frame is tracked against twice and fused once:
    // its pixels cross PCIe once.  The images must not be modified once they are on the device (Release() drops the copies).
    mutable std::shared_ptr<void> on_device;
};

} // namespace geometry
} // namespace one_piece
