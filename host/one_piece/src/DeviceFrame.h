// DeviceFrame.h -- device copies of an RGBDFrame's images (library-internal; see RGBDFrame::on_device).
#pragma once
#include <memory>

#include "Bridge.h"
#include "Geometry/RGBDFrame.h"

namespace one_piece {
namespace bridge {

struct DeviceImages {
    void* rgb = nullptr;
    void* depth = nullptr;
    int depth_fmt = 0, width = 0, height = 0, device = 0;
    ~DeviceImages() {
        if (rgb) op_device_release(rgb, device);
        if (depth) op_device_release(depth, device);
    }
};

// The frame's images on the device (uploaded at the first call, then shared by every copy of the frame), or null after a message when the
// images are not what the GPU path reads: 3 bytes of colour per pixel, depth as float32 or uint16, both continuous and of the same size.
inline std::shared_ptr<DeviceImages> OnDevice(const geometry::RGBDFrame& f, const char* where) {
    if (f.on_device) return std::static_pointer_cast<DeviceImages>(f.on_device);
    const bool depth_ok = f.depth.type() == CV_16UC1 || f.depth.type() == CV_32FC1;
    if (f.rgb.empty() || f.depth.empty() || f.rgb.type() != CV_8UC3 || !depth_ok || !f.rgb.isContinuous() || !f.depth.isContinuous() ||
        f.rgb.rows != f.depth.rows || f.rgb.cols != f.depth.cols) {
        std::cout << RED << "[ERROR]::[" << where << "]::the frame needs a continuous CV_8UC3 colour image and a CV_16UC1 / CV_32FC1 depth image of the same size" << RESET << std::endl;
        return std::shared_ptr<DeviceImages>();
    }
    std::shared_ptr<DeviceImages> d = std::make_shared<DeviceImages>();
    d->device = Device(); d->width = f.rgb.cols; d->height = f.rgb.rows; d->depth_fmt = DepthFormat(f.depth);
    const size_t npx = static_cast<size_t>(d->width) * d->height;
    if (Failed(op_device_upload(f.rgb.data, npx * 3, d->device, &d->rgb), where) ||
        Failed(op_device_upload(f.depth.data, npx * (d->depth_fmt == OP_DEPTH_U16 ? 2 : 4), d->device, &d->depth), where))
        return std::shared_ptr<DeviceImages>();
    f.on_device = d;
    return d;
}

} // namespace bridge
} // namespace one_piece
