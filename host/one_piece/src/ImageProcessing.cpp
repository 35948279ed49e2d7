// ImageProcessing.cpp -- tool::ConvertDepthTo32F / tool::BilateralFilter.
#include "Tool/ImageProcessing.h"

#include <cstdlib>
#include <iostream>

#include "Bridge.h"

namespace one_piece {
namespace tool {

// uint16 / depth_scale in float (negatives -> 0), float32 copied, anything else is fatal as in the reference
// (ImageProcessing.cpp:68-91).  Pure per-pixel conversion on the host; the fused GPU form is BilateralFilter below, which
// accepts the raw 16-bit image directly.
void ConvertDepthTo32F(const cv::Mat& depth, cv::Mat& refined_depth, float depth_scale) {
    refined_depth.create(depth.rows, depth.cols, CV_32FC1);
    const size_t n = static_cast<size_t>(depth.rows) * depth.cols;
    float* out = reinterpret_cast<float*>(refined_depth.data);
    if (depth.depth() == CV_32F) {
        const float* in = reinterpret_cast<const float*>(depth.data);
        for (size_t i = 0; i < n; ++i) out[i] = in[i];
    } else if (depth.depth() == CV_16U) {
        const unsigned short* in = reinterpret_cast<const unsigned short*>(depth.data);
        for (size_t i = 0; i < n; ++i) {
            const float z = in[i] / depth_scale;
            out[i] = z < 0 ? 0 : z;
        }
    } else {
        std::cout << RED << "[ImageProcessing]::[ERROR]::Unknown depth image type: " << depth.depth() << RESET << std::endl;
        std::exit(1);
    }
}

void BilateralFilter(const cv::Mat& source, cv::Mat& target, int range) { BilateralFilter(source, target, range, 1000.0f); }

void BilateralFilter(const cv::Mat& source, cv::Mat& target, int range, float depth_scale) {
    cv::Mat out(source.rows, source.cols, CV_32FC1); // source and target may be the same object
    if (bridge::Failed(op_bilateral_filter_depth(source.data, bridge::DepthFormat(source), depth_scale, source.cols, source.rows, 1, range, 0.03f, 4.5f,
                                                 OP_MEM_HOST, bridge::Device(), nullptr, reinterpret_cast<float*>(out.data)),
                       "BilateralFilter"))
        return;
    target = out;
}

} // namespace tool
} // namespace one_piece
