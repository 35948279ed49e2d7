// Tool/ImageProcessing.h -- the depth front end of the fusion drivers (reference: src/Tool/ImageProcessing.cpp:64-91,
// header default range = 7): ConvertDepthTo32F and BilateralFilter, on the GPU (op_bilateral_filter_depth).
// PARITY NOTE: the reference's BilateralFilter is cv::bilateralFilter (OpenCV, not vendored); this follows its
// documented definition (include/onepiece_hip.h).
#pragma once
#include "Geometry/Geometry.h"

namespace one_piece {
namespace tool {

void ConvertDepthTo32F(const cv::Mat& depth, cv::Mat& refined_depth, float depth_scale);
void BilateralFilter(const cv::Mat& source, cv::Mat& target, int range = 7);
// extension: raw CV_16UC1 depth in, the division by depth_scale fused into the filter kernel (saves the host pass of
// ConvertDepthTo32F); for CV_32FC1 input depth_scale is ignored
void BilateralFilter(const cv::Mat& source, cv::Mat& target, int range, float depth_scale);

} // namespace tool
} // namespace one_piece
