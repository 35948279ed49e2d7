// Odometry.cpp -- odometry::Odometry::DenseTracking over op_tracker_dense_tracking.
#include "Odometry/Odometry.h"

#include "Bridge.h"

#include <cstdint>
#include <utility>

namespace one_piece {
namespace odometry {

Odometry::Odometry() {}
Odometry::Odometry(const camera::PinholeCamera& _camera) : camera(_camera) {}
Odometry::Odometry(const Odometry& other)
    : camera(other.camera), multi_scale_level(other.multi_scale_level), iter_count_per_level(other.iter_count_per_level) {}
Odometry& Odometry::operator=(const Odometry& other) {
    if (this != &other) {
        camera = other.camera;
        multi_scale_level = other.multi_scale_level;
        iter_count_per_level = other.iter_count_per_level;
    }
    return *this;
}
Odometry::~Odometry() {
    if (tracker_) op_tracker_destroy(tracker_);
}

std::shared_ptr<DenseTrackingResult> Odometry::DenseTracking(const cv::Mat& source_color, const cv::Mat& target_color, const cv::Mat& source_depth,
                                                             const cv::Mat& target_depth, const geometry::TransformationMatrix& initial_T,
                                                             int term_type) {
    DenseTrackingResult result;
    result.T = initial_T;
    result.tracking_success = false;
    const int w = static_cast<int>(camera.GetWidth()), h = static_cast<int>(camera.GetHeight());
    const cv::Mat* imgs[4] = {&source_color, &target_color, &source_depth, &target_depth};
    for (int k = 0; k < 4; ++k)
        if (imgs[k]->rows != h || imgs[k]->cols != w || !imgs[k]->data) {
            std::cout << RED << "[ERROR]::[DenseTracking]::image " << k << " is " << imgs[k]->cols << " x " << imgs[k]->rows
                      << ", the camera " << w << " x " << h << RESET << std::endl;
            return std::make_shared<DenseTrackingResult>(result);
        }
    if (bridge::DepthFormat(source_depth) != bridge::DepthFormat(target_depth) || static_cast<int>(iter_count_per_level.size()) != multi_scale_level) {
        std::cout << RED << "[ERROR]::[DenseTracking]::the two depth images differ in type, or iter_count_per_level does not have multi_scale_level entries"
                  << RESET << std::endl;
        return std::make_shared<DenseTrackingResult>(result);
    }
    if (!tracker_ && bridge::Failed(op_tracker_create(bridge::Device(), &tracker_), "DenseTracking")) return std::make_shared<DenseTrackingResult>(result);
    float T0[16];
    bridge::RowMajor(initial_T, T0);
    std::vector<int32_t> iters(iter_count_per_level.begin(), iter_count_per_level.end());
    const size_t cap = static_cast<size_t>(w) * static_cast<size_t>(h);
    std::vector<int32_t> pix(4 * cap);
    std::vector<float> pts(6 * cap);
    op_track_result r;
    const op_camera pod = camera.Pod();
    if (bridge::Failed(op_tracker_dense_tracking(tracker_, &pod, multi_scale_level, iters.data(), source_color.data, target_color.data, source_depth.data,
                                                 target_depth.data, bridge::DepthFormat(source_depth), T0, term_type, OP_MEM_HOST, &r, pix.data(),
                                                 pts.data(), cap),
                       "DenseTracking"))
        return std::make_shared<DenseTrackingResult>(result);
    result.T = bridge::FromRowMajor(r.T);
    result.rmse = r.rmse;
    result.tracking_success = r.tracking_success != 0;
    const size_t n = static_cast<size_t>(r.n_correspondences) < cap ? static_cast<size_t>(r.n_correspondences) : cap;
    result.pixel_correspondence_set.reserve(n);
    result.correspondence_set.reserve(n);
    for (size_t k = 0; k < n; ++k) {
        result.pixel_correspondence_set.push_back(std::make_pair(geometry::Point2ui(static_cast<unsigned>(pix[4 * k]), static_cast<unsigned>(pix[4 * k + 1])),
                                                                 geometry::Point2ui(static_cast<unsigned>(pix[4 * k + 2]), static_cast<unsigned>(pix[4 * k + 3]))));
        result.correspondence_set.push_back(std::make_pair(geometry::Point3(pts[6 * k], pts[6 * k + 1], pts[6 * k + 2]),
                                                           geometry::Point3(pts[6 * k + 3], pts[6 * k + 4], pts[6 * k + 5])));
    }
    return std::make_shared<DenseTrackingResult>(std::move(result));
}

std::shared_ptr<DenseTrackingResult> Odometry::DenseTracking(geometry::RGBDFrame& source_frame, geometry::RGBDFrame& target_frame,
                                                             const geometry::TransformationMatrix& initial_T, int term_type) {
    return DenseTracking(source_frame.rgb, target_frame.rgb, source_frame.depth, target_frame.depth, initial_T, term_type);
}

} // namespace odometry
} // namespace one_piece
