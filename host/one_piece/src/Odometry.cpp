// Odometry.cpp -- odometry::Odometry::DenseTracking over op_tracker_dense_tracking.
#include "Odometry/Odometry.h"

#include "Bridge.h"
#include "DeviceFrame.h"

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
    while (!inflight_.empty()) { Finish(inflight_.front()); inflight_.pop_front(); } // the frames' device images must outlive the kernels that read them
    for (size_t i = 0; i < pipe_.size(); ++i)
        if (pipe_[i]) op_tracker_destroy(pipe_[i]);
    if (tracker_) op_tracker_destroy(tracker_);
}

void Odometry::SetPipelineDepth(int pairs_in_flight) {
    while (!inflight_.empty()) { ready_.push_back(Finish(inflight_.front())); inflight_.pop_front(); } // the slot rotation restarts: nothing may be in flight
    pipe_depth_ = pairs_in_flight < 1 ? 1 : (pairs_in_flight > 16 ? 16 : pairs_in_flight);
    enqueued_ = 0;
}

std::shared_ptr<DenseTrackingResult> Odometry::Finish(const InFlight& job) {
    std::shared_ptr<DenseTrackingResult> result = std::make_shared<DenseTrackingResult>();
    result->T = geometry::TransformationMatrix::Identity();
    result->tracking_success = false;
    op_track_result r;
    if (bridge::Failed(op_tracker_wait(pipe_[static_cast<size_t>(job.slot)], &r, nullptr, nullptr, 0), "DenseTrackingWait")) return result;
    result->T = bridge::FromRowMajor(r.T);
    result->rmse = r.rmse;
    result->tracking_success = r.tracking_success != 0;
    return result;
}

bool Odometry::DenseTrackingEnqueue(geometry::RGBDFrame& source_frame, geometry::RGBDFrame& target_frame, const geometry::TransformationMatrix& initial_T,
                                    int term_type) {
    const int w = static_cast<int>(camera.GetWidth()), h = static_cast<int>(camera.GetHeight());
    std::shared_ptr<bridge::DeviceImages> src = bridge::OnDevice(source_frame, "DenseTrackingEnqueue"), tgt = bridge::OnDevice(target_frame, "DenseTrackingEnqueue");
    if (!src || !tgt) return false;
    if (src->width != w || src->height != h || tgt->width != w || tgt->height != h || src->depth_fmt != tgt->depth_fmt ||
        static_cast<int>(iter_count_per_level.size()) != multi_scale_level) {
        std::cout << RED << "[ERROR]::[DenseTrackingEnqueue]::the frames do not match the camera or each other, or iter_count_per_level does not have multi_scale_level entries"
                  << RESET << std::endl;
        return false;
    }
    while (static_cast<int>(inflight_.size()) >= pipe_depth_) { // free the oldest pair's tracker; its result waits in ready_
        ready_.push_back(Finish(inflight_.front()));
        inflight_.pop_front();
    }
    if (static_cast<int>(pipe_.size()) < pipe_depth_) pipe_.resize(static_cast<size_t>(pipe_depth_), nullptr);
    const int slot = static_cast<int>(enqueued_ % static_cast<unsigned long long>(pipe_depth_));
    if (!pipe_[static_cast<size_t>(slot)] && bridge::Failed(op_tracker_create(bridge::Device(), &pipe_[static_cast<size_t>(slot)]), "DenseTrackingEnqueue")) return false;
    float T0[16];
    bridge::RowMajor(initial_T, T0);
    std::vector<int32_t> iters(iter_count_per_level.begin(), iter_count_per_level.end());
    const op_camera pod = camera.Pod();
    if (bridge::Failed(op_tracker_dense_tracking_enqueue(pipe_[static_cast<size_t>(slot)], &pod, multi_scale_level, iters.data(), static_cast<const uint8_t*>(src->rgb),
                                                         static_cast<const uint8_t*>(tgt->rgb), src->depth, tgt->depth, src->depth_fmt, T0, term_type, OP_MEM_DEVICE, 0),
                       "DenseTrackingEnqueue"))
        return false;
    ++enqueued_;
    InFlight job;
    job.slot = slot; job.source = src; job.target = tgt;
    inflight_.push_back(job);
    return true;
}

std::shared_ptr<DenseTrackingResult> Odometry::DenseTrackingWait() {
    if (!ready_.empty()) {
        std::shared_ptr<DenseTrackingResult> r = ready_.front();
        ready_.pop_front();
        return r;
    }
    if (inflight_.empty()) {
        std::cout << RED << "[ERROR]::[DenseTrackingWait]::nothing was enqueued" << RESET << std::endl;
        std::shared_ptr<DenseTrackingResult> none = std::make_shared<DenseTrackingResult>();
        none->T = geometry::TransformationMatrix::Identity();
        none->tracking_success = false;
        return none;
    }
    std::shared_ptr<DenseTrackingResult> r = Finish(inflight_.front());
    inflight_.pop_front();
    return r;
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
    // what the GPU path reads: 3 bytes of colour per pixel and float32 / uint16 depth, rows back to back -- anything else (a grey or float colour
    // image, a ROI view) would be read out of bounds
    const bool colour_ok = source_color.type() == CV_8UC3 && target_color.type() == CV_8UC3;
    const bool depth_ok = (source_depth.type() == CV_16UC1 || source_depth.type() == CV_32FC1) && (target_depth.type() == CV_16UC1 || target_depth.type() == CV_32FC1);
    if (!colour_ok || !depth_ok || !source_color.isContinuous() || !target_color.isContinuous() || !source_depth.isContinuous() || !target_depth.isContinuous()) {
        std::cout << RED << "[ERROR]::[DenseTracking]::colour images must be continuous CV_8UC3, depth images continuous CV_16UC1 or CV_32FC1" << RESET << std::endl;
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
