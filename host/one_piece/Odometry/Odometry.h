// Odometry.h -- the DENSE half of one_piece::odometry::Odometry (reference src/Odometry/Odometry.h:19-176) over the
// C-ABI's dense RGB-D tracker (op_tracker_*).  Same class name, namespace, result types, member signatures and defaults
// for DenseTracking (both overloads), the camera / pyramid setters and CreatePyramidCameras, so that
// example/DenseOdometry.cpp and DenseSlam-style callers compile unedited.
//
// Not declared: the sparse half (Find2DMathes, SparseTracking*, ComputeTransformation, GetLocalPointsFromKeyPoints,
// GetCorrespondencesFromMatches, SetFeatureNumber / GetFeatureNumber -- ORB features, brute-force / MILD matching and RANSAC
// on OpenCV, out of scope for this path) and the pyramid helpers whose images live on the GPU here
// (CreateImagePyramid, CreateImageXYZPyramid, MultiScaleComputing, InitializeRGBDDenseTracking: the C-ABI exposes them
// as op_tracker_track / op_tracker_read_pyramid).
#pragma once
#include <deque>
#include <iostream>
#include <memory>
#include <vector>

#include "Camera/Camera.h"
#include "Geometry/Geometry.h"
#include "Geometry/RGBDFrame.h"
#include "OdometryPredefined.h"
#include "Tool/ImageProcessing.h" // the reference's Odometry.h brings it in (Odometry.h:10), and example/DenseFusion/DenseFusion.cpp:92-93 relies on that

struct op_tracker; // include/onepiece_hip.h

namespace one_piece {
namespace odometry {

class SparseTrackingResult { // Odometry.h:21-29 (the type only: DenseSlam-style code stores it)
  public:
    geometry::TransformationMatrix T;
    geometry::FMatchSet correspondence_set_index;
    geometry::PointCorrespondenceSet correspondence_set;
    double rmse = 1e6;
    bool tracking_success;
};
class DenseTrackingResult { // Odometry.h:30-38
  public:
    geometry::TransformationMatrix T;
    geometry::PixelCorrespondenceSet pixel_correspondence_set; // ((v_s, u_s), (v_t, u_t)) in raster order of the source pixel
    geometry::PointCorrespondenceSet correspondence_set;       // (source xyz, target xyz), both read at the source pixel (Odometry.cpp:676-683)
    double rmse = 1e6;
    bool tracking_success;
};

class Odometry {
  public:
    Odometry();
    Odometry(const camera::PinholeCamera& _camera);
    Odometry(const Odometry& other);            // value semantics like the reference's: the copy gets its own device workspace
    Odometry& operator=(const Odometry& other);
    ~Odometry();

    // Odometry.cpp:463-524: intensity + NaN-depth conversion, Gaussian filtering, correspondences at the identity and
    // NormalizeIntensity, pyramids, MultiScaleComputing (coarse to fine, iter_count_per_level), result assembly -- one
    // op_tracker_dense_tracking call.  term_type 0 hybrid, 1 photometric, 2 geometric.
    std::shared_ptr<DenseTrackingResult> DenseTracking(const cv::Mat& source_color, const cv::Mat& target_color, const cv::Mat& source_depth,
                                                       const cv::Mat& target_depth, const geometry::TransformationMatrix& initial_T,
                                                       int term_type = 0);
    // Odometry.cpp:526-607.  The reference caches the prepared images inside the frames (and NormalizeIntensity then rescales
    // the cached intensity again on every call that reuses a frame); here every call prepares both frames afresh, like the
    // cv::Mat overload.
    std::shared_ptr<DenseTrackingResult> DenseTracking(geometry::RGBDFrame& source_frame, geometry::RGBDFrame& target_frame,
                                                       const geometry::TransformationMatrix& initial_T, int term_type = 0);

    // ---- beyond the reference's surface: several frame pairs in flight ---------------------------------------------------------------
    // One 640x480 track is 28 strictly sequential small problems that leave most of the GPU idle; four pairs in flight on four streams run at
    // ~3x the rate of one pair at a time.  DenseTrackingEnqueue uploads the two frames unless they are on the device
    // already (RGBDFrame::on_device: a frame is source once and target once, its pixels cross PCIe once), enqueues the whole call on a
    // tracker of its own and returns; DenseTrackingWait returns the OLDEST outstanding result (T, rmse, tracking_success; the correspondence
    // sets stay empty -- use DenseTracking when they are needed).  At most `SetPipelineDepth` (default 4) pairs are outstanding: enqueueing
    // one more waits for the oldest, whose result is then kept for the next DenseTrackingWait.  A caller that chains poses assumes
    // success when it enqueues pair (i, i+1) before pair (i-1, i) is known (examples/cpp/DenseFusion.cpp shows the resolution in order).
    void SetPipelineDepth(int pairs_in_flight);
    bool DenseTrackingEnqueue(geometry::RGBDFrame& source_frame, geometry::RGBDFrame& target_frame, const geometry::TransformationMatrix& initial_T,
                              int term_type = 0);
    std::shared_ptr<DenseTrackingResult> DenseTrackingWait();
    size_t DenseTrackingPending() const { return inflight_.size() + ready_.size(); }

    void SetCamera(const camera::PinholeCamera& _camera) { camera = _camera; }
    void SetCameraPara(float _fx, float _fy, float _cx, float _cy, int _width, int _height, float depthScale, float* _distortion = nullptr) {
        camera.SetPara(_fx, _fy, _cx, _cy, _width, _height, depthScale, _distortion);
    }
    void SetMultiScale(int layer_count) {
        multi_scale_level = layer_count;
        iter_count_per_level.resize(layer_count, 4);
    }
    std::vector<camera::PinholeCamera> CreatePyramidCameras() {
        std::vector<camera::PinholeCamera> result;
        for (int i = 0; i != multi_scale_level; ++i) {
            if (i == 0) result.push_back(camera);
            else result.push_back(result[i - 1].GenerateNextPyramid());
        }
        return result;
    }

  protected:
    camera::PinholeCamera camera;
    // for dense tracking
    int multi_scale_level = 3;
    std::vector<int> iter_count_per_level = {4, 8, 16};

  private:
    op_tracker* tracker_ = nullptr; // created at the first DenseTracking call
    struct InFlight { int slot; std::shared_ptr<void> source, target; };
    std::shared_ptr<DenseTrackingResult> Finish(const InFlight& job);
    std::vector<op_tracker*> pipe_;  // one tracker (one HIP stream) per pair in flight
    std::deque<InFlight> inflight_;  // oldest first
    std::deque<std::shared_ptr<DenseTrackingResult> > ready_; // results taken early to free a tracker
    int pipe_depth_ = 4;
    unsigned long long enqueued_ = 0;
};

} // namespace odometry
} // namespace one_piece
