// DenseFusion.cpp -- the tracking + fusion core of the reference's example/DenseFusion (DenseFusion.cpp:27-101 + DenseSlam.cpp:9-60)
// written against THIS repository's class surface only, the way an MI355X wants to be driven: every frame's images go to the device once
// (geometry::RGBDFrame::on_device), up to --pipeline frame pairs are tracked concurrently (odometry::Odometry::DenseTrackingEnqueue / Wait:
// the pair (i-1, i) is enqueued before (i-2, i-1) is known, assuming it will succeed -- a frame is tracked against the LAST TRACKED one,
// DenseSlam.cpp:24-33, so a failure drops the speculative pairs and resumes from the last tracked frame), results are taken in order, the pose
// is chained (global = global_last * T^-1, DenseSlam.cpp:31) and the frame is fused in place with its TRACKED pose (CubeHandler::
// IntegrateImage(const RGBDFrame&, pose)).  Poses, flags and the fused volume are those of the one-pair-at-a-time loop (--pipeline 1).
// The reference's own example/DenseFusion -- with submap registration and pose-graph optimisation -- compiles and runs unedited against the
// same surface (tests/test_reference_examples.py); this driver is the throughput-oriented form of its tracking + fusion part.
//
//   DenseFusion <dataset_path> [--voxel 0.01] [--stride 1] [--pipeline 4] [--preload] [--repeat 1] [--filter] [--sums reference|fp64] [--ply out.ply] [--poses out.txt]
//   --sums: how the trackers sum an iteration's normal equations: reference (the library's default: the reference's own sequential float32 order, every pose within 1e-4
//           of the CPU path) or fp64 (the order-free device reduction, ~15 x the frame rate; op_runtime_set_option(OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS, ...))
//   --preload: decode all PNGs before the clock starts (the rate then measures tracking + fusion, not the PNG decoder)
//   --repeat n: run the whole sequence n times into a cleared volume and report the LAST pass (with --preload): the first pass also creates the
//               volume (a 2.7 GB pool), the trackers, their streams and graphs -- ~0.1 s of one-time work that a 160-frame run would mostly measure
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>

#include "Geometry/Geometry.h"
#include "Geometry/RGBDFrame.h"
#include "Integration/CubeHandler.h"
#include "Odometry/Odometry.h"
#include "Tool/IO.h"
#include "Tool/ImageProcessing.h"
using namespace one_piece;

int main(int argc, char* argv[]) {
    if (argc < 2) {
        std::cout << "usage::DenseFusion [dataset_path] [--voxel v] [--stride n] [--pipeline k] [--preload] [--filter] [--ply file] [--poses file]" << std::endl;
        return 0;
    }
    float voxel = 0.01f;
    size_t stride = 1;
    int pipeline = 4;
    bool filter = false, preload = false;
    int repeat = 1;
    std::string ply_file, pose_file, sums;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--voxel") && i + 1 < argc) voxel = static_cast<float>(std::atof(argv[++i]));
        else if (!std::strcmp(argv[i], "--stride") && i + 1 < argc) stride = static_cast<size_t>(std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--pipeline") && i + 1 < argc) pipeline = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--filter")) filter = true;
        else if (!std::strcmp(argv[i], "--preload")) preload = true;
        else if (!std::strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--sums") && i + 1 < argc) sums = argv[++i];
        else if (!std::strcmp(argv[i], "--ply") && i + 1 < argc) ply_file = argv[++i];
        else if (!std::strcmp(argv[i], "--poses") && i + 1 < argc) pose_file = argv[++i];
    }
    if (stride < 1) stride = 1;
    if (pipeline < 1) pipeline = 1;
    if (repeat < 1 || !preload) repeat = 1;
    if (!sums.empty()) { // before the first tracker exists: the mode new trackers start in
        if (sums != "fp64" && sums != "reference") { std::cout << RED << "[ERROR]::--sums takes reference or fp64" << RESET << std::endl; return 1; }
        op_runtime_set_option(OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS, sums == "fp64" ? OP_TRACK_SUMS_FP64 : OP_TRACK_SUMS_REFERENCE_F32);
    }
    camera::PinholeCamera camera;
    camera.SetCameraType(camera::CameraType::OPEN3D_DATASET);
    odometry::Odometry rgbd_odometry(camera);
    rgbd_odometry.SetPipelineDepth(pipeline);
    integration::CubeHandler cube_handler(camera);
    cube_handler.SetVoxelResolution(voxel);
    std::vector<std::string> rgb_files, depth_files;
    tool::ReadImageSequence(argv[1], rgb_files, depth_files);
    std::vector<size_t> ids;
    for (size_t i = 0; i < rgb_files.size(); i += stride) ids.push_back(i);
    const size_t n = ids.size();
    std::vector<geometry::RGBDFrame> frames(n);
    auto load = [&](size_t k) {
        if (!frames[k].rgb.empty()) return true;
        frames[k] = geometry::RGBDFrame(cv::imread(rgb_files[ids[k]]), cv::imread(depth_files[ids[k]], -1), static_cast<int>(ids[k]));
        if (frames[k].rgb.empty() || frames[k].depth.empty()) {
            std::cout << RED << "[ERROR]::cannot read frame " << ids[k] << RESET << std::endl;
            return false;
        }
        return true;
    };
    const std::chrono::steady_clock::time_point t_load = std::chrono::steady_clock::now();
    if (preload)
        for (size_t k = 0; k < n; ++k)
            if (!load(k)) return 1;
    const double decode_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load).count();

    std::vector<geometry::TransformationMatrix> global_poses; // of the tracked frames, in order
    size_t used = 0, tracked = 0;
    auto fuse = [&](size_t k, const geometry::TransformationMatrix& pose) {
        ++tracked;
        global_poses.push_back(pose);
        if (!filter) { cube_handler.IntegrateImage(frames[k], pose); return; } // in place, from the device copy the tracker made (raw depth / depth_scale)
        cv::Mat refined_depth, filtered_depth;
        tool::ConvertDepthTo32F(frames[k].depth, refined_depth, camera.GetDepthScale()); // DenseFusion.cpp:92-94
        tool::BilateralFilter(refined_depth, filtered_depth);
        cube_handler.IntegrateImage(filtered_depth, frames[k].rgb, pose);
    };
    double t_enqueue = 0, t_wait = 0, t_fuse = 0, seconds = 0; // where the host thread spends the loop (printed with the result)
    auto secs = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    for (int pass = 0; pass < repeat; ++pass) {
    if (pass) { // again, into an empty volume; every frame's device copy is made again as well
        cube_handler.Clear();
        for (size_t k = 0; k < n; ++k) frames[k].on_device.reset();
        global_poses.clear();
        used = tracked = 0; t_enqueue = t_wait = t_fuse = 0;
    }
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    if (n) {
        if (!load(0)) return 1;
        ++used;
        fuse(0, geometry::TransformationMatrix::Identity());
    }
    size_t next = 1, last_ok = 0, released = 0;
    geometry::TransformationMatrix last_pose = geometry::TransformationMatrix::Identity();
    std::deque<std::pair<size_t, size_t> > pending; // (source, target) of the pairs in flight, oldest first
    while (next < n || !pending.empty()) {
        while (next < n && pending.size() < static_cast<size_t>(pipeline)) {
            if (!load(next)) return 1;
            const size_t target = pending.empty() ? last_ok : pending.back().first; // speculation: the previous pair will have tracked
            const std::chrono::steady_clock::time_point te = std::chrono::steady_clock::now();
            if (!rgbd_odometry.DenseTrackingEnqueue(frames[next], frames[target], geometry::TransformationMatrix::Identity(), 0)) return 1;
            t_enqueue += secs(te);
            pending.push_back(std::make_pair(next, target));
            ++next;
        }
        const std::pair<size_t, size_t> pair = pending.front();
        pending.pop_front();
        const std::chrono::steady_clock::time_point tw = std::chrono::steady_clock::now();
        std::shared_ptr<odometry::DenseTrackingResult> result = rgbd_odometry.DenseTrackingWait();
        t_wait += secs(tw);
        ++used;
        const std::chrono::steady_clock::time_point tf = std::chrono::steady_clock::now();
        if (result->tracking_success) { // DenseSlam.cpp:24-33: source = the new frame, target = the last tracked frame
            last_pose = last_pose * result->T.inverse();
            last_ok = pair.first;
            fuse(pair.first, last_pose);
        } else { // the reference starts a new submap here and relies on global registration to place it; without that the frame is skipped,
                 // and the pairs that assumed it tracked are taken back: the next frame is tracked against the last tracked one
            std::cout << YELLOW << "[WARNING]::tracking lost at frame " << ids[pair.first] << RESET << std::endl;
            while (!pending.empty()) { rgbd_odometry.DenseTrackingWait(); pending.pop_front(); }
            next = pair.first + 1;
        }
        t_fuse += secs(tf);
        { // frames before the last tracked one and before the oldest pair in flight are not read again: their images are dropped (with --preload only the
          // device copies, which then go back to the library's buffer cache; the tracker and the volume hold their own references to what they still read)
            const size_t safe = pending.empty() ? last_ok : (pending.front().second < last_ok ? pending.front().second : last_ok);
            for (; released < safe; ++released) {
                if (preload) frames[released].on_device.reset();
                else frames[released].Release();
            }
        }
    }
    cube_handler.Synchronize();
    seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const size_t blocks = cube_handler.GetCubeCount();
    size_t triangles = 0;
    if (!ply_file.empty()) {
        geometry::TriangleMesh mesh;
        cube_handler.ExtractTriangleMesh(mesh);
        triangles = mesh.GetTriangleSize();
        mesh.WriteToPLY(ply_file);
    }
    if (!pose_file.empty()) {
        std::ofstream ofs(pose_file.c_str());
        ofs.precision(9);
        for (size_t k = 0; k < global_poses.size(); ++k) {
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) ofs << global_poses[k](r, c) << (r == 3 && c == 3 ? "\n" : " ");
        }
    }
    std::cout << "{\"frames\": " << used << ", \"tracked\": " << tracked << ", \"pipeline\": " << pipeline << ", \"preloaded\": " << (preload ? "true" : "false") << ", \"passes\": " << repeat
              << ", \"decode_seconds\": " << decode_seconds << ", \"seconds\": " << seconds << ", \"frames_per_s\": " << (seconds > 0 ? used / seconds : 0.0)
              << ", \"host_seconds\": {\"enqueue\": " << t_enqueue << ", \"wait\": " << t_wait << ", \"fuse\": " << t_fuse << "}, \"blocks\": " << blocks << ", \"triangles\": " << triangles << "}" << std::endl;
    return 0;
}
