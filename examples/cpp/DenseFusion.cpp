// DenseFusion.cpp -- the tracking + fusion core of the reference's example/DenseFusion (DenseFusion.cpp:27-101 + DenseSlam.cpp:9-60)
// written against THIS repository's class surface only: read an RGB-D sequence directory (associate.txt; no poses needed), track every
// frame against the previous one with odometry::Odometry::DenseTracking (initial guess identity, hybrid term), chain
// global = global_last * T^-1 (DenseSlam.cpp:31), convert + (optionally) filter the depth and fuse the frame with its TRACKED pose
// (CubeHandler::IntegrateImage), extract the mesh.  Submap registration, loop closure and bundle adjustment (Registration/
// GlobalRegistration, Optimization) are outside this path and are left out: the poses are pure odometry.
//
//   DenseFusion <dataset_path> [--voxel 0.01] [--stride 1] [--filter] [--ply out.ply] [--poses out.txt]
#include <chrono>
#include <cstdlib>
#include <cstring>
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
        std::cout << "usage::DenseFusion [dataset_path] [--voxel v] [--stride n] [--filter] [--ply file] [--poses file]" << std::endl;
        return 0;
    }
    float voxel = 0.01f;
    size_t stride = 1;
    bool filter = false;
    std::string ply_file, pose_file;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--voxel") && i + 1 < argc) voxel = static_cast<float>(std::atof(argv[++i]));
        else if (!std::strcmp(argv[i], "--stride") && i + 1 < argc) stride = static_cast<size_t>(std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--filter")) filter = true;
        else if (!std::strcmp(argv[i], "--ply") && i + 1 < argc) ply_file = argv[++i];
        else if (!std::strcmp(argv[i], "--poses") && i + 1 < argc) pose_file = argv[++i];
    }
    camera::PinholeCamera camera;
    camera.SetCameraType(camera::CameraType::OPEN3D_DATASET);
    odometry::Odometry rgbd_odometry(camera);
    integration::CubeHandler cube_handler(camera);
    cube_handler.SetVoxelResolution(voxel);
    std::vector<std::string> rgb_files, depth_files;
    tool::ReadImageSequence(argv[1], rgb_files, depth_files);
    std::vector<geometry::TransformationMatrix> global_poses;
    geometry::RGBDFrame last;
    size_t used = 0, tracked = 0;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < rgb_files.size(); i += stride) {
        geometry::RGBDFrame frame(cv::imread(rgb_files[i]), cv::imread(depth_files[i], -1), static_cast<int>(i));
        if (frame.rgb.empty() || frame.depth.empty()) {
            std::cout << RED << "[ERROR]::cannot read frame " << i << RESET << std::endl;
            return 1;
        }
        geometry::TransformationMatrix pose = geometry::TransformationMatrix::Identity();
        bool ok = true;
        if (used > 0) { // DenseSlam.cpp:24-33: source = the new frame, target = the last tracked frame
            std::shared_ptr<odometry::DenseTrackingResult> result =
                rgbd_odometry.DenseTracking(frame, last, geometry::TransformationMatrix::Identity(), 0);
            ok = result->tracking_success;
            if (ok) pose = global_poses.back() * result->T.inverse();
        }
        ++used;
        if (!ok) { // the reference starts a new submap here and relies on global registration to place it; without that the frame is skipped
            std::cout << YELLOW << "[WARNING]::tracking lost at frame " << i << RESET << std::endl;
            continue;
        }
        ++tracked;
        global_poses.push_back(pose);
        cv::Mat refined_depth, filtered_depth;
        tool::ConvertDepthTo32F(frame.depth, refined_depth, camera.GetDepthScale()); // DenseFusion.cpp:92-94
        if (filter) tool::BilateralFilter(refined_depth, filtered_depth);
        else filtered_depth = refined_depth;
        cube_handler.IntegrateImage(filtered_depth, frame.rgb, pose);
        last = frame;
    }
    cube_handler.Synchronize();
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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
    std::cout << "{\"frames\": " << used << ", \"tracked\": " << tracked << ", \"seconds\": " << seconds << ", \"blocks\": " << blocks
              << ", \"triangles\": " << triangles << "}" << std::endl;
    return 0;
}
