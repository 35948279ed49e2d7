// ImageSequenceIntegration.cpp -- the flow of the reference's example/ImageSequenceIntegration.cpp:9-70 written against
// THIS repository's class surface (host/one_piece), using nothing else: read a sequence directory (associate.txt +
// trajectory.txt), convert (+ optionally filter) the depth, CubeHandler::IntegrateImage with the given poses,
// TransformNearest to the middle pose, ExtractTriangleMesh, write the mesh.  Visualisation and mesh simplification are
// outside the hot path and are left out.  Build: see examples/cpp/Makefile.
//
//   ImageSequenceIntegration <dataset_path> [--voxel 0.00625] [--stride 10] [--filter] [--map out.map] [--ply out.ply]
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "Geometry/Geometry.h"
#include "Integration/CubeHandler.h"
#include "Tool/IO.h"
#include "Tool/ImageProcessing.h"
using namespace one_piece;

int main(int argc, char* argv[]) {
    if (argc < 2) {
        std::cout << "usage::ImageSequenceIntegration [dataset_path] [--voxel v] [--stride n] [--filter] [--map file] [--ply file]" << std::endl;
        return 0;
    }
    float voxel = 0.00625f; // ImageSequenceIntegration.cpp:21
    size_t stride = 10;     // :29
    bool filter = false;
    std::string map_file, ply_file;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--voxel") && i + 1 < argc) voxel = static_cast<float>(std::atof(argv[++i]));
        else if (!std::strcmp(argv[i], "--stride") && i + 1 < argc) stride = static_cast<size_t>(std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--filter")) filter = true;
        else if (!std::strcmp(argv[i], "--map") && i + 1 < argc) map_file = argv[++i];
        else if (!std::strcmp(argv[i], "--ply") && i + 1 < argc) ply_file = argv[++i];
    }
    geometry::TriangleMesh mesh;
    camera::PinholeCamera camera;
    integration::CubeHandler cube_handler(camera);
    cube_handler.SetVoxelResolution(voxel);
    std::vector<std::string> rgb_files, depth_files;
    std::vector<geometry::TransformationMatrix> poses;
    tool::ReadImageSequenceWithPose(argv[1], rgb_files, depth_files, poses);
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    size_t used = 0;
    for (size_t i = 0; i < poses.size(); ++i) {
        if (i % stride != 0) continue;
        cv::Mat rgb = cv::imread(rgb_files[i]);
        cv::Mat depth = cv::imread(depth_files[i], -1);
        if (rgb.empty() || depth.empty()) {
            std::cout << RED << "[ERROR]::cannot read frame " << i << RESET << std::endl;
            return 1;
        }
        cv::Mat refined_depth, filtered_depth;
        tool::ConvertDepthTo32F(depth, refined_depth, camera.GetDepthScale());
        if (filter) tool::BilateralFilter(refined_depth, filtered_depth);
        else filtered_depth = refined_depth;
        cube_handler.IntegrateImage(filtered_depth, rgb, poses[i]);
        ++used;
    }
    cube_handler.Synchronize();
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const size_t blocks = cube_handler.GetCubeCount();
    if (!map_file.empty()) cube_handler.WriteToFile(map_file);
    size_t triangles = 0, transformed_blocks = 0;
    if (!ply_file.empty() && !poses.empty()) {
        std::cout << BLUE << "Transform the voxels ..." << RESET << std::endl;
        std::shared_ptr<integration::CubeHandler> transformed_cube_handler = cube_handler.TransformNearest(poses[poses.size() / 2]);
        transformed_blocks = transformed_cube_handler->GetCubeCount();
        transformed_cube_handler->ExtractTriangleMesh(mesh);
        triangles = mesh.GetTriangleSize();
        mesh.WriteToPLY(ply_file);
    }
    std::cout << "{\"frames\": " << used << ", \"of\": " << poses.size() << ", \"seconds\": " << seconds << ", \"blocks\": " << blocks
              << ", \"transformed_blocks\": " << transformed_blocks << ", \"triangles\": " << triangles << "}" << std::endl;
    return 0;
}
