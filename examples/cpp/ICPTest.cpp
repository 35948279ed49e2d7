// ICPTest.cpp -- the flow of the reference's example/ICPTest.cpp:6-50 against this repository's class surface: two
// clouds (here back-projected from two 16-bit depth PNGs of a sequence directory instead of PLY files -- the PLY
// readers are outside the hot path), EstimateNormals where missing, PointToPlane with threshold 0.01, print T.
//
//   ICPTest <source_depth.png> <target_depth.png> [--iterations 30] [--threshold 0.01]
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>

#include "Geometry/Geometry.h"
#include "Registration/ICP.h"
using namespace one_piece;

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cout << "Usage: ICPTest [source_depth_png] [target_depth_png]" << std::endl;
        return 1;
    }
    registration::ICPParameter icp_para;
    icp_para.threshold = 0.01; // ICPTest.cpp:31
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--iterations") && i + 1 < argc) icp_para.max_iteration = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--threshold") && i + 1 < argc) icp_para.threshold = std::atof(argv[++i]);
    }
    camera::PinholeCamera camera;
    geometry::PointCloud s_pcd, t_pcd;
    s_pcd.LoadFromDepth(cv::imread(argv[1], -1), camera);
    t_pcd.LoadFromDepth(cv::imread(argv[2], -1), camera);
    geometry::TransformationMatrix init_T = geometry::TransformationMatrix::Identity();
    if (!s_pcd.HasNormals()) s_pcd.EstimateNormals();
    if (!t_pcd.HasNormals()) t_pcd.EstimateNormals();
    std::shared_ptr<registration::RegistrationResult> result1 = registration::PointToPlane(s_pcd, t_pcd, init_T, icp_para);
    std::cout << result1->T << std::endl;
    std::cout << std::setprecision(9) << "{\"source_points\": " << s_pcd.GetSize() << ", \"target_points\": " << t_pcd.GetSize()
              << ", \"inliers\": " << result1->correspondence_set_index.size() << ", \"rmse\": " << result1->rmse << ", \"T\": [";
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) std::cout << (r + c ? ", " : "") << result1->T(r, c);
    std::cout << "], \"normals\": [";
    for (size_t i = 0; i < t_pcd.normals.size(); i += t_pcd.normals.size() / 7 + 1)
        std::cout << (i ? ", " : "") << "[" << t_pcd.normals[i](0) << ", " << t_pcd.normals[i](1) << ", " << t_pcd.normals[i](2) << "]";
    std::cout << "]}" << std::endl;
    return 0;
}
