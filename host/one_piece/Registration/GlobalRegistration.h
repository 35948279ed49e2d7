// Registration/GlobalRegistration.h -- feature-based global registration (reference: src/Registration/GlobalRegistration.h:12-38,
// GlobalRegistration.cpp:28-265): FPFH features, nearest-feature matching, three rounds of distance-consistency pruning, RANSAC over the
// surviving correspondences.  Same names, signatures, defaults and public members as the reference, so that example/DenseFusion (DenseSlam.h:8-9,
// DenseSlam.cpp:76,107) compiles against it unedited.  Host C++ (src/GlobalRegistration.cpp); it is what the reference runs between submaps, off
// the fusion / tracking hot path, and it is not accelerated.  Parity: unpinned -- the result depends on the RANSAC sampler's seed, which the
// reference takes from std::random_device.
#pragma once
#include <memory>
#include <random>
#include <tuple>

#include "3DFeature.h"
#include "Geometry/Geometry.h"
#include "Geometry/Ransac.h"
#include "RegistrationResult.h"

namespace one_piece {
namespace registration {

class RANSACParameter { // GlobalRegistration.h:12-25
  public:
    int max_iteration = 30;
    double threshold = 0.2;  // largest distance of an inlier correspondence
    double scaling = 1.0;
    int max_nn = 100;        // neighbours of a point in FPFH
    int max_nn_normal = 30;
    float search_radius_normal = 0.1;
    float voxel_len = 0.1;   // down-sampling cell
    float search_radius = 0.25;
};

// down-sample both clouds, estimate missing normals, compute features, then as below (GlobalRegistration.cpp:121-210)
std::shared_ptr<RegistrationResult> RansacRegistration(const geometry::PointCloud& source_pcd, const geometry::PointCloud& target_pcd,
                                                       const RANSACParameter& r_para = RANSACParameter());
// (source index, index of the nearest target feature) for every source feature (GlobalRegistration.cpp:28-78)
void FeatureMatching3D(const FeatureSet& source_feature, const FeatureSet& target_feature, geometry::FMatchSet& matching_index);
// keeps a match when one of candidate_num randomly drawn other matches preserves the distance between the two source points to within
// `difference` (relative) in the target (GlobalRegistration.cpp:80-113)
void RejectMatchesRanSaPC(const geometry::Point3List& source_points, const geometry::Point3List& target_points, std::default_random_engine& engine,
                          geometry::FMatchSet& init_matches, int candidate_num = 4, float difference = 0.1);
std::tuple<geometry::PointCloud, FeatureSet> DownSampleAndExtractFeature(const geometry::PointCloud& pcd, const RANSACParameter& r_para);
// features given (GlobalRegistration.cpp:219-265): match, prune three times, RANSAC; T, the inlier pairs and their indices, rmse over the inliers
std::shared_ptr<RegistrationResult> RansacRegistration(const geometry::PointCloud& source_feature_pcd, const geometry::PointCloud& target_feature_pcd,
                                                       const FeatureSet& source_features, const FeatureSet& target_features,
                                                       const RANSACParameter& r_para);

} // namespace registration
} // namespace one_piece
