// Geometry/Ransac.h -- geometry::EstimateRigidTransformationRANSAC (reference: src/Geometry/Ransac.h:12-13, Ransac.cpp:7-40, which drives the
// vendored GRANSAC template with src/Geometry/TransformationModel.hpp).  Host C++, on the path of example/DenseFusion only (submap-to-submap
// registration); not accelerated and NOT part of the pinned parity claim: the reference seeds its samplers from std::random_device, so two runs
// of the reference itself differ.
//
// What the reference computes, restated (src/RansacRigid.cpp): max_iteration times, draw 8 distinct correspondences, fit the rigid transform of
// those 8 (geometry::EstimateRigidTransformation), count the correspondences it maps to within `threshold` (Euclidean distance, strictly below);
// the draw with the largest inlier FRACTION wins (the first one on ties).  Returned: the transform fitted to the winning draw's 8
// correspondences -- not a refit on the inliers (Ransac.cpp:32-39) -- and the winner's inliers with their indices.  Fewer than 8
// correspondences: a warning and the zero matrix (Ransac.cpp:10-14); exactly 8: the reference's estimator declines and its caller dereferences
// a null model -- here that case returns the zero matrix as well.
#pragma once
#include <vector>

#include "Geometry/Geometry.h"

#define MIN_INLIER_SIZE_RANSAC_TRANSFORMATION 8 // TransformationModel.hpp:5

namespace one_piece {
namespace geometry {

TransformationMatrix EstimateRigidTransformationRANSAC(const PointCorrespondenceSet& correspondence_set, PointCorrespondenceSet& inliers,
                                                       std::vector<int>& inlier_ids, int max_iteration = 2000, float threshold = 0.1);

} // namespace geometry
} // namespace one_piece
