// Registration/RegistrationResult.h -- registration::RegistrationResult (reference: src/Registration/RegistrationResult.h:9-16):
// public data members read by the examples (DenseSlam.cpp:94-103).
#pragma once
#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"

namespace one_piece {
namespace registration {

class RegistrationResult {
  public:
    geometry::TransformationMatrix T;                    // Kabsch over the final inlier pairs (ICP.cpp:221)
    geometry::FMatchSet correspondence_set_index;        // (source id, target id), ascending source id
    geometry::PointCorrespondenceSet correspondence_set; // the same pairs as points
    double rmse;
};

} // namespace registration
} // namespace one_piece
