// Optimization/Optimizer.h -- optimization::Optimizer (reference: src/Optimization/Optimizer.h:10-26), the pose-only half: FastBA, which is
// what example/DenseFusion calls (DenseSlam.cpp:123).  The reference's BA member (points + poses, BundleAdjustment.cpp) is the sparse
// feature-based SLAM path and is not declared here (out of scope).
#pragma once
#include "Optimization/SimpleBA.h"

namespace one_piece {
namespace optimization {

class Optimizer {
  public:
    // only the camera poses are optimised
    void FastBA(const std::vector<Correspondence>& correspondences, geometry::SE3List& poses, int max_iteration = 5) {
        SimpleBA(correspondences, poses, max_iteration);
    }
};

} // namespace optimization
} // namespace one_piece
