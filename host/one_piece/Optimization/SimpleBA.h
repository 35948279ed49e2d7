// Optimization/SimpleBA.h -- optimization::SimpleBA (reference: src/Optimization/SimpleBA.h:16-19, SimpleBA.cpp:19-155): Gauss-Newton over the
// poses only, the first pose held fixed; residual of a pair = pose_s * p - pose_t * q, left-multiplied se(3) increments.  Host C++
// (src/SimpleBA.cpp); what example/DenseFusion calls after every registered submap (DenseSlam.cpp:121-125).  The reference assembles the normal
// equations as an Eigen sparse matrix and solves with SimplicialLDLT; here the (6 (n - 1))^2 system -- n = number of submaps, tens -- is dense
// and solved by an LDL^T in double.  Not accelerated; parity unpinned (its input comes from the RANSAC registration).
#pragma once
#include <vector>

#include "Geometry/Geometry.h"
#include "Optimization/Correspondence.h"

namespace one_piece {
namespace optimization {

void SimpleBA(const std::vector<Correspondence>& correspondences, geometry::SE3List& poses, int max_iteration = 5);

} // namespace optimization
} // namespace one_piece
