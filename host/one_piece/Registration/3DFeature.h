// Registration/3DFeature.h -- FPFH point features (reference: src/Registration/3DFeature.h:16-23, 3DFeature.cpp:9-130).  Host C++ for
// example/DenseFusion's submap registration; not accelerated, not part of the pinned parity claim (it feeds an RNG-driven RANSAC).
//
// A feature is a 33-bin histogram (3 angles x 11 bins) held in a geometry::VectorX.  ComputeFPFHFeature follows the reference's arithmetic as
// written (src/Feature3D.cpp): neighbours = the points whose SQUARED distance is below `radius` (the reference hands the radius to nanoflann's
// L2 adaptor unsquared, KDTree.h:133), nearest first, at most `knn` of them including the point itself; every neighbour adds the INTEGER quotient
// 100 / (n - 1) to one bin per angle (3DFeature.cpp:50: both operands are ints); the final feature is the point's own histogram plus the
// 1/distance-weighted histograms of its neighbours, each third rescaled to 100 by the UNWEIGHTED sum (3DFeature.cpp:104-124).  One deviation:
// a third whose neighbour sum is zero stays zero here (the reference multiplies by 100/0 and stores NaN).  The reference's ComputeSPFH takes
// its KDTree wrapper as an argument and is therefore not part of this surface.
#pragma once
#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"

namespace one_piece {
namespace registration {

typedef geometry::Vector4 PairDescriptor;
typedef geometry::VectorX Feature;        // 33 bins
typedef geometry::PointXList FeatureSet;

// the three Darboux-frame angles and the distance of an oriented point pair (3DFeature.cpp:9-27): (atan2(w.nt, u.nt), v.nt, u.d, |pt - ps|)
PairDescriptor ComputePairDescriptor(const geometry::Point3& ps, const geometry::Point3& ns, const geometry::Point3& pt, const geometry::Point3& nt);
void ComputeFPFHFeature(const geometry::PointCloud& pcd, FeatureSet& fpfh_features, int knn = 100, float radius = 0.1);

} // namespace registration
} // namespace one_piece
