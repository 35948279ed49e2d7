// Geometry/Geometry.h -- the geometry vocabulary of one_piece's hot path, re-declared for the MI355X library.
//
// Mirrors the declarations of /root/reference/src/Geometry/Geometry.h that the Integration / Registration surface and
// their callers use (type names :34-71, Se3ToSE3 / TransformPoints / TransformPoint / TransformNormals :83-88,
// VoxelGridHasher :101-112, EstimateRigidTransformation :129).  With -DONEPIECE_HAVE_EIGEN the types ARE the
// reference's (Eigen typedefs); without Eigen they are the look-alikes of compat/MiniEigen.h.  Images are cv::Mat:
// OpenCV's with -DONEPIECE_HAVE_OPENCV, compat/MiniCv.h's container otherwise.
// The arithmetic behind the functions lives in libonepiece_hip.so (C-ABI, include/onepiece_hip.h).
#pragma once
#include <cassert>  // <Eigen/Core> and <opencv2/opencv.hpp> bring <cassert> and <fstream> into the reference's Geometry.h; its examples rely on that
#include <cstddef>
#include <fstream>  // (example/MergeMultipleSubmaps.cpp:18,26)
#include <iostream> // the reference's Geometry.h brings it in, and its callers print with std::cout (example/ICPTest.cpp:11)
#include <memory>
#include <string>
#include <utility>
#include <vector>

#ifdef ONEPIECE_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "compat/MiniCv.h"
#endif

#ifdef ONEPIECE_HAVE_EIGEN
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/LU>
#include <Eigen/StdVector>
#else
#include "compat/MiniEigen.h"
#endif

#include "Tool/ConsoleColor.h"
#include "onepiece_hip.h"

namespace one_piece {
namespace camera {
class PinholeCamera;
}
namespace geometry {

typedef float scalar;

#ifdef ONEPIECE_HAVE_EIGEN
typedef Eigen::Matrix<scalar, 2, 1> Vector2;
typedef Eigen::Matrix<scalar, 3, 1> Vector3;
typedef Eigen::Matrix<scalar, 4, 1> Vector4;
typedef Eigen::Matrix<scalar, 6, 1> Vector6;
typedef Eigen::Matrix<scalar, 2, 2> Matrix2;
typedef Eigen::Matrix<scalar, 3, 3> Matrix3;
typedef Eigen::Matrix<scalar, 4, 4> Matrix4;
typedef Eigen::Matrix<scalar, 6, 6> Matrix6;
typedef Eigen::Matrix<scalar, Eigen::Dynamic, 1> VectorX;
typedef Eigen::Matrix<int, 2, 1> Point2i;
typedef Eigen::Matrix<int, 3, 1> Point3i;
typedef Eigen::Matrix<unsigned int, 2, 1> Point2ui;
typedef Eigen::Matrix<unsigned int, 3, 1> Point3ui;
#define ONEPIECE_ALIGNED_VECTOR(T) std::vector<T, Eigen::aligned_allocator<T> >
#else
typedef compat::Mat<scalar, 2, 1> Vector2;
typedef compat::Mat<scalar, 3, 1> Vector3;
typedef compat::Mat<scalar, 4, 1> Vector4;
typedef compat::Mat<scalar, 6, 1> Vector6;
typedef compat::Mat<scalar, 2, 2> Matrix2;
typedef compat::Mat<scalar, 3, 3> Matrix3;
typedef compat::Mat<scalar, 4, 4> Matrix4;
typedef compat::Mat<scalar, 6, 6> Matrix6;
typedef compat::VecX<scalar> VectorX;
typedef compat::Mat<int, 2, 1> Point2i;
typedef compat::Mat<int, 3, 1> Point3i;
typedef compat::Mat<unsigned int, 2, 1> Point2ui;
typedef compat::Mat<unsigned int, 3, 1> Point3ui;
#define ONEPIECE_ALIGNED_VECTOR(T) std::vector<T>
#endif

// Geometry.h:55-57,74-76: fixed-size column vector of any length and a list of them
#ifdef ONEPIECE_HAVE_EIGEN
template <int T>
using Vector = Eigen::Matrix<scalar, T, 1>;
template <int T>
using PointList = std::vector<Eigen::Matrix<scalar, T, 1>, Eigen::aligned_allocator<Eigen::Matrix<scalar, T, 1> > >;
#else
template <int T>
using Vector = compat::Mat<scalar, T, 1>;
template <int T>
using PointList = std::vector<compat::Mat<scalar, T, 1> >;
#endif
typedef Matrix4 TransformationMatrix;
typedef Vector3 Point3;
typedef Vector2 Point2;
typedef Vector6 Se3;
typedef Vector3 So3;
typedef Vector4 Plane; // (normal, d) with normal . p + d = 0 (Geometry.h:85)
typedef Matrix4 SE3;
typedef Matrix3 SO3;

typedef std::pair<Point3, Point3> PointCorrespondence;
typedef std::vector<PointCorrespondence> PointCorrespondenceSet;
typedef std::pair<int, int> FMatch;
typedef std::vector<FMatch> FMatchSet;
typedef std::vector<std::pair<Point2ui, Point2ui> > PixelCorrespondenceSet;
typedef ONEPIECE_ALIGNED_VECTOR(Point2) Point2List;
typedef ONEPIECE_ALIGNED_VECTOR(Point3) Point3List;
typedef ONEPIECE_ALIGNED_VECTOR(Point3i) Point3iList;
typedef ONEPIECE_ALIGNED_VECTOR(Point3ui) Point3uiList;
typedef ONEPIECE_ALIGNED_VECTOR(VectorX) PointXList; // Geometry.h:69
typedef ONEPIECE_ALIGNED_VECTOR(Matrix4) Mat4List;
typedef Mat4List SE3List;
typedef std::vector<Point3List> ImageXYZ; // Geometry.h:79

// Geometry.cpp:9-13 (Sophus SE3::exp; x = (upsilon, omega)) -> op_se3_exp
Matrix4 Se3ToSE3(const Vector6& input);
// Geometry.cpp:19-34,62-70 (4x4 * (p,1) then / w; normals * (n,0))
void TransformPoints(const Matrix4& T, Point3List& points);
Point3 TransformPoint(const Matrix4& T, const Point3& point);
void TransformNormals(const Matrix4& T, Point3List& normals);
// Geometry.cpp:165-171: unit normal of (p2 - p1) x (p3 - p1) and d = -p1 . normal
Plane GetPlane(const Point3& p1, const Point3& p2, const Point3& p3);
// Geometry.cpp:47-60: sqrt(mean squared distance of camera_pose * first to second), accumulated in double
double ComputeReprojectionError3D(const PointCorrespondenceSet& correspondence_set, const SE3& camera_pose);
// Geometry.cpp:107-151 (Kabsch in the reference's sequential float32 order) -> op_estimate_rigid_transformation
TransformationMatrix EstimateRigidTransformation(const PointCorrespondenceSet& correspondence_set);

// Geometry.h:101-112: 64-bit spatial hash of a block id (ints sign-extended to size_t before the multiply)
struct VoxelGridHasher {
    static constexpr size_t p1 = 73856093;
    static constexpr size_t p2 = 19349663;
    static constexpr size_t p3 = 83492791;
    std::size_t operator()(const Point3i& key) const {
        return static_cast<std::size_t>(op_hash_key(key(0), key(1), key(2)));
    }
};

} // namespace geometry
} // namespace one_piece
