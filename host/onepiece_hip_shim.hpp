// onepiece_hip_shim.hpp -- C++ glue between the reference's own types and the C-ABI of
// include/onepiece_hip.h.  Header-only templates: they name no Eigen / OpenCV / OnePiece header, so
// they compile inside the reference tree (where geometry::TransformationMatrix is Eigen::Matrix4f,
// images are cv::Mat, CubeMap is std::unordered_map<CubeID, VoxelCube, CubeHasher>) and, for the
// tests of this repository, against small look-alike types (tests/cpp/shim_check.cpp).
//
// INTEGRATION.md shows where each helper is called from inside the reference's
// CubeHandler (src/Integration/CubeHandler.{h,cpp}), ICP (src/Registration/ICP.cpp) and the dense
// tracker (src/Odometry/Odometry.cpp).
//
// Type requirements (all satisfied by the reference's types):
//   Mat4    : float operator()(int row, int col)                    geometry::TransformationMatrix
//   Image   : .data (byte pointer), int depth()                      cv::Mat (CV_32F == 5, CV_16U == 2)
//   CubeMap : operator[](CubeID) -> VoxelCube&, clear(), size(), iteration over (CubeID, VoxelCube)
//   CubeID  : CubeID(int,int,int), int operator()(int)               Eigen::Vector3i
//   VoxelCube : VoxelCube(CubeID), std::vector<TSDFVoxel> voxels (512, index x + 8y + 64z)
//   TSDFVoxel : TSDFVoxel(float sdf, float weight, Point3 color), .sdf, .weight, .color(k)
//   Point3  : Point3(float,float,float), float operator()(int), data() -> contiguous floats
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "onepiece_hip.h"

namespace one_piece {
namespace hip_shim {

constexpr int kCvDepth32F = 5; // CV_32F: the reference compares depth.depth() == CV_32FC1 (Integrator.cpp:26)

template <class Mat4>
inline void RowMajor(const Mat4& m, float out[16]) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = m(r, c);
}

template <class Mat4>
inline void FromRowMajor(const float in[16], Mat4& m) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m(r, c) = in[r * 4 + c];
}

// CubeHandler::IntegrateImage(depth, rgb, pose) (CubeHandler.cpp:197-210).  pose_inv: the caller's
// own pose.inverse() (Eigen), so block selection sees exactly the matrix the reference would use.
template <class Image, class Mat4>
inline int IntegrateImage(op_volume* vol, const Image& depth, const Image& rgb, const Mat4& pose, const Mat4& pose_inv) {
    float p[16], pi[16];
    RowMajor(pose, p);
    RowMajor(pose_inv, pi);
    const int fmt = depth.depth() == kCvDepth32F ? OP_DEPTH_F32 : OP_DEPTH_U16;
    return op_volume_integrate(vol, depth.data, fmt, reinterpret_cast<const uint8_t*>(rgb.data), OP_MEM_HOST, p, pi);
}

// tool::ConvertDepthTo32F(depth, refined, depth_scale) + tool::BilateralFilter(refined, target, range)
// (Tool/ImageProcessing.cpp:68-91, 64-67): `source` is the raw CV_16UC1 image or an already converted CV_32FC1
// one; `target_data` is target.data after target.create(source.rows, source.cols, CV_32FC1).
template <class Image>
inline int BilateralFilter(const Image& source, float depth_scale, float* target_data, int range = 7, int device = 0) {
    const int fmt = source.depth() == kCvDepth32F ? OP_DEPTH_F32 : OP_DEPTH_U16;
    return op_bilateral_filter_depth(source.data, fmt, depth_scale, source.cols, source.rows, 1, range, 0.03f, 4.5f, OP_MEM_HOST, device,
                                     nullptr, target_data);
}

// CubeHandler::IntegrateImage(const geometry::RGBDFrame&, pose) (CubeHandler.cpp:211-214): rgbd.depth / rgbd.rgb.
template <class Frame, class Mat4>
inline int IntegrateFrame(op_volume* vol, const Frame& rgbd, const Mat4& pose, const Mat4& pose_inv) {
    return IntegrateImage(vol, rgbd.depth, rgbd.rgb, pose, pose_inv);
}

// CubeHandler::PrepareCubes (CubeHandler.cpp:147-196): fills cube_id_list in the reference's order.
template <class CubeID, class Image, class Mat4>
inline int PrepareCubes(op_volume* vol, const Image& depth, const Mat4& pose, const Mat4& pose_inv, std::vector<CubeID>& cube_id_list) {
    float p[16], pi[16];
    RowMajor(pose, p);
    RowMajor(pose_inv, pi);
    const int fmt = depth.depth() == kCvDepth32F ? OP_DEPTH_F32 : OP_DEPTH_U16;
    size_t n = 0;
    std::vector<int32_t> ids(3 * 65536);
    int rc = op_volume_prepare_cubes(vol, depth.data, fmt, OP_MEM_HOST, p, pi, ids.data(), ids.size() / 3, &n, nullptr);
    if (rc == OP_OK && n > ids.size() / 3) { // blocks are already allocated: the second call only re-lists them
        ids.resize(3 * n);
        rc = op_volume_prepare_cubes(vol, depth.data, fmt, OP_MEM_HOST, p, pi, ids.data(), n, &n, nullptr);
    }
    cube_id_list.clear();
    if (rc != OP_OK) return rc;
    for (size_t i = 0; i < n; ++i) cube_id_list.push_back(CubeID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]));
    return OP_OK;
}

// device volume -> host CubeMap (what GetCubeMap / ExtractTriangleMesh / WriteToFile / ... read)
template <class CubeMap, class CubeID, class VoxelCube, class TSDFVoxel, class Point3>
inline int DownloadInto(op_volume* vol, CubeMap& cube_map) {
    size_t n = 0;
    int rc = op_volume_block_count(vol, &n);
    if (rc != OP_OK) return rc;
    std::vector<int32_t> keys(3 * n);
    std::vector<float> vox(n * 512 * 5);
    if (n) rc = op_volume_download(vol, keys.data(), vox.data(), n, &n);
    if (rc != OP_OK) return rc;
    cube_map.clear();
    for (size_t b = 0; b < n; ++b) {
        const CubeID id(keys[3 * b], keys[3 * b + 1], keys[3 * b + 2]);
        VoxelCube& cube = (cube_map[id] = VoxelCube(id));
        for (int v = 0; v < 512; ++v) {
            const float* p = &vox[(b * 512 + v) * 5];
            cube.voxels[v] = TSDFVoxel(p[0], p[1], Point3(p[2], p[3], p[4]));
        }
    }
    return OP_OK;
}

// host CubeMap -> device volume (after ReadFromFile / SetCubeMap / host-side edits)
template <class CubeMap>
inline int UploadFrom(op_volume* vol, const CubeMap& cube_map) {
    int rc = op_volume_clear(vol);
    if (rc != OP_OK) return rc;
    std::vector<int32_t> keys;
    std::vector<float> vox;
    keys.reserve(3 * cube_map.size());
    vox.reserve(cube_map.size() * 512 * 5);
    for (auto it = cube_map.begin(); it != cube_map.end(); ++it) {
        for (int c = 0; c < 3; ++c) keys.push_back(it->first(c));
        for (int v = 0; v < 512; ++v) {
            const auto& t = it->second.voxels[v];
            vox.push_back(t.sdf); vox.push_back(t.weight);
            vox.push_back(t.color(0)); vox.push_back(t.color(1)); vox.push_back(t.color(2));
        }
    }
    return op_volume_upload(vol, keys.data(), vox.data(), keys.size() / 3);
}

// registration::PointToPlane / PointToPoint (ICP.cpp:146-224 / :31-107) for clouds whose points are
// contiguous xyz floats (geometry::Point3List = std::vector<Eigen::Vector3f>).
// Result must offer: T (Mat4), rmse, correspondence_set_index (vector<pair<int,int>>),
// correspondence_set (vector<pair<Point3,Point3>>).
template <class Result, class Point3List, class Mat4>
inline int RunICP(int mode, const Point3List& source, const Point3List& target, const Point3List* target_normals, const Mat4& init_T,
                  int max_iteration, double threshold, int device, Result& result) {
    float T0[16];
    RowMajor(init_T, T0);
    op_icp_result r;
    std::vector<int32_t> pairs(2 * source.size() + 2);
    const float* nrm = (target_normals && !target_normals->empty()) ? (*target_normals)[0].data() : nullptr;
    const int rc = op_icp_register(mode, source.empty() ? nullptr : source[0].data(), source.size(), target.empty() ? nullptr : target[0].data(), nrm,
                                   target.size(), T0, max_iteration, threshold, device, &r, pairs.data(), source.size());
    if (rc != OP_OK) return rc;
    FromRowMajor(r.T, result.T);
    result.rmse = r.rmse;
    result.correspondence_set_index.clear();
    result.correspondence_set.clear();
    for (uint64_t k = 0; k < r.n_inliers; ++k) {
        result.correspondence_set_index.push_back(std::make_pair((int)pairs[2 * k], (int)pairs[2 * k + 1]));
        result.correspondence_set.push_back(std::make_pair(source[pairs[2 * k]], target[pairs[2 * k + 1]]));
    }
    return OP_OK;
}

// Odometry::MultiScaleComputing (Odometry.cpp:621-687): the eight pyramids + the camera pyramid in,
// pose / pixel pairs / point pairs out.  FloatImage: .ptr<float>() (cv::Mat CV_32FC1, continuous);
// Camera: GetWidth/GetHeight/GetFx/GetFy/GetCx/GetCy (camera::PinholeCamera).  PixelPairs:
// vector<pair<Point2ui,Point2ui>> with Point2ui(unsigned,unsigned); PointPairs:
// vector<pair<Point3,Point3>>.  Returns the status; *success = the reference's return value.
template <class Point2ui, class Point3, class FloatImage, class Camera, class Mat4, class PixelPairs, class PointPairs>
inline int MultiScaleComputing(op_tracker* tracker, const std::vector<FloatImage>& source_color, const std::vector<FloatImage>& target_color,
                               const std::vector<FloatImage>& source_depth, const std::vector<FloatImage>& target_depth,
                               const std::vector<FloatImage>& target_depth_dx, const std::vector<FloatImage>& target_depth_dy,
                               const std::vector<FloatImage>& target_color_dx, const std::vector<FloatImage>& target_color_dy,
                               const std::vector<Camera>& camera_pyramid, const std::vector<int>& iter_count_per_level, int term_type, Mat4& T,
                               PointPairs& correspondence_set, PixelPairs& pixel_correspondence_set, bool* success, double* rmse = nullptr) {
    const int n = (int)camera_pyramid.size();
    std::vector<op_track_level> lv(n);
    for (int i = 0; i < n; ++i) {
        const Camera& c = camera_pyramid[i];
        lv[i].width = (int32_t)c.GetWidth(); lv[i].height = (int32_t)c.GetHeight();
        lv[i].fx = c.GetFx(); lv[i].fy = c.GetFy(); lv[i].cx = c.GetCx(); lv[i].cy = c.GetCy();
        lv[i].source_color = source_color[i].template ptr<float>(); lv[i].source_depth = source_depth[i].template ptr<float>();
        lv[i].target_color = target_color[i].template ptr<float>(); lv[i].target_depth = target_depth[i].template ptr<float>();
        lv[i].target_color_dx = target_color_dx[i].template ptr<float>(); lv[i].target_color_dy = target_color_dy[i].template ptr<float>();
        lv[i].target_depth_dx = target_depth_dx[i].template ptr<float>(); lv[i].target_depth_dy = target_depth_dy[i].template ptr<float>();
    }
    const int W = n ? lv[0].width : 0, H = n ? lv[0].height : 0;
    float T0[16];
    RowMajor(T, T0);
    std::vector<int32_t> iters(iter_count_per_level.begin(), iter_count_per_level.end());
    iters.resize(n, 4); // SetMultiScale pads with 4 (Odometry.h:100-104)
    std::vector<int32_t> pix(4 * (size_t)W * H + 4);
    std::vector<float> pts(6 * (size_t)W * H + 6);
    op_track_result r;
    const int rc = op_tracker_track(tracker, lv.data(), n, iters.data(), W, H, term_type, T0, OP_MEM_HOST, &r, pix.data(), pts.data(),
                                    (size_t)W * H, nullptr, nullptr);
    if (rc != OP_OK) return rc;
    FromRowMajor(r.T, T);
    correspondence_set.clear();
    pixel_correspondence_set.clear();
    for (uint64_t k = 0; k < r.n_correspondences; ++k) {
        pixel_correspondence_set.push_back(std::make_pair(Point2ui((unsigned)pix[4 * k], (unsigned)pix[4 * k + 1]),
                                                          Point2ui((unsigned)pix[4 * k + 2], (unsigned)pix[4 * k + 3])));
        correspondence_set.push_back(std::make_pair(Point3(pts[6 * k], pts[6 * k + 1], pts[6 * k + 2]),
                                                    Point3(pts[6 * k + 3], pts[6 * k + 4], pts[6 * k + 5])));
    }
    if (success) *success = r.tracking_success != 0;
    if (rmse) *rmse = r.rmse;
    return OP_OK;
}

// CubeHandler::ExtractTriangleMesh (CubeHandler.cpp:9-44) / GenerateMeshByCube (:70-114).  tri_table / edge_pairs:
// the reference's own `&MCLookTable[0][0]` / `&EdgeIndexPairs[0][0]` (MarchingCubePredefined.h).  Mesh must offer
// points, colors (vectors of Point3) and triangles (vector of Point3ui); only_block: nullptr or int[3].
template <class Point3, class Point3ui, class Mesh>
inline int ExtractTriangleMesh(op_volume* vol, const int* tri_table, const int* edge_pairs, const int* only_block, Mesh& mesh) {
    size_t n = 0;
    int rc = op_volume_extract_mesh(vol, tri_table, edge_pairs, only_block, nullptr, nullptr, 0, &n);
    if (rc != OP_OK || n == 0) return rc;
    std::vector<float> p(3 * n), c(3 * n);
    rc = op_volume_extract_mesh(vol, tri_table, edge_pairs, only_block, p.data(), c.data(), n, &n);
    if (rc != OP_OK) return rc;
    unsigned index = (unsigned)mesh.points.size();   // MarchingCube.cpp:39: triangles index the running vertex list
    for (size_t k = 0; k < n; ++k) {
        mesh.points.push_back(Point3(p[3 * k], p[3 * k + 1], p[3 * k + 2]));
        mesh.colors.push_back(Point3(c[3 * k], c[3 * k + 1], c[3 * k + 2]));
        if (k % 3 == 2) { mesh.triangles.push_back(Point3ui(index, index + 1, index + 2)); index += 3; }
    }
    return OP_OK;
}

} // namespace hip_shim
} // namespace one_piece
