// PointCloud.cpp -- geometry::PointCloud / TriangleMesh members of the hot-path surface.
#include "Geometry/PointCloud.h"

#include <cstdio>
#include <fstream>

#include "Bridge.h"
#include "Geometry/RGBDFrame.h"
#include "Geometry/TriangleMesh.h"

namespace one_piece {
namespace geometry {

namespace {
op_camera SizedCamera(const camera::PinholeCamera& camera) { return camera.Pod(); }
} // namespace

void PointCloud::LoadFromDepth(const cv::Mat& depth, const camera::PinholeCamera& camera) {
    Reset();
    const op_camera cam = SizedCamera(camera);
    points.resize(static_cast<size_t>(cam.width) * cam.height);
    size_t n = 0;
    if (bridge::Failed(op_points_from_depth(&cam, depth.data, bridge::DepthFormat(depth), OP_MEM_HOST, bridge::Device(), bridge::Floats(points), &n),
                       "PointCloud::LoadFromDepth"))
        n = 0;
    points.resize(n);
}

void PointCloud::LoadFromRGBD(const cv::Mat& rgb, const cv::Mat& depth, const camera::PinholeCamera& camera) {
    Reset();
    const op_camera cam = SizedCamera(camera);
    points.resize(static_cast<size_t>(cam.width) * cam.height);
    colors.resize(points.size());
    size_t n = 0;
    if (bridge::Failed(op_points_from_rgbd(&cam, depth.data, bridge::DepthFormat(depth), rgb.data, OP_MEM_HOST, bridge::Device(),
                                           bridge::Floats(points), bridge::Floats(colors), &n),
                       "PointCloud::LoadFromRGBD"))
        n = 0;
    points.resize(n);
    colors.resize(n);
}

void PointCloud::LoadFromRGBD(const RGBDFrame& rgbd, const camera::PinholeCamera& camera) { LoadFromRGBD(rgbd.rgb, rgbd.depth, camera); }

void PointCloud::EstimateNormals(float radius, int knn) {
    normals.assign(points.size(), Point3(0, 0, 0));
    if (points.empty()) return;
    if (bridge::Failed(op_estimate_normals(bridge::Floats(points), points.size(), radius, knn, OP_MEM_HOST, bridge::Device(), bridge::Floats(normals)),
                       "PointCloud::EstimateNormals"))
        normals.clear();
}

void PointCloud::Transform(const TransformationMatrix& T) {
    TransformPoints(T, points);
    if (HasNormals()) TransformNormals(T, normals);
}

namespace {
// binary little-endian PLY: x y z [nx ny nz] [red green blue] per vertex, optional triangle list
bool WritePly(const std::string& file, const Point3List& pts, const Point3List& nrm, const Point3List& col, const Point3uiList* tri) {
    std::ofstream os(file.c_str(), std::ios::binary);
    if (!os) {
        std::cout << RED << "[ERROR]::[WriteToPLY]::cannot open " << file << RESET << std::endl;
        return false;
    }
    const bool has_n = nrm.size() == pts.size() && !pts.empty(), has_c = col.size() == pts.size() && !pts.empty();
    os << "ply\nformat binary_little_endian 1.0\nelement vertex " << pts.size() << "\nproperty float x\nproperty float y\nproperty float z\n";
    if (has_n) os << "property float nx\nproperty float ny\nproperty float nz\n";
    if (has_c) os << "property uchar red\nproperty uchar green\nproperty uchar blue\n";
    if (tri) os << "element face " << tri->size() << "\nproperty list uchar uint vertex_indices\n";
    os << "end_header\n";
    for (size_t i = 0; i < pts.size(); ++i) {
        os.write(reinterpret_cast<const char*>(pts[i].data()), 12);
        if (has_n) os.write(reinterpret_cast<const char*>(nrm[i].data()), 12);
        if (has_c) {
            unsigned char rgb[3];
            for (int k = 0; k < 3; ++k) {
                const float v = col[i](k) * 255.0f;
                rgb[k] = static_cast<unsigned char>(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
            os.write(reinterpret_cast<const char*>(rgb), 3);
        }
    }
    if (tri)
        for (size_t i = 0; i < tri->size(); ++i) {
            const unsigned char three = 3;
            os.write(reinterpret_cast<const char*>(&three), 1);
            os.write(reinterpret_cast<const char*>((*tri)[i].data()), 12);
        }
    return static_cast<bool>(os);
}
} // namespace

bool PointCloud::WriteToPLY(const std::string& fileName) const { return WritePly(fileName, points, normals, colors, nullptr); }

void TriangleMesh::Transform(const geometry::TransformationMatrix& T) {
    TransformPoints(T, points);
    if (HasNormals()) TransformNormals(T, normals);
}
std::shared_ptr<geometry::PointCloud> TriangleMesh::GetPointCloud() const {
    std::shared_ptr<PointCloud> pcd = std::make_shared<PointCloud>();
    pcd->points = points; pcd->normals = normals; pcd->colors = colors;
    return pcd;
}
bool TriangleMesh::WriteToPLY(const std::string& fileName) const { return WritePly(fileName, points, normals, colors, &triangles); }

} // namespace geometry
} // namespace one_piece
