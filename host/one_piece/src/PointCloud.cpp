// PointCloud.cpp -- geometry::PointCloud members (the loaders and EstimateNormals forward to the GPU library).
#include "Geometry/PointCloud.h"

#include <cmath>
#include <unordered_map>

#include "Bridge.h"
#include "Geometry/RGBDFrame.h"
#include "MeshIO.h"

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

bool PointCloud::WriteToPLY(const std::string& fileName) const { return meshio::WritePly(fileName, points, normals, colors, nullptr); }
bool PointCloud::WriteToOBJ(const std::string& filename) { return meshio::WriteObj(filename, points, normals, colors, nullptr); }
bool PointCloud::LoadFromPLY(const std::string& filename) { Reset(); return meshio::ReadPly(filename, points, normals, colors, nullptr); }
bool PointCloud::LoadFromOBJ(const std::string& filename) { Reset(); return meshio::ReadObj(filename, points, normals, colors, nullptr); }
bool PointCloud::LoadFromFile(const std::string& filename) {
    const size_t dot = filename.rfind('.');
    const std::string ext = dot == std::string::npos ? std::string() : filename.substr(dot + 1);
    if (ext == "obj") return LoadFromOBJ(filename);
    if (ext == "ply") return LoadFromPLY(filename);
    std::cout << YELLOW << "[WARNING]::[LoadFromFile]::only obj and ply files are supported." << RESET << std::endl;
    return false;
}

void PointCloud::LoadFromXYZ(const ImageXYZ& xyz) {
    Reset();
    for (size_t i = 0; i != xyz.size(); ++i)
        for (size_t j = 0; j != xyz[i].size(); ++j)
            if (xyz[i][j](2) > 0) points.push_back(xyz[i][j]);
}

void PointCloud::MergePCD(const PointCloud& another_pcd) {
    const size_t np = points.size() + another_pcd.points.size();
    const size_t nc = colors.size() + another_pcd.colors.size(), nn = normals.size() + another_pcd.normals.size();
    if (np != nc && nc > 0) { std::cout << RED << "[Error]::[MergePCD]::The color are not matching." << RESET << std::endl; return; }
    if (np != nn && nn > 0) { std::cout << RED << "[Error]::[MergePCD]::The normal are not matching." << RESET << std::endl; return; }
    points.insert(points.end(), another_pcd.points.begin(), another_pcd.points.end());
    colors.insert(colors.end(), another_pcd.colors.begin(), another_pcd.colors.end());
    normals.insert(normals.end(), another_pcd.normals.begin(), another_pcd.normals.end());
}

std::shared_ptr<PointCloud> PointCloud::DownSample(float grid_len) const {
    std::shared_ptr<PointCloud> out = std::make_shared<PointCloud>();
    const bool has_c = HasColors(), has_n = HasNormals();
    std::unordered_map<Point3i, std::pair<size_t, int>, VoxelGridHasher> cells; // cell -> (output index, members)
    for (size_t i = 0; i != points.size(); ++i) {
        const Point3& p = points[i];
        const Point3i id(static_cast<int>(std::floor(p(0) / grid_len)), static_cast<int>(std::floor(p(1) / grid_len)), static_cast<int>(std::floor(p(2) / grid_len)));
        std::unordered_map<Point3i, std::pair<size_t, int>, VoxelGridHasher>::iterator it = cells.find(id);
        if (it == cells.end()) {
            cells.insert(std::make_pair(id, std::make_pair(out->points.size(), 1)));
            out->points.push_back(p);
            if (has_c) out->colors.push_back(colors[i]);
            if (has_n) out->normals.push_back(normals[i]);
        } else {
            out->points[it->second.first] += p;
            if (has_c) out->colors[it->second.first] += colors[i];
            if (has_n) out->normals[it->second.first] += normals[i];
            it->second.second += 1;
        }
    }
    for (std::unordered_map<Point3i, std::pair<size_t, int>, VoxelGridHasher>::const_iterator it = cells.begin(); it != cells.end(); ++it) {
        const float n = static_cast<float>(it->second.second);
        out->points[it->second.first] /= n;
        if (has_c) out->colors[it->second.first] /= n;
        if (has_n) out->normals[it->second.first] /= n;
    }
    return out;
}

} // namespace geometry
} // namespace one_piece
