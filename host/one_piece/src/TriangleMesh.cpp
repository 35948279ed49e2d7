// TriangleMesh.cpp -- geometry::TriangleMesh members (host C++; see Geometry/TriangleMesh.h for what each one follows).
#include "Geometry/TriangleMesh.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <unordered_map>

#include "MeshIO.h"

namespace one_piece {
namespace geometry {

namespace {

std::string Extension(const std::string& filename) {
    const size_t dot = filename.rfind('.');
    return dot == std::string::npos ? std::string() : filename.substr(dot + 1);
}

Point3 Cross(const Point3& a, const Point3& b) {
    return Point3(a(1) * b(2) - a(2) * b(1), a(2) * b(0) - a(0) * b(2), a(0) * b(1) - a(1) * b(0));
}
void Normalize(Point3& v) {
    const float n = std::sqrt(v(0) * v(0) + v(1) * v(1) + v(2) * v(2));
    if (n > 0) { v(0) /= n; v(1) /= n; v(2) /= n; }
}

// keeps the triangles with keep[t], drops the vertices nothing refers to any more, re-indexes
void Compact(TriangleMesh& m, const std::vector<char>& keep) {
    const bool has_n = m.HasNormals(), has_c = m.HasColors();
    std::vector<long> remap(m.points.size(), -1);
    Point3uiList tri;
    tri.reserve(m.triangles.size());
    size_t n_pts = 0;
    for (size_t t = 0; t < m.triangles.size(); ++t) {
        if (!keep[t]) continue;
        Point3ui out;
        for (int k = 0; k < 3; ++k) {
            const unsigned v = m.triangles[t](k);
            if (remap[v] < 0) remap[v] = static_cast<long>(n_pts++);
            out(k) = static_cast<unsigned>(remap[v]);
        }
        tri.push_back(out);
    }
    Point3List pts(n_pts), nrm(has_n ? n_pts : 0), col(has_c ? n_pts : 0);
    for (size_t v = 0; v < remap.size(); ++v) {
        if (remap[v] < 0) continue;
        pts[remap[v]] = m.points[v];
        if (has_n) nrm[remap[v]] = m.normals[v];
        if (has_c) col[remap[v]] = m.colors[v];
    }
    m.points.swap(pts); m.normals.swap(nrm); m.colors.swap(col); m.triangles.swap(tri);
}

} // namespace

bool TriangleMesh::LoadFromPLY(const std::string& filename) { Reset(); return meshio::ReadPly(filename, points, normals, colors, &triangles); }
bool TriangleMesh::LoadFromOBJ(const std::string& filename) { Reset(); return meshio::ReadObj(filename, points, normals, colors, &triangles); }
bool TriangleMesh::LoadFromFile(const std::string& filename) {
    const std::string ext = Extension(filename);
    if (ext == "obj") return LoadFromOBJ(filename);
    if (ext == "ply") return LoadFromPLY(filename);
    std::cout << YELLOW << "[WARNING]::[LoadFromFile]::only obj and ply files are supported." << RESET << std::endl;
    return false;
}
bool TriangleMesh::WriteToPLY(const std::string& fileName) const { return meshio::WritePly(fileName, points, normals, colors, &triangles); }
bool TriangleMesh::WriteToOBJ(const std::string& fileName) const { return meshio::WriteObj(fileName, points, normals, colors, &triangles); }

void TriangleMesh::Transform(const geometry::TransformationMatrix& T) {
    TransformPoints(T, points);
    if (HasNormals()) TransformNormals(T, normals);
}

std::shared_ptr<geometry::PointCloud> TriangleMesh::GetPointCloud() const {
    std::shared_ptr<PointCloud> pcd = std::make_shared<PointCloud>();
    pcd->points = points; pcd->normals = normals; pcd->colors = colors;
    return pcd;
}

void TriangleMesh::ComputeNormals() {
    normals.assign(points.size(), Point3(0, 0, 0));
    for (size_t t = 0; t < triangles.size(); ++t) {
        const Point3ui& tri = triangles[t];
        Point3 n = Cross(points[tri(1)] - points[tri(0)], points[tri(2)] - points[tri(0)]);
        Normalize(n);
        for (int k = 0; k < 3; ++k) normals[tri(k)] += n;
    }
    for (size_t v = 0; v < normals.size(); ++v) Normalize(normals[v]);
}

void TriangleMesh::LoadFromMeshes(const std::vector<TriangleMesh>& meshes) {
    Reset();
    for (size_t m = 0; m < meshes.size(); ++m) {
        const unsigned base = static_cast<unsigned>(points.size());
        points.insert(points.end(), meshes[m].points.begin(), meshes[m].points.end());
        normals.insert(normals.end(), meshes[m].normals.begin(), meshes[m].normals.end());
        colors.insert(colors.end(), meshes[m].colors.begin(), meshes[m].colors.end());
        for (size_t t = 0; t < meshes[m].triangles.size(); ++t) {
            const Point3ui& s = meshes[m].triangles[t];
            triangles.push_back(Point3ui(s(0) + base, s(1) + base, s(2) + base));
        }
    }
}

std::shared_ptr<geometry::TriangleMesh> TriangleMesh::ClusteringSimplify(float grid_len) const {
    std::shared_ptr<TriangleMesh> out = std::make_shared<TriangleMesh>(*this);
    if (grid_len <= 0) {
        std::cout << RED << "[ClusteringMeshSimplification]::[ERROR]::Grid length cannot be less than 0." << RESET << std::endl;
        return out;
    }
    struct Cell { unsigned representative; unsigned count; double sum[3]; };
    std::unordered_map<Point3i, Cell, VoxelGridHasher> cells;
    cells.reserve(points.size() / 4 + 16);
    std::vector<char> keep(triangles.size(), 1);
    // pass 1: every triangle corner joins its cell; the first vertex seen in a cell represents it (and lends it its
    // colour / normal), the cell's position is the mean over all corners that fell into it
    for (size_t t = 0; t < triangles.size(); ++t) {
        unsigned rep[3];
        for (int k = 0; k < 3; ++k) {
            const unsigned v = triangles[t](k);
            const Point3& p = points[v];
            const Point3i id(static_cast<int>(std::floor(p(0) / grid_len)), static_cast<int>(std::floor(p(1) / grid_len)), static_cast<int>(std::floor(p(2) / grid_len)));
            std::unordered_map<Point3i, Cell, VoxelGridHasher>::iterator it = cells.find(id);
            if (it == cells.end()) {
                Cell c = {v, 0, {0, 0, 0}};
                it = cells.insert(std::make_pair(id, c)).first;
            }
            it->second.count += 1;
            for (int a = 0; a < 3; ++a) it->second.sum[a] += p(a);
            rep[k] = it->second.representative;
        }
        if (rep[0] == rep[1] || rep[0] == rep[2] || rep[1] == rep[2]) keep[t] = 0;
        else out->triangles[t] = Point3ui(rep[0], rep[1], rep[2]);
    }
    for (std::unordered_map<Point3i, Cell, VoxelGridHasher>::const_iterator it = cells.begin(); it != cells.end(); ++it) {
        const Cell& c = it->second;
        out->points[c.representative] = Point3(static_cast<float>(c.sum[0] / c.count), static_cast<float>(c.sum[1] / c.count), static_cast<float>(c.sum[2] / c.count));
    }
    Compact(*out, keep);
    std::cout << GREEN << "[ClusteringMeshSimplification]::[INFO]::Simplify done." << RESET << std::endl;
    return out;
}

std::shared_ptr<geometry::TriangleMesh> TriangleMesh::Prune(size_t min_points) const {
    std::shared_ptr<TriangleMesh> out = std::make_shared<TriangleMesh>(*this);
    // connected components over the triangles' edges: union-find on the vertices
    std::vector<unsigned> parent(points.size());
    for (size_t v = 0; v < parent.size(); ++v) parent[v] = static_cast<unsigned>(v);
    struct Find {
        std::vector<unsigned>& p;
        unsigned operator()(unsigned v) const { while (p[v] != v) { p[v] = p[p[v]]; v = p[v]; } return v; }
    } find = {parent};
    for (size_t t = 0; t < triangles.size(); ++t) {
        const unsigned a = find(triangles[t](0));
        for (int k = 1; k < 3; ++k) { const unsigned b = find(triangles[t](k)); if (a != b) parent[b] = a; }
    }
    std::vector<size_t> size(points.size(), 0);
    std::vector<char> referenced(points.size(), 0);
    for (size_t t = 0; t < triangles.size(); ++t)
        for (int k = 0; k < 3; ++k) referenced[triangles[t](k)] = 1;
    for (size_t v = 0; v < points.size(); ++v) if (referenced[v]) size[find(static_cast<unsigned>(v))] += 1;
    std::vector<char> keep(triangles.size(), 1);
    size_t pruned = 0;
    for (size_t t = 0; t < triangles.size(); ++t) if (size[find(triangles[t](0))] <= min_points) keep[t] = 0;
    for (size_t v = 0; v < points.size(); ++v) if (referenced[v] && size[find(static_cast<unsigned>(v))] <= min_points) ++pruned;
    Compact(*out, keep);
    std::cout << GREEN << "[MeshPruning]::[INFO]::Prune mesh done. " << pruned << " points are pruned. " << RESET << std::endl;
    return out;
}

} // namespace geometry
} // namespace one_piece
