// host_only_check.cpp -- the members of the class surface that run on the host (no GPU needed): integration::Frustum,
// geometry::GetPlane, TriangleMesh::{ComputeNormals, ClusteringSimplify, Prune, LoadFromMeshes, PLY / OBJ round trips},
// PointCloud::{DownSample, MergePCD, LoadFromXYZ, PLY round trip}, VoxelCube::ReadFromBufferFloat, tool::Timer.
// Prints "name value..." lines that tests/test_reference_examples.py compares with its own numpy evaluation.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>

#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"
#include "Geometry/TriangleMesh.h"
#include "Integration/CubeHandler.h"
#include "Tool/TickTock.h"

using namespace one_piece;

static void P3(const char* name, const geometry::Point3& p) { std::printf("%s %.9g %.9g %.9g\n", name, p(0), p(1), p(2)); }

int main(int argc, char** argv) {
    const std::string out = argc > 1 ? argv[1] : "/tmp";
    // ---- Frustum
    camera::PinholeCamera cam; // OPEN3D preset
    geometry::TransformationMatrix T = geometry::TransformationMatrix::Identity();
    T(0, 3) = 0.25f; T(1, 3) = -0.5f; T(2, 3) = 1.0f;
    T(0, 0) = 0.8f; T(0, 2) = 0.6f; T(2, 0) = -0.6f; T(2, 2) = 0.8f; // rotation about y
    integration::Frustum fr;
    fr.ComputeFromCamera(cam, T, 4.0f, 0.5f);
    const geometry::Plane pl[6] = {fr.GetTopPlane(), fr.GetLeftPlane(), fr.GetRightPlane(), fr.GetBottomPlane(), fr.GetNearPlane(), fr.GetFarPlane()};
    for (int k = 0; k < 6; ++k) std::printf("plane%d %.9g %.9g %.9g %.9g\n", k, pl[k](0), pl[k](1), pl[k](2), pl[k](3));
    for (int k = 0; k < 8; ++k) { char n[16]; std::snprintf(n, sizeof n, "corner%d", k); P3(n, fr.corners[k]); }
    for (int k = 0; k < 12; ++k) { std::printf("line%d %.9g %.9g %.9g %.9g %.9g %.9g\n", k, fr.lines[k].first(0), fr.lines[k].first(1), fr.lines[k].first(2),
                                               fr.lines[k].second(0), fr.lines[k].second(1), fr.lines[k].second(2)); }
    // centre of the frustum is inside, a point behind the camera is not
    const geometry::Point3 inside = geometry::TransformPoint(T, geometry::Point3(0, 0, 2)), behind = geometry::TransformPoint(T, geometry::Point3(0, 0, -1));
    std::printf("contain %d %d %d\n", (int)fr.ContainPoint(inside), (int)fr.ContainPoint(behind), (int)fr.ContainPoint(fr.corners[0] * 1.5f));
    std::shared_ptr<geometry::PointCloud> fp = fr.GetPointCloud();
    std::printf("frustum_cloud %zu %zu\n", fp->points.size(), fp->colors.size());
    P3("frustum_cloud_first", fp->points[0]); P3("frustum_cloud_last_of_edge0", fp->points[999]);
    integration::Frustum fv;
    fv.ComputeFromVectors(geometry::Point3(0, 0, 1), geometry::Point3(0, 0, 0), geometry::Point3(1, 0, 0), geometry::Point3(0, 1, 0), 2.0f, 1.0f, 1.0f, 1.5f);
    P3("vec_corner0", fv.corners[0]); P3("vec_corner4", fv.corners[4]);
    const geometry::Plane gp = geometry::GetPlane(geometry::Point3(1, 0, 0), geometry::Point3(0, 1, 0), geometry::Point3(0, 0, 1));
    std::printf("getplane %.9g %.9g %.9g %.9g\n", gp(0), gp(1), gp(2), gp(3));

    // ---- TriangleMesh: a 9 x 9 grid of unit squares in the plane z = 0.1 x (two triangles each) plus a far-away lone triangle
    geometry::TriangleMesh m;
    const int N = 9;
    for (int j = 0; j <= N; ++j)
        for (int i = 0; i <= N; ++i) { m.points.push_back(geometry::Point3(i * 0.1f, j * 0.1f, i * 0.01f)); m.colors.push_back(geometry::Point3(i / 9.0f, j / 9.0f, 0.5f)); }
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
            const unsigned a = j * (N + 1) + i, b = a + 1, c = a + N + 1, d = c + 1;
            m.triangles.push_back(geometry::Point3ui(a, b, c)); m.triangles.push_back(geometry::Point3ui(b, d, c));
        }
    const unsigned base = (unsigned)m.points.size();
    m.points.push_back(geometry::Point3(5, 5, 5)); m.points.push_back(geometry::Point3(5.1f, 5, 5)); m.points.push_back(geometry::Point3(5, 5.1f, 5));
    for (int k = 0; k < 3; ++k) m.colors.push_back(geometry::Point3(1, 0, 0));
    m.triangles.push_back(geometry::Point3ui(base, base + 1, base + 2));
    m.ComputeNormals();
    std::printf("mesh %zu %zu has_normals %d\n", m.GetPointSize(), m.GetTriangleSize(), (int)m.HasNormals());
    P3("normal_grid", m.normals[11]); P3("normal_lone", m.normals[base]);
    std::shared_ptr<geometry::TriangleMesh> pr = m.Prune(3);
    std::printf("pruned %zu %zu\n", pr->GetPointSize(), pr->GetTriangleSize());
    std::shared_ptr<geometry::TriangleMesh> cl = m.ClusteringSimplify(0.25f);
    std::printf("clustered %zu %zu colors %zu normals %zu\n", cl->GetPointSize(), cl->GetTriangleSize(), cl->colors.size(), cl->normals.size());
    for (size_t t = 0; t < cl->triangles.size(); ++t)
        for (int k = 0; k < 3; ++k) if (cl->triangles[t](k) >= cl->points.size()) { std::printf("clustered_bad_index\n"); return 1; }
    for (size_t v = 0; v < cl->points.size(); ++v) { char n[24]; std::snprintf(n, sizeof n, "clustered_p%zu", v); P3(n, cl->points[v]); }
    std::shared_ptr<geometry::TriangleMesh> bad = m.ClusteringSimplify(0.0f);
    std::printf("clustered_zero_grid %zu %zu\n", bad->GetPointSize(), bad->GetTriangleSize());
    // malformed PLY files are refused or repaired, never trusted: a vertex count far beyond the file, a face that names a missing vertex,
    // a negative list count
    {
        { std::ofstream f((out + "/huge.ply").c_str(), std::ios::binary); f << "ply\nformat binary_little_endian 1.0\nelement vertex 4000000000\nproperty float x\nproperty float y\nproperty float z\nend_header\nabc"; }
        { std::ofstream f((out + "/badface.ply").c_str()); f << "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n3 0 1 99\n"; }
        { std::ofstream f((out + "/neglist.ply").c_str()); f << "ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list int int vertex_indices\nend_header\n0 0 0\n-5 0 0 0\n"; }
        geometry::TriangleMesh h1, h2, h3;
        const bool r1 = h1.LoadFromPLY(out + "/huge.ply"), r2 = h2.LoadFromPLY(out + "/badface.ply"), r3 = h3.LoadFromPLY(out + "/neglist.ply");
        std::printf("malformed_ply %d %zu %d %zu %zu %d\n", r1 ? 1 : 0, h1.points.size(), r2 ? 1 : 0, h2.points.size(), h2.triangles.size(), r3 ? 1 : 0);
    }
    // PLY / OBJ round trips
    m.WriteToPLY(out + "/mesh.ply"); m.WriteToOBJ(out + "/mesh.obj");
    geometry::TriangleMesh rp, ro, rf;
    const bool okp = rp.LoadFromPLY(out + "/mesh.ply"), oko = ro.LoadFromOBJ(out + "/mesh.obj"), okf = rf.LoadFromFile(out + "/mesh.ply"), okx = rf.LoadFromFile(out + "/mesh.xyz");
    double dp = 0, dn = 0, dc = 0, dop = 0;
    bool tri_same = rp.triangles.size() == m.triangles.size() && ro.triangles.size() == m.triangles.size();
    for (size_t i = 0; i < m.points.size() && rp.points.size() == m.points.size() && ro.points.size() == m.points.size(); ++i) {
        dp = std::max(dp, (double)(rp.points[i] - m.points[i]).norm()); dn = std::max(dn, (double)(rp.normals[i] - m.normals[i]).norm());
        dc = std::max(dc, (double)(rp.colors[i] - m.colors[i]).norm()); dop = std::max(dop, (double)(ro.points[i] - m.points[i]).norm());
    }
    for (size_t t = 0; tri_same && t < m.triangles.size(); ++t) tri_same = rp.triangles[t] == m.triangles[t] && ro.triangles[t] == m.triangles[t];
    std::printf("roundtrip %d %d %d %d tri_same %d dp %.3g dn %.3g dc %.3g dop %.3g obj_normals %d obj_colors %d\n", (int)okp, (int)oko, (int)okf, (int)okx, (int)tri_same, dp, dn, dc, dop,
                (int)ro.HasNormals(), (int)ro.HasColors());
    std::vector<geometry::TriangleMesh> two(2, m);
    geometry::TriangleMesh joined; joined.LoadFromMeshes(two);
    std::printf("joined %zu %zu last %u %u %u\n", joined.GetPointSize(), joined.GetTriangleSize(), joined.triangles.back()(0), joined.triangles.back()(1), joined.triangles.back()(2));

    // ---- PointCloud
    geometry::PointCloud pc;
    for (int i = 0; i < 1000; ++i) { pc.points.push_back(geometry::Point3((i % 10) * 0.03f, ((i / 10) % 10) * 0.03f, (i / 100) * 0.03f)); pc.colors.push_back(geometry::Point3(i / 1000.0f, 0, 1)); }
    std::shared_ptr<geometry::PointCloud> ds = pc.DownSample(0.1f);
    std::printf("downsample %zu %zu\n", ds->points.size(), ds->colors.size());
    P3("downsample_p0", ds->points[0]); P3("downsample_c0", ds->colors[0]);
    geometry::PointCloud other; other.points.push_back(geometry::Point3(1, 2, 3));
    const size_t before = pc.points.size();
    pc.MergePCD(other); // refused: colours would not match
    std::printf("merge_refused %d\n", (int)(pc.points.size() == before));
    other.colors.push_back(geometry::Point3(0, 1, 0));
    pc.MergePCD(other);
    std::printf("merge_ok %zu %zu\n", pc.points.size(), pc.colors.size());
    geometry::ImageXYZ xyz(2, geometry::Point3List(3, geometry::Point3(1, 1, 1)));
    xyz[0][1] = geometry::Point3(0, 0, 0); xyz[1][2] = geometry::Point3(2, 2, -1);
    geometry::PointCloud fx; fx.LoadFromXYZ(xyz);
    std::printf("from_xyz %zu\n", fx.points.size());
    pc.WriteToPLY(out + "/cloud.ply");
    geometry::PointCloud back; const bool okc = back.LoadFromPLY(out + "/cloud.ply");
    std::printf("cloud_roundtrip %d %zu %d\n", (int)okc, back.points.size(), (int)back.HasColors());

    // ---- VoxelCube::ReadFromBufferFloat
    integration::VoxelCube cube(integration::CubeID(1, 2, 3));
    const float stream[] = {9, 5, 0.25f, 2, 7, -0.5f, 1, -2, 1, 5, 255, 127.5f, 0, 2};
    std::vector<float> buf(stream, stream + sizeof(stream) / sizeof(float));
    size_t ptr = 0;
    cube.ReadFromBufferFloat(buf, ptr);
    std::printf("cube_float ptr %zu v5 %.9g %.9g %.9g %.9g %.9g v7 %.9g %.9g\n", ptr, cube.voxels[5].sdf, cube.voxels[5].weight, cube.voxels[5].color(0), cube.voxels[5].color(1),
                cube.voxels[5].color(2), cube.voxels[7].sdf, cube.voxels[7].weight);

    // ---- tool::Timer
    tool::Timer timer;
    timer.TICK("a"); timer.TOCK("a"); timer.TOCK("never ticked");
    std::printf("timer %d\n", (int)(timer.Elapsed("a") >= 0));
    return 0;
}
