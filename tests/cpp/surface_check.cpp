// surface_check.cpp -- exercises the class surface of host/one_piece (one_piece::integration::CubeHandler,
// registration::PointToPlane, geometry::PointCloud, tool::*) the way the reference's callers use it, including the value
// semantics the header promises.  Reads a sequence directory written by the Python test, writes .map files the test
// compares with the CPU oracle, prints one JSON line.  Built by tests/test_cpp_surface.py with g++ -std=c++11.
//
//   surface_check <sequence_dir> <out_dir> <fx> <fy> <cx> <cy> <width> <height> <voxel>
#include <cstdlib>
#include <iostream>

#include "Geometry/Geometry.h"
#include "Integration/CubeHandler.h"
#include "Registration/ICP.h"
#include "Tool/IO.h"
#include "Tool/ImageProcessing.h"
using namespace one_piece;

struct Peek : integration::CubeHandler { // protected members stay reachable from a derived class, as in the reference
    explicit Peek(const camera::PinholeCamera& c) : integration::CubeHandler(c) {}
    float Truncation() const { return integrator.truncation; }
    float Resolution() const { return c_para.VoxelResolution; }
    float Far() const { return far; }
    // the reference's protected cube_map (CubeHandler.h:359), read and written the way a derived class would
    size_t MapSize() const { return cube_map.size(); }
    size_t MapObserved() const {
        size_t n = 0;
        for (integration::CubeMap::const_iterator it = cube_map.begin(); it != cube_map.end(); ++it)
            for (size_t v = 0; v < it->second.voxels.size(); ++v) n += it->second.voxels[v].weight > 0 ? 1 : 0;
        return n;
    }
    bool MapHas(const integration::CubeID& id) const { return cube_map.find(id) != cube_map.end(); }
    void MapDrop(const integration::CubeID& id) { cube_map.erase(id); }
    void MapPut(const integration::CubeID& id) { cube_map[id] = integration::VoxelCube(id); }
};

int main(int argc, char** argv) {
    if (argc < 10) return 2;
    const std::string seq = argv[1], out = argv[2];
    camera::PinholeCamera camera(std::atof(argv[3]), std::atof(argv[4]), std::atof(argv[5]), std::atof(argv[6]), std::atoi(argv[7]), std::atoi(argv[8]), 1000.0f);
    const float voxel = static_cast<float>(std::atof(argv[9]));
    std::vector<std::string> rgb_files, depth_files;
    std::vector<geometry::TransformationMatrix> poses;
    tool::ReadImageSequenceWithPose(seq, rgb_files, depth_files, poses);
    const size_t n = poses.size();
    std::vector<cv::Mat> rgb(n), depth(n);
    for (size_t i = 0; i < n; ++i) {
        rgb[i] = cv::imread(rgb_files[i]);
        cv::Mat raw = cv::imread(depth_files[i], -1);
        tool::ConvertDepthTo32F(raw, depth[i], camera.GetDepthScale());
    }
    // a: frames 0..n-2
    Peek a(camera);
    a.SetVoxelResolution(voxel);
    for (size_t i = 0; i + 1 < n; ++i) a.IntegrateImage(depth[i], rgb[i], poses[i]);
    const size_t n_a = a.GetCubeCount();
    // value semantics: b is a deep copy taken while a still has frames queued; a moves on, b does not
    integration::CubeHandler b(a);
    a.IntegrateImage(geometry::RGBDFrame(rgb[n - 1], depth[n - 1]), poses[n - 1]);
    const size_t n_a_after = a.GetCubeCount(), n_b = b.GetCubeCount();
    b.WriteToFile(out + "/b.map");      // == oracle after frames 0..n-2
    a.WriteToFile(out + "/a.map");      // == oracle after all frames
    // GetCubeMap returns a copy; SetCubeMap uploads one
    integration::CubeMap map_b = b.GetCubeMap();
    size_t present = 0, observed = 0;
    for (integration::CubeMap::const_iterator it = map_b.begin(); it != map_b.end(); ++it) {
        present += b.HasCube(it->first) ? 1 : 0;
        for (size_t v = 0; v < it->second.voxels.size(); ++v) observed += it->second.voxels[v].weight > 0 ? 1 : 0;
    }
    const bool far_absent = !b.HasCube(integration::CubeID(100000, -100000, 5));
    map_b.clear();                       // the copy is the caller's: the handler is unaffected
    integration::CubeHandler c(camera);
    c.SetVoxelResolution(voxel);
    c.SetCubeMap(b.GetCubeMap());
    c.WriteToFile(out + "/c.map");       // == b.map
    c.AddCube(integration::CubeID(1000, -1000, 7));
    c.AddCube(integration::CubeID(1000, -1000, 7));
    const size_t n_c = c.GetCubeCount();
    // assignment + Merge: d = b; d.Merge(a)  (the oracle: merge(b, a))
    integration::CubeHandler d;
    d = b;
    d.Merge(a);
    d.WriteToFile(out + "/d.map");
    integration::CubeHandler other_res(camera);
    other_res.SetVoxelResolution(voxel * 2);
    d.Merge(other_res);                  // refused with the reference's warning, d unchanged
    // resampling
    geometry::Se3 x;
    x(0) = 0.03f; x(1) = -0.02f; x(2) = 0.05f; x(3) = 0.02f; x(4) = 0.04f; x(5) = -0.03f;
    const geometry::TransformationMatrix T = geometry::Se3ToSE3(x);
    std::shared_ptr<integration::CubeHandler> tri = b.Transform(T), nea = b.TransformNearest(T);
    tri->WriteToFile(out + "/transform.map");
    nea->WriteToFile(out + "/nearest.map");
    const integration::CubeID probe = nea->GetCubeID(geometry::Point3(0.085f, -0.01f, 0.17f)); // default 0.01 resolution: (1, -1, 2)
    // file round trip
    integration::CubeHandler e(camera);
    e.SetVoxelResolution(voxel);
    e.ReadFromFile(out + "/a.map");
    const size_t n_e = e.GetCubeCount();
    e.Clear();
    const size_t n_e_cleared = e.GetCubeCount();
    // mesh (generated default tables) and point cloud
    geometry::TriangleMesh mesh;
    a.ExtractTriangleMesh(mesh);
    geometry::TriangleMesh one;
    a.GenerateMeshByCube(a.GetCubeID(geometry::Point3(0, 0, 0)), one);
    std::shared_ptr<geometry::PointCloud> band = a.GetPointCloud();
    mesh.WriteToPLY(out + "/mesh.ply");
    // PrepareCubes / ComputeBounding on a fresh handler
    integration::CubeHandler f(camera);
    f.SetVoxelResolution(voxel);
    std::vector<integration::CubeID> list;
    f.PrepareCubes(depth[0], poses[0], list);
    geometry::Point3 mx, mn;
    f.ComputeBounding(depth[0], poses[0], mx, mn);
    // the reference's public per-cube members: Integrator::IntegrateImage on ONE host VoxelCube (two frames), Integrator::GetSDF,
    // and the allocation pass of Transform for one cube
    integration::Integrator integrator;
    integration::CubePara c_para;
    c_para.SetVoxelResolution(voxel);
    const integration::CubeID single_id = list[list.size() / 2];
    integration::VoxelCube single(single_id);
    integrator.IntegrateImage(depth[0], rgb[0], poses[0], camera, single, c_para);
    integrator.IntegrateImage(depth[1], rgb[1], poses[1], camera, single, c_para);
    integration::CubeHandler g(camera);
    g.SetVoxelResolution(voxel);
    integration::CubeMap one_cube;
    one_cube[single_id] = single;
    g.SetCubeMap(one_cube);
    g.WriteToFile(out + "/single.map");
    const float sdf_centre = integrator.GetSDF(c_para.GetGlobalPoint(single_id, 292), camera, poses[0], depth[0]);
    const float sdf_off = integrator.GetSDF(geometry::TransformPoint(poses[0], geometry::Point3(50, 0, 1)), camera, poses[0], depth[0]);
    integration::CubeHandler h(camera);
    h.SetVoxelResolution(voxel);
    h.AddTransformedCube(single, T);
    const size_t n_h = h.GetCubeCount();
    h.AddTransformedCubeNearest(single, T);
    const size_t n_h2 = h.GetCubeCount();
    // registration through the class surface
    geometry::PointCloud s_pcd, t_pcd;
    s_pcd.LoadFromDepth(depth[1], camera);
    t_pcd.LoadFromDepth(depth[0], camera);
    t_pcd.EstimateNormals(0.1f, 30);
    registration::ICPParameter para;
    para.max_iteration = 8;
    para.threshold = 0.05;
    std::shared_ptr<registration::RegistrationResult> plane = registration::PointToPlane(s_pcd, t_pcd, geometry::TransformationMatrix::Identity(), para);
    std::shared_ptr<registration::RegistrationResult> point = registration::PointToPoint(s_pcd, t_pcd);
    geometry::PointCloud no_normals;
    no_normals.points = t_pcd.points;
    std::shared_ptr<registration::RegistrationResult> refused = registration::PointToPlane(s_pcd, no_normals);
    const geometry::TransformationMatrix kab = geometry::EstimateRigidTransformation(plane->correspondence_set);

    // the protected cube_map mirror through a derived class: follows the device volume, and edits reach the device
    Peek m(camera);
    m.SetVoxelResolution(voxel);
    const size_t mirror_empty = m.MapSize();
    m.IntegrateImage(depth[0], rgb[0], poses[0]);
    const size_t mirror_n1 = m.MapSize(), mirror_obs1 = m.MapObserved();
    m.IntegrateImage(depth[1], rgb[1], poses[1]);               // the mirror is stale now: the next look downloads again
    const size_t mirror_n2 = m.MapSize(), mirror_count2 = m.GetCubeCount();
    const integration::CubeID drop_id = list[0], put_id(12345, -2, 7);
    const bool had = m.MapHas(drop_id);
    m.MapDrop(drop_id);
    m.MapPut(put_id);
    const size_t mirror_after_edit = m.GetCubeCount();          // a member call: the edited mirror is uploaded first
    const bool dropped = !m.HasCube(drop_id), put = m.HasCube(put_id);
    m.WriteToFile(out + "/mirror.map");

    std::cout.precision(9);
    std::cout << "{\"mirror\": [" << mirror_empty << ", " << mirror_n1 << ", " << mirror_obs1 << ", " << mirror_n2 << ", " << mirror_count2 << ", " << (had ? 1 : 0)
              << ", " << mirror_after_edit << ", " << (dropped ? 1 : 0) << ", " << (put ? 1 : 0) << "], \"n_a\": " << n_a << ", \"n_a_after\": " << n_a_after << ", \"n_b\": " << n_b << ", \"map_present\": " << present
              << ", \"observed\": " << observed << ", \"far_absent\": " << (far_absent ? 1 : 0) << ", \"n_c\": " << n_c << ", \"n_e\": " << n_e
              << ", \"n_e_cleared\": " << n_e_cleared << ", \"probe\": [" << probe(0) << ", " << probe(1) << ", " << probe(2) << "]"
              << ", \"transform_blocks\": " << tri->GetCubeCount() << ", \"nearest_blocks\": " << nea->GetCubeCount()
              << ", \"mesh_triangles\": " << mesh.GetTriangleSize() << ", \"mesh_points\": " << mesh.GetPointSize() << ", \"one_block_triangles\": "
              << one.GetTriangleSize() << ", \"band_points\": " << band->GetSize() << ", \"list\": " << list.size() << ", \"list0\": [" << (list.empty() ? 0 : list[0](0))
              << ", " << (list.empty() ? 0 : list[0](1)) << ", " << (list.empty() ? 0 : list[0](2)) << "], \"bound_max\": [" << mx(0) << ", " << mx(1) << ", " << mx(2)
              << "], \"trunc\": " << a.Truncation() << ", \"res\": " << a.Resolution() << ", \"far\": " << a.Far() << ", \"plane_inliers\": "
              << plane->correspondence_set_index.size() << ", \"plane_pairs\": " << plane->correspondence_set.size() << ", \"plane_rmse\": " << plane->rmse
              << ", \"refused_inliers\": " << refused->correspondence_set_index.size() << ", \"point_inliers\": " << point->correspondence_set_index.size()
              << ", \"single_id\": [" << single_id(0) << ", " << single_id(1) << ", " << single_id(2) << "], \"sdf_centre\": " << sdf_centre << ", \"sdf_off\": " << sdf_off
              << ", \"added_trilinear\": " << n_h << ", \"added_nearest\": " << n_h2 << ", \"plane_T\": [";
    for (int r = 0; r < 4; ++r) for (int cc = 0; cc < 4; ++cc) std::cout << (r + cc ? ", " : "") << plane->T(r, cc);
    std::cout << "], \"kabsch_T\": [";
    for (int r = 0; r < 4; ++r) for (int cc = 0; cc < 4; ++cc) std::cout << (r + cc ? ", " : "") << kab(r, cc);
    std::cout << "], \"point_T\": [";
    for (int r = 0; r < 4; ++r) for (int cc = 0; cc < 4; ++cc) std::cout << (r + cc ? ", " : "") << point->T(r, cc);
    std::cout << "]}" << std::endl;
    return 0;
}
