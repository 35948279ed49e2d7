// shim_check.cpp -- exercises host/onepiece_hip_shim.hpp with look-alikes of the reference's types
// (same member names and shapes as Eigen::Matrix4f / cv::Mat / CubeMap / VoxelCube / TSDFVoxel).
// Built and run by tests/test_cpp_shim.py; with a GPU it fuses the 5-frame survey "wall" scene
// through the shim and prints the statistics the survey recorded from the reference itself.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "onepiece_hip_shim.hpp"

struct Mat4 { // column-major like Eigen::Matrix4f
    float m[16];
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    float& operator()(int r, int c) { return m[c * 4 + r]; }
    static Mat4 Identity() { Mat4 t; std::memset(t.m, 0, sizeof(t.m)); t(0, 0) = t(1, 1) = t(2, 2) = t(3, 3) = 1; return t; }
};
struct Image { // the cv::Mat members the hot path touches
    int rows = 0, cols = 0, type = 0;
    unsigned char* data = nullptr;
    int depth() const { return type & 7; }
};
struct Vec3i {
    int v[3];
    Vec3i() : v{0, 0, 0} {}
    Vec3i(int x, int y, int z) : v{x, y, z} {}
    int operator()(int i) const { return v[i]; }
    bool operator==(const Vec3i& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};
struct Vec3f {
    float v[3];
    Vec3f() : v{0, 0, 0} {}
    Vec3f(float x, float y, float z) : v{x, y, z} {}
    float operator()(int i) const { return v[i]; }
    const float* data() const { return v; }
};
struct Hasher { // geometry::VoxelGridHasher
    size_t operator()(const Vec3i& k) const { return (size_t)op_hash_key(k(0), k(1), k(2)); }
};
struct Voxel {
    float sdf = 999, weight = 0;
    Vec3f color = Vec3f(-1, -1, -1);
    Voxel() = default;
    Voxel(float s, float w, const Vec3f& c) : sdf(s), weight(w), color(c) {}
};
struct Cube {
    std::vector<Voxel> voxels;
    Vec3i cube_id;
    Cube() : voxels(512) {}
    explicit Cube(const Vec3i& id) : voxels(512), cube_id(id) {}
};
typedef std::unordered_map<Vec3i, Cube, Hasher> Map;
struct FloatMat { // cv::Mat CV_32FC1 look-alike
    std::vector<float> buf;
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(buf.data()); }
};
struct Cam { // camera::PinholeCamera getters (GetWidth/GetHeight return float in the reference, Camera.h:61-62)
    float fx, fy, cx, cy; int w, h;
    float GetWidth() const { return (float)w; } float GetHeight() const { return (float)h; }
    float GetFx() const { return fx; } float GetFy() const { return fy; } float GetCx() const { return cx; } float GetCy() const { return cy; }
};
struct Pt2u { unsigned v[2]; Pt2u() : v{0, 0} {} Pt2u(unsigned a, unsigned b) : v{a, b} {} unsigned operator()(int i) const { return v[i]; } };
struct Tri3 { unsigned v[3]; Tri3() : v{0, 0, 0} {} Tri3(unsigned a, unsigned b, unsigned c) : v{a, b, c} {} };
struct Mesh { std::vector<Vec3f> points, colors; std::vector<Tri3> triangles; };
struct Result {
    Mat4 T;
    double rmse = 0;
    std::vector<std::pair<int, int>> correspondence_set_index;
    std::vector<std::pair<Vec3f, Vec3f>> correspondence_set;
};

int main(int argc, char** argv) {
    namespace sh = one_piece::hip_shim;
    int ndev = 0;
    op_device_count(&ndev);
    if (argc > 1 && std::string(argv[1]) == "--compile-only" ) { std::printf("{\"compiled\": true, \"devices\": %d}\n", ndev); return 0; }
    op_camera cam;
    op_camera_preset(1, &cam);
    op_volume* vol = nullptr;
    if (op_volume_create(&cam, 0.005f, 0.1f, 5.0f, 0.5f, 0, 1u << 16, &vol) != OP_OK) { std::printf("{\"error\": \"%s\"}\n", op_last_error()); return ndev ? 1 : 0; }
    // survey scene: d(u,v) = 2 + 0.3 sin(u/40) cos(v/50) in float32, pose = identity + i cm in x
    std::vector<float> d(640 * 480);
    std::vector<unsigned char> c(640 * 480 * 3, 128);
    for (int v = 0; v < 480; ++v)
        for (int u = 0; u < 640; ++u) d[v * 640 + u] = 2.0f + 0.3f * sinf((float)u / 40.0f) * cosf((float)v / 50.0f);
    Image depth; depth.rows = 480; depth.cols = 640; depth.type = 5; depth.data = (unsigned char*)d.data();
    Image rgb; rgb.rows = 480; rgb.cols = 640; rgb.type = 16; rgb.data = c.data();
    std::vector<Vec3i> list;
    size_t first_list = 0;
    for (int i = 0; i < 5; ++i) {
        Mat4 pose = Mat4::Identity(), inv = Mat4::Identity();
        pose(0, 3) = (float)(0.01 * i);
        float pr[16], pi[16];
        sh::RowMajor(pose, pr);
        op_mat4_inverse(pr, pi);
        sh::FromRowMajor(pi, inv);
        if (i == 0) { if (sh::PrepareCubes(vol, depth, pose, inv, list) != OP_OK) return 1; first_list = list.size(); }
        if (sh::IntegrateImage(vol, depth, rgb, pose, inv) != OP_OK) { std::printf("{\"error\": \"%s\"}\n", op_last_error()); return 1; }
    }
    // the drivers' depth front end through the shim: a smooth surface stays within a millimetre, the image stays valid
    std::vector<float> filtered(640 * 480);
    const int frc = sh::BilateralFilter(depth, 1000.0f, filtered.data());
    double fmax = 0;
    for (size_t k = 0; k < filtered.size(); ++k) fmax = std::fmax(fmax, std::fabs((double)filtered[k] - d[k]));
    Map map;
    if ((sh::DownloadInto<Map, Vec3i, Cube, Voxel, Vec3f>(vol, map)) != OP_OK) return 1;
    unsigned long long observed = 0, x = 0;
    double wsum = 0;
    for (auto& kv : map) {
        x ^= op_hash_key(kv.first(0), kv.first(1), kv.first(2));
        for (auto& t : kv.second.voxels) if (t.weight > 0) { ++observed; wsum += t.weight; }
    }
    // round trip through UploadFrom
    op_volume* vol2 = nullptr;
    op_volume_create(&cam, 0.005f, 0.1f, 5.0f, 0.5f, 0, 1u << 16, &vol2);
    sh::UploadFrom(vol2, map);
    size_t n2 = 0;
    op_volume_block_count(vol2, &n2);
    // tiny ICP through the shim: a cloud against itself shifted by 2 mm
    std::vector<Vec3f> tgt, src, nrm;
    for (int v = 0; v < 60; ++v) for (int u = 0; u < 80; ++u) {
        float z = 1.5f + 0.1f * sinf(u * 0.2f) * cosf(v * 0.15f);
        tgt.push_back(Vec3f(u * 0.01f, v * 0.01f, z)); nrm.push_back(Vec3f(0, 0, 1)); src.push_back(Vec3f(u * 0.01f + 0.002f, v * 0.01f, z));
    }
    Result res; res.T = Mat4::Identity();
    const int rc = sh::RunICP(OP_ICP_POINT_TO_POINT, src, tgt, (const std::vector<Vec3f>*)nullptr, Mat4::Identity(), 10, 0.05, 0, res);
    // mesh extraction through the shim with a procedural table (a fan over the sign-changing edges)
    static const int EP[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    static int TT[256][16];
    for (int cs = 0; cs < 256; ++cs) {
        int act[12], na = 0, k = 0;
        for (int e2 = 0; e2 < 12; ++e2) if (((cs >> EP[e2][0]) & 1) != ((cs >> EP[e2][1]) & 1)) act[na++] = e2;
        for (int i = 0; i < 16; ++i) TT[cs][i] = -1;
        for (int q = 1; q + 1 < na && k < 15; ++q) { TT[cs][k++] = act[0]; TT[cs][k++] = act[q]; TT[cs][k++] = act[q + 1]; }
    }
    Mesh mesh;
    const int mrc = (sh::ExtractTriangleMesh<Vec3f, Tri3>(vol, &TT[0][0], &EP[0][0], (const int*)nullptr, mesh));
    // dense tracker through the shim: the wall scene against itself, one level, identity expected
    op_tracker* trk = nullptr;
    int trc = op_tracker_create(0, &trk);
    std::vector<FloatMat> sc(1), sdp(1), zero(1);
    sc[0].buf.resize(640 * 480); sdp[0].buf = d; zero[0].buf.assign(640 * 480, 0.0f);
    for (int k = 0; k < 640 * 480; ++k) sc[0].buf[k] = 0.5f + 0.25f * sinf((float)(k % 640) / 17.0f);
    std::vector<Cam> cams(1); cams[0] = Cam{cam.fx, cam.fy, cam.cx, cam.cy, 640, 480};
    std::vector<int> it(1, 2);
    Mat4 Tt = Mat4::Identity();
    std::vector<std::pair<Vec3f, Vec3f>> pt_pairs;
    std::vector<std::pair<Pt2u, Pt2u>> px_pairs;
    bool ok = false;
    double rmse = -1;
    if (trc == OP_OK)
        trc = sh::MultiScaleComputing<Pt2u, Vec3f>(trk, sc, sc, sdp, sdp, zero, zero, zero, zero, cams, it, OP_TRACK_HYBRID, Tt, pt_pairs, px_pairs, &ok, &rmse);
    op_tracker_destroy(trk);
    std::printf("{\"blocks\": %zu, \"observed\": %llu, \"weight_sum\": %.0f, \"xor\": \"0x%llx\", \"first_list\": %zu, \"reupload_blocks\": %zu, "
                "\"icp_rc\": %d, \"icp_pairs\": %zu, \"icp_tx\": %.6f, \"track_rc\": %d, \"track_pairs\": %zu, \"track_ok\": %d, "
                "\"track_tx\": %.6f, \"track_rmse\": %.6g, \"track_first_pair\": [%u, %u, %u, %u], \"mesh_rc\": %d, \"mesh_triangles\": %zu, \"mesh_vertices\": %zu, \"filter_rc\": %d, \"filter_max_change\": %.6g}\n",
                map.size(), observed, wsum, x, first_list, n2, rc, res.correspondence_set_index.size(), res.T(0, 3), trc, px_pairs.size(), (int)ok,
                Tt(0, 3), rmse, px_pairs.empty() ? 0u : px_pairs[0].first(0), px_pairs.empty() ? 0u : px_pairs[0].first(1),
                px_pairs.empty() ? 0u : px_pairs[0].second(0), px_pairs.empty() ? 0u : px_pairs[0].second(1), mrc, mesh.triangles.size(), mesh.points.size(), frc, fmax);
    op_volume_destroy(vol); op_volume_destroy(vol2);
    return 0;
}
