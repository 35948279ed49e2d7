// ops_driver.cpp -- torch-free driver of the C-ABI's volume operations for rocprofv3 runs (like prof_driver.cpp: PMC collection
// segfaults under python + torch).  Reads a frame dump of tools/dump_frames.py, fuses every frame into one volume, then runs the
// operations named on the command line `reps` times each and prints wall-clock figures plus the sizes the byte models need:
//   raycast       op_volume_raycast, depth only, outputs in HBM (views = the poses of evenly spaced frames of the dump)
//   raycast_nc    the same with normals + colours
//   pointcloud    op_volume_point_cloud            (CubeHandler::GetPointCloud)
//   mesh          op_volume_extract_mesh           (CubeHandler::ExtractTriangleMesh; tables from libone_piece_hip_host.so)
//   transform     op_volume_transform, trilinear   (CubeHandler::Transform)
//   transform_nn  op_volume_transform, nearest     (CubeHandler::TransformNearest)
//   normals       op_points_from_depth + op_estimate_normals(0.1, 30) of frame 0 (PointCloud::EstimateNormals)
//   bilateral     op_bilateral_filter_depth(d = 7, 0.03, 4.5) of every frame (tool::BilateralFilter)
//   create        op_volume_create + op_volume_destroy of an empty volume of the default capacity (a CubeHandler per submap / per Transform)
//   all           every one of the above
// Usage: ops_driver.bin frames.bin voxel reps op [op ...]
// Build: hipcc --offload-arch=gfx950 -O2 -I include tools/ops_driver.cpp -L onepiece_amd -lonepiece_hip -L host/one_piece -lone_piece_hip_host -o tools/ops_driver.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>
#include <vector>
#include "onepiece_hip.h"

extern "C" void op_host_generate_mc_tables(int* tri_table_256x16, int* edge_pairs_12x2);

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, op_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s frames.bin voxel reps op [op ...]\n", argv[0]); return 2; }
    if (op_runtime_configure(16) != OP_OK) return 1;
    const char* path = argv[1];
    const float voxel = (float)atof(argv[2]);
    const int reps = std::max(1, atoi(argv[3]));
    std::set<std::string> ops;
    for (int i = 4; i < argc; ++i) ops.insert(argv[i]);
    const bool all = ops.count("all") != 0;
    auto want = [&](const char* o) { return all || ops.count(o) != 0; };
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    int hdr[3];
    if (fread(hdr, 4, 3, f) != 3) return 1;
    const int n = hdr[0], w = hdr[1], h = hdr[2];
    const size_t npx = (size_t)w * h;
    std::vector<float> poses((size_t)n * 16), depth(npx * n);
    std::vector<unsigned char> rgb(npx * 3 * n);
    for (int i = 0; i < n; ++i) {
        if (fread(&poses[(size_t)i * 16], 4, 16, f) != 16) return 1;
        if (fread(&depth[npx * i], 4, npx, f) != npx) return 1;
        if (fread(&rgb[npx * 3 * i], 1, npx * 3, f) != npx * 3) return 1;
    }
    fclose(f);
    float* d_depth; unsigned char* d_rgb;
    HK(hipMalloc((void**)&d_depth, depth.size() * 4));
    HK(hipMalloc((void**)&d_rgb, rgb.size()));
    HK(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_rgb, rgb.data(), rgb.size(), hipMemcpyHostToDevice));
    op_camera cam; CK(op_camera_preset(1, &cam));
    cam.width = w; cam.height = h;
    op_volume* v; CK(op_volume_create(&cam, voxel, 0.1f, 5.0f, 0.5f, 0, 1u << 18, &v));
    double t0 = now();
    CK(op_volume_integrate_sequence(v, d_depth, npx * 4, OP_DEPTH_F32, d_rgb, npx * 3, poses.data(), (size_t)n));
    CK(op_volume_sync(v));
    size_t nb; CK(op_volume_block_count(v, &nb));
    printf("fused %d frames of %d x %d at %.4f m in %.1f ms: %zu blocks (%.1f MB of voxels)\n", n, w, h, voxel, (now() - t0) * 1e3, nb, nb * 10240.0 / 1e6);

    if (want("raycast") || want("raycast_nc")) {
        float *d_out, *d_nrm, *d_col;
        HK(hipMalloc((void**)&d_out, npx * 4)); HK(hipMalloc((void**)&d_nrm, npx * 12)); HK(hipMalloc((void**)&d_col, npx * 12));
        std::vector<float> hd(npx);
        const int views = std::min(n, 8);
        for (int nc = 0; nc < 2; ++nc) {
            if (!want(nc ? "raycast_nc" : "raycast")) continue;
            // cold = every view loads every visible block (what the first view after a change of the volume costs: OP_VOLUME_OPT_RAYCAST_PRUNE 0);
            // warm = later views of the unchanged volume, which drop blocks by what earlier views learnt about them (the default)
            for (int warm = 0; warm < 2; ++warm) {
                CK(op_volume_set_option(v, OP_VOLUME_OPT_RAYCAST_PRUNE, warm));
                double best = 1e9, sum = 0; size_t hits = 0; int cnt = 0;
                uint64_t st[4] = {0, 0, 0, 0};
                for (int r = 0; r < reps + 1; ++r)              // (the first round warms up -- and, with pruning on, is the round that learns)
                    for (int k = 0; k < views; ++k) {
                        const float* p = &poses[(size_t)(k * n / views) * 16];
                        t0 = now();
                        CK(op_volume_raycast(v, &cam, p, d_out, nc ? d_nrm : nullptr, nc ? d_col : nullptr, OP_MEM_DEVICE));
                        const double dt = now() - t0;
                        if (r == 0) {
                            HK(hipMemcpy(hd.data(), d_out, npx * 4, hipMemcpyDeviceToHost));
                            for (float z : hd) hits += z > 0;
                        } else {
                            best = std::min(best, dt); sum += dt; ++cnt;
                            uint64_t s4[4]; CK(op_volume_raycast_stats(v, &s4[0], &s4[1], &s4[2], &s4[3]));
                            for (int q = 0; q < 4; ++q) st[q] += s4[q];
                        }
                    }
                printf("%s, %s: %d views x %d reps of %zu rays: mean %.3f ms, best %.3f ms per view (call to completion), %.1f M rays/s, hit fraction %.4f | per view: "
                       "%.0f visible blocks, %.0f dropped unloaded, %.0f loaded, %.0f marched\n",
                       nc ? "raycast_nc (depth + normals + colours)" : "raycast (depth only)", warm ? "warm" : "cold", views, reps, npx, sum / cnt * 1e3, best * 1e3,
                       npx / (sum / cnt) / 1e6, (double)hits / ((double)npx * views), (double)st[0] / cnt, (double)st[1] / cnt, (double)st[2] / cnt, (double)st[3] / cnt);
            }
        }
        (void)hipFree(d_out); (void)hipFree(d_nrm); (void)hipFree(d_col);
    }
    if (want("pointcloud")) {
        size_t np = 0; CK(op_volume_point_cloud(v, nullptr, nullptr, 0, &np));
        std::vector<float> xyz(np * 3), col(np * 3);
        double sum = 0;
        for (int r = 0; r < reps; ++r) { t0 = now(); CK(op_volume_point_cloud(v, xyz.data(), col.data(), np, &np)); sum += now() - t0; }
        printf("pointcloud: %zu points from %zu blocks, %.3f ms per call incl. the %.1f MB download\n", np, nb, sum / reps * 1e3, np * 24.0 / 1e6);
    }
    if (want("mesh")) {
        std::vector<int> tri(256 * 16), edge(24);
        op_host_generate_mc_tables(tri.data(), edge.data());
        size_t nv = 0; CK(op_volume_extract_mesh(v, tri.data(), edge.data(), nullptr, nullptr, nullptr, 0, &nv));
        std::vector<float> pts(nv * 3), col(nv * 3);
        double sum = 0;
        for (int r = 0; r < reps; ++r) { t0 = now(); CK(op_volume_extract_mesh(v, tri.data(), edge.data(), nullptr, pts.data(), col.data(), nv, &nv)); sum += now() - t0; }
        printf("mesh: %zu triangles from %zu blocks, %.3f ms per call incl. the %.1f MB download\n", nv / 3, nb, sum / reps * 1e3, nv * 24.0 / 1e6);
    }
    for (int nearest = 0; nearest < 2; ++nearest) {
        if (!want(nearest ? "transform_nn" : "transform")) continue;
        // a small rigid motion: 3 degrees about y, 2 cm / -1 cm / 3 cm
        const float c = 0.99862953f, s = 0.05233596f;
        const float T[16] = {c, 0, s, 0.02f, 0, 1, 0, -0.01f, -s, 0, c, 0.03f, 0, 0, 0, 1};
        double sum = 0; size_t nout = 0;
        for (int r = 0; r < reps + 1; ++r) { // (the first call pays for the result pool's hipMalloc -- tens of milliseconds; later calls take it from the library's buffer cache)
            op_volume* o = nullptr;
            t0 = now();
            CK(op_volume_transform(v, T, nullptr, nearest, 0, &o));
            CK(op_volume_sync(o));
            if (r) sum += now() - t0;
            CK(op_volume_block_count(o, &nout));
            CK(op_volume_destroy(o));
        }
        printf("%s: %zu -> %zu blocks, %.3f ms per call (new volume included)\n", nearest ? "transform_nn" : "transform", nb, nout, sum / reps * 1e3);
    }
    if (want("create")) {
        double sum = 0, worst = 0;
        for (int r = 0; r < reps + 1; ++r) {
            op_volume* o = nullptr;
            t0 = now();
            CK(op_volume_create(&cam, voxel, 0.1f, 5.0f, 0.5f, 0, 0, &o));
            CK(op_volume_destroy(o));
            const double dt = now() - t0;
            if (r) { sum += dt; worst = std::max(worst, dt); }
        }
        printf("create: empty volume of the default capacity (2^18 blocks, 2.7 GB) created and destroyed in %.3f ms (worst of %d: %.3f ms)\n", sum / reps * 1e3, reps, worst * 1e3);
    }
    if (want("normals")) {
        std::vector<float> xyz(npx * 3), nrm(npx * 3);
        size_t np = 0;
        CK(op_points_from_depth(&cam, depth.data(), OP_DEPTH_F32, OP_MEM_HOST, 0, xyz.data(), &np));
        float *d_xyz, *d_n;
        HK(hipMalloc((void**)&d_xyz, np * 12)); HK(hipMalloc((void**)&d_n, np * 12));
        HK(hipMemcpy(d_xyz, xyz.data(), np * 12, hipMemcpyHostToDevice));
        double sum = 0;
        for (int r = 0; r < reps + 1; ++r) { t0 = now(); CK(op_estimate_normals(d_xyz, np, 0.1f, 30, OP_MEM_DEVICE, 0, d_n)); if (r) sum += now() - t0; }
        printf("normals: %zu points, radius 0.1, knn 30: %.3f ms per call (device arrays)\n", np, sum / reps * 1e3);
        (void)hipFree(d_xyz); (void)hipFree(d_n);
    }
    if (want("bilateral")) {
        float* d_o; HK(hipMalloc((void**)&d_o, npx * 4 * (size_t)n));
        double sum = 0;
        for (int r = 0; r < reps + 1; ++r) {
            t0 = now();
            CK(op_bilateral_filter_depth(d_depth, OP_DEPTH_F32, 1.0f, w, h, n, 7, 0.03f, 4.5f, OP_MEM_DEVICE, 0, nullptr, d_o));
            HK(hipDeviceSynchronize());
            if (r) sum += now() - t0;
        }
        printf("bilateral: %d images of %d x %d, d = 7: %.3f ms per call, %.2f us per image\n", n, w, h, sum / reps * 1e3, sum / reps / n * 1e6);
        (void)hipFree(d_o);
    }
    CK(op_volume_destroy(v));
    return 0;
}
