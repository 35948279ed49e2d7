// prof_driver.cpp -- torch-free driver of the C-ABI for rocprofv3 runs (PMC collection segfaults
// when the profiled process is python+torch).  Reads a frame dump written by tools/dump_frames.py:
//   int32 n, w, h;  then n x { float pose[16]; float depth[w*h]; uint8 rgb[w*h*3] }
// uploads the frames to HBM once, then fuses them `reps` times into a fresh 5 mm volume.
// With "track=K": tracking + fusion with K frame pairs in flight (one tracker each).
// With a 5th argument "track": instead tracks every consecutive frame pair (op_tracker_dense_tracking, device
// frames) and fuses each frame with its TRACKED pose -- the config-4 pipeline, one pair at a time.
// With "icp": registration::PointToPlane of frame 1's cloud onto frame 0's (LoadFromDepth, EstimateNormals, 30 iterations,
// threshold 0.01 -- ICPTest.cpp's configuration), `reps` times.
// With "host": one op_volume_integrate call per frame with pageable HOST images (float32 and uint16 depth), the
// reference's call pattern (PCIe-inclusive rate).
// With "batch=N" (N = 1..32): the default fusion loop, but the sequence is handed over N frames per call, so every
// k_integrate launch fuses N frames (batch=1: one frame per launch, where SURVEY 8(d)'s byte model is a lower bound of
// the launch's HBM traffic).
// Build: hipcc --offload-arch=gfx950 -O2 -I include tools/prof_driver.cpp -L onepiece_amd -lonepiece_hip -o tools/prof_driver.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "onepiece_hip.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, op_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (op_runtime_configure(16) != OP_OK) return 1; // one hardware queue per tracker stream in the track=K mode (before the first HIP call; the library sets nothing on its own)
    const char* path = argc > 1 ? argv[1] : "/tmp/frames.bin";
    const int reps = argc > 2 ? atoi(argv[2]) : 1;
    const float voxel = argc > 3 ? (float)atof(argv[3]) : 0.005f;
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
    if (hipMalloc((void**)&d_depth, depth.size() * 4) != hipSuccess || hipMalloc((void**)&d_rgb, rgb.size()) != hipSuccess) return 1;
    if (hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (hipMemcpy(d_rgb, rgb.data(), rgb.size(), hipMemcpyHostToDevice) != hipSuccess) return 1;
    op_camera cam; CK(op_camera_preset(1, &cam));
    cam.width = w; cam.height = h;
    op_volume* v; CK(op_volume_create(&cam, voxel, 0.1f, 5.0f, 0.5f, 0, 1u << 18, &v));
    if (const char* e = getenv("PD_UPDATE")) // PD_UPDATE=sum_form: the opt-in once-per-batch update (OP_VOLUME_UPDATE_SUM_FORM)
        if (std::string(e) == "sum_form") CK(op_volume_set_option(v, OP_VOLUME_OPT_UPDATE, OP_VOLUME_UPDATE_SUM_FORM));
    if (argc > 4 && std::string(argv[4]).rfind("track=", 0) == 0) {
        // "track=K": tracking + fusion with K frame pairs in flight, each on its own tracker (stream): pair i is enqueued while pairs
        // i-K+1 .. i-1 are still running; results are taken in order, the pose is chained and the frame fused -- the C-ABI pipeline behind
        // onepiece_amd/dense_slam.py, without the interpreter in the loop.
        const int K = std::max(1, atoi(argv[4] + 6));
        std::vector<op_tracker*> trk((size_t)K);
        for (auto& t : trk) CK(op_tracker_create(0, &t));
        {   // this driver profiles the fp64-reduction mode unless PD_TRACK_SUMS=reference_f32 asks for the library's default (every iteration's sums in the
            // reference's sequential float32 order, OP_TRACK_SUMS_REFERENCE_F32)
            const char* e = getenv("PD_TRACK_SUMS");
            const int sums = e && std::string(e) == "reference_f32" ? OP_TRACK_SUMS_REFERENCE_F32 : OP_TRACK_SUMS_FP64;
            for (auto t : trk) CK(op_tracker_set_option(t, OP_TRACK_OPT_SUMS, sums));
        }
        const int32_t iters[3] = {4, 8, 16};
        const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int r = 0; r < reps; ++r) {
            CK(op_volume_clear(v));
            float g[16]; for (int k = 0; k < 16; ++k) g[k] = I4[k];
            int ok = 1;
            auto resolve = [&](int j) {                         // frame j >= 1: result of the pair (j - 1, j)
                op_track_result res;
                CK(op_tracker_wait(trk[(size_t)((j - 1) % K)], &res, nullptr, nullptr, 0));
                ok += res.tracking_success;
                float inv[16], ng[16];                          // global = global_last * T^-1 (DenseSlam.cpp:31)
                CK(op_mat4_inverse(res.T, inv));
                for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) ng[a * 4 + b] = ((g[a * 4] * inv[b] + g[a * 4 + 1] * inv[4 + b]) + g[a * 4 + 2] * inv[8 + b]) + g[a * 4 + 3] * inv[12 + b];
                for (int k = 0; k < 16; ++k) g[k] = ng[k];
                CK(op_volume_integrate(v, d_depth + npx * j, OP_DEPTH_F32, d_rgb + npx * 3 * j, OP_MEM_DEVICE, g, nullptr));
                return 0;
            };
            auto t0 = std::chrono::steady_clock::now();
            CK(op_volume_integrate(v, d_depth, OP_DEPTH_F32, d_rgb, OP_MEM_DEVICE, g, nullptr));
            for (int i = 1; i < n; ++i) {
                if (i - 1 >= K && resolve(i - K)) return 1;     // frees the tracker this pair will use
                CK(op_tracker_dense_tracking_enqueue(trk[(size_t)((i - 1) % K)], &cam, 3, iters, d_rgb + npx * 3 * (i - 1), d_rgb + npx * 3 * i, d_depth + npx * (i - 1),
                                                     d_depth + npx * i, OP_DEPTH_F32, I4, OP_TRACK_HYBRID, OP_MEM_DEVICE, 0));
            }
            for (int j = std::max(1, n - K); j < n; ++j) if (resolve(j)) return 1;
            CK(op_volume_sync(v));
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            size_t nb; CK(op_volume_block_count(v, &nb));
            printf("rep %d: %d pairs in flight, tracked %d/%d frames, %.1f frames/s (tracking + fusion), blocks %zu, final t = (%.4f %.4f %.4f)\n", r, K, ok, n, n / dt, nb, g[3], g[7], g[11]);
        }
        for (auto t : trk) op_tracker_destroy(t);
        op_volume_destroy(v);
        return 0;
    }
    const bool track = argc > 4 && std::string(argv[4]) == "track";
    if (track) {
        op_tracker* trk; CK(op_tracker_create(0, &trk));
        CK(op_tracker_set_option(trk, OP_TRACK_OPT_SUMS, OP_TRACK_SUMS_FP64)); // (the profiled mode; the library's default is the reference-order one)
        const int32_t iters[3] = {4, 8, 16};
        const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int r = 0; r < reps; ++r) {
            CK(op_volume_clear(v));
            float g[16]; for (int k = 0; k < 16; ++k) g[k] = I4[k];
            auto t0 = std::chrono::steady_clock::now();
            CK(op_volume_integrate(v, d_depth, OP_DEPTH_F32, d_rgb, OP_MEM_DEVICE, g, nullptr));
            int ok = 1;
            for (int i = 1; i < n; ++i) {
                op_track_result res;
                CK(op_tracker_dense_tracking(trk, &cam, 3, iters, d_rgb + npx * 3 * (i - 1), d_rgb + npx * 3 * i, d_depth + npx * (i - 1), d_depth + npx * i,
                                             OP_DEPTH_F32, I4, OP_TRACK_HYBRID, OP_MEM_DEVICE, &res, nullptr, nullptr, 0));
                ok += res.tracking_success;
                float inv[16], ng[16];                      // global = global_last * T^-1 (DenseSlam.cpp:31)
                CK(op_mat4_inverse(res.T, inv));
                for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) ng[a * 4 + b] = ((g[a * 4] * inv[b] + g[a * 4 + 1] * inv[4 + b]) + g[a * 4 + 2] * inv[8 + b]) + g[a * 4 + 3] * inv[12 + b];
                for (int k = 0; k < 16; ++k) g[k] = ng[k];
                CK(op_volume_integrate(v, d_depth + npx * i, OP_DEPTH_F32, d_rgb + npx * 3 * i, OP_MEM_DEVICE, g, nullptr));
            }
            CK(op_volume_sync(v));
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            size_t nb; CK(op_volume_block_count(v, &nb));
            printf("rep %d: tracked %d/%d frames, %.3f ms/frame (tracking + fusion), blocks %zu, final t = (%.4f %.4f %.4f)\n", r, ok, n, dt / n * 1e3, nb, g[3], g[7], g[11]);
        }
        op_tracker_destroy(trk);
        op_volume_destroy(v);
        return 0;
    }
    if (argc > 4 && std::string(argv[4]) == "icp") {
        if (n < 2) return 1;
        std::vector<float> tgt(npx * 3), src(npx * 3), nrm(npx * 3);
        size_t nt = 0, ns = 0;
        CK(op_points_from_depth(&cam, depth.data(), OP_DEPTH_F32, OP_MEM_HOST, 0, tgt.data(), &nt));
        CK(op_points_from_depth(&cam, depth.data() + npx, OP_DEPTH_F32, OP_MEM_HOST, 0, src.data(), &ns));
        for (int r = 0; r < 3; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            CK(op_points_from_depth(&cam, depth.data() + npx, OP_DEPTH_F32, OP_MEM_HOST, 0, src.data(), &ns));
            printf("op_points_from_depth %.3f ms (host depth image -> host points)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
        }
        auto secs = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
        CK(op_estimate_normals(tgt.data(), nt, 0.1f, 30, OP_MEM_HOST, 0, nrm.data()));
        op_icp* icp = nullptr;
        for (int r = 0; r < 3; ++r) { // the fixed costs of one registration::PointToPlane call around the loop
            if (icp) { auto t0 = std::chrono::steady_clock::now(); CK(op_icp_destroy(icp)); printf("op_icp_destroy %.3f ms; ", secs(t0) * 1e3); }
            auto t0 = std::chrono::steady_clock::now();
            CK(op_estimate_normals(tgt.data(), nt, 0.1f, 30, OP_MEM_HOST, 0, nrm.data()));
            const double tn = secs(t0); t0 = std::chrono::steady_clock::now();
            CK(op_icp_create(tgt.data(), nrm.data(), nt, 0.01, OP_MEM_HOST, 0, &icp));
            const double tc = secs(t0); t0 = std::chrono::steady_clock::now();
            if (const char* e = getenv("PD_ICP_TIES")) CK(op_icp_set_option(icp, OP_ICP_OPT_TIES, atoi(e))); // 1 = OP_ICP_TIES_REFERENCE
            CK(op_icp_set_source(icp, src.data(), ns, OP_MEM_HOST));
            printf("op_estimate_normals %.3f ms, op_icp_create %.3f ms, op_icp_set_source %.3f ms (host arrays, %zu / %zu points)\n", tn * 1e3, tc * 1e3, secs(t0) * 1e3, nt, ns);
        }
        const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int r = 0; r < reps; ++r)
            for (int mode = 1; mode >= 0; --mode) {
                op_icp_result res;
                auto t0 = std::chrono::steady_clock::now();
                CK(op_icp_run(icp, mode, I4, 30, &res, nullptr, 0, nullptr, nullptr));
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                printf("rep %d %s: %zu x %zu points, 30 iterations in %.3f ms (%.0f it/s), inliers %llu, rmse %.6g\n", r, mode ? "point-to-plane" : "point-to-point",
                       ns, nt, dt * 1e3, 30 / dt, (unsigned long long)res.n_inliers, res.rmse);
            }
        // the loop alone: the difference of a 100- and a 40-iteration call (the final pass and the finish cancel; both calls end on converged
        // steps -- a loop stopped after LARGE steps has its final correspondences re-decided in the host tree, which a 10-iteration call would add)
        for (int mode = 1; mode >= 0; --mode) {
            double best[2] = {1e9, 1e9};
            const int its[2] = {40, 100};
            for (int r = 0; r < 5; ++r)
                for (int k = 0; k < 2; ++k) {
                    op_icp_result res;
                    auto t0 = std::chrono::steady_clock::now();
                    CK(op_icp_run(icp, mode, I4, its[k], &res, nullptr, 0, nullptr, nullptr));
                    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (dt < best[k]) best[k] = dt;
                }
            printf("%s loop only: %.2f us/iteration (%.0f it/s); final pass + finish %.3f ms\n", mode ? "point-to-plane" : "point-to-point",
                   (best[1] - best[0]) / (its[1] - its[0]) * 1e6, (its[1] - its[0]) / (best[1] - best[0]), (best[0] - its[0] * (best[1] - best[0]) / (its[1] - its[0])) * 1e3);
        }
        op_icp_destroy(icp);
        op_volume_destroy(v);
        return 0;
    }
    if (argc > 4 && std::string(argv[4]).rfind("icpk=", 0) == 0) {
        // "icpk=K": K registration contexts, each with its own consecutive frame pair (2k -> 2k + 1... wrapped), 60 point-to-plane iterations each, enqueued together
        // (op_icp_run_enqueue: own stream + host thread per context) -- ICP's replica axis.  Prints the aggregate iteration rate.
        const int K = std::max(1, atoi(argv[4] + 5));
        if (n < 2) return 1;
        std::vector<op_icp*> ctx((size_t)K, nullptr);
        std::vector<float> tgt(npx * 3), src(npx * 3), nrm(npx * 3);
        for (int k = 0; k < K; ++k) {
            const int a = (2 * k) % (n - 1);
            size_t nt = 0, ns = 0;
            CK(op_points_from_depth(&cam, depth.data() + npx * a, OP_DEPTH_F32, OP_MEM_HOST, 0, tgt.data(), &nt));
            CK(op_points_from_depth(&cam, depth.data() + npx * (a + 1), OP_DEPTH_F32, OP_MEM_HOST, 0, src.data(), &ns));
            CK(op_estimate_normals(tgt.data(), nt, 0.1f, 30, OP_MEM_HOST, 0, nrm.data()));
            CK(op_icp_create(tgt.data(), nrm.data(), nt, 0.01, OP_MEM_HOST, 0, &ctx[(size_t)k]));
            CK(op_icp_set_source(ctx[(size_t)k], src.data(), ns, OP_MEM_HOST));
        }
        const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        std::vector<op_icp_result> res((size_t)K);
        for (int r = 0; r < reps + 1; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; ++k) CK(op_icp_run_enqueue(ctx[(size_t)k], 1, I4, 60, &res[(size_t)k], nullptr, 0));
            for (int k = 0; k < K; ++k) CK(op_icp_wait(ctx[(size_t)k]));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (r) printf("rep %d: %d contexts in flight x 60 iterations in %.3f ms: aggregate %.0f it/s (%.2f us per iteration chip-wide), inliers of context 0: %llu\n", r, K, dt * 1e3,
                          K * 60 / dt, dt / (K * 60) * 1e6, (unsigned long long)res[0].n_inliers);
        }
        for (auto c : ctx) op_icp_destroy(c);
        op_volume_destroy(v);
        return 0;
    }
    if (argc > 4 && std::string(argv[4]) == "host") {
        // the reference's own call pattern: one CubeHandler::IntegrateImage(cv::Mat depth, cv::Mat rgb, pose) per frame with
        // PAGEABLE host images (std::vector storage here, like cv::Mat's), float32 depth and raw uint16 depth
        std::vector<unsigned short> d16(depth.size());
        for (size_t i = 0; i < depth.size(); ++i) { const float z = depth[i] * 1000.0f + 0.5f; d16[i] = (unsigned short)(z < 0 ? 0 : (z > 65535 ? 65535 : z)); }
        for (int r = 0; r < reps; ++r)
            for (int fmt = 0; fmt < 2; ++fmt) {
                CK(op_volume_clear(v));
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; ++i) {
                    const void* dp = fmt == 0 ? (const void*)&depth[npx * i] : (const void*)&d16[npx * i];
                    CK(op_volume_integrate(v, dp, fmt == 0 ? OP_DEPTH_F32 : OP_DEPTH_U16, &rgb[npx * 3 * i], OP_MEM_HOST, &poses[(size_t)i * 16], nullptr));
                }
                CK(op_volume_sync(v));
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                size_t nb; CK(op_volume_block_count(v, &nb));
                printf("rep %d host images, %s depth: %d frames, %.1f frames/s (%.1f us/frame, %.2f GB/s over PCIe), blocks %zu\n", r, fmt == 0 ? "float32" : "uint16", n,
                       n / dt, dt / n * 1e6, n * (double)npx * (fmt == 0 ? 7 : 5) / dt / 1e9, nb);
            }
        op_volume_destroy(v);
        return 0;
    }
    int batch = n;
    bool per_call_launch = false; // batch=N: every call's frames are launched as their own batch (else they join the library's queue: 32 per launch)
    if (argc > 4 && std::string(argv[4]).rfind("batch=", 0) == 0) { batch = atoi(argv[4] + 6); per_call_launch = true; }
    if (batch < 1) batch = 1;
    for (int r = 0; r < reps; ++r) {
        CK(op_volume_clear(v));
        auto t0 = std::chrono::steady_clock::now();
        for (int f0 = 0; f0 < n; f0 += batch) {
            const int nf = n - f0 < batch ? n - f0 : batch;
            CK(op_volume_integrate_sequence(v, d_depth + npx * f0, npx * 4, OP_DEPTH_F32, d_rgb + npx * 3 * f0, npx * 3, poses.data() + (size_t)f0 * 16, (size_t)nf));
            if (per_call_launch) CK(op_volume_flush(v));
        }
        CK(op_volume_sync(v));
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t fr, sel, vis, upd; CK(op_volume_stats(v, &fr, &sel, &vis, &upd));
        uint64_t ln, br, vw, sc; CK(op_volume_stats_launches(v, &ln, &br, &vw, &sc));
        size_t nb; CK(op_volume_block_count(v, &nb));
        printf("rep %d: %d frames %.3f ms/frame  sel/frame %.0f  upd/frame %.0f  blocks %zu | k_integrate launches %llu (%.2f frames each): blocks read %.0f, "
               "voxels written %.0f, %.0f shader cycles per launch\n", r, n, dt / n * 1e3, (double)sel / fr, (double)upd / fr, nb, (unsigned long long)ln, (double)fr / ln, (double)br / ln, (double)vw / ln, (double)sc / ln);
    }
    op_volume_destroy(v);
    return 0;
}
