// MultiGpuSequenceIntegration.cpp -- BASELINE configs[4] from a C++ host: the frames of a sequence directory are
// sharded contiguously over N GPUs of one node (one host thread + one CubeHandler-style volume per device, zero
// communication while fusing), then merged into GPU 0's volume over RCCL (op_volume_merge_rccl: the owner-partitioned
// exchange by default, --dense for the one sliced reduce of the whole union).
// Single process, ncclCommInitAll.  With N = 1 the merge is the identity (or, with --force-exchange, the whole
// exchange with a one-rank communicator, which exercises the RCCL path on a single-GPU box).
//
//   MultiGpuSequenceIntegration <dataset_path> [--gpus N] [--voxel 0.005] [--map out.map] [--dense] [--slice-blocks n] [--force-exchange]
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

#include "Geometry/Geometry.h"
#include "Tool/IO.h"
#include "Tool/ImageProcessing.h"
#include "onepiece_hip.h"
using namespace one_piece;

int main(int argc, char* argv[]) {
    if (argc < 2) {
        std::cout << "usage::MultiGpuSequenceIntegration [dataset_path] [--gpus N] [--voxel v] [--map file]" << std::endl;
        return 0;
    }
    int gpus = 1;
    float voxel = 0.005f;
    std::string map_file;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--voxel") && i + 1 < argc) voxel = static_cast<float>(std::atof(argv[++i]));
        else if (!std::strcmp(argv[i], "--map") && i + 1 < argc) map_file = argv[++i];
        else if (!std::strcmp(argv[i], "--dense")) op_runtime_set_option(OP_RUNTIME_OPT_MERGE_ALGORITHM, OP_MERGE_DENSE_REDUCE);
        else if (!std::strcmp(argv[i], "--slice-blocks") && i + 1 < argc) op_runtime_set_option(OP_RUNTIME_OPT_MERGE_SLICE_BLOCKS, std::atoll(argv[++i]));
        else if (!std::strcmp(argv[i], "--force-exchange")) op_runtime_set_option(OP_RUNTIME_OPT_MERGE_FORCE_SINGLE_RANK, 1);
    }
    int available = 0;
    op_device_count(&available);
    if (gpus < 1 || gpus > available) {
        std::cout << RED << "[ERROR]::" << gpus << " GPUs requested, " << available << " available" << RESET << std::endl;
        return 1;
    }
    std::vector<std::string> rgb_files, depth_files;
    std::vector<geometry::TransformationMatrix> poses;
    tool::ReadImageSequenceWithPose(argv[1], rgb_files, depth_files, poses);
    op_camera cam;
    op_camera_preset(1, &cam);
    std::vector<ncclComm_t> comms(gpus);
    std::vector<int> devs(gpus);
    for (int g = 0; g < gpus; ++g) devs[g] = g;
    if (ncclCommInitAll(comms.data(), gpus, devs.data()) != ncclSuccess) {
        std::cout << RED << "[ERROR]::ncclCommInitAll failed" << RESET << std::endl;
        return 1;
    }
    std::vector<op_volume*> vols(gpus, nullptr);
    std::vector<int> status(gpus, 0);
    std::vector<size_t> local_blocks(gpus, 0);
    std::vector<op_merge_stats> mstats(gpus);
    std::vector<double> fuse_ms(gpus, 0.0);
    size_t n_union = 0;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int g = 0; g < gpus; ++g)
        workers.emplace_back([&, g] {
            int rc = op_volume_create(&cam, voxel, 0.1f, 5.0f, 0.5f, g, 0, &vols[g]);
            const size_t n = poses.size(), base = n / gpus, rem = n % gpus;
            const size_t lo = g * base + (static_cast<size_t>(g) < rem ? g : rem), hi = lo + base + (static_cast<size_t>(g) < rem ? 1 : 0);
            for (size_t i = lo; i < hi && rc == OP_OK; ++i) {
                cv::Mat rgb = cv::imread(rgb_files[i]), depth = cv::imread(depth_files[i], -1), refined;
                tool::ConvertDepthTo32F(depth, refined, cam.depth_scale);
                float p[16];
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) p[r * 4 + c] = poses[i](r, c);
                rc = op_volume_integrate(vols[g], refined.data, OP_DEPTH_F32, rgb.data, OP_MEM_HOST, p, nullptr);
            }
            if (rc == OP_OK) rc = op_volume_block_count(vols[g], &local_blocks[g]);
            fuse_ms[g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            size_t merged = 0;
            // every rank calls the merge, also one whose fusion failed: it announces the failure there and all ranks leave together
            const int mrc = op_volume_merge_rccl_stats(vols[g], comms[g], 0, &merged, &mstats[g]);
            if (rc == OP_OK) rc = mrc;
            if (g == 0) n_union = merged;
            status[g] = rc;
        });
    for (size_t g = 0; g < workers.size(); ++g) workers[g].join();
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int bad = 0;
    for (int g = 0; g < gpus; ++g)
        if (status[g] != OP_OK) { std::cout << RED << "[ERROR]::rank " << g << ": " << status[g] << RESET << std::endl; bad = 1; }
    size_t final_blocks = 0;
    if (!bad) op_volume_block_count(vols[0], &final_blocks);
    if (!bad && !map_file.empty()) op_volume_write_file(vols[0], map_file.c_str());
    std::cout << "{\"gpus\": " << gpus << ", \"frames\": " << poses.size() << ", \"seconds\": " << seconds << ", \"union_blocks\": " << n_union
              << ", \"root_blocks\": " << final_blocks << ", \"rank0_local_blocks\": " << local_blocks[0] << ", \"per_rank\": [";
    for (int g = 0; g < gpus; ++g)
        std::cout << (g ? ", " : "") << "{\"rank\": " << mstats[g].rank << ", \"rccl_ranks\": " << mstats[g].ranks << ", \"local_blocks\": " << local_blocks[g]
                  << ", \"fusion_ms\": " << fuse_ms[g] << ", \"merge_ms\": " << mstats[g].total_ms << ", \"merge_prepare_ms\": " << mstats[g].prepare_ms
                  << ", \"merge_transfer_ms\": " << mstats[g].transfer_ms << ", \"merge_bytes\": " << mstats[g].reduce_bytes << ", \"slices\": " << mstats[g].slices
                  << ", \"algorithm\": " << mstats[g].algorithm << ", \"held_blocks\": " << mstats[g].held_blocks << ", \"owned_blocks\": " << mstats[g].owned_blocks
                  << ", \"wire_bytes_sent\": " << mstats[g].wire_bytes_sent << ", \"wire_bytes_received\": " << mstats[g].wire_bytes_received << "}";
    std::cout << "], \"ok\": " << (bad ? "false" : "true") << "}" << std::endl;
    for (int g = 0; g < gpus; ++g) { if (vols[g]) op_volume_destroy(vols[g]); ncclCommDestroy(comms[g]); }
    return bad;
}
