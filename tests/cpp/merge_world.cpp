// merge_world.cpp -- TEST DRIVER: op_volume_merge_rccl with world > 1 on ONE GPU.
//
// N host threads = N ranks, every rank with its own volume on device 0, fuse their shards of a frame file (the format of
// tools/dump_frames.py) through op_volume_integrate (host images) and then call op_volume_merge_rccl_stats on communicators
// created with ncclCommInitAll of the library --rccl-library names (handed to the product with op_runtime_set_rccl_library) --
// tests/cpp/librccl_double.so on a one-GPU box (the real RCCL refuses two ranks on one device), the real librccl on a multi-GPU
// node (then rank r runs on device r).  The root writes its merged volume as a .map file; with --root -1 (owner exchange without a
// gather) every rank writes its owned partition as <out.map>.rank<r>.  The caller compares with a sequential CubeHandler::Merge chain.
//
//   merge_world <frames.bin> <out.map> --shards "0-10,10-13,13-13,13-40" [--voxel 0.02] [--root 0] [--fail-rank r] [--devices N]
//               [--rccl-library path] [--algorithm owner|dense] [--slice-blocks n]
//
// --shards: one "first-last" (last exclusive) per rank; an empty range = a rank with nothing to contribute.
// --fail-rank r: rank r fuses a frame no volume can hold (a bounding box of 10 km), so that it ENTERS the merge with a failed volume:
//   every rank must come back with an error instead of waiting in a collective.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "onepiece_hip.h"

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: merge_world frames.bin out.map --shards a-b,c-d,... [--voxel v] [--root r] [--fail-rank r] [--devices n]\n"); return 2; }
    std::string shards_arg;
    float voxel = 0.02f;
    int root = 0, fail_rank = -1, devices = 1;
    std::string rccl_library, algorithm = "owner";
    long long slice_blocks = 0, fault = 0;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--shards") && i + 1 < argc) shards_arg = argv[++i];
        else if (!strcmp(argv[i], "--voxel") && i + 1 < argc) voxel = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--root") && i + 1 < argc) root = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fail-rank") && i + 1 < argc) fail_rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rccl-library") && i + 1 < argc) rccl_library = argv[++i];
        else if (!strcmp(argv[i], "--algorithm") && i + 1 < argc) algorithm = argv[++i];
        else if (!strcmp(argv[i], "--slice-blocks") && i + 1 < argc) slice_blocks = atoll(argv[++i]);
        else if (!strcmp(argv[i], "--fault") && i + 1 < argc) fault = atoll(argv[++i]);   // OP_RUNTIME_OPT_MERGE_FAULT: stage * 1024 + rank + 1
    }
    std::vector<std::pair<int, int>> shards;
    for (size_t p = 0; p < shards_arg.size();) {
        size_t q = shards_arg.find(',', p);
        if (q == std::string::npos) q = shards_arg.size();
        int a = 0, b = 0;
        if (sscanf(shards_arg.substr(p, q - p).c_str(), "%d-%d", &a, &b) != 2) { fprintf(stderr, "bad shard list\n"); return 2; }
        shards.push_back({a, b});
        p = q + 1;
    }
    const int world = (int)shards.size();
    if (world < 1 || root < -1 || root >= world) { fprintf(stderr, "bad world / root\n"); return 2; }
    if (op_runtime_set_option(OP_RUNTIME_OPT_MERGE_ALGORITHM, algorithm == "dense" ? OP_MERGE_DENSE_REDUCE : OP_MERGE_OWNER_EXCHANGE) != OP_OK ||
        op_runtime_set_option(OP_RUNTIME_OPT_MERGE_SLICE_BLOCKS, slice_blocks) != OP_OK ||
        op_runtime_set_option(OP_RUNTIME_OPT_MERGE_FAULT, fault) != OP_OK ||
        op_runtime_set_rccl_library(rccl_library.empty() ? nullptr : rccl_library.c_str()) != OP_OK) { fprintf(stderr, "%s\n", op_last_error()); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int hdr[3];
    if (fread(hdr, 4, 3, f) != 3) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2];
    const size_t npx = (size_t)w * h;
    std::vector<float> poses((size_t)n * 16), depth(npx * n);
    std::vector<unsigned char> rgb(npx * 3 * n);
    for (int i = 0; i < n; ++i) {
        if (fread(&poses[(size_t)i * 16], 4, 16, f) != 16 || fread(&depth[npx * i], 4, npx, f) != npx || fread(&rgb[npx * 3 * i], 1, npx * 3, f) != npx * 3) return 2;
    }
    fclose(f);
    // the communicators come from the same library op_volume_merge_rccl will bind (dlopen returns the same handle)
    void* lib = dlopen(rccl_library.empty() ? "librccl.so.1" : rccl_library.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    auto init_all = (ncclResult_t(*)(ncclComm_t*, int, const int*))dlsym(lib, "ncclCommInitAll");
    auto destroy = (ncclResult_t(*)(ncclComm_t))dlsym(lib, "ncclCommDestroy");
    if (!init_all || !destroy) { fprintf(stderr, "ncclCommInitAll / ncclCommDestroy missing\n"); return 2; }
    std::vector<ncclComm_t> comms((size_t)world);
    std::vector<int> devs((size_t)world);
    for (int r = 0; r < world; ++r) devs[(size_t)r] = r % devices;
    if (init_all(comms.data(), world, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 2; }
    op_camera cam;
    op_camera_preset(1, &cam);
    cam.fx *= (float)w / cam.width; cam.fy *= (float)h / cam.height; cam.cx *= (float)w / cam.width; cam.cy *= (float)h / cam.height; // dump_frames.py writes full-size frames; scaled ones keep the ratio
    cam.width = w; cam.height = h;
    std::vector<op_volume*> vols((size_t)world, nullptr);
    std::vector<int> status((size_t)world, 0);
    std::vector<std::string> errors((size_t)world);
    std::vector<size_t> local_blocks((size_t)world, 0), merged((size_t)world, 0);
    std::vector<op_merge_stats> ms((size_t)world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            int rc = op_volume_create(&cam, voxel, 0.1f, 5.0f, 0.5f, devs[(size_t)r], 1u << 14, &vols[(size_t)r]);
            for (int i = shards[(size_t)r].first; i < shards[(size_t)r].second && i < n && rc == OP_OK; ++i)
                rc = op_volume_integrate(vols[(size_t)r], &depth[npx * i], OP_DEPTH_F32, &rgb[npx * 3 * i], OP_MEM_HOST, &poses[(size_t)i * 16], nullptr);
            if (rc == OP_OK) rc = op_volume_block_count(vols[(size_t)r], &local_blocks[(size_t)r]);
            if (r == fail_rank && rc == OP_OK) { // a frame whose bounding box spans 10 km: the next look at the volume reports it
                std::vector<float> far(npx, 1.0f);
                for (size_t k = 0; k < npx; k += 2) far[k] = 10000.0f;
                rc = op_volume_set_near_far(vols[(size_t)r], 0.5f, 1e9f);
                if (rc == OP_OK) rc = op_volume_integrate(vols[(size_t)r], far.data(), OP_DEPTH_F32, &rgb[0], OP_MEM_HOST, &poses[0], nullptr);
            }
            const int mrc = op_volume_merge_rccl_stats(vols[(size_t)r], comms[(size_t)r], root, &merged[(size_t)r], &ms[(size_t)r]);
            if (rc == OP_OK) rc = mrc;
            status[(size_t)r] = rc;
            if (rc != OP_OK) errors[(size_t)r] = op_last_error(); // (thread-local: read on the rank's own thread)
        });
    for (auto& t : th) t.join();
    int bad = 0;
    for (int r = 0; r < world; ++r) bad |= status[(size_t)r] != OP_OK;
    size_t root_blocks = 0;
    if (!bad && root >= 0) { op_volume_block_count(vols[(size_t)root], &root_blocks); if (op_volume_write_file(vols[(size_t)root], argv[2]) != OP_OK) bad = 1; }
    if (!bad && root < 0)
        for (int r = 0; r < world; ++r) {
            size_t nb = 0;
            op_volume_block_count(vols[(size_t)r], &nb);
            root_blocks += nb;
            if (op_volume_write_file(vols[(size_t)r], (std::string(argv[2]) + ".rank" + std::to_string(r)).c_str()) != OP_OK) bad = 1;
        }
    printf("{\"world\": %d, \"root\": %d, \"ok\": %s, \"root_blocks\": %zu, \"per_rank\": [", world, root, bad ? "false" : "true", root_blocks);
    for (int r = 0; r < world; ++r) {
        std::string e = errors[(size_t)r];
        for (auto& ch : e) if (ch == '"' || ch == '\\') ch = '\'';
        printf("%s{\"rank\": %d, \"status\": %d, \"error\": \"%s\", \"local_blocks\": %zu, \"union_blocks\": %zu, \"rccl_ranks\": %d, \"rccl_rank\": %d, \"slices\": %llu, \"bytes\": %llu, "
               "\"algorithm\": %d, \"held_blocks\": %llu, \"owned_blocks\": %llu, \"wire_bytes_sent\": %llu, \"wire_bytes_received\": %llu}",
               r ? ", " : "", r, status[(size_t)r], e.c_str(), local_blocks[(size_t)r], merged[(size_t)r], ms[(size_t)r].ranks, ms[(size_t)r].rank,
               (unsigned long long)ms[(size_t)r].slices, (unsigned long long)ms[(size_t)r].reduce_bytes, ms[(size_t)r].algorithm, (unsigned long long)ms[(size_t)r].held_blocks,
               (unsigned long long)ms[(size_t)r].owned_blocks, (unsigned long long)ms[(size_t)r].wire_bytes_sent, (unsigned long long)ms[(size_t)r].wire_bytes_received);
    }
    printf("]}\n");
    for (int r = 0; r < world; ++r) { if (vols[(size_t)r]) op_volume_destroy(vols[(size_t)r]); destroy(comms[(size_t)r]); }
    return bad;
}
