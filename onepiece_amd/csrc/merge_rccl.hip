// merge_rccl.hip -- op_volume_merge_rccl: the frame-sharded multi-GPU merge as ONE library call.
//
// Distributed form of CubeHandler::Merge (reference: src/Integration/CubeHandler.h:145-167 -- key union + per-voxel
// weighted mean): every rank fused its own contiguous shard of the frames into a private volume with zero
// communication; this call merges all of them into `root`'s volume:
//   1. ncclAllGather of the per-rank block counts, then of the (padded) int32x3 key arrays        12 B / block
//   2. every rank builds the SAME sorted union: keys packed into one u64 (3 x 21 bits, the packing of the device
//      hash table), radix-sorted and made unique on the device (rocPRIM; a plain library sort of <= ~1e6 keys)
//   3. k_pack_sum writes the rank's blocks in union order in SUM form [w*sdf, w, w*c]             10 KiB / block
//   4. ncclReduce(sum, float32) to the root -- the only bulk transfer; over xGMI's point-to-point links RCCL's ring is
//      per-link bound, so it is issued in few large slices (32 768 blocks = 320 MB) on a communication stream: slice i + 1
//      is packed while slice i is on the wire
//   5. the root normalises back to mean form (k_unpack_sum), slice by slice behind the reduces still in flight.
// Temporaries come from the library's buffer cache (a steady stream of merges allocates nothing); rank-local failures are
// agreed on over the communicator before the bulk transfer, so no rank is left waiting inside RCCL.
// Keys and weights are exact for any rank count; sdf / colour differ from a sequential Merge chain only in fp32
// summation order (<= 1e-6 relative).  RCCL is bound at run time (dlopen "librccl.so.1", or the library ONEPIECE_RCCL_LIBRARY names): a host
// that never merges -- or a Python process whose torch already carries its own RCCL -- does not need it at link time.
// Threading: call from one host thread (or process) per rank, like any NCCL collective without group semantics.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

namespace {

using op::fail;

struct Rccl {
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    char why[256] = "missing symbols";   // captured ONCE, where dlopen failed (dlerror() clears itself when read)
};

const Rccl& rccl() {
    static Rccl r = [] {
        Rccl t;
        void* h = nullptr;
        // ONEPIECE_RCCL_LIBRARY names the library to bind instead of the system's RCCL (a site-specific build; the test suite points it at
        // tests/cpp/librccl_double.so to run several ranks on ONE device, which the real RCCL refuses).  When set it is the only candidate.
        const char* forced = std::getenv("ONEPIECE_RCCL_LIBRARY");
        if (forced && *forced) h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        else
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) {
            const char* e = dlerror();
            std::snprintf(t.why, sizeof(t.why), "%s", e ? e : "dlopen failed");
            return t;
        }
        t.CommCount = (decltype(t.CommCount))dlsym(h, "ncclCommCount");
        t.CommUserRank = (decltype(t.CommUserRank))dlsym(h, "ncclCommUserRank");
        t.AllGather = (decltype(t.AllGather))dlsym(h, "ncclAllGather");
        t.AllReduce = (decltype(t.AllReduce))dlsym(h, "ncclAllReduce");
        t.Reduce = (decltype(t.Reduce))dlsym(h, "ncclReduce");
        t.GetErrorString = (decltype(t.GetErrorString))dlsym(h, "ncclGetErrorString");
        t.ok = t.CommCount && t.CommUserRank && t.AllGather && t.AllReduce && t.Reduce && t.GetErrorString;
        return t;
    }();
    return r;
}

// A failing collective leaves the communicator in an undefined state: nothing can be agreed over it any more (`fatal`).
#define OP_NCCL(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) { rc = fail(OP_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_)); fatal = true; goto done; } \
    } while (0)
// A rank-LOCAL failure (allocation, a kernel of this volume): recorded, and the rank keeps taking part in the collectives
// with empty payloads until the next agreement point, where all ranks leave together -- nobody is left waiting in RCCL.
#define OP_LOCAL(expr)                                                                                    \
    do {                                                                                                  \
        if (rc == OP_OK) { hipError_t e_ = (expr); if (e_ != hipSuccess) rc = fail(OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)

constexpr int kOff = 1 << 20; // block coordinates are within +-2^20 (the device hash key's 21-bit fields)

// gathered keys (world x mx x 3, the first counts[r] rows of rank r valid) -> packed u64; padding rows -> ~0 (sorts last)
__global__ void k_pack_keys(const int* __restrict__ keys, const int* __restrict__ counts, int world, size_t mx, unsigned long long* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)world * mx) return;
    const size_t r = i / mx, j = i - r * mx;
    unsigned long long p = ~0ull;
    if (j < (size_t)counts[r]) {
        const int* k = keys + 3 * i;
        p = ((unsigned long long)(k[0] + kOff) << 42) | ((unsigned long long)(k[1] + kOff) << 21) | (unsigned long long)(k[2] + kOff);
    }
    out[i] = p;
}
__global__ void k_unpack_keys(const unsigned long long* __restrict__ packed, size_t n, int* __restrict__ keys) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long p = packed[i];
    keys[3 * i] = (int)(p >> 42) - kOff;
    keys[3 * i + 1] = (int)((p >> 21) & 0x1FFFFFull) - kOff;
    keys[3 * i + 2] = (int)(p & 0x1FFFFFull) - kOff;
}

} // namespace

#ifndef OP_MERGE_SLICE_BLOCKS
#define OP_MERGE_SLICE_BLOCKS 32768 // union blocks per reduce slice (320 MB): large enough for the ring to reach its per-link rate, small enough to pipeline
#endif

extern "C" int op_volume_merge_rccl_stats(op_volume* v, void* nccl_comm, int root, size_t* n_union_out, op_merge_stats* stats) {
    if (n_union_out) *n_union_out = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (!v || !nccl_comm) return fail(OP_ERR_INVALID, "null argument");
    if (!rccl().ok) return fail(OP_ERR_NO_DEVICE, "RCCL is not available (librccl.so.1: %s)", rccl().why);
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    int world = 0, rank = 0;
    if (rccl().CommCount(comm, &world) != ncclSuccess || rccl().CommUserRank(comm, &rank) != ncclSuccess || world < 1)
        return fail(OP_ERR_INVALID, "invalid RCCL communicator");
    if (root < 0 || root >= world) return fail(OP_ERR_INVALID, "root %d outside the communicator (%d ranks)", root, world);
    if (stats) { stats->ranks = world; stats->rank = rank; }
    size_t n_local = 0;
    int rc = op_volume_block_count(v, &n_local); // flushes queued frames, synchronises, selects nothing yet
    void* sv = nullptr;
    { // the stream is needed ALSO when the volume has failed: the failure is announced to the other ranks over it (counts all-gather below)
        const int src = op_volume_stream(v, &sv);
        if (rc == OP_OK) rc = src;
    }
    hipStream_t stream = (hipStream_t)sv, cstream = nullptr;
    hipDevice_t dev = 0;
    // (also for a volume that has failed: its announcement needs a few bytes of memory on ITS device)
    if (stream && hipStreamGetDevice(stream, &dev) != hipSuccess && rc == OP_OK) rc = fail(OP_ERR_HIP, "hipStreamGetDevice failed");
    if (stream && hipSetDevice((int)dev) != hipSuccess && rc == OP_OK) rc = fail(OP_ERR_HIP, "hipSetDevice failed");
    // one rank: nothing to merge.  (ONEPIECE_RCCL_FORCE=1 runs the whole exchange anyway -- a one-rank all-gather and
    // reduce -- so that the RCCL path can be exercised on a single-GPU box; sdf / colour then pass through the sum form,
    // (w*s)/w, and may move by one rounding.)
    if (world == 1 && !std::getenv("ONEPIECE_RCCL_FORCE")) { if (rc == OP_OK && n_union_out) *n_union_out = n_local; if (stats) stats->union_blocks = n_local; return rc; }
    if (stream == nullptr) return rc; // not even a stream to run the agreement on: a broken volume handle

    bool fatal = false;
    int* d_small = nullptr;    // [0] my count or -1 (this rank has failed), [1 .. world] everybody's, [world + 1] agreement flag
    int *d_keys = nullptr, *d_all = nullptr, *d_union = nullptr;
    unsigned long long *d_pk = nullptr, *d_sorted = nullptr, *d_uniq = nullptr;
    unsigned* d_nuniq = nullptr;
    void* d_tmp = nullptr;
    float* d_buf = nullptr;
    std::vector<int> counts(world + 2);
    std::vector<hipEvent_t> packed, reduced;
    size_t mx = 1, total = 0, n_union = 0, tmp_a = 0, tmp_b = 0, n_slices = 0;
    unsigned nuniq = 0;
    size_t slice = OP_MERGE_SLICE_BLOCKS;
    if (const char* e = std::getenv("ONEPIECE_MERGE_SLICE_BLOCKS")) { const long n = std::atol(e); if (n > 0) slice = (size_t)n; } // test hook: several slices on a small volume
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };

    // 1. counts (a rank that has already failed announces -1 and everybody leaves), then padded keys
    // (unconditionally -- not OP_LOCAL, which does nothing once rc is set: a rank that ENTERS with a failed volume must still announce it)
    if (op::cached_malloc((void**)&d_small, (size_t)(world + 2) * sizeof(int)) != hipSuccess) d_small = nullptr;
    if (!d_small) return rc != OP_OK ? rc : fail(OP_ERR_HIP, "no device memory for the merge's counters"); // nothing was communicated yet, but without device memory this rank cannot say so: the caller must abort the communicator
    {
        const int n_mine = rc == OP_OK ? (int)n_local : -1;
        if (hipMemcpyAsync(d_small, &n_mine, sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "count upload failed"); fatal = true; goto done; }
        OP_NCCL(rccl().AllGather(d_small, d_small + 1, 1, ncclInt32, comm, stream));
        if (hipMemcpyAsync(counts.data(), d_small + 1, world * sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            rc = fail(OP_ERR_HIP, "reading the gathered block counts failed"); fatal = true; goto done;
        }
    }
    for (int r = 0; r < world; ++r) {
        if (counts[r] < 0) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "rank %d entered the merge with a failed volume", r); goto done; } // every rank sees it: a common exit
        if ((size_t)counts[r] > mx) mx = (size_t)counts[r];
        total += (size_t)counts[r];
    }
    if (total == 0) goto done;
    OP_LOCAL(op::cached_malloc((void**)&d_keys, mx * 3 * sizeof(int)));
    OP_LOCAL(op::cached_malloc((void**)&d_all, (size_t)world * mx * 3 * sizeof(int)));
    if (rc == OP_OK) {
        OP_LOCAL(hipMemsetAsync(d_keys, 0, mx * 3 * sizeof(int), stream));
        size_t got = 0;
        if (rc == OP_OK) rc = op_volume_keys_device(v, d_keys, mx, &got);
    }
    // agreement: did every rank get its buffers?  (all ranks take part, whatever happened locally)
    {
        const int bad = rc != OP_OK;
        if (hipMemcpyAsync(d_small + world + 1, &bad, sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) { rc = fail(OP_ERR_HIP, "status upload failed"); fatal = true; goto done; }
        OP_NCCL(rccl().AllReduce(d_small + world + 1, d_small + world + 1, 1, ncclInt32, ncclMax, comm, stream));
        int any = 0;
        if (hipMemcpyAsync(&any, d_small + world + 1, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            rc = fail(OP_ERR_HIP, "status download failed"); fatal = true; goto done;
        }
        if (any) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "another rank could not allocate its merge buffers"); goto done; }
    }
    OP_NCCL(rccl().AllGather(d_keys, d_all, mx * 3, ncclInt32, comm, stream));
    // 2. identical sorted union on every rank
    {
        const size_t n_all = (size_t)world * mx;
        OP_LOCAL(op::cached_malloc((void**)&d_pk, n_all * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_sorted, n_all * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_uniq, n_all * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_nuniq, sizeof(unsigned)));
        if (rc == OP_OK) {
            hipLaunchKernelGGL(k_pack_keys, dim3((unsigned)((n_all + 255) / 256)), dim3(256), 0, stream, (const int*)d_all, (const int*)(d_small + 1), world, mx, d_pk);
            OP_LOCAL(rocprim::radix_sort_keys(nullptr, tmp_a, d_pk, d_sorted, n_all, 0, 64, stream));
            OP_LOCAL(rocprim::unique(nullptr, tmp_b, d_sorted, d_uniq, d_nuniq, n_all, rocprim::equal_to<unsigned long long>(), stream));
            OP_LOCAL(op::cached_malloc(&d_tmp, tmp_a > tmp_b ? tmp_a : tmp_b));
            OP_LOCAL(rocprim::radix_sort_keys(d_tmp, tmp_a, d_pk, d_sorted, n_all, 0, 64, stream));
            OP_LOCAL(rocprim::unique(d_tmp, tmp_b, d_sorted, d_uniq, d_nuniq, n_all, rocprim::equal_to<unsigned long long>(), stream));
            OP_LOCAL(hipMemcpyAsync(&nuniq, d_nuniq, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            OP_LOCAL(hipStreamSynchronize(stream));
        }
        // every rank sorted the same gathered array: n_union is the same everywhere (a rank whose sort failed computes it from the counts'
        // upper bound instead and only sends zeros, see below)
        n_union = rc == OP_OK ? (size_t)nuniq - (total < n_all ? 1 : 0) : 0; // the padding value ~0 is the last unique entry
        if (rc == OP_OK && n_union) {
            OP_LOCAL(op::cached_malloc((void**)&d_union, n_union * 3 * sizeof(int)));
            if (rc == OP_OK) hipLaunchKernelGGL(k_unpack_keys, dim3((unsigned)((n_union + 255) / 256)), dim3(256), 0, stream, (const unsigned long long*)d_uniq, n_union, d_union);
        }
        if (rc == OP_OK && n_union) OP_LOCAL(op::cached_malloc((void**)&d_buf, n_union * 5 * 512 * sizeof(float)));
        if (rc == OP_OK) OP_LOCAL(op::cached_stream(&cstream));
    }
    // agreement before the bulk transfer: the union size every rank will reduce (0 = somebody failed: nobody reduces)
    {
        long long mine = rc == OP_OK ? (long long)n_union : -1, lo = 0;
        long long* d_agree = (long long*)d_small; // the counts (world + 2 >= 2 ints = 8 bytes) have been consumed: k_pack_keys ran before the sort's synchronisation
        if (hipMemcpyAsync(d_agree, &mine, sizeof(mine), hipMemcpyHostToDevice, stream) != hipSuccess) { rc = fail(OP_ERR_HIP, "status upload failed"); fatal = true; goto done; }
        OP_NCCL(rccl().AllReduce(d_agree, d_agree, 1, ncclInt64, ncclMin, comm, stream));
        if (hipMemcpyAsync(&lo, d_agree, sizeof(lo), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            rc = fail(OP_ERR_HIP, "status download failed"); fatal = true; goto done;
        }
        if (lo < 0) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "another rank failed while building the union"); goto done; }
    }
    if (n_union == 0) goto done;
    if (stats) { stats->union_blocks = n_union; stats->reduce_bytes = n_union * 5 * 512 * sizeof(float); stats->prepare_ms = ms_since(t_begin); }
    // 3.-5. in slices: pack slice i in sum form on the volume's stream, reduce it on the communication stream as soon as it is packed
    // (slice i + 1 is packed while slice i is on the wire), and -- on the root, once ALL of its own slices are packed, because its
    // volume is source and destination -- normalise slice i while later ones are still reducing.
    {
        const auto t_xfer = std::chrono::steady_clock::now();
        n_slices = (n_union + slice - 1) / slice;
        packed.assign(n_slices, nullptr); reduced.assign(n_slices, nullptr);
        for (size_t i = 0; i < n_slices && rc == OP_OK; ++i) { OP_LOCAL(op::cached_event(&packed[i])); OP_LOCAL(op::cached_event(&reduced[i])); }
        if (rc != OP_OK) { fatal = true; goto done; } // events are tiny: failing here, after the agreement, leaves the other ranks in the reduce
        for (size_t i = 0; i < n_slices; ++i) {
            const size_t lo = i * slice, cnt = std::min(slice, n_union - lo);
            float* part = d_buf + lo * 5 * 512;
            const int prc = op_volume_pack_sum(v, d_union + 3 * lo, cnt, part); // enqueues on the volume's stream
            if (prc != OP_OK) { rc = prc; fatal = true; goto done; }
            if (hipEventRecord(packed[i], stream) != hipSuccess || hipStreamWaitEvent(cstream, packed[i], 0) != hipSuccess) { rc = fail(OP_ERR_HIP, "event failed"); fatal = true; goto done; }
            OP_NCCL(rccl().Reduce(part, part, cnt * 5 * 512, ncclFloat32, ncclSum, root, comm, cstream));
            if (hipEventRecord(reduced[i], cstream) != hipSuccess) { rc = fail(OP_ERR_HIP, "event failed"); fatal = true; goto done; }
        }
        if (rank == root) {
            rc = op_volume_unpack_sum_begin(v, d_union, n_union); // waits for the packs, then clears and enters the union
            for (size_t i = 0; i < n_slices && rc == OP_OK; ++i) {
                const size_t lo = i * slice, cnt = std::min(slice, n_union - lo);
                if (hipStreamWaitEvent(stream, reduced[i], 0) != hipSuccess) { rc = fail(OP_ERR_HIP, "event failed"); break; }
                rc = op_volume_unpack_sum_chunk(v, lo, cnt, d_buf + lo * 5 * 512);
            }
        }
        if (hipStreamSynchronize(cstream) != hipSuccess && rc == OP_OK) rc = fail(OP_ERR_HIP, "the reduce did not complete");
        if (hipStreamSynchronize(stream) != hipSuccess && rc == OP_OK) rc = fail(OP_ERR_HIP, "the merge kernels did not complete");
        if (stats) { stats->slices = n_slices; stats->transfer_ms = ms_since(t_xfer); }
    }
done:
    if (fatal) (void)hipDeviceSynchronize(); // whatever was enqueued must not outlive the buffers
    else { (void)hipStreamSynchronize(stream); if (cstream) (void)hipStreamSynchronize(cstream); }
    for (auto e : packed) if (e) op::release_event(e, (int)dev);
    for (auto e : reduced) if (e) op::release_event(e, (int)dev);
    if (cstream) op::release_stream(cstream, (int)dev);
    for (void* p : {(void*)d_small, (void*)d_keys, (void*)d_all, (void*)d_union, (void*)d_pk, (void*)d_sorted, (void*)d_uniq, (void*)d_nuniq, d_tmp, (void*)d_buf})
        if (p) op::cached_free(p);
    if (stats) stats->total_ms = ms_since(t_begin);
    if (rc == OP_OK && n_union_out) *n_union_out = n_union;
    return rc;
}

extern "C" int op_volume_merge_rccl(op_volume* v, void* nccl_comm, int root, size_t* n_union_out) {
    return op_volume_merge_rccl_stats(v, nccl_comm, root, n_union_out, nullptr);
}
