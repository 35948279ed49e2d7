// merge_rccl.hip -- op_volume_merge_rccl: the frame-sharded multi-GPU merge as ONE library call.
//
// Distributed form of CubeHandler::Merge (reference: src/Integration/CubeHandler.h:145-167 -- key union + per-voxel
// weighted mean; like the reference's, it touches only blocks somebody HOLDS): every rank fused its own contiguous shard of
// the frames into a private volume with zero communication; this call merges all of them.  Two algorithms
// (op_runtime_set_option(OP_RUNTIME_OPT_MERGE_ALGORITHM, ...)):
//
// OP_MERGE_OWNER_EXCHANGE (default) -- every block key has an OWNER rank (a hash of the key mod the number of ranks):
//   1. a rank sorts its keys by owner and packs its blocks in that order in SUM form [w*sdf, w, w*c]        10 KiB / block held
//   2. ncclAllGather of the world x world matrix of counts, then ONE group of ncclSend / ncclRecv: every rank sends each
//      other rank the keys (8 B) and sum-form blocks it holds of that rank's partition and receives its own partition's --
//      (world - 1) / world of the blocks a rank HOLDS cross the wire, spread over its links to all peers at once (xGMI is
//      point to point: 7 links per GPU; a ring reduce keeps one of them busy with the whole union, zeros included)
//   3. the owner builds the sorted union of its partition (rocPRIM sort + unique), and adds the received blocks into it
//      source by source in rank order (deterministic; no atomics: a source holds a key once)
//   4. root >= 0: the owners send their summed partitions to the root in a second group (the root receives (world - 1) /
//      world of the UNION over all its links), which normalises them into its volume (k_unpack_sum);
//      root < 0: nobody gathers -- every rank's volume becomes its owned, merged partition of the map.
// OP_MERGE_DENSE_REDUCE (rounds 1-4, kept as the fallback) -- all-gather of the keys, the same sorted union on every rank,
//   every rank packs the WHOLE union (zeros where it holds nothing), one sliced ncclReduce(sum) to the root, normalise.
//
// Temporaries come from the library's buffer cache (a steady stream of merges allocates nothing); rank-local failures are
// agreed on over the communicator before EVERY bulk transfer (the exchange and the gather), so no rank is left waiting inside RCCL.
// Keys and weights are exact for any rank count; sdf / colour differ from a sequential Merge chain only in fp32
// summation order (<= 1e-6 relative).  RCCL is bound at run time (dlopen "librccl.so.1", or the library
// op_runtime_set_rccl_library named before the first merge): a host that never merges -- or a Python process whose torch
// already carries its own RCCL -- does not need it at link time.  Nothing here reads the environment.
// Threading: call from one host thread (or process) per rank, like any NCCL collective without group semantics.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

namespace {

using op::fail;

// which RCCL to bind (op_runtime_set_rccl_library; nothing comes from the environment); the other process-wide settings: op::runtime_options()
struct RcclChoice {
    std::mutex mu;
    std::string path;      // empty = the system's RCCL
    bool bound = false;
};
RcclChoice& rccl_choice() { static RcclChoice c; return c; }

struct Rccl {
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
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
        // op_runtime_set_rccl_library names the library to bind instead of the system's RCCL (a site-specific build; the test suite points it at
        // tests/cpp/librccl_double.so to run several ranks on ONE device, which the real RCCL refuses).  When set it is the only candidate.
        std::string forced;
        { std::lock_guard<std::mutex> lk(rccl_choice().mu); forced = rccl_choice().path; rccl_choice().bound = true; }
        if (!forced.empty()) h = dlopen(forced.c_str(), RTLD_NOW | RTLD_GLOBAL);
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
        t.Send = (decltype(t.Send))dlsym(h, "ncclSend");
        t.Recv = (decltype(t.Recv))dlsym(h, "ncclRecv");
        t.GroupStart = (decltype(t.GroupStart))dlsym(h, "ncclGroupStart");
        t.GroupEnd = (decltype(t.GroupEnd))dlsym(h, "ncclGroupEnd");
        t.ok = t.CommCount && t.CommUserRank && t.AllGather && t.AllReduce && t.Reduce && t.GetErrorString && t.Send && t.Recv && t.GroupStart && t.GroupEnd;
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


// ---- owner-partitioned exchange --------------------------------------------------------------------------------------------
// owner of a block: a finalising mix of its packed id, mod the number of ranks (the table's own hash keeps the low bits of the
// coordinates: eight ranks would each own whole 8 x 8 x 8-block lattices of it -- fine -- but three or five would not divide them)
__host__ __device__ inline unsigned owner_of(unsigned long long p, unsigned world) {
    p ^= p >> 33; p *= 0xff51afd7ed558ccdULL; p ^= p >> 33; p *= 0xc4ceb9fe1a85ec53ULL; p ^= p >> 33;
    return (unsigned)(p % world);
}
// int32x3 keys -> packed ids + their owners
__global__ void k_owner_keys(const int* __restrict__ keys, size_t n, unsigned world, unsigned long long* __restrict__ packed, unsigned* __restrict__ owner) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int* k = keys + 3 * i;
    const unsigned long long p = ((unsigned long long)(k[0] + kOff) << 42) | ((unsigned long long)(k[1] + kOff) << 21) | (unsigned long long)(k[2] + kOff);
    packed[i] = p; owner[i] = owner_of(p, world);
}
// first position of every owner in the owner-sorted array: bounds[r] = lower_bound(r), bounds[world] = n
__global__ void k_owner_bounds(const unsigned* __restrict__ sorted_owner, size_t n, unsigned world, int* __restrict__ bounds) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > world) return;
    size_t lo = 0, hi = n;
    while (lo < hi) { const size_t m = (lo + hi) / 2; if (sorted_owner[m] < r) lo = m + 1; else hi = m; }
    bounds[r] = (int)lo;
}
// position of every received key in the owner's sorted union
__global__ void k_find_in_union(const unsigned long long* __restrict__ keys, size_t n, const unsigned long long* __restrict__ uni, size_t n_uni, unsigned* __restrict__ idx) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    size_t lo = 0, hi = n_uni;
    while (lo < hi) { const size_t m = (lo + hi) / 2; if (uni[m] < k) lo = m + 1; else hi = m; }
    idx[i] = (unsigned)lo;
}
// acc[idx[b]] += part[b] for the blocks of ONE source rank (a source holds a key once: no two blocks of a launch meet)
__global__ __launch_bounds__(512) void k_accumulate_blocks(float* __restrict__ acc, const float* __restrict__ part, const unsigned* __restrict__ idx) {
    const size_t b = blockIdx.x;
    float* d = acc + (size_t)idx[b] * 2560;
    const float* q = part + b * 2560;
#pragma unroll
    for (int i = 0; i < 5; ++i) d[threadIdx.x + 512 * i] += q[threadIdx.x + 512 * i];
}
} // namespace

#ifndef OP_MERGE_SLICE_BLOCKS
#define OP_MERGE_SLICE_BLOCKS 32768 // union blocks per reduce slice (320 MB): large enough for the ring to reach its per-link rate, small enough to pipeline
#endif

static int merge_dense_reduce(op_volume* v, void* nccl_comm, int root, size_t* n_union_out, op_merge_stats* stats) {
    if (n_union_out) *n_union_out = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (!v || !nccl_comm) return fail(OP_ERR_INVALID, "null argument");
    if (!rccl().ok) return fail(OP_ERR_NO_DEVICE, "RCCL is not available (librccl.so.1: %s)", rccl().why);
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    int world = 0, rank = 0;
    if (rccl().CommCount(comm, &world) != ncclSuccess || rccl().CommUserRank(comm, &rank) != ncclSuccess || world < 1)
        return fail(OP_ERR_INVALID, "invalid RCCL communicator");
    if (root < 0 || root >= world) return fail(OP_ERR_INVALID, "root %d outside the communicator (%d ranks)", root, world);
    if (stats) { stats->ranks = world; stats->rank = rank; stats->algorithm = OP_MERGE_DENSE_REDUCE; }
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
    // one rank: nothing to merge.  (OP_RUNTIME_OPT_MERGE_FORCE_SINGLE_RANK runs the whole exchange anyway -- a one-rank all-gather and
    // reduce -- so that the RCCL path can be exercised on a single-GPU box; sdf / colour then pass through the sum form,
    // (w*s)/w, and may move by one rounding.)
    if (world == 1 && !op::runtime_options().merge_force_single_rank.load()) { if (rc == OP_OK && n_union_out) *n_union_out = n_local; if (stats) stats->union_blocks = n_local; return rc; }
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
    { const long long n = op::runtime_options().merge_slice_blocks.load(); if (n > 0) slice = (size_t)n; } // OP_RUNTIME_OPT_MERGE_SLICE_BLOCKS (tests: several slices on a small volume)
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
    if (stats) {
        stats->union_blocks = n_union; stats->reduce_bytes = n_union * 5 * 512 * sizeof(float); stats->prepare_ms = ms_since(t_begin);
        stats->held_blocks = n_local; stats->owned_blocks = rank == root ? n_union : 0;
        // a reduce moves the whole buffer out of every rank but the root and into every rank but the chain's first: counted once per rank, plus the key all-gather
        stats->wire_bytes_sent = (world > 1 && rank != root ? stats->reduce_bytes : 0) + (uint64_t)(world - 1) * mx * 12;
        stats->wire_bytes_received = (world > 1 ? stats->reduce_bytes : 0) + (uint64_t)(world - 1) * mx * 12;
    }
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

static int merge_owner_exchange(op_volume* v, void* nccl_comm, int root, size_t* n_union_out, op_merge_stats* stats) {
    if (n_union_out) *n_union_out = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (!v || !nccl_comm) return fail(OP_ERR_INVALID, "null argument");
    if (!rccl().ok) return fail(OP_ERR_NO_DEVICE, "RCCL is not available (librccl.so.1: %s)", rccl().why);
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    int world = 0, rank = 0;
    if (rccl().CommCount(comm, &world) != ncclSuccess || rccl().CommUserRank(comm, &rank) != ncclSuccess || world < 1)
        return fail(OP_ERR_INVALID, "invalid RCCL communicator");
    if (root >= world || world > 1023) return fail(OP_ERR_INVALID, "root %d outside the communicator (%d ranks; at most 1023)", root, world);
    if (stats) { stats->ranks = world; stats->rank = rank; stats->algorithm = OP_MERGE_OWNER_EXCHANGE; }
    size_t n_local = 0;
    int rc = op_volume_block_count(v, &n_local); // flushes queued frames, synchronises
    void* sv = nullptr;
    { // the stream is needed ALSO when the volume has failed: the failure is announced to the other ranks over it
        const int src = op_volume_stream(v, &sv);
        if (rc == OP_OK) rc = src;
    }
    hipStream_t stream = (hipStream_t)sv;
    hipDevice_t dev = 0;
    if (stream && hipStreamGetDevice(stream, &dev) != hipSuccess && rc == OP_OK) rc = fail(OP_ERR_HIP, "hipStreamGetDevice failed");
    if (stream && hipSetDevice((int)dev) != hipSuccess && rc == OP_OK) rc = fail(OP_ERR_HIP, "hipSetDevice failed");
    if (world == 1 && !op::runtime_options().merge_force_single_rank.load()) { // one rank: it owns everything it holds
        if (rc == OP_OK && n_union_out) *n_union_out = n_local;
        if (stats) { stats->union_blocks = n_local; stats->held_blocks = n_local; stats->owned_blocks = n_local; }
        return rc;
    }
    if (stream == nullptr) return rc;

    bool fatal = false;
    const unsigned uw = (unsigned)world;
    int* d_small = nullptr;      // [0 .. world): counts per destination (or -1: this rank has failed), [world .. world + world^2): everybody's, then 2 words for the agreements
    int *d_keys = nullptr, *d_bounds = nullptr, *d_sorted_keys = nullptr, *d_allkeys = nullptr;
    unsigned long long *d_pk = nullptr, *d_pk_sorted = nullptr, *d_rkeys = nullptr, *d_uni_sorted = nullptr, *d_own = nullptr, *d_gkeys = nullptr;
    unsigned *d_owner = nullptr, *d_owner_sorted = nullptr, *d_idx = nullptr, *d_nuniq = nullptr;
    void* d_tmp = nullptr;
    float *d_send = nullptr, *d_recv = nullptr, *d_acc = nullptr, *d_gather = nullptr;
    std::vector<int> matrix((size_t)world * world, 0), bounds((size_t)world + 1, 0), owned((size_t)world, 0);
    std::vector<size_t> roff((size_t)world + 1, 0); // where source s's blocks start in the receive buffers
    size_t n_recv = 0, n_own = 0, n_union = 0, tmp_a = 0, tmp_b = 0, tmp_c = 0;
    unsigned nuniq = 0;
    uint64_t sent = 0, received = 0;
    const size_t kBlk = 2560;    // floats per block in sum form
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    auto t_xfer = t_begin;
    // OP_RUNTIME_OPT_MERGE_FAULT (test hook): does `stage`'s allocation of THIS rank "fail"?
    auto injected = [&](int stage) { return op::runtime_options().merge_fault.load() == (long long)stage * 1024 + rank + 1; };

    // 1. this rank's keys, sorted by owner; its blocks packed in that order
    if (op::cached_malloc((void**)&d_small, ((size_t)world + (size_t)world * world + 4) * sizeof(int)) != hipSuccess) d_small = nullptr;
    if (!d_small) return rc != OP_OK ? rc : fail(OP_ERR_HIP, "no device memory for the merge's counters"); // nothing was communicated yet
    if (rc == OP_OK && n_local) {
        OP_LOCAL(op::cached_malloc((void**)&d_keys, n_local * 3 * sizeof(int)));
        OP_LOCAL(op::cached_malloc((void**)&d_pk, n_local * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_pk_sorted, n_local * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_owner, n_local * 4));
        OP_LOCAL(op::cached_malloc((void**)&d_owner_sorted, n_local * 4));
        OP_LOCAL(op::cached_malloc((void**)&d_sorted_keys, n_local * 3 * sizeof(int)));
        OP_LOCAL(op::cached_malloc((void**)&d_send, n_local * kBlk * sizeof(float)));
    }
    OP_LOCAL(op::cached_malloc((void**)&d_bounds, ((size_t)world + 1) * sizeof(int)));
    if (rc == OP_OK && n_local) {
        size_t got = 0;
        rc = op_volume_keys_device(v, d_keys, n_local, &got);
        if (rc == OP_OK) {
            hipLaunchKernelGGL(k_owner_keys, dim3((unsigned)((n_local + 255) / 256)), dim3(256), 0, stream, (const int*)d_keys, n_local, uw, d_pk, d_owner);
            OP_LOCAL(rocprim::radix_sort_pairs(nullptr, tmp_a, d_owner, d_owner_sorted, d_pk, d_pk_sorted, n_local, 0, 16, stream));
            OP_LOCAL(op::cached_malloc(&d_tmp, tmp_a));
            OP_LOCAL(rocprim::radix_sort_pairs(d_tmp, tmp_a, d_owner, d_owner_sorted, d_pk, d_pk_sorted, n_local, 0, 16, stream));
            if (rc == OP_OK) {
                hipLaunchKernelGGL(k_owner_bounds, dim3(1), dim3(1024), 0, stream, (const unsigned*)d_owner_sorted, n_local, uw, d_bounds);
                hipLaunchKernelGGL(k_unpack_keys, dim3((unsigned)((n_local + 255) / 256)), dim3(256), 0, stream, (const unsigned long long*)d_pk_sorted, n_local, d_sorted_keys);
                OP_LOCAL(hipMemcpyAsync(bounds.data(), d_bounds, ((size_t)world + 1) * sizeof(int), hipMemcpyDeviceToHost, stream));
                OP_LOCAL(hipStreamSynchronize(stream));
            }
            if (rc == OP_OK) rc = op_volume_pack_sum(v, d_sorted_keys, n_local, d_send); // on the volume's stream
        }
    }
    // 2. the matrix of counts (a rank that has failed announces -1 in every column and everybody leaves)
    {
        std::vector<int> mine((size_t)world, rc == OP_OK ? 0 : -1);
        if (rc == OP_OK) for (int d = 0; d < world; ++d) mine[(size_t)d] = bounds[(size_t)d + 1] - bounds[(size_t)d];
        if (hipMemcpyAsync(d_small, mine.data(), (size_t)world * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            if (rc == OP_OK) rc = fail(OP_ERR_HIP, "count upload failed"); fatal = true; goto done;
        }
        OP_NCCL(rccl().AllGather(d_small, d_small + world, (size_t)world, ncclInt32, comm, stream));
        if (hipMemcpyAsync(matrix.data(), d_small + world, (size_t)world * world * sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            rc = fail(OP_ERR_HIP, "reading the gathered block counts failed"); fatal = true; goto done;
        }
    }
    for (int r = 0; r < world; ++r)
        if (matrix[(size_t)r * world] < 0) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "rank %d entered the merge with a failed volume", r); goto done; } // every rank sees it: a common exit
    for (int s = 0; s < world; ++s) { roff[(size_t)s] = n_recv; n_recv += (size_t)matrix[(size_t)s * world + rank]; } // column `rank`: what every source holds of MY partition
    roff[(size_t)world] = n_recv;
    // 3. receive buffers, then the agreement: did every rank get its memory?
    if (n_recv) {
        OP_LOCAL(op::cached_malloc((void**)&d_rkeys, n_recv * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_recv, n_recv * kBlk * sizeof(float)));
        OP_LOCAL(op::cached_malloc((void**)&d_uni_sorted, n_recv * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_own, n_recv * 8));
        OP_LOCAL(op::cached_malloc((void**)&d_idx, n_recv * 4));
    }
    OP_LOCAL(op::cached_malloc((void**)&d_nuniq, sizeof(unsigned)));
    {
        int* d_flag = d_small + world + world * world;
        const int bad = rc != OP_OK;
        if (hipMemcpyAsync(d_flag, &bad, sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) { rc = fail(OP_ERR_HIP, "status upload failed"); fatal = true; goto done; }
        OP_NCCL(rccl().AllReduce(d_flag, d_flag, 1, ncclInt32, ncclMax, comm, stream));
        int any = 0;
        if (hipMemcpyAsync(&any, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            rc = fail(OP_ERR_HIP, "status download failed"); fatal = true; goto done;
        }
        if (any) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "another rank could not prepare its part of the merge"); goto done; }
    }
    if (stats) stats->prepare_ms = ms_since(t_begin);
    t_xfer = std::chrono::steady_clock::now();
    // 4. the exchange: keys and sum-form blocks to their owners, all pairs in one group (every link of the node at once)
    OP_NCCL(rccl().GroupStart());
    for (int p = 0; p < world; ++p) {
        if (p == rank) continue;
        const size_t ns = (size_t)(bounds[(size_t)p + 1] - bounds[(size_t)p]), nr = (size_t)matrix[(size_t)p * world + rank];
        if (ns) {
            OP_NCCL(rccl().Send(d_pk_sorted + bounds[(size_t)p], ns, ncclUint64, p, comm, stream));
            OP_NCCL(rccl().Send(d_send + (size_t)bounds[(size_t)p] * kBlk, ns * kBlk, ncclFloat32, p, comm, stream));
            sent += ns * (8 + kBlk * 4);
        }
        if (nr) {
            OP_NCCL(rccl().Recv(d_rkeys + roff[(size_t)p], nr, ncclUint64, p, comm, stream));
            OP_NCCL(rccl().Recv(d_recv + roff[(size_t)p] * kBlk, nr * kBlk, ncclFloat32, p, comm, stream));
            received += nr * (8 + kBlk * 4);
        }
    }
    OP_NCCL(rccl().GroupEnd());
    {   // this rank's own share of its partition stays on the device
        const size_t nself = (size_t)matrix[(size_t)rank * world + rank];
        if (nself) {
            if (hipMemcpyAsync(d_rkeys + roff[(size_t)rank], d_pk_sorted + bounds[(size_t)rank], nself * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                hipMemcpyAsync(d_recv + roff[(size_t)rank] * kBlk, d_send + (size_t)bounds[(size_t)rank] * kBlk, nself * kBlk * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) {
                rc = fail(OP_ERR_HIP, "copying the rank's own partition failed"); fatal = true; goto done;
            }
        }
    }
    // 5. the owner's sorted union, and the sum over the sources in rank order.  A failure here (the sort, the sort's scratch, the partition's
    //    sums -- n_own x 10 KiB, only known now) is rank-LOCAL: it is announced in step 6's all-gather (-1), so that all ranks leave together.
    if (n_recv) {
        OP_LOCAL(rocprim::radix_sort_keys(nullptr, tmp_b, d_rkeys, d_uni_sorted, n_recv, 0, 64, stream));
        OP_LOCAL(rocprim::unique(nullptr, tmp_c, d_uni_sorted, d_own, d_nuniq, n_recv, rocprim::equal_to<unsigned long long>(), stream));
        if (rc == OP_OK && d_tmp && (tmp_b > tmp_a || tmp_c > tmp_a)) { (void)hipStreamSynchronize(stream); op::cached_free(d_tmp); d_tmp = nullptr; }
        if (rc == OP_OK && !d_tmp) OP_LOCAL(op::cached_malloc(&d_tmp, std::max(tmp_a, std::max(tmp_b, tmp_c))));
        OP_LOCAL(rocprim::radix_sort_keys(d_tmp, tmp_b, d_rkeys, d_uni_sorted, n_recv, 0, 64, stream));
        OP_LOCAL(rocprim::unique(d_tmp, tmp_c, d_uni_sorted, d_own, d_nuniq, n_recv, rocprim::equal_to<unsigned long long>(), stream));
        OP_LOCAL(hipMemcpyAsync(&nuniq, d_nuniq, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        OP_LOCAL(hipStreamSynchronize(stream));
        if (rc == OP_OK) {
            n_own = nuniq;
            if (injected(1)) rc = fail(OP_ERR_HIP, "no memory for the partition's sums (injected)");
            OP_LOCAL(op::cached_malloc((void**)&d_acc, n_own * kBlk * sizeof(float)));
            OP_LOCAL(hipMemsetAsync(d_acc, 0, n_own * kBlk * sizeof(float), stream));
        }
        if (rc == OP_OK) {
            hipLaunchKernelGGL(k_find_in_union, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, stream, (const unsigned long long*)d_rkeys, n_recv, (const unsigned long long*)d_own, n_own, d_idx);
            for (int s = 0; s < world; ++s) {
                const size_t ns = roff[(size_t)s + 1] - roff[(size_t)s];
                if (ns) hipLaunchKernelGGL(k_accumulate_blocks, dim3((unsigned)ns), dim3(512), 0, stream, d_acc, (const float*)(d_recv + roff[(size_t)s] * kBlk), (const unsigned*)(d_idx + roff[(size_t)s]));
            }
        }
    }
    // 6. sizes of the partitions (a rank whose step 5 failed announces -1: every rank sees it and leaves -- a common exit, like step 2's)
    {
        const int mine = rc == OP_OK ? (int)n_own : -1;
        if (hipMemcpyAsync(d_small, &mine, sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "count upload failed"); fatal = true; goto done; }
        OP_NCCL(rccl().AllGather(d_small, d_small + world, 1, ncclInt32, comm, stream));
        if (hipMemcpyAsync(owned.data(), d_small + world, (size_t)world * sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            rc = fail(OP_ERR_HIP, "reading the partition sizes failed"); fatal = true; goto done;
        }
        for (int r = 0; r < world; ++r) {
            if (owned[(size_t)r] < 0) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "rank %d could not sum its partition of the merge", r); goto done; }
            n_union += (size_t)owned[(size_t)r];
        }
    }
    if (root < 0) {
        // 7a. no gather: this rank's volume becomes its owned, merged partition (no collective follows: what fails from here on is this rank's own error)
        if (n_own) {
            if (op::cached_malloc((void**)&d_allkeys, n_own * 3 * sizeof(int)) != hipSuccess) { rc = fail(OP_ERR_HIP, "no memory for the partition's keys"); goto done; }
            hipLaunchKernelGGL(k_unpack_keys, dim3((unsigned)((n_own + 255) / 256)), dim3(256), 0, stream, (const unsigned long long*)d_own, n_own, d_allkeys);
            rc = op_volume_unpack_sum_begin(v, d_allkeys, n_own);
            if (rc == OP_OK) rc = op_volume_unpack_sum_chunk(v, 0, n_own, d_acc);
        } else
            rc = op_volume_clear(v);
    } else {
        // 7b. the owners send their summed partitions to the root, which normalises the whole map into its volume.  The root's gather buffers
        //     (n_union x 10 KiB: the largest allocation of the merge) come first, and ONE more agreement: if the root has no room for them,
        //     no owner enters the group -- everybody returns the error instead of waiting in ncclSend.
        std::vector<size_t> goff((size_t)world + 1, 0);
        for (int r = 0; r < world; ++r) goff[(size_t)r + 1] = goff[(size_t)r] + (size_t)owned[(size_t)r];
        if (rank == root && n_union) {
            if (injected(2)) rc = fail(OP_ERR_HIP, "no memory for the gathered map (injected)");
            OP_LOCAL(op::cached_malloc((void**)&d_gkeys, n_union * 8));
            OP_LOCAL(op::cached_malloc((void**)&d_gather, n_union * kBlk * sizeof(float)));
            OP_LOCAL(op::cached_malloc((void**)&d_allkeys, n_union * 3 * sizeof(int)));
        }
        {
            int* d_flag = d_small + world + world * world + 1;
            const int bad = rc != OP_OK;
            if (hipMemcpyAsync(d_flag, &bad, sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) { rc = fail(OP_ERR_HIP, "status upload failed"); fatal = true; goto done; }
            OP_NCCL(rccl().AllReduce(d_flag, d_flag, 1, ncclInt32, ncclMax, comm, stream));
            int any = 0;
            if (hipMemcpyAsync(&any, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
                rc = fail(OP_ERR_HIP, "status download failed"); fatal = true; goto done;
            }
            if (any) { if (rc == OP_OK) rc = fail(OP_ERR_HIP, "the root could not allocate the gathered map"); goto done; }
        }
        OP_NCCL(rccl().GroupStart());
        if (rank == root) {
            for (int r = 0; r < world; ++r) {
                const size_t nr = (size_t)owned[(size_t)r];
                if (r == rank || !nr) continue;
                OP_NCCL(rccl().Recv(d_gkeys + goff[(size_t)r], nr, ncclUint64, r, comm, stream));
                OP_NCCL(rccl().Recv(d_gather + goff[(size_t)r] * kBlk, nr * kBlk, ncclFloat32, r, comm, stream));
                received += nr * (8 + kBlk * 4);
            }
        } else if (n_own) {
            OP_NCCL(rccl().Send(d_own, n_own, ncclUint64, root, comm, stream));
            OP_NCCL(rccl().Send(d_acc, n_own * kBlk, ncclFloat32, root, comm, stream));
            sent += n_own * (8 + kBlk * 4);
        }
        OP_NCCL(rccl().GroupEnd());
        if (rank == root && n_union) {
            if (n_own && (hipMemcpyAsync(d_gkeys + goff[(size_t)rank], d_own, n_own * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                          hipMemcpyAsync(d_gather + goff[(size_t)rank] * kBlk, d_acc, n_own * kBlk * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)) { rc = fail(OP_ERR_HIP, "copy failed"); goto done; }
            hipLaunchKernelGGL(k_unpack_keys, dim3((unsigned)((n_union + 255) / 256)), dim3(256), 0, stream, (const unsigned long long*)d_gkeys, n_union, d_allkeys);
            rc = op_volume_unpack_sum_begin(v, d_allkeys, n_union);
            if (rc == OP_OK) rc = op_volume_unpack_sum_chunk(v, 0, n_union, d_gather);
        } else if (rank == root)
            rc = op_volume_clear(v); // nobody holds anything
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == OP_OK) rc = fail(OP_ERR_HIP, "the merge did not complete");
    if (stats) {
        stats->union_blocks = n_union; stats->held_blocks = n_local; stats->owned_blocks = n_own; stats->wire_bytes_sent = sent; stats->wire_bytes_received = received;
        stats->reduce_bytes = sent; stats->slices = 1; stats->transfer_ms = ms_since(t_xfer);
    }
done:
    if (fatal) (void)hipDeviceSynchronize(); // whatever was enqueued must not outlive the buffers
    else (void)hipStreamSynchronize(stream);
    for (void* p : {(void*)d_small, (void*)d_keys, (void*)d_bounds, (void*)d_sorted_keys, (void*)d_allkeys, (void*)d_pk, (void*)d_pk_sorted, (void*)d_rkeys, (void*)d_uni_sorted, (void*)d_own,
                    (void*)d_gkeys, (void*)d_owner, (void*)d_owner_sorted, (void*)d_idx, (void*)d_nuniq, d_tmp, (void*)d_send, (void*)d_recv, (void*)d_acc, (void*)d_gather})
        if (p) op::cached_free(p);
    if (stats) stats->total_ms = ms_since(t_begin);
    if (rc == OP_OK && n_union_out) *n_union_out = n_union;
    return rc;
}

extern "C" int op_volume_merge_rccl_stats(op_volume* v, void* nccl_comm, int root, size_t* n_union_out, op_merge_stats* stats) {
    if (op::runtime_options().merge_algorithm.load() == OP_MERGE_DENSE_REDUCE) {
        if (root < 0) return fail(OP_ERR_INVALID, "the dense reduce needs a root (root = -1, no gather, is the owner exchange's)");
        return merge_dense_reduce(v, nccl_comm, root, n_union_out, stats);
    }
    return merge_owner_exchange(v, nccl_comm, root, n_union_out, stats);
}

extern "C" int op_runtime_set_rccl_library(const char* path) {
    std::lock_guard<std::mutex> lk(rccl_choice().mu);
    if (rccl_choice().bound) return fail(OP_ERR_INVALID, "op_runtime_set_rccl_library: RCCL has already been bound by an earlier merge");
    rccl_choice().path = path ? path : "";
    return OP_OK;
}

extern "C" int op_volume_merge_rccl(op_volume* v, void* nccl_comm, int root, size_t* n_union_out) {
    return op_volume_merge_rccl_stats(v, nccl_comm, root, n_union_out, nullptr);
}
