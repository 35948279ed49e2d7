// merge_rccl.hip -- op_volume_merge_rccl: the frame-sharded multi-GPU merge as ONE library call.
//
// Distributed form of CubeHandler::Merge (reference: src/Integration/CubeHandler.h:145-167 -- key union + per-voxel
// weighted mean): every rank fused its own contiguous shard of the frames into a private volume with zero
// communication; this call merges all of them into `root`'s volume:
//   1. ncclAllGather of the per-rank block counts, then of the (padded) int32x3 key arrays        12 B / block
//   2. every rank builds the SAME sorted union: keys packed into one u64 (3 x 21 bits, the packing of the device
//      hash table), radix-sorted and made unique on the device (rocPRIM; a plain library sort of <= ~1e6 keys)
//   3. k_pack_sum writes the rank's blocks in union order in SUM form [w*sdf, w, w*c]             10 KiB / block
//   4. ONE ncclReduce(sum, float32) to the root -- the only bulk transfer; over xGMI's point-to-point links RCCL's
//      ring is per-link bound, so one large reduce is the right shape
//   5. the root normalises back to mean form (k_unpack_sum).
// Keys and weights are exact for any rank count; sdf / colour differ from a sequential Merge chain only in fp32
// summation order (<= 1e-6 relative).  RCCL is bound at run time (dlopen "librccl.so.1"): a host that never merges --
// or a Python process whose torch already carries its own RCCL -- does not need it at link time.
// Threading: call from one host thread (or process) per rank, like any NCCL collective without group semantics.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

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
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static Rccl r = [] {
        Rccl t;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) return t;
        t.CommCount = (decltype(t.CommCount))dlsym(h, "ncclCommCount");
        t.CommUserRank = (decltype(t.CommUserRank))dlsym(h, "ncclCommUserRank");
        t.AllGather = (decltype(t.AllGather))dlsym(h, "ncclAllGather");
        t.Reduce = (decltype(t.Reduce))dlsym(h, "ncclReduce");
        t.GetErrorString = (decltype(t.GetErrorString))dlsym(h, "ncclGetErrorString");
        t.ok = t.CommCount && t.CommUserRank && t.AllGather && t.Reduce && t.GetErrorString;
        return t;
    }();
    return r;
}

#define OP_NCCL(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) { rc = fail(OP_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_)); goto done; } \
    } while (0)
#define OP_HIPG(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) { rc = fail(OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } \
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

extern "C" int op_volume_merge_rccl(op_volume* v, void* nccl_comm, int root, size_t* n_union_out) {
    if (n_union_out) *n_union_out = 0;
    if (!v || !nccl_comm) return fail(OP_ERR_INVALID, "null argument");
    if (!rccl().ok) return fail(OP_ERR_NO_DEVICE, "RCCL is not available (dlopen librccl.so.1 failed: %s)", dlerror() ? dlerror() : "missing symbols");
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    int world = 0, rank = 0;
    if (rccl().CommCount(comm, &world) != ncclSuccess || rccl().CommUserRank(comm, &rank) != ncclSuccess || world < 1)
        return fail(OP_ERR_INVALID, "invalid RCCL communicator");
    if (root < 0 || root >= world) return fail(OP_ERR_INVALID, "root %d outside the communicator (%d ranks)", root, world);
    size_t n_local = 0;
    OP_TRY(op_volume_block_count(v, &n_local)); // flushes queued frames, synchronises, selects nothing yet
    void* sv = nullptr;
    OP_TRY(op_volume_stream(v, &sv));
    hipStream_t stream = (hipStream_t)sv;
    hipDevice_t dev = 0;
    OP_HIP(hipStreamGetDevice(stream, &dev));
    OP_HIP(hipSetDevice((int)dev));
    // one rank: nothing to merge.  (ONEPIECE_RCCL_FORCE=1 runs the whole exchange anyway -- a one-rank all-gather and
    // reduce -- so that the RCCL path can be exercised on a single-GPU box; sdf / colour then pass through the sum form,
    // (w*s)/w, and may move by one rounding.)
    if (world == 1 && !std::getenv("ONEPIECE_RCCL_FORCE")) { if (n_union_out) *n_union_out = n_local; return OP_OK; }

    int rc = OP_OK;
    int *d_cnt = nullptr, *d_cnts = nullptr, *d_keys = nullptr, *d_all = nullptr, *d_union = nullptr;
    unsigned long long *d_pk = nullptr, *d_sorted = nullptr, *d_uniq = nullptr;
    unsigned* d_nuniq = nullptr;
    void* d_tmp = nullptr;
    float* d_buf = nullptr;
    std::vector<int> counts(world);
    size_t mx = 1, total = 0, n_union = 0, tmp_a = 0, tmp_b = 0;
    unsigned nuniq = 0;
    const int n_mine = (int)n_local;

    // 1. counts, then padded keys
    OP_HIPG(hipMalloc((void**)&d_cnt, sizeof(int)));
    OP_HIPG(hipMalloc((void**)&d_cnts, world * sizeof(int)));
    OP_HIPG(hipMemcpyAsync(d_cnt, &n_mine, sizeof(int), hipMemcpyHostToDevice, stream));
    OP_NCCL(rccl().AllGather(d_cnt, d_cnts, 1, ncclInt32, comm, stream));
    OP_HIPG(hipMemcpyAsync(counts.data(), d_cnts, world * sizeof(int), hipMemcpyDeviceToHost, stream));
    OP_HIPG(hipStreamSynchronize(stream));
    for (int r = 0; r < world; ++r) { if ((size_t)counts[r] > mx) mx = (size_t)counts[r]; total += (size_t)counts[r]; }
    if (total == 0) goto done;
    OP_HIPG(hipMalloc((void**)&d_keys, mx * 3 * sizeof(int)));
    OP_HIPG(hipMemsetAsync(d_keys, 0, mx * 3 * sizeof(int), stream));
    OP_HIPG(hipMalloc((void**)&d_all, (size_t)world * mx * 3 * sizeof(int)));
    {
        size_t got = 0;
        rc = op_volume_keys_device(v, d_keys, mx, &got);
        if (rc != OP_OK) goto done;
    }
    OP_NCCL(rccl().AllGather(d_keys, d_all, mx * 3, ncclInt32, comm, stream));
    // 2. identical sorted union on every rank
    {
        const size_t n_all = (size_t)world * mx;
        OP_HIPG(hipMalloc((void**)&d_pk, n_all * 8));
        OP_HIPG(hipMalloc((void**)&d_sorted, n_all * 8));
        OP_HIPG(hipMalloc((void**)&d_uniq, n_all * 8));
        OP_HIPG(hipMalloc((void**)&d_nuniq, sizeof(unsigned)));
        hipLaunchKernelGGL(k_pack_keys, dim3((unsigned)((n_all + 255) / 256)), dim3(256), 0, stream, (const int*)d_all, (const int*)d_cnts, world, mx, d_pk);
        OP_HIPG(rocprim::radix_sort_keys(nullptr, tmp_a, d_pk, d_sorted, n_all, 0, 64, stream));
        OP_HIPG(rocprim::unique(nullptr, tmp_b, d_sorted, d_uniq, d_nuniq, n_all, rocprim::equal_to<unsigned long long>(), stream));
        OP_HIPG(hipMalloc(&d_tmp, tmp_a > tmp_b ? tmp_a : tmp_b));
        OP_HIPG(rocprim::radix_sort_keys(d_tmp, tmp_a, d_pk, d_sorted, n_all, 0, 64, stream));
        OP_HIPG(rocprim::unique(d_tmp, tmp_b, d_sorted, d_uniq, d_nuniq, n_all, rocprim::equal_to<unsigned long long>(), stream));
        OP_HIPG(hipMemcpyAsync(&nuniq, d_nuniq, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        OP_HIPG(hipStreamSynchronize(stream));
        n_union = nuniq;
        if (total < n_all) --n_union; // the padding value ~0 is the last unique entry
        OP_HIPG(hipMalloc((void**)&d_union, (n_union ? n_union : 1) * 3 * sizeof(int)));
        if (n_union) hipLaunchKernelGGL(k_unpack_keys, dim3((unsigned)((n_union + 255) / 256)), dim3(256), 0, stream, (const unsigned long long*)d_uniq, n_union, d_union);
        OP_HIPG(hipStreamSynchronize(stream));
    }
    if (n_union == 0) goto done;
    // 3.-5. pack in sum form, one reduce, normalise on the root
    OP_HIPG(hipMalloc((void**)&d_buf, n_union * 5 * 512 * sizeof(float)));
    rc = op_volume_pack_sum(v, d_union, n_union, d_buf);
    if (rc != OP_OK) goto done;
    OP_NCCL(rccl().Reduce(d_buf, d_buf, n_union * 5 * 512, ncclFloat32, ncclSum, root, comm, stream));
    OP_HIPG(hipStreamSynchronize(stream));
    if (rank == root) rc = op_volume_unpack_sum(v, d_union, n_union, d_buf);
done:
    for (void* p : {(void*)d_cnt, (void*)d_cnts, (void*)d_keys, (void*)d_all, (void*)d_union, (void*)d_pk, (void*)d_sorted, (void*)d_uniq, (void*)d_nuniq, d_tmp, (void*)d_buf})
        if (p) (void)hipFree(p);
    if (rc == OP_OK && n_union_out) *n_union_out = n_union;
    return rc;
}
