// common.hpp -- shared plumbing of the C-ABI library (error reporting, HIP checks, device helpers).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/onepiece_hip.h"

namespace op {

extern thread_local char g_last_error[512];

// Process-wide settings (op_runtime_set_option; defined in volume.hip).  The library reads NOTHING from the environment and changes nothing in it:
// what rounds 1-4 took from ONEPIECE_* variables or set at load time is set through the C-ABI by whoever wants it.
struct RuntimeOptions {
    std::atomic<int> merge_algorithm{OP_MERGE_OWNER_EXCHANGE};
    std::atomic<long long> merge_slice_blocks{0};      // union blocks per reduce slice of the dense merge; 0 = the built-in 32 768
    std::atomic<int> merge_force_single_rank{0};       // run the whole exchange with one rank too (how a one-GPU box exercises the RCCL path)
    std::atomic<int> tracker_graph{1};                 // dense tracker: replay the captured hipGraph of a track (0: plain launches)
    std::atomic<int> copy_threads{2};                  // helper threads of the pageable -> pinned staging copies (read when the first host image arrives)
    std::atomic<int> hw_queues_requested{0};           // op_runtime_configure
    std::atomic<int> icp_default_sums{OP_ICP_SUMS_REFERENCE_F32}; // OP_RUNTIME_OPT_ICP_DEFAULT_SUMS: the OP_ICP_OPT_SUMS of contexts created afterwards (op_icp_create, op_icp_register)
    std::atomic<int> tracker_default_sums{OP_TRACK_SUMS_REFERENCE_F32}; // OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS: the OP_TRACK_OPT_SUMS of trackers created afterwards
    std::atomic<int> tracker_batch_sums{0};            // OP_RUNTIME_OPT_TRACKER_BATCH_SUMS: 1 = twelve or more reference-order trackers running at the same time sum in one launch per round (measured: no gain, off by default)
    std::atomic<int> icp_many_in_flight{4};            // OP_RUNTIME_OPT_ICP_MANY_IN_FLIGHT: iterations op_icp_run_many keeps enqueued at a time over its fp64-mode contexts
    std::atomic<long long> merge_fault{0};             // TEST HOOK (OP_RUNTIME_OPT_MERGE_FAULT): stage * 1024 + rank + 1 -- that rank's allocation of that merge stage "fails"; 0 = off
    std::atomic<long long> cache_device_bytes{32ll << 30}; // released device buffers kept for reuse, per device (buffer cache below); 0 = keep none
};
RuntimeOptions& runtime_options();

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

#define OP_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ::op::fail(OP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                              __FILE__, __LINE__);                                                \
    } while (0)

#define OP_TRY(expr)                  \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != OP_OK) return rc_; \
    } while (0)

// Selects `device` after checking that a usable GPU exists; there is no CPU fallback anywhere.
inline int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(OP_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(OP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    OP_HIP(hipSetDevice(device));
    return OP_OK;
}

// ---- buffer cache -------------------------------------------------------------------------------------------------
// The reference's callers create and drop objects per call (registration::PointToPlane builds its kd-tree inside the
// call); the GPU equivalents -- a 216 MB cell table, an 11 MB pinned row buffer, a stream, events -- cost milliseconds
// to allocate and free (hipFree synchronises the device), several times the 1-2 ms such a call computes for.  Released
// buffers therefore go to a per-process cache and are handed out again to the next request of a similar size (best fit,
// at most 25 % larger), so a steady stream of calls allocates nothing.  That includes a volume's block pool (2.7 GB at the default
// capacity): the reference's drivers make a CubeHandler per submap and a new one per Transform, and a hipMalloc of gigabytes
// is not only milliseconds -- every few calls the driver takes 1.2-1.5 s over one (measured: tools/ops_driver.bin transform,
// profiles/r05_transform_pool.txt).  The cache holds at most RuntimeOptions::cache_device_bytes per device (32 GB of the
// 288 GB; OP_RUNTIME_OPT_CACHE_DEVICE_BYTES) and kCacheHostBytes of pinned memory per device, no single buffer above half the
// device limit (beyond that a released buffer is really freed); op_release_cached_memory() empties it.  Owners synchronise their
// stream before releasing, so a cached buffer is idle.  Contents are NOT cleared.
constexpr size_t kCacheHostBytes = 1ull << 30;
struct BufferCache {
    struct Slot { void* p; size_t bytes; int device; bool host; };
    std::mutex mu;
    std::vector<Slot> free_slots, live_slots;
    std::vector<hipStream_t> streams;   // per device would be more precise; streams are tied to the device current at creation,
    std::vector<int> stream_device;     // so the device is kept next to each
    std::vector<hipEvent_t> events;
    std::vector<int> event_device;
    size_t cached(int device, bool host) const {
        size_t t = 0;
        for (const Slot& s : free_slots) if (s.host == host && s.device == device) t += s.bytes;
        return t;
    }
};
inline BufferCache& buffer_cache() { static BufferCache c; return c; }

inline hipError_t cache_alloc(void** out, size_t bytes, bool host) {
    *out = nullptr;
    if (!bytes) bytes = 1;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    BufferCache& c = buffer_cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < c.free_slots.size(); ++i) {
            const BufferCache::Slot& s = c.free_slots[i];
            // pinned memory is handed back to the device it was allocated under as well (it was mapped in that context)
            if (s.host != host || s.device != device || s.bytes < bytes || s.bytes > bytes + bytes / 4 + 4096) continue;
            if (best == (size_t)-1 || s.bytes < c.free_slots[best].bytes) best = i;
        }
        if (best != (size_t)-1) {
            c.live_slots.push_back(c.free_slots[best]);
            *out = c.free_slots[best].p;
            c.free_slots.erase(c.free_slots.begin() + (long)best);
            return hipSuccess;
        }
    }
    void* p = nullptr;
    e = host ? hipHostMalloc(&p, bytes, hipHostMallocMapped) : hipMalloc(&p, bytes);
    if (e != hipSuccess) { // out of memory: give the cache back and try once more
        std::vector<BufferCache::Slot> drop;
        { std::lock_guard<std::mutex> lock(c.mu); drop.swap(c.free_slots); }
        for (const BufferCache::Slot& s : drop) { if (s.host) (void)hipHostFree(s.p); else (void)hipFree(s.p); }
        (void)hipGetLastError();
        e = host ? hipHostMalloc(&p, bytes, hipHostMallocMapped) : hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lock(c.mu);
    c.live_slots.push_back({p, bytes, device, host});
    *out = p;
    return hipSuccess;
}
inline hipError_t cached_malloc(void** out, size_t bytes) { return cache_alloc(out, bytes, false); }
inline hipError_t cached_host_malloc(void** out, size_t bytes) { return cache_alloc(out, bytes, true); } // pinned + mapped

// returns the buffer to the cache (or frees it when the cache is full); pointers not handed out by cache_alloc are freed directly
inline void cached_free(void* p) {
    if (!p) return;
    BufferCache& c = buffer_cache();
    BufferCache::Slot s{nullptr, 0, 0, false};
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (size_t i = 0; i < c.live_slots.size(); ++i)
            if (c.live_slots[i].p == p) { s = c.live_slots[i]; c.live_slots.erase(c.live_slots.begin() + (long)i); break; }
        const size_t limit = s.host ? kCacheHostBytes : (size_t)runtime_options().cache_device_bytes.load();
        if (s.p && s.bytes <= limit / 2 && c.cached(s.device, s.host) + s.bytes <= limit) {
            c.free_slots.push_back(s);
            return;
        }
    }
    if (!s.p) { (void)hipFree(p); return; } // not ours
    if (s.host) (void)hipHostFree(p); else (void)hipFree(p);
}
inline hipError_t cached_stream(hipStream_t* out) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    BufferCache& c = buffer_cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (size_t i = 0; i < c.streams.size(); ++i)
            if (c.stream_device[i] == device) {
                *out = c.streams[i];
                c.streams.erase(c.streams.begin() + (long)i); c.stream_device.erase(c.stream_device.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
inline void release_stream(hipStream_t s, int device) { // the owner has synchronised it
    if (!s) return;
    BufferCache& c = buffer_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    if (c.streams.size() < 64) { c.streams.push_back(s); c.stream_device.push_back(device); } else (void)hipStreamDestroy(s);
}
inline hipError_t cached_event(hipEvent_t* out) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    BufferCache& c = buffer_cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (size_t i = 0; i < c.events.size(); ++i)
            if (c.event_device[i] == device) {
                *out = c.events[i];
                c.events.erase(c.events.begin() + (long)i); c.event_device.erase(c.event_device.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipEventCreateWithFlags(out, hipEventDisableTiming);
}
inline void release_event(hipEvent_t ev, int device) {
    if (!ev) return;
    BufferCache& c = buffer_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    if (c.events.size() < 256) { c.events.push_back(ev); c.event_device.push_back(device); } else (void)hipEventDestroy(ev);
}
// frees everything the cache holds (buffers in use are unaffected)
inline void release_cached_memory() {
    BufferCache& c = buffer_cache();
    std::vector<BufferCache::Slot> drop;
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> ev;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        drop.swap(c.free_slots); st.swap(c.streams); ev.swap(c.events);
        c.stream_device.clear(); c.event_device.clear();
    }
    for (const BufferCache::Slot& s : drop) { if (s.host) (void)hipHostFree(s.p); else (void)hipFree(s.p); }
    for (hipStream_t s : st) (void)hipStreamDestroy(s);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
}

constexpr int kWave = 64; // gfx950 wavefront width

// XCD-aware workgroup order.  Workgroup p of a 1-D grid is dispatched to XCD p % 8 and every XCD has its own 4 MiB
// L2 (MI355X_MICROARCH.md), so with the plain order eight neighbouring workgroups -- which gather neighbouring data --
// land on eight different L2s and each L2 sees the whole working set.  This bijection of [0, n) hands XCD x the
// contiguous slab of logical indices [start_x, start_x + q + (x < r)), q = n / 8, r = n % 8.
__device__ __forceinline__ unsigned xcd_slab_index(unsigned p, unsigned n) {
    const unsigned x = p & 7u, slot = p >> 3, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + slot;
}

// Wave-wide sums of 32 doubles per lane by recursive halving ("reduce-scatter"): at the step with lane
// mask M a lane keeps one half of its values and receives the partner's copy of that half, so the work
// halves every step (16+8+4+2+1+1 = 32 fp64 adds per lane instead of 32 x 6 for a butterfly per value).
// On return lane L holds in v[0] the wave total of element (L >> 1) & 31.
//
// How the halves change hands matters more than the 32 adds (tools/valu_ubench.hip, MI355X): `up ? a : b` on a double
// is two VOP2 v_cndmask_b32 reading VCC, and BACK-TO-BACK VOP2 v_cndmask issue at ~11 cycles each instead of ~1.4 --
// the four selects of a step cost 43 cycles, 1300 per reduction.  So:
//   * lane masks 32 and 16 (24 of the 31 steps) use gfx950's v_permlane32_swap / v_permlane16_swap: swapping the odd
//     rows of a (the value the lower half keeps) with the even rows of b (the value the upper half keeps) leaves
//     {own a, partner's a} in the lower rows and {partner's b, own b} in the upper ones -- a' + b' is the step, with no
//     select and no LDS permute;
//   * the other masks select with v_bfi_b32 against an all-ones / all-zeros lane word.
__device__ __forceinline__ double pack_d(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ double select_d(unsigned m, double if_set, double if_clear) { // m: all ones or all zeros
    const unsigned sl = (unsigned)__double2loint(if_set), sh = (unsigned)__double2hiint(if_set);
    const unsigned cl = (unsigned)__double2loint(if_clear), ch = (unsigned)__double2hiint(if_clear);
    return pack_d((m & sl) | (~m & cl), (m & sh) | (~m & ch));
}
template <int M>
__device__ __forceinline__ double halve_step(double a /* kept where (lane & M) == 0 */, double b /* kept elsewhere */, unsigned up_mask) {
    if constexpr (M == 32 || M == 16) {
        (void)up_mask;
        const unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
        const unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
        if constexpr (M == 32) {
            const auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false), h = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
            return pack_d(l[0], h[0]) + pack_d(l[1], h[1]);
        } else {
            const auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false), h = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
            return pack_d(l[0], h[0]) + pack_d(l[1], h[1]);
        }
    } else {
        const double keep = select_d(up_mask, b, a), give = select_d(up_mask, a, b);
        return keep + __shfl_xor(give, M, 64);
    }
}
template <int H, int M, int N>
__device__ __forceinline__ void wave_halve(double (&v)[N], int lane) {
    const unsigned up_mask = (lane & M) != 0 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < H; ++i) v[i] = halve_step<M>(v[i], v[i + H], up_mask);
}
__device__ __forceinline__ void wave_reduce_scatter32(double (&v)[32]) {
    const int lane = threadIdx.x & 63;
    wave_halve<16, 32>(v, lane);
    wave_halve<8, 16>(v, lane);
    wave_halve<4, 8>(v, lane);
    wave_halve<2, 4>(v, lane);
    wave_halve<1, 2>(v, lane);
    v[0] += __shfl_xor(v[0], 1, 64);
}
// The same reduction when the 32 per-lane values are cheap to (re)compute: `val(k)` (k a compile-time constant after
// unrolling) is evaluated inside the first halving step, so only 16 doubles are ever live (32 VGPRs instead of 64).
// Returns what wave_reduce_scatter32 leaves in v[0]; bit-identical to it.
template <int K> struct index_c { static constexpr int value = K; };
template <int I, class F>
__device__ __forceinline__ void lazy_first_step(F& val, bool up, double (&v)[16]) {
    if constexpr (I < 16) {
        const double lo = val(index_c<I>{}), hi = val(index_c<I + 16>{});
        (void)up;
        v[I] = halve_step<32>(lo, hi, 0u);
        lazy_first_step<I + 1>(val, up, v);
    }
}
// val is called as val(index_c<k>{}) so that k is a constant expression inside it
template <class F>
__device__ __forceinline__ double wave_reduce_scatter32_lazy(F val) {
    const int lane = threadIdx.x & 63;
    const bool up = (lane & 32) != 0;
    double v[16];
    lazy_first_step<0>(val, up, v);
    wave_halve<8, 16>(v, lane);
    wave_halve<4, 8>(v, lane);
    wave_halve<2, 4>(v, lane);
    wave_halve<1, 2>(v, lane);
    return v[0] + __shfl_xor(v[0], 1, 64);
}

} // namespace op
