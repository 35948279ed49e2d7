// volume.hip -- device-resident voxel-block-hashed TSDF volume for gfx950 (MI355X) and the C-ABI
// entry points op_volume_* declared in include/onepiece_hip.h.
//
// What it replaces (file:line under /root/reference/src):
//   integration::CubeHandler::IntegrateImage      Integration/CubeHandler.cpp:197-210
//   CubeHandler::ComputeBounding  (kernel KA)     Integration/CubeHandler.cpp:116-145
//   CubeHandler::PrepareCubes     (kernel KB)     Integration/CubeHandler.cpp:147-196
//   Integrator::GetSDF                            Integration/Integrator.cpp:8-35
//   Integrator::IntegrateImage    (kernel KC)     Integration/Integrator.cpp:36-94
//   TSDFVoxel::operator+                          Integration/TSDFVoxel.h:24-39
//   CubeHandler::Merge            (k_merge_blocks, K4 k_pack_sum/k_unpack_sum)  Integration/CubeHandler.h:145-167
//   CubeHandler::Transform / TransformNearest / GetPointCloud                   Integration/CubeHandler.h:199-338, CubeHandler.cpp:45-69
//   CubeHandler::WriteToFile / ReadFromFile / ReadFromFileFloat                 Integration/CubeHandler.h:40-128
// plus a raycaster that north_star asks for and the reference does not have (k_raycast).
//
// Data layout in HBM:
//   pool   : max_blocks x [5 planes x 512 floats]; plane order sdf, weight, c0, c1, c2; in-plane
//            index = the reference's voxel id x + 8y + 64z.  A wave64 therefore owns one z-slice
//            and reads/writes 256 contiguous bytes per plane -- fully coalesced, unlike the
//            reference's 20-byte AoS TSDFVoxel.
//   keys   : max_blocks x int32[3] (block id), indexed by pool slot.
//   tkeys / tvals : open-addressing hash table (packed 64-bit block id -> pool slot); the probe start
//            is the low bits of the reference's 64-bit VoxelGridHasher value (Geometry/Geometry.h:101-112).
//   bmask / blist : per table slot, the frames of the current batch that selected the block; the
//            list of slots the batch touched.
//   pimg   : the batch's frames packed as {depth, rgba} per pixel.
// Frames are fused in batches of up to kMaxBatch: KA (prepare + bounding) -> KB (select) -> KC
// (integrate), enqueued without any host synchronisation: every kernel is launched with a fixed
// grid and reads its trip counts from device memory.
//
// Floating point: compiled with -ffp-contract=off; every expression keeps the reference's operand
// order and intermediate types (the CPU restatement in oracle/ is what the tests compare against --
// bit for bit on block selection and, in practice, on every voxel value).
#include <cfloat>
#include <climits>
#include <algorithm>
#include <numeric>
#include <vector>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "common.hpp"

#include <type_traits>
#include "host_math.hpp"
#include "px_round.hpp"

namespace op {
thread_local char g_last_error[512] = "";
}

// Load-time defaults of the HIP runtime this library relies on.  Every volume and every tracker owns a HIP stream; the runtime maps
// streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and streams that share a queue serialise: four trackers in
// flight + the fusing volume reach 2.6 k frames/s on 4 queues and 4.5 k on 8.  The runtime reads the variable when
// it initialises (its first API call), so setting it when this library is LOADED is early enough for every consumer that links it or
// dlopens it before touching HIP; a value the caller has set is never overwritten.  op_runtime_hw_queues() reports what is in force.
__attribute__((constructor)) static void op_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

namespace {

using op::fail;

constexpr int kVox = 512;            // voxels per block (CUBE_SIZE^3, VoxelCube.h:4)
constexpr int kBlockFloats = 5 * kVox;
constexpr unsigned long long kEmptyKey = ~0ULL; // table slot never used
constexpr int kPending = -1;         // slot claimed, pool slot not published yet
constexpr int kDead = -3;            // slot claimed but the pool was full
constexpr int kKaW = 64, kKaH = 16;   // KA: a workgroup's pixel rectangle (256 threads x 4 consecutive pixels of a row)
constexpr int kTile = 16;             // KA -> KB: min / max valid depth per 16 x 16 pixel tile (k_select's coarse test); a KA rectangle = 4 tiles
#ifndef KB_GRID
#define KB_GRID 384
#endif
constexpr int kSelectGrid = KB_GRID; // KB persistent grid.x (256-thread workgroups) per frame
static_assert(kSelectGrid % 8 == 0 && kSelectGrid >= 8, "k_select deals a frame's workgroups to the 8 XCDs in equal shares");
#ifndef OP_MAX_BATCH
#define OP_MAX_BATCH 32
#endif
constexpr int kMaxBatch = OP_MAX_BATCH; // frames fused per launch by op_volume_integrate_sequence (<= 32: one bit of the batch mask each)
constexpr int kKaFrames = 16;        // frames per k_prepare_frames launch (its poses + frustum planes travel as kernel arguments: 160 B per frame)
static_assert(kMaxBatch <= 64 && kMaxBatch % kKaFrames == 0, "one bit of the batch mask per frame; KA takes kKaFrames frames per launch");
typedef std::conditional<(kMaxBatch > 32), unsigned long long, unsigned>::type bmask_t; // a block's batch mask: which frames of the batch selected it
__host__ __device__ inline int mask_ctz(unsigned m) { return __builtin_ctz(m); }
__host__ __device__ inline int mask_ctz(unsigned long long m) { return __builtin_ctzll(m); }
__device__ inline unsigned mask_popc(unsigned m) { return (unsigned)__popc(m); }
__device__ inline unsigned mask_popc(unsigned long long m) { return (unsigned)__popcll(m); }
constexpr int kAccSlots = 16;        // see State::acc
// KC (k_integrate): ZT voxels of one (x, y) column of a block per thread (a workgroup of 8 / ZT waves owns a block), the waves
// per SIMD it is compiled for, and its grid = exactly the workgroups that are resident then (they draw blocks of the batch's
// list from per-XCD counters; a multiple of 8 = the XCDs).  Measured per 16-frame launch (tools/ab_variants.sh,
// profiles/r03_ab_column_kernel.txt): ZT 1 at 8 waves 380 us, ZT 2 at 8 waves 357 us, ZT 2 at 6 waves 367 us, ZT 4 at 5 waves 381 us,
// ZT 4 at 4 waves 416 us, ZT 8 at 3 waves 446 us -- the instructions a bigger ZT saves are lost again to the lower occupancy (a wave
// issues at most one instruction every ~5 cycles, so instruction throughput needs the eight waves).
#ifndef KC_ZT
#define KC_ZT 2
#endif
#ifndef KC_COL_MIN_WAVES
#define KC_COL_MIN_WAVES 8
#endif
#ifndef KC_COL_GRID
#define KC_COL_GRID (256 * KC_COL_MIN_WAVES * 4 / (8 / KC_ZT)) // resident workgroups: 256 CUs x 4 SIMDs x waves per SIMD / waves per workgroup
#endif
constexpr int kColGrid = KC_COL_GRID;
// the sum-form variant of k_integrate keeps fewer values per voxel alive and may own more voxels per thread (profiles/r04_ab_sumform_zt.txt)
#ifndef KC_ZT_SUM
#define KC_ZT_SUM 2
#endif
#ifndef KC_SUM_MIN_WAVES
#define KC_SUM_MIN_WAVES KC_COL_MIN_WAVES
#endif
constexpr int kColGridSum = 256 * KC_SUM_MIN_WAVES * 4 / (8 / KC_ZT_SUM);
static_assert(kColGridSum % 8 == 0, "one drawing workgroup per XCD slab at least");
static_assert(kColGrid % 8 == 0, "one drawing workgroup per XCD slab at least");
constexpr int kPartialGrid = 1024; // slots of the counter arrays k_integrate's workgroups add to (workgroup b -> slot b % 1024).  Not more: the host reads
                                   // them with small pageable copies, and a 16 KB device-to-host copy takes the runtime's pinned-staging path (milliseconds)
constexpr int kCoordLimit = 1 << 20; // |block coordinate| < 2^20 (40 km at 4 cm blocks)

struct CamParams {
    float fx, fy, cx, cy, depth_scale, res, trunc;
    int width, height, depth_u16;
    PxAxis ax, ay;  // exact thresholds of the fp32 in-image pixel rounding (px_round.hpp), x and y axis
    int fast_px;    // both exact -> use px_pixel_sp, else the double formula
};
struct PoseFwd { float pose[16]; float planes[24]; }; // planes: top, left, right, bottom, near, far
struct PoseInv { float m[12]; };                      // rows 0..2 of pose^-1
struct BatchFwd { PoseFwd f[kMaxBatch]; };
struct KaFwd { PoseFwd f[kKaFrames]; };            // the slice of a BatchFwd one KA launch gets
struct BatchInv { PoseInv f[kMaxBatch]; };
struct BatchPtrs { const void* depth[kMaxBatch]; const unsigned char* rgb[kMaxBatch]; }; // device images of each frame

#ifndef KC_SHARES
#define KC_SHARES 8
#endif
constexpr int kKcShares = KC_SHARES; // k_integrate: the batch list is dealt to this many draw counters (a multiple of 8: workgroup b draws from share b % kKcShares, on XCD b % 8)
static_assert(kKcShares % 8 == 0 && kKcShares <= 256, "whole XCDs");
// The batch's block list can be kept as kBands lists (-DKC_BANDS=1).  A FULL batch (>= KC_STEAL_MIN_FRAMES frames) then files a block under the
// horizontal image band its first selecting frame sees it in (k_select), and k_integrate's XCD x starts on list x: the workgroups of one XCD -- one
// 4 MiB L2 -- gather from one eighth of every packed frame of the batch instead of from all of them.  Measured (round 4, profiles/r04_ab_bands.txt,
// 32-frame launches of the bench scene): L2 misses fall by 16 % (exact update: FETCH_SIZE 469 -> 393 MB x 2 per launch) to 20 % (sum form: 410 -> 327),
// the launch takes the SAME time with the exact update (676 us both ways) and 5 % LONGER with the sum form (524 -> 551 us): the kernel is bound by
// instruction issue, not by its L2 misses (which the 256 MB MALL serves), and lists of unequal length drain less evenly than equal shares of one
// list.  Not the default.  Short batches use list 0 only in either build, dealt to the XCDs in chunks.
#ifndef KC_BANDS
#define KC_BANDS 0
#endif
#ifndef KC_STEAL_MIN_FRAMES
#define KC_STEAL_MIN_FRAMES 24
#endif
constexpr int kBands = 8;
static_assert(kBands == kKcShares, "one list per draw counter");
constexpr int kKcTSlots = 256; // k_integrate's workgroup b reports its duration to slot b % 256 (atomics on one address serialise at ~100 ns each)
struct State {
    // (the first 32 bytes are what the host's synchronous paths read: StateHead below)
    unsigned n_batch;   // (unused since the batch list became kBands lists: n_list below)
    unsigned overflow;  // bit0 pool full, bit1 table full, bit2 bbox too large, bit3 coordinate range
    unsigned n_rec;     // PrepareCubes record mode: entries in sel_list / sel_cand
    unsigned fail_seq;  // sequence number of the batch that first ran out of pool / table space (valid while overflow & 3)
    unsigned cur_seq;   // sequence number of the batch whose kernels are running (written by KA)
    unsigned pad[3];
    unsigned long long stat_frames;
    unsigned long long stat_launches; // k_integrate launches that fused something (a poisoned launch does not count)
    // Shader-clock duration of k_integrate (s_memtime counts shader cycles on this part, tools/valu_ubench.hip; its value is
    // not synchronised between CUs, so every workgroup measures ITSELF): the longest s_memtime span of a workgroup of the
    // running launch -- the workgroups are resident from the kernel's start to its end -- kKcTSlots slots, folded into
    // stat_kc_ticks by the next batch's KA or by the host.
    unsigned long long kc_t[kKcTSlots];
    unsigned long long stat_kc_ticks;
    unsigned long long n_cand[kMaxBatch];
    float bbox[kMaxBatch][6]; // max xyz, min xyz
    unsigned n_inside[kMaxBatch];
    // ComputeBounding of the batch's frames, accumulated by KA's workgroups with atomicMax / atomicAdd: [0..2] max xyz and
    // [3..5] min xyz of the in-frustum points as order-preserving words (the minima complemented, so that 0 is the identity
    // of all six), [6] their number.  Zeroed by whoever consumed them last (KC, k_finish_select) and by vol_reset.
    unsigned acc[kMaxBatch][kAccSlots][8]; // kAccSlots sets per frame (workgroup x uses set x % kAccSlots): atomics on ONE
                                            // address serialise at ~100 ns each, 300 of them cost KA 35 us
    unsigned kc_next[kKcShares * 16]; // KC dynamic scheduling: next list position of each share of the batch list (one cache line each)
    unsigned n_list[kBands];          // lengths of the batch's block lists (list b = blist + b * max_blocks); a short batch only fills list 0
    int sel_rng[kMaxBatch][8];        // k_select_vote -> k_select_merge: first super-block (absolute) and extent in super-blocks of a frame's words ([3..5] = 0: none)
};

struct StateHead { unsigned n_batch, overflow, n_rec, fail_seq, cur_seq, pad[3]; }; // = the first 32 bytes of State
static_assert(sizeof(StateHead) == 32 && offsetof(State, stat_frames) == 32, "StateHead mirrors the head of State");

struct VolView {
    unsigned long long* tkeys; // packed block id or kEmptyKey
    int* tvals;                // pool slot, kPending or kDead
    unsigned table_mask;
    int* keys;                 // block id by pool slot
    float* pool;
    unsigned max_blocks;
    unsigned* n_blocks;
    bmask_t* bmask;            // per TABLE slot: which frames of the current batch selected the block
    int* blist;                // table slots touched by the current batch: kBands lists of max_blocks entries each (State::n_list)
    int* sel_list;             // record mode (PrepareCubes): table slot (translated to pool slot by k_finish_select) + candidate rank
    unsigned long long* sel_cand;
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long hash_key_dev(int x, int y, int z) {
    return ((unsigned long long)(long long)x * 73856093ULL) ^ ((unsigned long long)(long long)y * 19349663ULL) ^
           ((unsigned long long)(long long)z * 83492791ULL);
}
__device__ __forceinline__ unsigned long long pack_key(int x, int y, int z) {
    return ((unsigned long long)(unsigned)(x + kCoordLimit) << 42) | ((unsigned long long)(unsigned)(y + kCoordLimit) << 21) |
           (unsigned long long)(unsigned)(z + kCoordLimit);
}
__device__ __forceinline__ bool key_in_range(int x, int y, int z) {
    return x >= -kCoordLimit && x < kCoordLimit && y >= -kCoordLimit && y < kCoordLimit && z >= -kCoordLimit && z < kCoordLimit;
}

// Eigen's 3-term reduction order a0 + (a1 + a2).
__device__ __forceinline__ float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }

// The two quotients (fx*X)/Z and (fy*Y)/Z of one projection with ONE reciprocal.  An IEEE float division is, on this
// hardware, v_div_scale x2, v_rcp, two FMAs refining the reciprocal, mul + three FMAs for the quotient, v_div_fmas,
// v_div_fixup (11 instructions; the six divisions of a voxel update were 37 % of KC's VALU work).  When v_div_scale does
// not rescale, the result IS fma(r1, y, q1) of the sequence below, so both quotients can share y: 13 instead of 22
// instructions, bit-identical.  The hardware rescales only when the divisor or the quotient leaves the range where these
// plain FMAs are exact (|Z| or |q| beyond ~2^+-96, denormals); outside the window tested here -- 2^-60 <= |Z| < 2^60 --
// the three operands are first rescaled by 2^+-96 (exact; an operand that over- or underflows in that belongs to a
// quotient beyond 2^+-90), and inside it a quotient that differs can only be one of magnitude < 2^-36 (which every pixel
// rounding maps to the same pixel: the thresholds of px_round are >= 2^-23 away from 0) or > 2^36 (which no image
// contains: both forms are rejected by the caller's bounds test).  Z = 0, inf, NaN give NaN here and +-inf / 0 / NaN
// there: rejected, or a pixel whose sdf = d - Z cannot pass the truncation test.  tests/test_integration_gpu.py compares this function
// with the plain division on the device over dense random and boundary operands (op_debug_project_uv).
__device__ __forceinline__ float div_shared_rcp(float n, float z, float y) {
    float q = n * y;
    float r = __builtin_fmaf(-z, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-z, q, n);
    return __builtin_fmaf(r, y, q);
}
// Pixel index v * width + u of the projection, or -1 when it falls outside the image (Integrator.cpp:20-21,61-63).
template <bool FAST>
__device__ __forceinline__ int project_pixel(const CamParams& C, float X, float Y, float Z) {
    float nx = C.fx * X, ny = C.fy * Y, z = Z;
    const unsigned ez = (__float_as_uint(Z) >> 23) & 0xffu;    // biased exponent of Z
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ez - 67u < 120u)) != 0ull, 0)) { // some lane outside 2^-60 <= |Z| < 2^60 (never, for a camera):
        if (!(ez - 67u < 120u)) {
            const float sc = ez < 67u ? 0x1p96f : 0x1p-96f;     // rescale all three by an exact power of two, which is
            z *= sc; nx *= sc; ny *= sc;                        // what v_div_scale does; the quotients are unchanged
        }
    }
    float y = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    const float ax = div_shared_rcp(nx, z, y), ay = div_shared_rcp(ny, z, y);
    int u, v;
    const bool in_u = FAST ? px_pixel_sp(ax, C.ax, u) : px_pixel_dp(ax, C.cx, C.width, u);
    const bool in_v = FAST ? px_pixel_sp(ay, C.ay, v) : px_pixel_dp(ay, C.cy, C.height, v);
    return (in_u && in_v) ? (int)__umul24((unsigned)v, (unsigned)C.width) + u : -1; // both factors < 2^20: one full-rate 24-bit multiply
}

// test hook: both forms of the projection for n operand triples (see op_debug_project_uv)
__global__ void k_debug_project_uv(CamParams C, const float* __restrict__ X, const float* __restrict__ Y,
                                   const float* __restrict__ Z, size_t n, int* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pix = C.fast_px ? project_pixel<true>(C, X[i], Y[i], Z[i]) : project_pixel<false>(C, X[i], Y[i], Z[i]);
    out[4 * i] = pix < 0 ? INT_MIN : pix % C.width;
    out[4 * i + 1] = pix < 0 ? INT_MIN : pix / C.width;
    out[4 * i + 2] = px_round_dp((C.fx * X[i]) / Z[i], C.cx);   // the reference's formula with the plain division
    out[4 * i + 3] = px_round_dp((C.fy * Y[i]) / Z[i], C.cy);
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Concurrent find-or-claim, wait-free.  The 64-bit packed key is claimed with one CAS, which also
// publishes it, so concurrent claims of the SAME key (frames of one batch) simply agree on the
// table slot.  Only the CAS winner allocates the pool block and stores its index in tvals[slot]
// with a plain store: nobody reads tvals in the launch that inserts -- callers work with the TABLE
// SLOT and translate slot -> pool block in the next kernel (kernel boundaries make it visible on
// every XCD).  Returns the table slot, or -1 when the table is full (flagged in st->overflow).
__device__ int table_claim(const VolView& V, State* st, int x, int y, int z, bool* created) {
    *created = false;
    const unsigned long long key = pack_key(x, y, z);
    unsigned s = (unsigned)hash_key_dev(x, y, z) & V.table_mask;
    for (unsigned probe = 0; probe <= V.table_mask; ++probe, s = (s + 1) & V.table_mask) {
        unsigned long long k = V.tkeys[s];
        if (k == kEmptyKey) {
            // (a stale cached "empty" is harmless: the CAS is resolved at the coherence point)
            k = atomicCAS(&V.tkeys[s], kEmptyKey, key);
            if (k == kEmptyKey) { // slot is ours: allocate a pool block
                const unsigned idx = atomicAdd(V.n_blocks, 1u);
                if (idx >= V.max_blocks) {
                    if ((atomicOr(&st->overflow, 1u) & 3u) == 0u) st->fail_seq = st->cur_seq;
                    V.tvals[s] = kDead;
                } else {
                    V.keys[3 * idx] = x; V.keys[3 * idx + 1] = y; V.keys[3 * idx + 2] = z;
                    V.tvals[s] = (int)idx;
                }
                *created = true;
                return (int)s;
            }
        }
        if (k == key) return (int)s;
    }
    if ((atomicOr(&st->overflow, 2u) & 3u) == 0u) st->fail_seq = st->cur_seq;
    return -1;
}

// Read-only lookup (no concurrent inserts running).
__device__ int table_find(const VolView& V, int x, int y, int z) {
    if (!key_in_range(x, y, z)) return -1;
    const unsigned long long key = pack_key(x, y, z);
    unsigned s = (unsigned)hash_key_dev(x, y, z) & V.table_mask;
    for (unsigned probe = 0; probe <= V.table_mask; ++probe, s = (s + 1) & V.table_mask) {
        const unsigned long long k = V.tkeys[s];
        if (k == kEmptyKey) return -1;
        if (k == key) { const int v = V.tvals[s]; return v >= 0 ? v : -1; }
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
// pool / table maintenance
// ---------------------------------------------------------------------------------------------
__global__ void k_fill_pool(float* pool, size_t first_block, size_t n_blocks) {
    // default TSDFVoxel {sdf 999, weight 0, color (-1,-1,-1)} (TSDFVoxel.h:79-81)
    const size_t total = n_blocks * kBlockFloats;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int plane = (int)((i % kBlockFloats) / kVox);
        pool[first_block * kBlockFloats + i] = plane == 0 ? 999.0f : (plane == 1 ? 0.0f : -1.0f);
    }
}

__global__ void k_clear_table(unsigned long long* tkeys, int* tvals, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        tkeys[i] = kEmptyKey;
        tvals[i] = kPending;
    }
}

// Growth: the n allocated blocks (distinct keys, pool slot = index) re-enter a freshly cleared, larger table.
__global__ void k_rehash(unsigned long long* __restrict__ tkeys, int* __restrict__ tvals, unsigned mask, const int* __restrict__ keys, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
    const unsigned long long key = pack_key(x, y, z);
    for (unsigned s = (unsigned)hash_key_dev(x, y, z) & mask;; s = (s + 1) & mask)
        if (atomicCAS(&tkeys[s], kEmptyKey, key) == kEmptyKey) { tvals[s] = (int)i; return; }
}

// After a select-only launch (PrepareCubes API): clear the batch masks again and translate the
// recorded table slots into pool slots.
__global__ void k_finish_select(VolView V, State* st) {
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < (unsigned)(kMaxBatch * kAccSlots * 8); k += gridDim.x * blockDim.x) (&st->acc[0][0][0])[k] = 0u; // as KC does
    const unsigned n = st->n_list[0] < V.max_blocks ? st->n_list[0] : V.max_blocks; // a one-frame batch: list 0 only
    const unsigned nr = st->n_rec < V.max_blocks ? st->n_rec : V.max_blocks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) V.bmask[V.blist[i]] = (bmask_t)0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x) V.sel_list[i] = V.tvals[V.sel_list[i]];
}

// Integrator::IntegrateImage for a caller-chosen cube list (op_volume_integrate_cubes): takes KB's place in a one-frame
// batch -- every listed cube is found or allocated and put on the batch list with the frame's bit, no selection test.
__global__ void k_mark_cubes(VolView V, State* st, const int* __restrict__ keys, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (st->overflow & 3u)) return;
    const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
    if (!key_in_range(x, y, z)) { atomicOr(&st->overflow, 8u); return; }
    bool created;
    const int slot = table_claim(V, st, x, y, z, &created);
    if (slot < 0) return;
    if (atomicOr(&V.bmask[slot], (bmask_t)1) == (bmask_t)0) { // a key listed twice is fused once
        const unsigned pos = atomicAdd(&st->n_list[0], 1u);
        if (pos < V.max_blocks) V.blist[pos] = slot;
    }
}

__device__ __forceinline__ unsigned ord_enc(float f) { // order-preserving float -> unsigned
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// ---------------------------------------------------------------------------------------------
// KA: per-frame preparation = ComputeBounding (CubeHandler.cpp:116-145: back-project, transform,
// frustum test, min/max) + packing of the frame into one {depth, rgba} record per pixel so that
// the later gathers are single 8-byte loads + the smallest and largest valid depth of every 16 x 16
// pixel tile (what k_select's coarse test looks at).  grid = (ka_grid(W, H), n_frames); a workgroup
// owns a 64 x 16 pixel rectangle, a thread 4 consecutive pixels of one row.
// One bounding partial per workgroup (no atomics): [max x,y,z, min x,y,z, inside, pad].
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int ka_grid(int w, int h) { return ((w + kKaW - 1) / kKaW) * ((h + kKaH - 1) / kKaH); }
__host__ __device__ inline int tiles_w(int w) { return (w + kTile - 1) / kTile; }
__host__ __device__ inline int tiles_h(int h) { return (h + kTile - 1) / kTile; }

__global__ __launch_bounds__(256) void k_prepare_frames(KaFwd B, int f0, CamParams C, BatchPtrs Q, uint2* __restrict__ pimg, float2* __restrict__ ptile,
                                                        float* __restrict__ partial, State* st, unsigned seq,
                                                        const unsigned* __restrict__ n_blocks, unsigned* __restrict__ hstat) {
    __shared__ float s_red[4][6];
    __shared__ unsigned s_cnt[4];
    __shared__ float s_tile[4][4][2];
    const int tid = threadIdx.x, f = f0 + (int)blockIdx.y; // frame of the batch
    if (blockIdx.x == 0 && f == 0 && tid == 0) {
        st->n_rec = 0; // new batch: empty lists
        for (int b = 0; b < kBands; ++b) st->n_list[b] = 0;
        st->cur_seq = seq;
        // Progress report for the host (host-mapped pinned memory, read without any synchronisation): this kernel starting
        // means every earlier batch has finished; unless the stream is poisoned by an overflow they all completed.  The host
        // uses it to retire its replay log / staging slots and to grow the pool BEFORE it runs full.
        if (hstat && (st->overflow & 3u) == 0u) {
            hstat[1] = *n_blocks;
            __threadfence_system();
            __hip_atomic_store(&hstat[0], seq - 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (blockIdx.x == 0 && f == 0 && tid < kKcShares) st->kc_next[tid * 16] = 0u;
    if (blockIdx.x == 0 && f == 0) { // the previous launch's k_integrate duration in shader cycles: its longest workgroup
        static_assert(kKcTSlots == 256, "one slot per thread");
        unsigned long long kc = st->kc_t[tid];
        st->kc_t[tid] = 0ull;
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(kc, o, 64); kc = x > kc ? x : kc; }
        __shared__ unsigned long long s_kc[4];
        if ((tid & 63) == 0) s_kc[tid >> 6] = kc;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) kc = s_kc[w] > kc ? s_kc[w] : kc;
            st->stat_kc_ticks += kc;
        }
    }
    const PoseFwd& P = B.f[blockIdx.y];
    const int npix = C.width * C.height;
    const void* dptr = Q.depth[f];
    const unsigned char* cptr = Q.rgb[f];
    uint2* out = pimg + (size_t)f * npix;
    const int wgx = (C.width + kKaW - 1) / kKaW;
    const int gy = (int)blockIdx.x / wgx, gx = (int)blockIdx.x - gy * wgx;
    const int row = gy * kKaH + (tid >> 4), col0 = gx * kKaW + (tid & 15) * 4;
    float mx0 = -FLT_MAX, mx1 = -FLT_MAX, mx2 = -FLT_MAX, mn0 = FLT_MAX, mn1 = FLT_MAX, mn2 = FLT_MAX;
    float tmin = __builtin_inff(), tmax = -__builtin_inff(); // valid depths of the thread's pixels
    unsigned inside = 0;
    if (row < C.height && col0 < C.width) {
        const int pix0 = row * C.width + col0;
        float zz[4];
        unsigned cc[4] = {0u, 0u, 0u, 0u};
        const int nv = C.width - col0 < 4 ? C.width - col0 : 4; // pixels of the row this thread has
        // Integrator.cpp:26-29 / PointCloud.cpp:83-86: float depth, or uint16 / depth_scale.  Aligned rows take one wide load per thread.
        const bool wide = nv == 4 && (C.width & 3) == 0 && ((size_t)dptr & 15u) == 0 && ((size_t)cptr & 3u) == 0;
        if (wide) {
            if (C.depth_u16) {
                const ushort4 d = *reinterpret_cast<const ushort4*>((const unsigned short*)dptr + pix0);
                zz[0] = (float)d.x / C.depth_scale; zz[1] = (float)d.y / C.depth_scale; zz[2] = (float)d.z / C.depth_scale; zz[3] = (float)d.w / C.depth_scale;
            } else {
                const float4 d = *reinterpret_cast<const float4*>((const float*)dptr + pix0);
                zz[0] = d.x; zz[1] = d.y; zz[2] = d.z; zz[3] = d.w;
            }
            if (cptr) {
                const unsigned* c3 = reinterpret_cast<const unsigned*>(cptr + 3 * (size_t)pix0); // 12 bytes = 4 pixels, 4-byte aligned
                const unsigned d0 = c3[0], d1 = c3[1], d2 = c3[2];
                cc[0] = d0 & 0xffffffu; cc[1] = (d0 >> 24) | ((d1 & 0xffffu) << 8); cc[2] = (d1 >> 16) | ((d2 & 0xffu) << 16); cc[3] = d2 >> 8;
            }
            uint4* o4 = reinterpret_cast<uint4*>(out + pix0); // byte 3 of the colour word = 1: the observation count k_integrate's sum form adds up
            o4[0] = make_uint4(__float_as_uint(zz[0]), cc[0] | 0x01000000u, __float_as_uint(zz[1]), cc[1] | 0x01000000u);
            o4[1] = make_uint4(__float_as_uint(zz[2]), cc[2] | 0x01000000u, __float_as_uint(zz[3]), cc[3] | 0x01000000u);
        } else {
            for (int e = 0; e < nv; ++e) {
                const int pix = pix0 + e;
                zz[e] = C.depth_u16 ? (float)((const unsigned short*)dptr)[pix] / C.depth_scale : ((const float*)dptr)[pix];
                if (cptr) cc[e] = (unsigned)cptr[3 * (size_t)pix] | ((unsigned)cptr[3 * (size_t)pix + 1] << 8) | ((unsigned)cptr[3 * (size_t)pix + 2] << 16);
                out[pix] = make_uint2(__float_as_uint(zz[e]), cc[e] | 0x01000000u);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= nv) break;
            const float z = zz[e];
            if (!(z > 0)) continue;
            tmin = fminf(tmin, z); tmax = fmaxf(tmax, z);
            const int i = row, j = col0 + e;
            const float x = ((float)j - C.cx) * z / C.fx; // PointCloud.cpp:90-93
            const float y = ((float)i - C.cy) * z / C.fy;
            const float* M = P.pose;                       // Geometry.cpp:19-27
            const float q0 = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * 1.0f;
            const float q1 = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * 1.0f;
            const float q2 = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * 1.0f;
            const float q3 = ((M[12] * x + M[13] * y) + M[14] * z) + M[15] * 1.0f;
            // TransformPoints divides by w (Geometry.cpp:24-26).  A rigid pose has the bottom row (0, 0, 0, 1), so w is exactly 1 and x / 1 = x:
            // the three IEEE divisions (33 of the ~220 instructions per pixel) only run when some lane's w is not 1 (a projective "pose", NaN).
            float p0 = q0, p1 = q1, p2 = q2;
            if (__builtin_amdgcn_ballot_w64(q3 != 1.0f) != 0ull) { p0 = q0 / q3; p1 = q1 / q3; p2 = q2 / q3; }
            bool in = true; // Frustum::ContainPoint incl. its early "== 0 -> true" (Frustum.h:74-103)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float dist = sum3(P.planes[4 * k] * p0, P.planes[4 * k + 1] * p1, P.planes[4 * k + 2] * p2) + P.planes[4 * k + 3];
                if (dist < 0) { in = false; break; }
                if (dist == 0) break;
            }
            if (in) {
                ++inside;
                mx0 = p0 > mx0 ? p0 : mx0; mx1 = p1 > mx1 ? p1 : mx1; mx2 = p2 > mx2 ? p2 : mx2;
                mn0 = p0 < mn0 ? p0 : mn0; mn1 = p1 < mn1 ? p1 : mn1; mn2 = p2 < mn2 ? p2 : mn2;
            }
        }
    }
    // tiles: 4 lanes share a 16-pixel row segment, lane bits 4 and 5 are the wave's 4 rows, the 4 waves are the tile's 16 rows
    tmin = fminf(tmin, __shfl_xor(tmin, 1, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 1, 64));
    tmin = fminf(tmin, __shfl_xor(tmin, 2, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 2, 64));
    tmin = fminf(tmin, __shfl_xor(tmin, 16, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmin = fminf(tmin, __shfl_xor(tmin, 32, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    mx0 = wave_max(mx0); mx1 = wave_max(mx1); mx2 = wave_max(mx2);
    mn0 = wave_min(mn0); mn1 = wave_min(mn1); mn2 = wave_min(mn2);
    inside = wave_sum(inside);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
        s_red[wave][0] = mx0; s_red[wave][1] = mx1; s_red[wave][2] = mx2;
        s_red[wave][3] = mn0; s_red[wave][4] = mn1; s_red[wave][5] = mn2;
        s_cnt[wave] = inside;
    }
    if ((lane & 0x33) == 0) { s_tile[wave][lane >> 2][0] = tmin; s_tile[wave][lane >> 2][1] = tmax; }
    __syncthreads();
    if (tid < 4) {
        const int tx = gx * (kKaW / kTile) + tid, tw = tiles_w(C.width);
        if (tx < tw && gy < tiles_h(C.height)) {
            float lo = s_tile[0][tid][0], hi = s_tile[0][tid][1];
            for (int w = 1; w < 4; ++w) { lo = fminf(lo, s_tile[w][tid][0]); hi = fmaxf(hi, s_tile[w][tid][1]); }
            ptile[((size_t)f * tiles_h(C.height) + gy) * tw + tx] = make_float2(lo, hi);
        }
    }
    float* pout = partial + ((size_t)f * gridDim.x + blockIdx.x) * 8;
    if (tid < 6) {
        float v = s_red[0][tid];
        for (int w = 1; w < 4; ++w) v = tid < 3 ? fmaxf(v, s_red[w][tid]) : fminf(v, s_red[w][tid]);
        pout[tid] = v;
    } else if (tid == 6) {
        ((unsigned*)pout)[6] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    // ... and into the frame's accumulators, from which every KB workgroup takes the candidate range (they used to fold
    // the frame's 300 partial rows each: a third of that kernel's time).  The rows stay for op_volume_compute_bounding.
    if (tid < 7) {
        const unsigned cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (cnt) { // a workgroup without an in-frustum point contributes nothing (its row is the identity)
            unsigned* a = st->acc[f][blockIdx.x % kAccSlots];
            if (tid < 3) atomicMax(&a[tid], ord_enc(pout[tid]));
            else if (tid < 6) atomicMax(&a[tid], ~ord_enc(pout[tid]));
            else atomicAdd(&a[6], cnt);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// KB: PrepareCubes (CubeHandler.cpp:147-196) for every frame of the batch (blockIdx.y = frame).
// The reference probes EVERY block of the bbox +-1 range (8 corner-voxel GetSDF probes each); ~13 % are selected.  Here the range is cut
// into super-blocks of kSB^3 blocks, and a super-block is first tested as a whole against the tiles' depth range (k_prepare_frames):
//   * its 8 extreme voxel centres are transformed into the camera; all of them farther than 5 cm in front of it => every voxel centre of
//     the super-block projects inside the pixel bounding box of those 8 projections (a projective map keeps convex hulls while z > 0);
//   * the box widened by 2 px + 0.1 % lies outside the image => every probe is off-image (GetSDF = 999), nothing is selected;
//   * else, with [dmin, dmax] the valid depths of the tiles the box touches: dmin - zmax >= truncation + 1 mm or zmin - dmax >= truncation
//     + 1 mm (or no valid depth at all) => every probe has |sdf| >= truncation, nothing is selected.
// The margins are orders of magnitude above the float rounding of either side, so a super-block is only ever dropped when the exact
// per-block test below would reject every one of its blocks; the selected set is the reference's, bit for bit (parity suite, fuzz).
// Surviving super-blocks go through the exact test: one wave per super-block, one lane per block, 8 probes per lane.  A selected block
// is looked up / inserted in the hash table, its batch mask gets the frame's bit, and the first selection in the batch appends it to
// the batch list (collected in LDS, one global append per workgroup).  record != 0 (single-frame PrepareCubes API): also emits
// (table slot, candidate rank) pairs.
// ---------------------------------------------------------------------------------------------
#ifndef KB_SB
#define KB_SB 4
#endif
constexpr int kSB = KB_SB;                      // super-block edge in blocks
constexpr int kSBVol = kSB * kSB * kSB;         // 64 blocks = one wave
#ifndef KB_SBPERWG
#define KB_SBPERWG 8
#endif
constexpr int kSBPerWg = KB_SBPERWG;                    // super-blocks a workgroup tests at a time (8 lanes each)
constexpr int kSelTiles = 64;                   // a super-block whose pixel box touches more tiles skips the depth test (it is close to the camera)
static_assert(kSBVol == 64, "one lane per block of a super-block");

// -- the three steps of the selection, shared by k_select and k_select_vote ---------------------------------------------------------
// Finish ComputeBounding from the frame's accumulators (k_prepare_frames) and turn it into the candidate range (CubeHandler.cpp:147-163): wave 0 of a
// workgroup calls this, lane 0 leaves {i0, j0, k0, ni, nj, nk} in range[] (shared memory; all 0: no candidates) and, if `publish`, the frame's statistics in State.
__device__ __forceinline__ void frame_candidate_range(State* st, const CamParams& C, int f, int lane, bool publish, int* range) {
    unsigned tot = 0, e[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    const unsigned poisoned = lane == 0 ? (st->overflow & 3u) : 0u; // issued together with the accumulator loads: one round trip, not two
    if (lane < kAccSlots) { // lane k < kAccSlots reads set k (one round trip), then a 16-lane fold
        const unsigned* a = st->acc[f][lane];
        tot = a[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) e[c] = a[c];
    }
#pragma unroll
    for (int o = kAccSlots / 2; o > 0; o >>= 1) {
        tot += __shfl_xor(tot, o, 64);
#pragma unroll
        for (int c = 0; c < 6; ++c) { const unsigned x = __shfl_xor(e[c], o, 64); e[c] = x > e[c] ? x : e[c]; }
    }
    if (lane == 0) {
        float b[6];
        for (int c = 0; c < 6; ++c) // nothing in the frustum: the reference's lowest() / max() start values (CubeHandler.cpp:129-130)
            b[c] = tot ? ord_dec(c < 3 ? e[c] : ~e[c]) : (c < 3 ? -FLT_MAX : FLT_MAX);
        // A batch that ran out of pool / table space poisons the stream: its KC and every later batch do nothing (an empty
        // candidate range here), so that the host can grow the volume and REPLAY from the failing batch on -- no frame is
        // ever partially fused (vol_recover).  Read by one thread per workgroup: a per-thread load of this hot line next to
        // the candidate loop doubled the kernel's time.
        if (tot == 0 || poisoned) {
            for (int c = 0; c < 6; ++c) range[c] = 0;
        } else {
            for (int c = 0; c < 3; ++c) {
                // GetCubeID (VoxelCube.h:63-74): floor(p/res) in float -> int, then
                // floor((pb + 0.0)/8) in double == arithmetic shift by 3.
                const int hi = ((int)floorf(b[c] / C.res)) >> 3;
                const int lo = ((int)floorf(b[3 + c] / C.res)) >> 3;
                range[c] = lo - 1;
                range[3 + c] = hi - lo + 3;
            }
        }
        if (publish) {
            for (int c = 0; c < 6; ++c) st->bbox[f][c] = b[c];
            st->n_inside[f] = tot;
        }
    }
}

// Coarse test of the blocks [bi0..bi1] x [bj0..bj1] x [bk0..bk1] (part of a super-block) against frame M / tiles: false when the exact test below would
// reject every one of them (see the comment above k_select).  Called by 8 consecutive lanes, one per corner of the box, with the same arguments otherwise.
__device__ __forceinline__ bool superblock_survives(const CamParams& C, const float* __restrict__ M, const float2* __restrict__ tiles, int tw, int corner,
                                                    int bi0, int bi1, int bj0, int bj1, int bk0, int bk1, float cube_res, float o_lo, float o_hi) {
    const float px = (corner & 1) ? (float)bi1 * cube_res + o_hi : (float)bi0 * cube_res + o_lo;
    const float py = (corner & 2) ? (float)bj1 * cube_res + o_hi : (float)bj0 * cube_res + o_lo;
    const float pz = (corner & 4) ? (float)bk1 * cube_res + o_hi : (float)bk0 * cube_res + o_lo;
    const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    const float qy = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    const float qz = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    float zmin = qz, zmax = qz;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { zmin = fminf(zmin, __shfl_xor(zmin, o, 64)); zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64)); }
    if (!(zmin > 0.05f)) return true; // not all 8 extreme centres well in front of the camera: no shortcut
    // (qz > 0.05 on all 8 lanes.  v_rcp_f32 instead of the exact quotient: 1 ulp against margins of 2 px + 0.1 %, and two divisions were a sixth of this function)
    const float rz = __builtin_amdgcn_rcpf(qz);
    const float uf = (C.fx * q0) * rz + C.cx, vf = (C.fy * qy) * rz + C.cy;
    float umin = uf, umax = uf, vmin = vf, vmax = vf;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        umin = fminf(umin, __shfl_xor(umin, o, 64)); umax = fmaxf(umax, __shfl_xor(umax, o, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    }
    const float mu = 2.0f + 1e-3f * fmaxf(fabsf(umin), fabsf(umax)), mv = 2.0f + 1e-3f * fmaxf(fabsf(vmin), fabsf(vmax));
    const float u_lo = umin - mu, u_hi = umax + mu, v_lo = vmin - mv, v_hi = vmax + mv;
    const float wmax = (float)(C.width - 1), hmax = (float)(C.height - 1);
    if (!(u_hi >= 0.0f && u_lo <= wmax && v_hi >= 0.0f && v_lo <= hmax))
        return !(u_hi < 0.0f || u_lo > wmax || v_hi < 0.0f || v_lo > hmax); // NaN somewhere: no shortcut
    const int x0 = (int)fmaxf(u_lo, 0.0f), x1 = (int)fminf(u_hi, wmax), y0 = (int)fmaxf(v_lo, 0.0f), y1 = (int)fminf(v_hi, hmax);
    const int tx0 = x0 / kTile, tx1 = x1 / kTile, ty0 = y0 / kTile, ty1 = y1 / kTile;
    const int ntx = tx1 - tx0 + 1, nt = ntx * (ty1 - ty0 + 1);
    if (nt > kSelTiles) return true;
    float dmin = __builtin_inff(), dmax = -__builtin_inff();
    static_assert(kSelTiles == 64, "8 tiles per lane at most");
    float2 d[8]; // the lane's tiles t = corner, corner + 8, ...: independent loads, one round trip (a loop with one dependent load per trip was most of this function's time)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int t = corner + 8 * k;
        const int ty = (int)((float)t * (1.0f / (float)ntx) + 1e-4f); // t / ntx for 0 <= t < 64, 1 <= ntx <= 64 (the quotient's fractional part is 0 or >= 1/64)
        const int tx = t - ty * ntx;
        d[k] = t < nt ? tiles[(ty0 + ty) * tw + tx0 + tx] : make_float2(__builtin_inff(), -__builtin_inff());
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { dmin = fminf(dmin, d[k].x); dmax = fmaxf(dmax, d[k].y); }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { dmin = fminf(dmin, __shfl_xor(dmin, o, 64)); dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64)); }
    const float guard = C.trunc + 1e-3f;
    // (no valid depth in the tiles: dmin = +inf, dmax = -inf, both differences are +inf)
    return !(dmin - zmax >= guard || zmin - dmax >= guard);
}

// Integrator::GetSDF (Integrator.cpp:8-35) probes of the 8 corner voxels {0,7,56,63,448,455,504,511} of the block at (bx, by, bz) for the frame
// with inverse pose rows M and packed image img: all 8 projections first, then all 8 gathers in flight together, then the min.  True when the
// block is selected (CubeHandler.cpp:176-190: min |sdf| < truncation).  pmax = the largest pixel index among the corners (-1: none on the image).
template <bool FAST>
__device__ __forceinline__ bool block_selected(const CamParams& C, const float* __restrict__ M, const uint2* __restrict__ img, float bx, float by, float bz,
                                               float o_lo, float o_hi, int& pmax) {
    int pix[8];
    float zc[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float px = bx + ((corner & 1) ? o_hi : o_lo);
        const float py = by + ((corner & 2) ? o_hi : o_lo);
        const float pz = bz + ((corner & 4) ? o_hi : o_lo);
        const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
        const float q1c = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
        const float q2c = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
        zc[corner] = q2c;
        pix[corner] = project_pixel<FAST>(C, q0, q1c, q2c);
    }
    float dd[8];
    pmax = pix[0];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) { dd[corner] = pix[corner] >= 0 ? __uint_as_float(img[(unsigned)pix[corner]].x) : 0.0f; pmax = max(pmax, pix[corner]); }
    float min_sdf = FLT_MAX;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float sdf = dd[corner] <= 0 ? 999.0f : dd[corner] - zc[corner]; // off-image or d <= 0 -> 999
        const float a = fabsf(sdf);
        if (min_sdf > a) min_sdf = a;
    }
    return min_sdf < C.trunc;
}

#ifndef KB_MINWAVES
#define KB_MINWAVES 7
#endif
template <bool FAST>
__global__ __launch_bounds__(256, KB_MINWAVES) void k_select(BatchInv B, CamParams C, VolView V, const uint2* __restrict__ pimg, const float2* __restrict__ ptile,
                                                State* st, int record) {
    __shared__ int s_range[6]; // i0, j0, k0, ni, nj, nk
    __shared__ unsigned s_nsurv, s_nfirst, s_nrec, s_base[2];
    __shared__ unsigned s_surv[kSBPerWg];
    __shared__ int s_first[kSBPerWg * kSBVol];
    __shared__ unsigned short s_fpos[kSBPerWg * kSBVol]; // band (3 bits) | rank within the workgroup's entries of that band << 3
    __shared__ unsigned s_bcnt[kBands], s_bbase[kBands];
    __shared__ int s_rslot[kSBPerWg * kSBVol];
    __shared__ unsigned long long s_rcand[kSBPerWg * kSBVol];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup -> (frame, slot within the frame).  Consecutive workgroups land on consecutive XCDs, each with its own 4 MiB
    // L2, and a candidate's 8 corner probes gather from its frame's 2.4 MB packed image.
    // The batch is nf x 8 work units (a frame's chunks c with c % 8 == q); XCD x takes units [x nf, (x + 1) nf) in frame-major
    // order, i.e. exactly nf / 8 frames' worth whatever nf is, and walks them frame after frame (dispatch order ~ j), so that its L2 holds ONE
    // 2.4 MB image at a time.  (Whole frames per XCD -- frames x, x + 8, ... -- left some XCDs with two frames and others with one whenever
    // nf is not a multiple of 8: a 14-frame batch took as long as a 16-frame one.)
    int f, wslot, wstride;
    {
        const int nf = (int)gridDim.y, id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int per_unit = (int)gridDim.x >> 3;        // workgroups per unit
        const int x = id & 7, j = id >> 3;               // j = 0 .. nf * per_unit - 1 on this XCD
        const int u = x * nf + j / per_unit;             // global unit
        f = u >> 3;
        wslot = (u & 7) + 8 * (j % per_unit);            // 0 .. grid.x - 1; slot 0 of a frame also publishes the frame's statistics
        wstride = (int)gridDim.x;
    }
    const float* M = B.f[f].m;
    const uint2* img = pimg + (size_t)f * C.width * C.height;
    const int tw = tiles_w(C.width), th = tiles_h(C.height);
    const float2* tiles = ptile + (size_t)f * tw * th;

    if (wave == 0) frame_candidate_range(st, C, f, lane, wslot == 0, s_range);
    if (tid == 0) {
        s_nsurv = 0u; s_nfirst = 0u; s_nrec = 0u;
    }
    if (tid < kBands) s_bcnt[tid] = 0u;
    __syncthreads();
    // full batches file a block under the image band (eighths of the image height) it is first seen in; see kBands
    const bool bands = KC_BANDS != 0 && (int)gridDim.y >= KC_STEAL_MIN_FRAMES && record == 0;
    const float band_scale = 8.0f / (float)(C.width * C.height);
    const int i0 = s_range[0], j0 = s_range[1], k0 = s_range[2];
    int ni = s_range[3], nj = s_range[4], nk = s_range[5];
    unsigned long long ncand = (unsigned long long)((long long)ni * nj * nk);
    if (ni > 4096 || nj > 4096 || nk > 4096) { // > 160 m at 5 mm: treat as a bad frame, select nothing
        if (wslot == 0 && tid == 0) atomicOr(&st->overflow, 4u);
        ncand = 0; ni = nj = nk = 0;
    }
    if (wslot == 0 && tid == 0) st->n_cand[f] = ncand;

    const float cube_res = C.res * 8.0f; // CubeHandler.cpp:164
    const float half = C.res / 2;        // VoxelCube.h:47
    const float o_lo = 0.0f * C.res + half, o_hi = 7.0f * C.res + half; // VoxelCentroidOffSet of x = 0 / 7
    const bmask_t fbit = (bmask_t)1 << f;
    // super-block grid of the frame's range (<= 1024^3 < 2^32 entries)
    const unsigned nsi = (unsigned)(ni + kSB - 1) / kSB, nsj = (unsigned)(nj + kSB - 1) / kSB, nsk = (unsigned)(nk + kSB - 1) / kSB;
    const unsigned n_super = nsi * nsj * nsk;

    for (unsigned chunk = (unsigned)wslot; (unsigned long long)chunk * kSBPerWg < n_super; chunk += (unsigned)wstride) {
        // ---- coarse test: 8 lanes per super-block (one per corner), waves 0 and 1
        if (tid < kSBPerWg * 8) {
            const unsigned sb = chunk * kSBPerWg + (unsigned)(tid >> 3);
            const int corner = tid & 7;
            bool survive = false;
            if (sb < n_super) {
                const unsigned q1 = sb / nsk, q2 = q1 / nsj;
                const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
                // first and last block of the super-block inside the range, per axis
                const int bi0 = i0 + si * kSB, bj0 = j0 + sj * kSB, bk0 = k0 + sk * kSB;
                const int bi1 = min(bi0 + kSB - 1, i0 + ni - 1), bj1 = min(bj0 + kSB - 1, j0 + nj - 1), bk1 = min(bk0 + kSB - 1, k0 + nk - 1);
                survive = superblock_survives(C, M, tiles, tw, corner, bi0, bi1, bj0, bj1, bk0, bk1, cube_res, o_lo, o_hi);
            }
            if (survive && corner == 0) s_surv[atomicAdd(&s_nsurv, 1u)] = chunk * kSBPerWg + (unsigned)(tid >> 3);
        }
        __syncthreads();
        const unsigned nsurv = s_nsurv;
        // ---- exact test: one wave per surviving super-block, one lane per block
        for (unsigned sv = (unsigned)wave; sv < nsurv; sv += 4u) {
            const unsigned sb = s_surv[sv];
            const unsigned q1 = sb / nsk, q2 = q1 / nsj;
            const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
            const int ci = si * kSB + (lane >> 4), cj = sj * kSB + ((lane >> 2) & 3), ck = sk * kSB + (lane & 3); // position in the range
            bool first = false, rec = false;
            int pool_idx = -1, band = 0;
            if (ci < ni && cj < nj && ck < nk) {
                const int bi = i0 + ci, bj = j0 + cj, bk = k0 + ck;
                const float bx = (float)bi * cube_res, by = (float)bj * cube_res, bz = (float)bk * cube_res;
                int pmax; // the lowest on-image corner (largest pixel index): files the block under an image band below
                if (block_selected<FAST>(C, M, img, bx, by, bz, o_lo, o_hi, pmax)) {
                    if (!key_in_range(bi, bj, bk)) {
                        atomicOr(&st->overflow, 8u);
                    } else {
                        bool created;
                        pool_idx = table_claim(V, st, bi, bj, bk, &created); // table slot; KC translates it
                        if (pool_idx >= 0) {
                            first = atomicOr(&V.bmask[pool_idx], fbit) == (bmask_t)0;
                            rec = record != 0;
                            // the image row of the block's lowest on-image corner (pixel index / pixels per band; a heuristic, any band is correct)
                            if (bands) band = min(kBands - 1, (int)((float)max(pmax, 0) * band_scale));
                        }
                    }
                }
            }
            // wave-aggregated appends to the workgroup's lists: batch list (first selection in this batch) and record list
            const unsigned long long m_a = __ballot(first), m_b = __ballot(rec);
            const unsigned long long below = (1ULL << lane) - 1ULL;
            if (m_a) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_nfirst, (unsigned)__popcll(m_a));
                base = __shfl(base, 0, 64);
                if (first) {
                    const unsigned k = base + __popcll(m_a & below);
                    s_first[k] = pool_idx;
                    s_fpos[k] = (unsigned short)((unsigned)band | ((bands ? atomicAdd(&s_bcnt[band], 1u) : k) << 3));
                }
            }
            if (m_b) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_nrec, (unsigned)__popcll(m_b));
                base = __shfl(base, 0, 64);
                if (rec) {
                    const unsigned r = base + __popcll(m_b & below);
                    s_rslot[r] = pool_idx;
                    // candidate rank == position in the reference's i, j, k loop nest, k fastest (CubeHandler.cpp:170-173)
                    s_rcand[r] = ((unsigned long long)ci * (unsigned long long)nj + (unsigned long long)cj) * (unsigned long long)nk + (unsigned long long)ck;
                }
            }
        }
        __syncthreads();
        // ---- one global append per workgroup and list
        const unsigned nfirst = s_nfirst, nrec = s_nrec;
        if (tid < kBands) { // one global append per list
            const unsigned c = bands ? s_bcnt[tid] : (tid == 0 ? nfirst : 0u);
            s_bbase[tid] = c ? atomicAdd(&st->n_list[tid], c) : 0u;
        }
        if (tid == 0) {
            s_base[1] = nrec ? atomicAdd(&st->n_rec, nrec) : 0u;
            s_nsurv = 0u;
        }
        __syncthreads();
        for (unsigned k = (unsigned)tid; k < nfirst; k += 256u) {
            const unsigned fp = s_fpos[k], b = fp & 7u;
            const unsigned pos = s_bbase[b] + (fp >> 3);
            if (pos < V.max_blocks) V.blist[(size_t)b * V.max_blocks + pos] = s_first[k];
        }
        for (unsigned k = (unsigned)tid; k < nrec; k += 256u) {
            const unsigned pos = s_base[1] + k;
            if (pos < V.max_blocks) { V.sel_list[pos] = s_rslot[k]; V.sel_cand[pos] = s_rcand[k]; }
        }
        __syncthreads(); // the lists are reused by the next chunk
        if (tid == 0) { s_nfirst = 0u; s_nrec = 0u; }
        if (tid < kBands) s_bcnt[tid] = 0u;
        // (the next chunk's coarse test does not touch s_nfirst / s_nrec / s_bcnt; its __syncthreads orders the reset before their next use)
    }
}

// ---------------------------------------------------------------------------------------------
// KB for a batch of several frames, in two steps.  Consecutive frames select nearly the same blocks -- a block of the bench scene is selected by ~25 of a
// batch's 32 frames -- and in k_select every one of those selections is a hash probe + a returning atomicOr on the block's batch mask + a list append by
// whichever frame came first, ~650 k dependent round trips per batch, with the lists in LDS that force four barriers on every chunk of super-blocks.
// k_select_vote only RECORDS a frame's selections: one 64-bit word per super-block of its range (bit = lane = block; 0 for a super-block the coarse
// test dropped), plain stores into sbits[f][super-block].  Nothing is shared between the waves of a workgroup any more, so every wave walks chunks of 8
// super-blocks on its own -- coarse test (8 lanes per super-block), then the exact test of each survivor (one lane per block) -- without a barrier.
// The super-blocks are aligned to absolute block coordinates (block >> 2), so that the frames of a batch cut space into the SAME super-blocks: the
// first and last super-block of an axis may be partly outside the frame's range (bits of blocks outside it stay 0; a block is a candidate of frame f
// iff it lies in f's range, as in k_select).  k_select_merge then ORs the frames' words per super-block and claims every selected block once.
// A frame whose range has more super-blocks than a row of sbits holds (kVoteCap) claims directly, like k_select; the two mix freely (both OR into bmask).
// ---------------------------------------------------------------------------------------------
#ifndef KB_VOTE
#define KB_VOTE 1            // 0: every batch goes through k_select
#endif
#ifndef KB_VOTE_MIN_FRAMES
#define KB_VOTE_MIN_FRAMES 20 // per batch, k_select against k_select_vote + k_select_merge (profiles/r04_ab_kb_select.txt): 24 / 33 us at 4 frames, 33 / 39 at 8, 51 / 51 at 16, 82 / 75 at 32
#endif
#ifndef KB_VOTE_WGS
#define KB_VOTE_WGS 1792     // workgroups of a k_select_vote launch (all resident: 7 per CU), shared out among the frames
#endif
constexpr unsigned kVoteCap = 1u << 18; // super-blocks per frame in sbits (2 MB per frame; 16.8 M blocks = 1000 m^3 at 5 mm voxels)
static_assert(kSB == 4, "k_select_vote / k_select_merge: super-block = block >> 2");

#ifdef KB_TRACE // development aid (make EXTRA=-DKB_TRACE, tools/kb_trace.sh): per-wave phase times of the last k_select_vote launch, dumped by op_volume_destroy
__device__ unsigned long long g_kb_trace[kSelectGrid * kMaxBatch * 4 * 8];
#define KB_T(K) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr_[K] += now_ - tr_last_; tr_last_ = now_; } while (0)
#define KB_N(K, V) do { tr_[K] += (V); } while (0)
#else
#define KB_T(K) do { } while (0)
#define KB_N(K, V) do { } while (0)
#endif
template <bool FAST>
__global__ __launch_bounds__(256, KB_MINWAVES) void k_select_vote(BatchInv B, CamParams C, VolView V, const uint2* __restrict__ pimg, const float2* __restrict__ ptile,
                                                                  State* st, unsigned long long* __restrict__ sbits, unsigned vote_cap) {
    __shared__ int s_range[6]; // i0, j0, k0, ni, nj, nk
    __shared__ unsigned s_vn[3], s_vsb[3][32]; // the survivors of three consecutive rounds (one barrier per round)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) s_vn[tid] = 0u;
    int f, wslot, wstride; // workgroup -> (frame, slot within the frame): whole frames per XCD, as in k_select
    {
        const int nf = (int)gridDim.y, id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int per_unit = (int)gridDim.x >> 3;
        const int x = id & 7, j = id >> 3;
        const int u = x * nf + j / per_unit;
        f = u >> 3;
        wslot = (u & 7) + 8 * (j % per_unit);
        wstride = (int)gridDim.x;
    }
    const float* M = B.f[f].m;
    const uint2* img = pimg + (size_t)f * C.width * C.height;
    const int tw = tiles_w(C.width), th = tiles_h(C.height);
    const float2* tiles = ptile + (size_t)f * tw * th;
#ifdef KB_TRACE
    unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last_ = __builtin_amdgcn_s_memtime();
#endif
    if (wave == 0) frame_candidate_range(st, C, f, lane, wslot == 0, s_range);
    __syncthreads();
    KB_T(0);
    const int i0 = s_range[0], j0 = s_range[1], k0 = s_range[2];
    int ni = s_range[3], nj = s_range[4], nk = s_range[5];
    unsigned long long ncand = (unsigned long long)((long long)ni * nj * nk);
    if (ni > 4096 || nj > 4096 || nk > 4096) { // > 160 m at 5 mm: treat as a bad frame, select nothing
        if (wslot == 0 && tid == 0) atomicOr(&st->overflow, 4u);
        ncand = 0; ni = nj = nk = 0;
    }
    if (wslot == 0 && tid == 0) st->n_cand[f] = ncand;
    const float cube_res = C.res * 8.0f; // CubeHandler.cpp:164
    const float half = C.res / 2;        // VoxelCube.h:47
    const float o_lo = 0.0f * C.res + half, o_hi = 7.0f * C.res + half; // VoxelCentroidOffSet of x = 0 / 7
    const bmask_t fbit = (bmask_t)1 << f;
    const int oi = (i0 >> 2) * kSB, oj = (j0 >> 2) * kSB, ok = (k0 >> 2) * kSB; // first block of super-block 0 (<= the first block of the range)
    const unsigned nsi = ni > 0 ? (unsigned)(i0 + ni - oi + kSB - 1) / kSB : 0u, nsj = nj > 0 ? (unsigned)(j0 + nj - oj + kSB - 1) / kSB : 0u,
                   nsk = nk > 0 ? (unsigned)(k0 + nk - ok + kSB - 1) / kSB : 0u;
    const unsigned n_super = nsi * nsj * nsk; // (<= 1026^3 < 2^32)
    const bool vote = n_super <= vote_cap; // (vote_cap <= kVoteCap, a row of sbits)
    unsigned long long* bits = sbits + (size_t)f * kVoteCap;
    if (wslot == 0 && tid == 0) { // for k_select_merge: the super-blocks this frame's words are laid out over (extent 0: it has none)
        int* r = st->sel_rng[f];
        r[0] = oi >> 2; r[1] = oj >> 2; r[2] = ok >> 2;
        r[3] = vote ? (int)nsi : 0; r[4] = vote ? (int)nsj : 0; r[5] = vote ? (int)nsk : 0;
        r[6] = !vote && n_super != 0u; // this frame claims directly: the merge step must expect batch masks that are already set
    }
    // A round = 32 super-blocks: every wave runs the coarse test of 8 of them (8 lanes per super-block, one per corner), the survivors of the four waves
    // are pooled in LDS and dealt out again for the exact test (one wave per super-block, one lane per block) -- a wave's own 8 super-blocks hold anything
    // from 0 to 8 survivors.  The pool of round r + 2 is emptied while round r runs, so one barrier per round is enough.
    // The 32 super-blocks of a round are spread evenly over the range (slot s of round r = super-block s * n_rounds + r), not adjacent: survivors come in
    // clusters -- a round of 32 neighbours has anything from 0 to 32 of them, and the busiest workgroup decided the kernel's length.
    // (Rounds drawn from a per-frame counter instead of the fixed stride: measured, no gain -- a workgroup has two rounds, the draw for the second is
    // made before the first one's weight is known.)
    const unsigned n_rounds = (n_super + 31u) / 32u;
    const float inv_nsk = 1.0f / (float)nsk, inv_nsj = 1.0f / (float)nsj;
    auto div_small = [](unsigned a, unsigned d, float inv_d) { // floor(a / d) for a < 2^22: the float quotient is off by one at most
        unsigned q = (unsigned)((float)a * inv_d);
        const unsigned r = q * d;
        if (r > a) --q; else if (a - r >= d) ++q;
        return q;
    };
    unsigned vc = 0;
    for (unsigned round = (unsigned)wslot; round < n_rounds; round += (unsigned)wstride, vc = vc == 2u ? 0u : vc + 1u) {
        {
            const unsigned sb = (unsigned)(tid >> 3) * n_rounds + round;
            const int corner = lane & 7;
            bool survive = false;
            if (sb < n_super) {
                const unsigned q1 = vote ? div_small(sb, nsk, inv_nsk) : sb / nsk, q2 = vote ? div_small(q1, nsj, inv_nsj) : q1 / nsj;
                const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
                // first and last block of the super-block inside the range, per axis
                const int bi0 = max(oi + si * kSB, i0), bj0 = max(oj + sj * kSB, j0), bk0 = max(ok + sk * kSB, k0);
                const int bi1 = min(oi + si * kSB + kSB - 1, i0 + ni - 1), bj1 = min(oj + sj * kSB + kSB - 1, j0 + nj - 1), bk1 = min(ok + sk * kSB + kSB - 1, k0 + nk - 1);
                survive = superblock_survives(C, M, tiles, tw, corner, bi0, bi1, bj0, bj1, bk0, bk1, cube_res, o_lo, o_hi);
                if (corner == 0) {
                    if (survive) s_vsb[vc][atomicAdd(&s_vn[vc], 1u)] = sb;
                    else if (vote) bits[sb] = 0ull; // dropped as a whole: no block of it is selected
                }
            }
        }
        KB_T(1); KB_N(4, 1);
        __syncthreads();
        const unsigned n_todo = s_vn[vc];
        if (tid == 0) s_vn[vc == 0u ? 2u : vc - 1u] = 0u; // the pool of the round after the next (its last readers have passed the barrier above)
        KB_T(3);
        // ---- exact test: one lane per block of a surviving super-block
        for (unsigned sv = (unsigned)wave; sv < n_todo; sv += 4u) {
            KB_N(5, 1);
            const unsigned sb = s_vsb[vc][sv];
            const unsigned q1 = vote ? div_small(sb, nsk, inv_nsk) : sb / nsk, q2 = vote ? div_small(q1, nsj, inv_nsj) : q1 / nsj;
            const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
            const int bi = oi + si * kSB + (lane >> 4), bj = oj + sj * kSB + ((lane >> 2) & 3), bk = ok + sk * kSB + (lane & 3);
            bool selected = false;
            if (bi >= i0 && bi < i0 + ni && bj >= j0 && bj < j0 + nj && bk >= k0 && bk < k0 + nk) { // a candidate of this frame
                int pmax;
                selected = block_selected<FAST>(C, M, img, (float)bi * cube_res, (float)bj * cube_res, (float)bk * cube_res, o_lo, o_hi, pmax);
            }
            if (vote) { // the frame's word for this super-block
                const unsigned long long word = __ballot(selected);
                if (lane == 0) bits[sb] = word;
                KB_T(2);
                continue;
            }
            // (a range too large for sbits: claim directly)
            int slot = -1;
            if (selected) {
                if (!key_in_range(bi, bj, bk)) {
                    atomicOr(&st->overflow, 8u);
                } else {
                    bool created;
                    const int ts = table_claim(V, st, bi, bj, bk, &created);
                    if (ts >= 0 && atomicOr(&V.bmask[ts], fbit) == (bmask_t)0) slot = ts;
                }
            }
            const unsigned long long got = __ballot(slot >= 0);
            unsigned base = 0;
            if (lane == 0 && got) base = atomicAdd(&st->n_list[0], (unsigned)__popcll(got));
            base = __shfl(base, 0, 64);
            if (slot >= 0) {
                const unsigned pos = base + (unsigned)__popcll(got & ((1ULL << lane) - 1ULL));
                if (pos < V.max_blocks) V.blist[pos] = slot;
            }
        }
    }
#ifdef KB_TRACE
    if (lane == 0) for (int k = 0; k < 8; ++k) g_kb_trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + k] = tr_[k];
#endif
}

// ---------------------------------------------------------------------------------------------
// KB, second step of a voting batch.  The frames' words are laid out over their own ranges, but on the same absolute super-block grid, so the words of
// different frames for one super-block can be put side by side: one wave per super-block S, lane e fetches frame e's word for S (0 where S is outside e's
// range) -- one round trip for the whole batch.  The 32 x 64 bit matrix is transposed with one ballot per selected block (lane = block gets the mask of the
// frames that selected it), each block is claimed once, its complete mask ORed into its batch mask, and -- unless a frame on the direct path listed it
// first -- it joins the batch list.  The walk covers the bounding range of the frames' ranges; if the frames lie so far apart that this has more
// super-blocks than the frames' words together, the words are walked instead (frame by frame) and the wave of the LOWEST frame whose word for S is not 0
// deals with S.  A workgroup handles kMergeWords consecutive super-blocks and appends their blocks as ONE segment in super-block / lane order: the batch
// list comes out in runs of spatially adjacent blocks (super-blocks k fastest), which k_integrate rewards -- its workgroups draw consecutive entries, and
// neighbours gather from the same image lines at the same time.
// ---------------------------------------------------------------------------------------------
#ifndef KB_MERGE_WORDS
#define KB_MERGE_WORDS 8
#endif
constexpr int kMergeWords = KB_MERGE_WORDS; // = waves per workgroup
#ifndef KB_MERGE_GRID
#define KB_MERGE_GRID 2048
#endif
__global__ __launch_bounds__(64 * kMergeWords) void k_select_merge(VolView V, State* st, const unsigned long long* __restrict__ sbits, int nf) {
    __shared__ int s_r[kMaxBatch][6];          // first super-block (absolute) and extent in super-blocks of every frame's words
    __shared__ unsigned s_pre[kMaxBatch + 1];  // words before frame f (only for the walk over the words)
    __shared__ unsigned s_wc[kMergeWords], s_wp[kMergeWords], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this lane's frame (lane < nf; every wave loads the 32 rows itself: no barrier in the usual case), for the gathers below
    int e0 = 0, e1 = 0, e2 = 0, en0 = 0, en1 = 0, en2 = 0, direct = 0;
    if (lane < nf) {
        const int* r = st->sel_rng[lane];
        e0 = r[0]; e1 = r[1]; e2 = r[2]; en0 = r[3]; en1 = r[4]; en2 = r[5]; direct = r[6];
    }
    const bool any_direct = __ballot(direct != 0) != 0ull;
    unsigned total = (unsigned)en0 * (unsigned)en1 * (unsigned)en2; // this frame's words (<= kVoteCap), then all frames' (<= 64 x 2^18)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    // the bounding range of the frames' words (every wave computes it: six 64-lane reductions)
    const bool has_words = en0 > 0 && en1 > 0 && en2 > 0;
    int lo0 = has_words ? e0 : INT_MAX, lo1 = has_words ? e1 : INT_MAX, lo2 = has_words ? e2 : INT_MAX;
    int hi0 = has_words ? e0 + en0 : INT_MIN, hi1 = has_words ? e1 + en1 : INT_MIN, hi2 = has_words ? e2 + en2 : INT_MIN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo0 = min(lo0, __shfl_xor(lo0, o, 64)); lo1 = min(lo1, __shfl_xor(lo1, o, 64)); lo2 = min(lo2, __shfl_xor(lo2, o, 64));
        hi0 = max(hi0, __shfl_xor(hi0, o, 64)); hi1 = max(hi1, __shfl_xor(hi1, o, 64)); hi2 = max(hi2, __shfl_xor(hi2, o, 64));
    }
    const unsigned un0 = (unsigned)hi0 - (unsigned)lo0, un1 = (unsigned)hi1 - (unsigned)lo1, un2 = (unsigned)hi2 - (unsigned)lo2; // (meaningless without words: total == 0)
    const unsigned long long n_union = total ? (unsigned long long)un0 * (unsigned long long)un1 * (unsigned long long)un2 : 0ull;
    const bool by_union = n_union <= (unsigned long long)total; // (the usual case: consecutive frames of one camera)
    const unsigned n_units = by_union ? (unsigned)n_union : total;
    if (!by_union) { // the walk over the words needs to know where a frame's words start
        if (tid < nf) {
#pragma unroll
            for (int c = 0; c < 6; ++c) s_r[tid][c] = st->sel_rng[tid][c];
        }
        __syncthreads();
        if (tid == 0) {
            s_pre[0] = 0u;
            for (int f = 0; f < nf; ++f) s_pre[f + 1] = s_pre[f] + (unsigned)s_r[f][3] * (unsigned)s_r[f][4] * (unsigned)s_r[f][5];
        }
        __syncthreads();
    }

    for (unsigned run = blockIdx.x; (unsigned long long)run * kMergeWords < n_units; run += gridDim.x) {
        const unsigned g = run * kMergeWords + (unsigned)wave;
        int slot = -1;
        if (g < n_units) {
            int S0, S1, S2, f = -1; // the super-block (absolute); per-word walk: the frame whose word this is
            if (by_union) {
                const unsigned q1 = g / un2, q2 = q1 / un1;
                S0 = lo0 + (int)q2; S1 = lo1 + (int)(q1 - q2 * un1); S2 = lo2 + (int)(g - q1 * un2);
            } else {
                f = (int)__popcll(__ballot(lane < nf && s_pre[lane + 1] <= g)); // frames whose words end at or before g (kMaxBatch <= 64 lanes)
                const unsigned sb = g - s_pre[f];
                const int* r = s_r[f];
                const unsigned nsj = (unsigned)r[4], nsk = (unsigned)r[5];
                const unsigned q1 = sb / nsk, q2 = q1 / nsj;
                S0 = r[0] + (int)q2; S1 = r[1] + (int)(q1 - q2 * nsj); S2 = r[2] + (int)(sb - q1 * nsk);
            }
            // lane e: frame e's word for S
            unsigned long long w = 0ull;
            const int d0 = S0 - e0, d1 = S1 - e1, d2 = S2 - e2;
            if ((unsigned)d0 < (unsigned)en0 && (unsigned)d1 < (unsigned)en1 && (unsigned)d2 < (unsigned)en2)
                w = sbits[(size_t)lane * kVoteCap + (((unsigned)d0 * (unsigned)en1 + (unsigned)d1) * (unsigned)en2 + (unsigned)d2)];
            const unsigned long long voters = __ballot(w != 0ull);
            if (voters != 0ull && (by_union || (int)__builtin_ctzll(voters) == f)) {
                // transpose: which blocks are selected at all, then one ballot per selected block
                unsigned long long any = w;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o, 64);
                bmask_t m = (bmask_t)0;
                for (unsigned long long rem = any; rem != 0ull; rem &= rem - 1ull) {
                    const int l = (int)__builtin_ctzll(rem);
                    const unsigned long long col = __ballot((w >> l) & 1ull);
                    if (lane == l) m = (bmask_t)col;
                }
                if (m != (bmask_t)0) {
                    const int bi = S0 * kSB + (lane >> 4), bj = S1 * kSB + ((lane >> 2) & 3), bk = S2 * kSB + (lane & 3);
                    if (!key_in_range(bi, bj, bk)) {
                        atomicOr(&st->overflow, 8u);
                    } else {
                        bool created;
                        const int ts = table_claim(V, st, bi, bj, bk, &created); // table slot; KC translates it
                        if (ts >= 0) {
                            // this wave is the only one that sees this block -- unless a frame claims directly, then the batch mask tells who listed it
                            if (!any_direct) { V.bmask[ts] = m; slot = ts; }
                            else if (atomicOr(&V.bmask[ts], m) == (bmask_t)0) slot = ts;
                        }
                    }
                }
            }
        }
        const unsigned long long got = __ballot(slot >= 0);
        if (lane == 0) s_wc[wave] = (unsigned)__popcll(got);
        if (__syncthreads_or(got != 0ull) == 0) continue; // nothing selected in these super-blocks
        if (tid == 0) {
            unsigned n = 0;
            for (int w = 0; w < kMergeWords; ++w) { s_wp[w] = n; n += s_wc[w]; }
            s_base = atomicAdd(&st->n_list[0], n);
        }
        __syncthreads();
        if (slot >= 0) {
            const unsigned pos = s_base + s_wp[wave] + (unsigned)__popcll(got & ((1ULL << lane) - 1ULL));
            if (pos < V.max_blocks) V.blist[pos] = slot;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// KC: Integrator::IntegrateImage (Integrator.cpp:36-94) for all frames of the batch (k_integrate below).
// ---------------------------------------------------------------------------------------------
typedef unsigned int kc_v2u __attribute__((ext_vector_type(2)));

// PLAIN: the volume's content has only ever been written by this kernel since it was created / cleared (no upload, merge,
// resampling or file in between; the host tracks it).  Then every stored voxel is either the default or a running mean
// of finite in-band observations: weights are integers >= 1, colours are means of byte/255 values (non-negative, so their
// numerators w*c + n never cancel), and the update can be evaluated
//   * branch-free: an invalid voxel (TSDFVoxel::IsValid false) is the valid formula with weight 0 --
//     (0*s + new)/(0 + 1) = new exactly for finite s -- instead of a second code path with five selects;
//   * with ONE refined reciprocal of wsum shared by the four quotients (the compiler's own v_rcp + FMA sequence, spelled
//     out as in project_uv): bit-identical to the IEEE division whenever v_div_scale would not rescale, i.e. for
//     wsum in [1, 2^25] and a numerator that is 0 or >= 2^-100 in magnitude.  Colour numerators are 0 or >= 2^-32
//     (no cancellation); the sdf numerator CAN cancel to something tiny, so it alone is guarded: a lane whose
//     |w*s + new| is non-zero and below 2^-100 takes the plain division (never, in practice).
// Without PLAIN (arbitrary uploaded data: NaN, infinities, denormals, fractional weights) the update is the reference's
// two-branch form with four true divisions.
// ---------------------------------------------------------------------------------------------
// KC.  One workgroup per block of the batch list; the resident workgroups draw blocks from per-XCD counters.  Every voxel is
// read ONCE, every frame that selected the block is applied to it in frame order in registers (bit-identical to the
// reference's frame-by-frame running mean) and it is written once -- HBM traffic per voxel drops from 40 B per frame to
// 40 B per batch; block ownership is exclusive, so the read-modify-write needs no atomics.  A thread owns ZT voxels
// of one (x, y) column of the block (z = zg*ZT .. zg*ZT + ZT-1), a wave owns ZT z-slices, a workgroup of 8/ZT waves owns
// the block.  What that buys, per voxel and frame:
//   * everything that is uniform over the wave -- the frame's bit test, the three s_load_dwordx4 of its pose rows, the buffer
//     resource of its packed image, the loop control -- is paid once per ZT voxels instead of once per voxel (a third of the
//     issue slots of a one-voxel-per-thread kernel go to scalar and branch instructions, profiles/r03_issue_costs.json);
//   * the partial sums M[r][0]*px + M[r][1]*py of the three pose rows depend on x and y only and are shared by the ZT voxels
//     (the same two rounded products and one rounded sum the reference forms for each of them: bit-identical);
//   * ZT independent dependency chains per thread hide the VALU and gather latencies that eight waves per SIMD hid before,
//     so the kernel runs at a lower occupancy with a larger register budget.
// A plane row of a z-slice is still one 256-byte wave access.  Frames are applied in ascending order; the gathers of the
// NEXT selected frame are issued before the current frame's updates (two record sets, the frame loop unrolled by two).
// ---------------------------------------------------------------------------------------------
template <bool PLAIN>
__device__ __forceinline__ void voxel_update(float& s, float& w, float& c0, float& c1, float& c2, float new_sdf, unsigned rgba, const float* s_c255) {
    const float n0 = s_c255[rgba & 0xffu], n1 = s_c255[(rgba >> 8) & 0xffu], n2 = s_c255[(rgba >> 16) & 0xffu];
    if (PLAIN) {
        // TSDFVoxel::IsValid (TSDFVoxel.h:75-78) false -> weight 0 in the same formula (see the PLAIN comment above)
        const float wv = (s >= 1 || w <= 0) ? 0.0f : w;
        const float wsum = wv + 1.0f;
        float y = __builtin_amdgcn_rcpf(wsum);
        const float e = __builtin_fmaf(-wsum, y, 1.0f);
        y = __builtin_fmaf(e, y, y);
        const float ns = wv * s + 1.0f * new_sdf;
        const float m0 = wv * c0 + 1.0f * n0, m1 = wv * c1 + 1.0f * n1, m2 = wv * c2 + 1.0f * n2;
        float qs = div_shared_rcp(ns, wsum, y);
        const bool tiny = !(fabsf(ns) >= 0x1p-100f) && ns != 0.0f;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tiny) != 0ull, 0)) {
            if (tiny) qs = ns / wsum;
        }
        s = qs;
        c0 = div_shared_rcp(m0, wsum, y);
        c1 = div_shared_rcp(m1, wsum, y);
        c2 = div_shared_rcp(m2, wsum, y);
        w = wsum;
    } else if (!(s >= 1 || w <= 0)) { // TSDFVoxel::IsValid (TSDFVoxel.h:75-78)
        const float wsum = w + 1.0f;  // TSDFVoxel::operator+ with other = (new_sdf, 1.0, c) (TSDFVoxel.h:24-39)
        s = (w * s + 1.0f * new_sdf) / wsum;
        c0 = (w * c0 + 1.0f * n0) / wsum;
        c1 = (w * c1 + 1.0f * n1) / wsum;
        c2 = (w * c2 + 1.0f * n2) / wsum;
        w = wsum;
    } else {
        s = new_sdf; w = 1.0f; c0 = n0; c1 = n1; c2 = n2;
    }
}

// SUMF (opt-in, OP_VOLUME_UPDATE_SUM_FORM): the frames of the batch are not applied one by one.  Per voxel the kernel keeps the NUMBER of in-band
// observations of the batch, the sum of their sdf values and the sums of their colour bytes (exact integers), and forms the weighted mean with the
// stored voxel ONCE per batch: s' = (w s + sum sdf) / (w + n), c' = (w c + sum bytes / 255) / (w + n), w' = w + n -- TSDFVoxel::operator+
// (TSDFVoxel.h:24-39) applied n times in exact arithmetic.  Same blocks, same pixels, same weights (integers); sdf and colour differ from the
// frame-by-frame running mean by float rounding only (a few 1e-7 relative; north_star's bar is 1e-4).  Per voxel and frame the ~35 instructions
// of the exactly rounded update shrink to 7 (two selects, one float add, two byte-pair adds with their masks).
#ifdef KC_TRACE // development aid (make EXTRA=-DKC_TRACE): where a workgroup of the last k_integrate launch spent its time, per wave; dumped by op_volume_destroy
__device__ unsigned long long g_kc_trace[4096 * 4 * 8];
#define KC_T(K) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); kt_[K] += now_ - kt_last_; kt_last_ = now_; } while (0)
#define KC_N(K, V) do { kt_[K] += (V); } while (0)
#else
#define KC_T(K) do { } while (0)
#define KC_N(K, V) do { } while (0)
#endif
template <bool FAST, bool PLAIN, int ZT, bool SUMF = false>
__global__ __launch_bounds__(512 / ZT, (SUMF ? KC_SUM_MIN_WAVES : KC_COL_MIN_WAVES)) void k_integrate(BatchInv B, CamParams C, VolView V, const uint2* __restrict__ pimg, State* st,
                                                                            int n_frames, unsigned long long* __restrict__ upd_partial,
                                                                            unsigned long long* __restrict__ sel_partial, unsigned long long* __restrict__ chg_partial,
                                                                            unsigned plain_from) {
    constexpr int kWaves = 8 / ZT;            // waves per workgroup = z-groups per block
    __shared__ unsigned s_cnt[kWaves][2];
    __shared__ float s_c255[256];             // (float)b / 255.0f for every byte (Integrator.cpp:78), correctly rounded once
    __shared__ unsigned s_next[2];
    const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#ifdef KC_TRACE
    unsigned long long kt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kt_last_ = t_in;
#endif
    // KB has consumed the frames' bounding accumulators: back to the identity for the next batch (also when poisoned)
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < (unsigned)(kMaxBatch * kAccSlots * 8); k += gridDim.x * blockDim.x) (&st->acc[0][0][0])[k] = 0u;
    if (st->overflow & 3u) return; // pool / table exhausted in this or an earlier batch: nothing is fused, the host replays
    const int tid = threadIdx.x, lane = tid & 63, zg = tid >> 6;
    if (!SUMF) for (int k = tid; k < 256; k += blockDim.x) s_c255[k] = (float)k / 255.0f;
    const unsigned npix = (unsigned)(C.width * C.height);
    const float half = C.res / 2;
    // VoxelCentroidOffSet (VoxelCube.h:48-61): x*res + half with x = lane & 7, y = lane >> 3
    const float ox = (float)(lane & 7) * C.res + half;
    const float oy = (float)(lane >> 3) * C.res + half;
    const float __attribute__((address_space(4)))* kargs =
        (const float __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr(); // BatchInv B = offset 0 of the kernarg segment
    (void)B;
    unsigned upd = 0, sel = 0, chg = 0, nblk = 0;
#ifndef KC_CHUNK_LOG2
#define KC_CHUNK_LOG2 5
#endif
    // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own 4 MiB L2; blocks that gather the same pixels should meet in one L2)
    // and dynamic scheduling (blocks differ in work: 1..32 frames touch them; the workgroups of an XCD DRAW list positions from one counter,
    // the next one before the current block is processed so that the atomic's round trip is hidden).  Two ways of dealing the batch's blocks
    // to the eight draw counters:
    //  * full batches (>= KC_STEAL_MIN_FRAMES frames): XCD x starts on the x-th contiguous eighth of the list (with -DKC_BANDS=1 on list x, see
    //    kBands); the eighths hold the same number of blocks but not the same work, so a workgroup whose share is exhausted reads all eight
    //    counters (one round trip) and goes on with the share that has the most left.
    //    Per 32-frame launch: eighths alone 689 us, with stealing 627-631 us;
    //  * short batches: list 0, chunks of 32 blocks dealt round-robin (every XCD a sample of the whole list), no stealing: for ONE frame per
    //    launch, where the kernel is HBM-bound and the work per block uniform, 79 us against 95 us with stealing (its last look costs a short
    //    launch more than it can win).  Measured crossover (tools/prof_driver.bin batch=N under the tracer, stealing vs chunks): 8 frames
    //    207 vs 194 us, 16: 360 vs 356, 24: 524 vs 527, 32: 677 vs 687.
    const bool eighths = n_frames >= KC_STEAL_MIN_FRAMES;
    const bool lists = eighths && KC_BANDS != 0;              // one list per share
    constexpr unsigned kChunk = 1u << KC_CHUNK_LOG2;
    const unsigned n0 = st->n_list[0] < V.max_blocks ? st->n_list[0] : V.max_blocks;
    const unsigned n_chunks = (n0 + kChunk - 1u) >> KC_CHUNK_LOG2;
    const unsigned per0 = eighths ? (n0 + (unsigned)kKcShares - 1u) / (unsigned)kKcShares : ((n_chunks + (unsigned)kKcShares - 1u) / (unsigned)kKcShares) << KC_CHUNK_LOG2;
    unsigned xcd = blockIdx.x % (unsigned)kKcShares;          // the share this workgroup draws from: its own first
    for (;;) {
    unsigned* ctr = &st->kc_next[xcd * 16u];
    if (tid == 0) s_next[0] = atomicAdd(ctr, 1u);
    // positions j < per_xcd of share xcd; its blocks are list[j] (one list per share) or positions of list 0
    unsigned per_xcd = per0, n = n0;
    const int* list = V.blist;
    if (lists) {
        const unsigned nl = st->n_list[xcd];
        per_xcd = n = nl < V.max_blocks ? nl : V.max_blocks;
        list = V.blist + (size_t)xcd * V.max_blocks;
    }
    __syncthreads();
    unsigned slot = 0u;
    for (unsigned j = s_next[0]; j < per_xcd;) {
        if (tid == 0) s_next[slot ^ 1u] = atomicAdd(ctr, 1u);
        const unsigned b = lists ? j : (eighths ? xcd * per_xcd + j : (((j >> KC_CHUNK_LOG2) * (unsigned)kKcShares + xcd) << KC_CHUNK_LOG2) + (j & (kChunk - 1u)));
        KC_T(0);
        const int tslot = b < n ? list[b] : -1;
        const int idx = tslot >= 0 ? V.tvals[tslot] : -1; // idx < 0: pool overflow (reported through st->overflow)
        KC_N(6, 1);
        if (idx >= 0) {
            const bmask_t mask = V.bmask[tslot];
            if (zg == 0) { sel += mask_popc(mask); ++nblk; }
            const int kx = V.keys[3 * idx], ky = V.keys[3 * idx + 1], kz = V.keys[3 * idx + 2];
            float* vox = V.pool + (size_t)idx * kBlockFloats + (zg * ZT) * 64 + lane;
            float s[ZT], w[ZT], c0[ZT], c1[ZT], c2[ZT], pz[ZT];
#pragma unroll
            for (int z = 0; z < ZT; ++z) {
                // (the sum form needs the stored voxel only after the frames: it is loaded there, and the registers are free until then)
                if (!SUMF) { s[z] = vox[z * 64]; w[z] = vox[kVox + z * 64]; c0[z] = vox[2 * kVox + z * 64]; c1[z] = vox[3 * kVox + z * 64]; c2[z] = vox[4 * kVox + z * 64]; }
                // GetGlobalPoint (VoxelCube.h:75-80): Point3(id) * CUBE_SIZE * VoxelResolution + offset
                pz[z] = ((float)kz * 8.0f) * C.res + ((float)(zg * ZT + z) * C.res + half);
            }
            const float px = ((float)kx * 8.0f) * C.res + ox;
            const float py = ((float)ky * 8.0f) * C.res + oy;
            unsigned changed = 0u;
            // sum form: sdf sum, byte sums of colour channels 0 and 2 in the two halves of one word, of channel 1 in the low half of another whose
            // high half counts the observations (<= 64 frames x 255 < 2^16)
            float ssum[ZT];
            unsigned acc02[ZT], acc1n[ZT];
#pragma unroll
            for (int z = 0; z < ZT; ++z) { ssum[z] = 0.0f; acc02[z] = 0u; acc1n[z] = 0u; }
            // one selected frame: projections of the thread's ZT voxels and their {depth, rgba} gathers
            auto project = [&](int f, kc_v2u (&rec)[ZT], float (&zc)[ZT]) {
                int fo = f;
                asm volatile("" : "+s"(fo));
                const float __attribute__((address_space(4)))* M = kargs + fo * 12;
                // (frame base in 32 bits: check_cam keeps kMaxBatch x npix x 8 below 2^32)
                const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)pimg + (unsigned)fo * (npix * 8u)), 0, (int)(npix * 8u), 0x00020000);
                const float a0 = M[0] * px + M[1] * py, a1 = M[4] * px + M[5] * py, a2 = M[8] * px + M[9] * py;
#pragma unroll
                for (int z = 0; z < ZT; ++z) {
                    const float q0 = (a0 + M[2] * pz[z]) + M[3] * 1.0f;
                    const float q1 = (a1 + M[6] * pz[z]) + M[7] * 1.0f;
                    const float q2 = (a2 + M[10] * pz[z]) + M[11] * 1.0f;
                    zc[z] = q2;
                    const int pix = project_pixel<FAST>(C, q0, q1, q2); // off-image: pixel -1 = an offset the buffer answers with zeros
                    rec[z] = __builtin_amdgcn_raw_buffer_load_b64(frame, pix * 8, 0, 0);
                }
            };
            // PLAIN: every block of the volume was written by this kernel only.  Otherwise (the volume has seen an upload, a merge, a
            // sum-form unpack or a file): the blocks that existed then (pool slots below plain_from) hold arbitrary data and take the
            // general update; blocks allocated since are this kernel's own and keep the fast one.  Uniform per block.
            const bool plain_block = PLAIN || (unsigned)idx >= plain_from;
            auto apply = [&](const kc_v2u (&rec)[ZT], const float (&zc)[ZT], auto plain_c) {
                constexpr bool kPlain = decltype(plain_c)::value;
#pragma unroll
                for (int z = 0; z < ZT; ++z) {
                    const float d = __uint_as_float(rec[z].x); // off-image pixels carry d == 0 -> skipped like `continue`
                    const float new_sdf = d - zc[z];
                    // Integrator.cpp:70,74 (d > 0 and |sdf| < truncation) as ONE compare and one divergent region: an absent
                    // observation takes the place of an out-of-band one
                    const float band = d > 0 ? fabsf(new_sdf) : C.trunc;
                    const bool hit = band < C.trunc;
                    upd += hit ? 1u : 0u;                              // per lane; summed over the wave at the end
                    if (SUMF) { // branch-free: an observation that misses adds zeros
                        ssum[z] += hit ? new_sdf : 0.0f;
                        const unsigned t = hit ? rec[z].y : 0u;           // byte 3 of a packed pixel is 1 (k_prepare_frames): the count
                        acc02[z] += t & 0x00ff00ffu;
                        acc1n[z] += (t >> 8) & 0x00ff00ffu;
                    } else if (hit) {
                        changed |= 1u << z;
                        voxel_update<kPlain>(s[z], w[z], c0[z], c1[z], c2[z], new_sdf, rec[z].y, s_c255);
                    }
                }
            };
            kc_v2u recA[ZT], recB[ZT];
            float zcA[ZT], zcB[ZT];
#ifdef KC_TRACE
            { float keep_ = px + py + pz[0]; if (!SUMF) keep_ += s[0]; asm volatile("" :: "v"(keep_)); KC_T(1); } // (the block's metadata and voxels have arrived)
#endif
            auto frames = [&](auto plain_c) {
                bmask_t m = mask;                                 // wave-uniform
                if (!m) return;
                int f = mask_ctz(m); m &= m - 1u;
                project(f, recA, zcA);
                for (;;) {
                    const bool more1 = m != 0u;
                    if (more1) { f = mask_ctz(m); m &= m - 1u; project(f, recB, zcB); }
                    apply(recA, zcA, plain_c);
                    if (!more1) break;
                    const bool more2 = m != 0u;
                    if (more2) { f = mask_ctz(m); m &= m - 1u; project(f, recA, zcA); }
                    apply(recB, zcB, plain_c);
                    if (!more2) break;
                }
            };
            if (SUMF || PLAIN || plain_block) frames(std::true_type{}); else frames(std::false_type{});
            KC_T(2); KC_N(7, mask_popc(mask));
            if (SUMF) {
#pragma unroll
                for (int z = 0; z < ZT; ++z)
                    if (acc1n[z] >> 16) { s[z] = vox[z * 64]; w[z] = vox[kVox + z * 64]; c0[z] = vox[2 * kVox + z * 64]; c1[z] = vox[3 * kVox + z * 64]; c2[z] = vox[4 * kVox + z * 64]; }
#pragma unroll
                for (int z = 0; z < ZT; ++z) {
                    const unsigned cnt = acc1n[z] >> 16;
                    if (cnt) {
                        changed |= 1u << z;
                        // TSDFVoxel::operator+ (TSDFVoxel.h:24-39) for the batch's observations at once; an invalid voxel (IsValid false,
                        // :75-78) is replaced by their mean, as the first observation would have replaced it
                        const bool valid = !(s[z] >= 1 || w[z] <= 0);
                        const float wv = valid ? w[z] : 0.0f, nf = (float)cnt, wsum = wv + nf;
                        const float b0 = (float)(acc02[z] & 0xffffu) / 255.0f, b1 = (float)(acc1n[z] & 0xffffu) / 255.0f, b2 = (float)(acc02[z] >> 16) / 255.0f;
                        s[z] = ((valid ? wv * s[z] : 0.0f) + ssum[z]) / wsum;
                        c0[z] = ((valid ? wv * c0[z] : 0.0f) + b0) / wsum;
                        c1[z] = ((valid ? wv * c1[z] : 0.0f) + b1) / wsum;
                        c2[z] = ((valid ? wv * c2[z] : 0.0f) + b2) / wsum;
                        w[z] = wsum;
                    }
                }
            }
#pragma unroll
            for (int z = 0; z < ZT; ++z)
                if ((changed >> z) & 1u) { vox[z * 64] = s[z]; vox[kVox + z * 64] = w[z]; vox[2 * kVox + z * 64] = c0[z]; vox[3 * kVox + z * 64] = c1[z]; vox[4 * kVox + z * 64] = c2[z]; }
            chg += (unsigned)__popc(changed);
        }
        KC_T(3);
        __syncthreads();                                   // every wave has read the mask; s_next[slot ^ 1] is visible
        KC_T(4);
        if (tslot >= 0 && tid == 0) V.bmask[tslot] = (bmask_t)0; // the owner clears it for the next batch
        slot ^= 1u;
        j = s_next[slot];
    }
    if (!eighths) break;
    // the share is exhausted: look at all the draw counters at once (one round trip) and go on with the share that has the most left
    __syncthreads();                                          // everybody has read the last draw
    if (tid < 64) {
        unsigned left = 0u;
        if (tid < kKcShares) {
            const unsigned c = __hip_atomic_load(&st->kc_next[(unsigned)tid * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned len = per0;
            if (lists) { const unsigned nl = st->n_list[tid]; len = nl < V.max_blocks ? nl : V.max_blocks; }
            left = c < len ? len - c : 0u;
        }
        unsigned key = ((left < 0x7fffffu ? left : 0x7fffffu) << 8) | (unsigned)tid; // most left, ties to the higher share index (any fixed rule)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned x = __shfl_xor(key, o, 64); key = x > key ? x : key; }
        if (tid == 0) s_next[0] = key;
    }
    __syncthreads();
    const unsigned key = s_next[0];
    if ((key >> 8) == 0u) break;                              // nothing left anywhere
    xcd = key & 0xffu;
    __syncthreads();                                          // s_next[0] is written again at the top
    }
#ifdef KC_TRACE
    KC_T(5);
    if (lane == 0 && blockIdx.x < 4096) for (int k = 0; k < 8; ++k) g_kc_trace[((size_t)blockIdx.x * 4 + zg) * 8 + k] = kt_[k];
#endif
    // per-workgroup counters into kPartialGrid slots
    upd = wave_sum(upd); chg = wave_sum(chg);
    if (lane == 0) { s_cnt[zg][0] = upd; s_cnt[zg][1] = chg; }
    __syncthreads();
    if (tid == 0) {
        unsigned t = 0, c = 0;
        for (int k = 0; k < kWaves; ++k) { t += s_cnt[k][0]; c += s_cnt[k][1]; }
        const unsigned slot_c = blockIdx.x % (unsigned)kPartialGrid;
        atomicAdd(&upd_partial[slot_c], (unsigned long long)t);
        atomicAdd(&sel_partial[slot_c], (unsigned long long)sel);
        atomicAdd(&chg_partial[slot_c], (unsigned long long)c);
        atomicAdd(&chg_partial[kPartialGrid + slot_c], (unsigned long long)nblk);
        if (blockIdx.x == 0) { st->stat_frames += (unsigned long long)n_frames; st->stat_launches += 1ull; }
        atomicMax(&st->kc_t[blockIdx.x % (unsigned)kKcTSlots], (unsigned long long)__builtin_amdgcn_s_memtime() - t_in);
    }
}

// ---------------------------------------------------------------------------------------------
// export / import / merge kernels
// ---------------------------------------------------------------------------------------------
// SoA pool -> AoS {sdf,w,c0,c1,c2} x 512 for blocks [first, first+count)
__global__ __launch_bounds__(512) void k_export_aos(const float* __restrict__ pool, size_t first, float* __restrict__ out) {
    const size_t b = blockIdx.x;
    const float* src = pool + (first + b) * kBlockFloats + threadIdx.x;
    float* dst = out + (b * kVox + threadIdx.x) * 5;
#pragma unroll
    for (int p = 0; p < 5; ++p) dst[p] = src[p * kVox];
}

// insert keys; slots[i] receives the TABLE slot of key i, encoded -(slot+2) when newly created; the
// consumers below translate it to the pool slot through tvals (next kernel => visible)
__global__ void k_insert_keys(VolView V, const int* __restrict__ keys, size_t n, int* __restrict__ slots, State* st) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
    if (!key_in_range(x, y, z)) { atomicOr(&st->overflow, 8u); slots[i] = -1; return; }
    bool created;
    const int slot = table_claim(V, st, x, y, z, &created);
    slots[i] = slot < 0 ? -1 : (created ? -(slot + 2) : slot);
}

// AoS voxels -> pool planes for the given slots (SetCubeMap / AddCube + assignment)
__global__ __launch_bounds__(512) void k_import_aos(float* __restrict__ pool, const int* __restrict__ slots,
                                                    const int* __restrict__ tvals, const float* __restrict__ in) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    if (idx <= -2) idx = -(idx + 2);
    idx = tvals[idx]; // table slot -> pool slot
    if (idx < 0) return;
    const float* src = in + ((size_t)blockIdx.x * kVox + threadIdx.x) * 5;
    float* dst = pool + (size_t)idx * kBlockFloats + threadIdx.x;
#pragma unroll
    for (int p = 0; p < 5; ++p) dst[p * kVox] = src[p];
}

// CubeHandler::Merge (CubeHandler.h:145-167): dst block (slots) += src block (TSDFVoxel::operator+,
// general weights), or plain copy when the block was just created in dst.
__global__ __launch_bounds__(512) void k_merge_blocks(float* __restrict__ dpool, const float* __restrict__ spool,
                                                      const int* __restrict__ slots, const int* __restrict__ tvals) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    const bool fresh = idx <= -2;
    if (fresh) idx = -(idx + 2);
    idx = tvals[idx]; // table slot -> pool slot
    if (idx < 0) return;
    const float* a = spool + (size_t)blockIdx.x * kBlockFloats + threadIdx.x; // src block i lives in src pool slot i
    float* t = dpool + (size_t)idx * kBlockFloats + threadIdx.x;
    const float bs = a[0], bw = a[kVox], b0 = a[2 * kVox], b1 = a[3 * kVox], b2 = a[4 * kVox];
    const float tw = t[kVox];
    if (fresh || tw == 0) { // copy / "weight == 0 -> return other"
        t[0] = bs; t[kVox] = bw; t[2 * kVox] = b0; t[3 * kVox] = b1; t[4 * kVox] = b2;
        return;
    }
    if (bw == 0) return;
    const float w = tw + bw;
    if (w != 0) {
        const float ts = t[0], t0 = t[2 * kVox], t1 = t[3 * kVox], t2 = t[4 * kVox];
        t[0] = (tw * ts + bw * bs) / w;
        t[2 * kVox] = (tw * t0 + bw * b0) / w;
        t[3 * kVox] = (tw * t1 + bw * b1) / w;
        t[4 * kVox] = (tw * t2 + bw * b2) / w;
    } else {
        t[0] = 999.0f; t[2 * kVox] = t[3 * kVox] = t[4 * kVox] = -1.0f;
    }
    t[kVox] = w;
}

// K4a: sum-form pack for the RCCL reduce: [w*sdf, w, w*c0, w*c1, w*c2] planes per union key.
__global__ __launch_bounds__(512) void k_pack_sum(VolView V, const int* __restrict__ ukeys, float* __restrict__ out) {
    __shared__ int s_idx;
    if (threadIdx.x == 0) s_idx = table_find(V, ukeys[3 * blockIdx.x], ukeys[3 * blockIdx.x + 1], ukeys[3 * blockIdx.x + 2]);
    __syncthreads();
    const int idx = s_idx;
    float* o = out + (size_t)blockIdx.x * kBlockFloats + threadIdx.x;
    float s = 0, w = 0, c0 = 0, c1 = 0, c2 = 0;
    if (idx >= 0) {
        const float* t = V.pool + (size_t)idx * kBlockFloats + threadIdx.x;
        w = t[kVox];
        if (w > 0) { s = w * t[0]; c0 = w * t[2 * kVox]; c1 = w * t[3 * kVox]; c2 = w * t[4 * kVox]; }
        else w = 0;
    }
    o[0] = s; o[kVox] = w; o[2 * kVox] = c0; o[3 * kVox] = c1; o[4 * kVox] = c2;
}

// K4b: normalise the reduced sums back to mean form into the (re-keyed) volume.
__global__ __launch_bounds__(512) void k_unpack_sum(float* __restrict__ pool, const int* __restrict__ slots,
                                                    const int* __restrict__ tvals, const float* __restrict__ sum) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    if (idx <= -2) idx = -(idx + 2);
    idx = tvals[idx]; // table slot -> pool slot
    if (idx < 0) return;
    const float* a = sum + (size_t)blockIdx.x * kBlockFloats + threadIdx.x;
    float* t = pool + (size_t)idx * kBlockFloats + threadIdx.x;
    const float w = a[kVox];
    if (w > 0) {
        t[0] = a[0] / w; t[kVox] = w; t[2 * kVox] = a[2 * kVox] / w; t[3 * kVox] = a[3 * kVox] / w; t[4 * kVox] = a[4 * kVox] / w;
    } else {
        t[0] = 999.0f; t[kVox] = 0.0f; t[2 * kVox] = t[3 * kVox] = t[4 * kVox] = -1.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// Volume resampling (CubeHandler::Transform / TransformNearest, CubeHandler.h:199-338) and
// GetPointCloud (CubeHandler.cpp:45-69)
// ---------------------------------------------------------------------------------------------
struct Mat4 { float m[16]; };
struct Vox5 { float s, w, c0, c1, c2; };

__device__ __forceinline__ Vox5 default_voxel() { return Vox5{999.0f, 0.0f, -1.0f, -1.0f, -1.0f}; }

// cube_map.find(GetCubeID(p)) + GetVoxel(GetVoxelID(p)) (VoxelCube.h:63-67,81-86); default voxel if absent
__device__ Vox5 fetch_voxel(const VolView& S, int px, int py, int pz) {
    const int cx = px >> 3, cy = py >> 3, cz = pz >> 3; // floor((p + 0.0) / 8)
    const int idx = table_find(S, cx, cy, cz);
    if (idx < 0) return default_voxel();
    const int vid = (px - cx * 8) + (py - cy * 8) * 8 + (pz - cz * 8) * 64;
    const float* t = S.pool + (size_t)idx * kBlockFloats + vid;
    return Vox5{t[0], t[kVox], t[2 * kVox], t[3 * kVox], t[4 * kVox]};
}
// TSDFVoxel::operator*(float) (TSDFVoxel.h:56-67)
__device__ __forceinline__ Vox5 vox_scale(const Vox5& a, float wgt) {
    if (wgt == 0 || a.w == 0) return default_voxel();
    return Vox5{a.s * wgt, a.w * wgt, a.c0 * wgt, a.c1 * wgt, a.c2 * wgt};
}
// TSDFVoxel::add (TSDFVoxel.h:40-51)
__device__ __forceinline__ Vox5 vox_add_direct(const Vox5& a, const Vox5& b) {
    if (a.w == 0) return b;
    if (b.w == 0) return a;
    return Vox5{a.s + b.s, a.w + b.w, a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2};
}
// one stage of ReadVoxelInterpolate (VoxelCube.cpp:17-20):
// ((a * (1 - t)).add(b * t)) / ((1 - t) * (a.weight != 0) + t * (b.weight != 0))
__device__ __forceinline__ Vox5 interp_stage(const Vox5& a, const Vox5& b, float t) {
    if (!(a.w != 0 || b.w != 0)) return default_voxel();
    const Vox5 sum = vox_add_direct(vox_scale(a, 1 - t), vox_scale(b, t));
    const float d = (1 - t) * (float)(a.w != 0) + t * (float)(b.w != 0);
    return vox_scale(sum, 1 / d); // operator/(w) = operator*(1 / w) (TSDFVoxel.h:68-71)
}

// pass 1: AddTransformedCube / AddTransformedCubeNearest (CubeHandler.h:199-241), executed with the
// RESULT's CubePara (alloc_res).  One workgroup per source block.
template <bool NEAREST>
__global__ __launch_bounds__(512) void k_transform_alloc(VolView S, VolView D, State* dst_state, Mat4 T, float alloc_res) {
    const int b = blockIdx.x, vid = threadIdx.x;
    const int kx = S.keys[3 * b], ky = S.keys[3 * b + 1], kz = S.keys[3 * b + 2];
    const float half = alloc_res / 2;
    const float px = ((float)kx * 8.0f) * alloc_res + ((float)(vid & 7) * alloc_res + half);
    const float py = ((float)ky * 8.0f) * alloc_res + ((float)((vid >> 3) & 7) * alloc_res + half);
    const float pz = ((float)kz * 8.0f) * alloc_res + ((float)(vid >> 6) * alloc_res + half);
    const float* M = T.m;
    const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    const float q1 = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    const float q2 = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    const float q3 = ((M[12] * px + M[13] * py) + M[14] * pz) + M[15] * 1.0f;
    const float n0 = NEAREST ? q0 / q3 : q0 / q3 - half, n1 = NEAREST ? q1 / q3 : q1 / q3 - half,
                n2 = NEAREST ? q2 / q3 : q2 / q3 - half;
    const int p0 = (int)floorf(n0 / alloc_res), p1 = (int)floorf(n1 / alloc_res), p2 = (int)floorf(n2 / alloc_res);
    int lx = INT_MIN, ly = INT_MIN, lz = INT_MIN;
#pragma unroll
    for (int k = 0; k < (NEAREST ? 1 : 8); ++k) {
        const int cx = (p0 + (k & 1)) >> 3, cy = (p1 + ((k >> 1) & 1)) >> 3, cz = (p2 + ((k >> 2) & 1)) >> 3;
        if (cx == lx && cy == ly && cz == lz) continue;
        lx = cx; ly = cy; lz = cz;
        if (!key_in_range(cx, cy, cz)) { atomicOr(&dst_state->overflow, 8u); continue; }
        bool created;
        table_claim(D, dst_state, cx, cy, cz, &created); // AddCube
    }
}

// pass 2: every voxel of the result reads the source through trans^-1 (CubeHandler.h:257-294 /
// :312-334) with the SOURCE's CubePara (this->c_para).  One workgroup per result block; result
// voxels are still default, so `voxels[voxel_id] += result` stores `result` (weight == 0 -> other).
template <bool NEAREST>
__global__ __launch_bounds__(512) void k_transform_fill(VolView S, VolView D, Mat4 Tinv, float src_res) {
    const int b = blockIdx.x, vid = threadIdx.x;
    const int kx = D.keys[3 * b], ky = D.keys[3 * b + 1], kz = D.keys[3 * b + 2];
    const float half = src_res / 2;
    const float px = ((float)kx * 8.0f) * src_res + ((float)(vid & 7) * src_res + half);
    const float py = ((float)ky * 8.0f) * src_res + ((float)((vid >> 3) & 7) * src_res + half);
    const float pz = ((float)kz * 8.0f) * src_res + ((float)(vid >> 6) * src_res + half);
    const float* M = Tinv.m;
    const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    const float q1 = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    const float q2 = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    const float q3 = ((M[12] * px + M[13] * py) + M[14] * pz) + M[15] * 1.0f;
    const float n0 = NEAREST ? q0 / q3 : q0 / q3 - half, n1 = NEAREST ? q1 / q3 : q1 / q3 - half,
                n2 = NEAREST ? q2 / q3 : q2 / q3 - half;
    const int p0 = (int)floorf(n0 / src_res), p1 = (int)floorf(n1 / src_res), p2 = (int)floorf(n2 / src_res);
    Vox5 r;
    if (NEAREST) {
        r = fetch_voxel(S, p0, p1, p2);
    } else {
        Vox5 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fetch_voxel(S, p0 + (k & 1), p1 + ((k >> 1) & 1), p2 + ((k >> 2) & 1));
        // ReadVoxelInterpolate (VoxelCube.cpp:6-50)
        const float xw = (n0 - (float)p0 * src_res) / src_res, yw = (n1 - (float)p1 * src_res) / src_res,
                    zw = (n2 - (float)p2 * src_res) / src_res;
        const Vox5 z1 = interp_stage(interp_stage(v[0], v[1], xw), interp_stage(v[2], v[3], xw), yw);
        const Vox5 z2 = interp_stage(interp_stage(v[4], v[5], xw), interp_stage(v[6], v[7], xw), yw);
        r = interp_stage(z1, z2, zw);
    }
    float* t = D.pool + (size_t)b * kBlockFloats + vid;
    t[0] = r.s; t[kVox] = r.w; t[2 * kVox] = r.c0; t[3 * kVox] = r.c1; t[4 * kVox] = r.c2;
}

// GetPointCloud: voxels with weight != 0 and |sdf| < truncation, in the reference's x,y,z loop order
// inside a block.  counts == nullptr: emit using offsets; else count only.
__global__ __launch_bounds__(512) void k_point_cloud(VolView V, float res, float trunc, unsigned* __restrict__ counts,
                                                     const unsigned* __restrict__ offsets, float* __restrict__ xyz,
                                                     float* __restrict__ col) {
    __shared__ unsigned s_w[8];
    const int b = blockIdx.x, o = threadIdx.x;
    const int x = o >> 6, y = (o >> 3) & 7, z = o & 7; // loop nest: x outer, y, z inner
    const int vid = x + y * 8 + z * 64;
    const float* t = V.pool + (size_t)b * kBlockFloats + vid;
    const float sdf = t[0], w = t[kVox];
    const bool ok = w != 0 && fabsf(sdf) < trunc;
    const unsigned long long m = __ballot(ok);
    const int lane = o & 63, wave = o >> 6;
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    if (counts) {
        if (o == 0) { unsigned tot = 0; for (int k = 0; k < 8; ++k) tot += s_w[k]; counts[b] = tot; }
        return;
    }
    if (!ok) return;
    unsigned rank = __popcll(m & ((1ULL << lane) - 1ULL));
    for (int k = 0; k < wave; ++k) rank += s_w[k];
    const size_t pos = (size_t)offsets[b] + rank;
    const float cube_res = 8.0f * res, half = res / 2; // VoxelCube.h:150, :47
    const float f = fabsf(sdf) / trunc;
    xyz[3 * pos] = (float)V.keys[3 * b] * cube_res + ((float)x * res + half);
    xyz[3 * pos + 1] = (float)V.keys[3 * b + 1] * cube_res + ((float)y * res + half);
    xyz[3 * pos + 2] = (float)V.keys[3 * b + 2] * cube_res + ((float)z * res + half);
    col[3 * pos] = f; col[3 * pos + 1] = f; col[3 * pos + 2] = f;
}


// ---------------------------------------------------------------------------------------------
// Mesh extraction: CubeHandler::ExtractTriangleMesh / GenerateMeshByCube (CubeHandler.cpp:9-114) +
// MarchingCube (MarchingCube.cpp:8-74).  One workgroup per block, one thread per voxel in the reference's
// x, y, z loop order; the 7 neighbour blocks a voxel on the +x/+y/+z faces needs are looked up once per
// workgroup.  The 256 x 16 triangle table and the 12 x 2 edge table are the CALLER'S data (the reference
// keeps them in MarchingCubePredefined.h; its shim passes them through the C-ABI), staged in LDS.
// Two passes with the same kernel: counts (triangles per block) and, after a scan, the ordered emit of
// three unshared vertices per triangle, exactly as MarchingCube() pushes them.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_mesh(VolView V, float res, const int* __restrict__ tri_table, const int* __restrict__ edge_pairs,
                                              const unsigned* __restrict__ blocks, unsigned* __restrict__ counts,
                                              const unsigned* __restrict__ offsets, float* __restrict__ pts, float* __restrict__ col) {
    __shared__ int s_tri[256 * 16];
    __shared__ int s_edge[24];
    __shared__ int s_nb[8];
    __shared__ unsigned s_w[8];
    const int b = (int)blocks[blockIdx.x], o = threadIdx.x;
    for (int k = o; k < 256 * 16; k += 512) s_tri[k] = tri_table[k];
    if (o < 24) s_edge[o] = edge_pairs[o];
    const int kx = V.keys[3 * b], ky = V.keys[3 * b + 1], kz = V.keys[3 * b + 2];
    if (o < 8) s_nb[o] = o == 0 ? b : table_find(V, kx + (o & 1), ky + ((o >> 1) & 1), kz + ((o >> 2) & 1)); // HasCube(neighbor_cube_id)
    __syncthreads();
    const int x = o >> 6, y = (o >> 3) & 7, z = o & 7;    // loop nest: x outer, y, z inner
    const int ox = x == 7, oy = y == 7, oz = z == 7;      // NeighborCubeIDOffset[index]
    const float cube_res = 8.0f * res, half = res / 2;    // VoxelCube.h:149-153, :48-61
    float cp[8][3], cs[8], cc[8][3];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int xo = (i == 1 || i == 2 || i == 5 || i == 6), yo = (i == 2 || i == 3 || i == 6 || i == 7), zo = i >= 4; // CornerXYZOffset, VoxelCube.h:45-47
        const int sel = (xo & ox) | ((yo & oy) << 1) | ((zo & oz) << 2);
        const int nb = s_nb[sel];
        const int vx = (x + xo) & 7, vy = (y + yo) & 7, vz = (z + zo) & 7;
        if (ok && nb < 0) ok = false;
        if (ok) {
            const float* t = V.pool + (size_t)nb * kBlockFloats + (vx + vy * 8 + vz * 64);
            const float sdf = t[0], w = t[kVox];
            cs[i] = sdf; cc[i][0] = t[2 * kVox]; cc[i][1] = t[3 * kVox]; cc[i][2] = t[4 * kVox];
            cp[i][0] = (float)(kx + (xo & ox)) * cube_res + ((float)vx * res + half);
            cp[i][1] = (float)(ky + (yo & oy)) * cube_res + ((float)vy * res + half);
            cp[i][2] = (float)(kz + (zo & oz)) * cube_res + ((float)vz * res + half);
            if (sdf >= 1 || w <= 0) ok = false;            // !IsValid (TSDFVoxel.h:75-78)
        }
    }
    int ci = 0;
    unsigned ntri = 0;
    if (ok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ci |= cs[i] > 0 ? 1 << i : 0;  // DetermineCase
        for (int i = 0; i < 16 && s_tri[16 * ci + i] != -1; i += 3) ++ntri;
    }
    // exclusive scan of ntri over the workgroup in thread (= reference loop) order
    unsigned incl = ntri;
    const int lane = o & 63, wave = o >> 6;
    for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    if (counts) {
        if (o == 0) { unsigned tot = 0; for (int k = 0; k < 8; ++k) tot += s_w[k]; counts[blockIdx.x] = tot; }
        return;
    }
    if (!ntri) return;
    unsigned first = incl - ntri;
    for (int k = 0; k < wave; ++k) first += s_w[k];
    size_t vtx = ((size_t)offsets[blockIdx.x] + first) * 3;
    for (int i = 0; i < 16 && s_tri[16 * ci + i] != -1; i += 3)
        for (int j = 0; j < 3; ++j, ++vtx) {
            const int e = s_tri[16 * ci + i + j], a = s_edge[2 * e], c = s_edge[2 * e + 1];
            // InterpolateEdgeVetex (MarchingCube.cpp:8-16); corners picked by dynamic index -> select chains
            float pa[3] = {0, 0, 0}, pc[3] = {0, 0, 0}, ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0}, sa = 0, sc = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q == a) { sa = cs[q]; pa[0] = cp[q][0]; pa[1] = cp[q][1]; pa[2] = cp[q][2]; ca[0] = cc[q][0]; ca[1] = cc[q][1]; ca[2] = cc[q][2]; }
                if (q == c) { sc = cs[q]; pc[0] = cp[q][0]; pc[1] = cp[q][1]; pc[2] = cp[q][2]; cb[0] = cc[q][0]; cb[1] = cc[q][1]; cb[2] = cc[q][2]; }
            }
            const float sdf_diff = sc - sa;
            const float t = sa / sdf_diff;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pts[3 * vtx + k] = pa[k] - t * (pc[k] - pa[k]);
                col[3 * vtx + k] = (ca[k] + cb[k]) / 2.0f;  // (c1 + c2) / 2
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Ray casting (north_star "integrate/raycast").  The reference has NO raycast; the
// definition is this implementation's own and is validated against the analytic synthetic scene:
// march every pixel ray from near to far through the voxel-block hash, sample the sdf
// trilinearly over the 8 surrounding voxel centres (all 8 must be observed, weight > 0), step one
// voxel inside allocated blocks (valid sample or not: IntegrateImage only writes |sdf| < truncation, so the
// free space in front of a surface is unobserved voxels INSIDE allocated blocks and must not be leapt over), jump
// to the exit face of a block that is absent from the hash, and report the first
// + -> - crossing by linear interpolation as z-depth.  Normal = normalised central difference of the
// trilinear sdf (+-res/2), colour = trilinear colour at the hit.  16x16 pixel tiles per workgroup
// keep neighbouring rays -- which walk the same blocks -- on one CU.
// ---------------------------------------------------------------------------------------------
struct BlockCache { int cx, cy, cz, idx; };

template <bool COL>
__device__ __forceinline__ bool rc_fetch(const VolView& V, BlockCache& bc, int px, int py, int pz, Vox5* out) {
    const int cx = px >> 3, cy = py >> 3, cz = pz >> 3;
    if (!(cx == bc.cx && cy == bc.cy && cz == bc.cz)) { bc.cx = cx; bc.cy = cy; bc.cz = cz; bc.idx = table_find(V, cx, cy, cz); }
    if (bc.idx < 0) return false;
    const int vid = (px - cx * 8) + (py - cy * 8) * 8 + (pz - cz * 8) * 64;
    const float* t = V.pool + (size_t)bc.idx * kBlockFloats + vid;
    out->s = t[0]; out->w = t[kVox];
    if (COL) { out->c0 = t[2 * kVox]; out->c1 = t[3 * kVox]; out->c2 = t[4 * kVox]; } // colour planes only at the hit
    else { out->c0 = out->c1 = out->c2 = 0.0f; }
    return out->w > 0;
}

template <bool COL>
__device__ bool rc_sample_t(const VolView& V, BlockCache& bc, float res, float x, float y, float z, float* sdf, float* col) {
    const float gx = x / res - 0.5f, gy = y / res - 0.5f, gz = z / res - 0.5f;
    const float fx0 = floorf(gx), fy0 = floorf(gy), fz0 = floorf(gz);
    const int ix = (int)fx0, iy = (int)fy0, iz = (int)fz0;
    const float fx = gx - fx0, fy = gy - fy0, fz = gz - fz0;
    float acc = 0, a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        Vox5 t;
        if (!rc_fetch<COL>(V, bc, ix + (k & 1), iy + ((k >> 1) & 1), iz + ((k >> 2) & 1), &t)) return false;
        const float wx = (k & 1) ? fx : 1.0f - fx, wy = (k & 2) ? fy : 1.0f - fy, wz = (k & 4) ? fz : 1.0f - fz;
        const float w = (wx * wy) * wz;
        acc += w * t.s; a0 += w * t.c0; a1 += w * t.c1; a2 += w * t.c2;
    }
    *sdf = acc;
    if (COL) { col[0] = a0; col[1] = a1; col[2] = a2; }
    return true;
}
// marching and normal samples read only the sdf and weight planes (2 of the 5)
__device__ __forceinline__ bool rc_sample(const VolView& V, BlockCache& bc, float res, float x, float y, float z, float* sdf, float* col) {
    return col ? rc_sample_t<true>(V, bc, res, x, y, z, sdf, col) : rc_sample_t<false>(V, bc, res, x, y, z, sdf, nullptr);
}

__global__ __launch_bounds__(256) void k_raycast(VolView V, op_camera cam, Mat4 P, float res, float near_d, float far_d,
                                                 float* __restrict__ depth_out, float* __restrict__ normals_out, float* __restrict__ colors_out) {
    const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= cam.width || py >= cam.height) return;
    const size_t pix = (size_t)py * cam.width + px;
    const float dcx = ((float)px - cam.cx) / cam.fx, dcy = ((float)py - cam.cy) / cam.fy;
    const float* M = P.m;
    const float d0 = (M[0] * dcx + M[1] * dcy) + M[2], d1 = (M[4] * dcx + M[5] * dcy) + M[6], d2 = (M[8] * dcx + M[9] * dcy) + M[10];
    const float o0 = M[3], o1 = M[7], o2 = M[11];
    const float fine = res, coarse = res * 8.0f;
    BlockCache bc{INT_MIN, INT_MIN, INT_MIN, -1};
    float t = near_d, t_prev = 0, s_prev = 0, hit = 0;
    bool have_prev = false;
    while (t <= far_d) {
        float sdf;
        if (rc_sample(V, bc, res, o0 + t * d0, o1 + t * d1, o2 + t * d2, &sdf, nullptr)) {
            if (have_prev && s_prev > 0 && sdf <= 0) { hit = t_prev + (t - t_prev) * (s_prev / (s_prev - sdf)); break; }
            have_prev = true; s_prev = sdf; t_prev = t;
            t += fine;
        } else {
            have_prev = false;
            // an invalid sample inside an allocated block is an unobserved voxel: step one voxel.  Only a block that
            // is absent from the hash is skipped, up to its exit face (no valid sample can lie in it: all 8 voxel
            // centres around a point of an absent block cannot be observed)
            const float p0 = o0 + t * d0, p1 = o1 + t * d1, p2 = o2 + t * d2;
            const float b0 = floorf(p0 / coarse), b1 = floorf(p1 / coarse), b2 = floorf(p2 / coarse);
            const int bx = (int)b0, by = (int)b1, bz = (int)b2;
            if (!(bx == bc.cx && by == bc.cy && bz == bc.cz)) { bc.cx = bx; bc.cy = by; bc.cz = bz; bc.idx = table_find(V, bx, by, bz); }
            float step = fine;
            if (bc.idx < 0) {
                float t_exit = FLT_MAX;
                if (d0 > 0) t_exit = fminf(t_exit, ((b0 + 1.0f) * coarse - p0) / d0); else if (d0 < 0) t_exit = fminf(t_exit, (b0 * coarse - p0) / d0);
                if (d1 > 0) t_exit = fminf(t_exit, ((b1 + 1.0f) * coarse - p1) / d1); else if (d1 < 0) t_exit = fminf(t_exit, (b1 * coarse - p1) / d1);
                if (d2 > 0) t_exit = fminf(t_exit, ((b2 + 1.0f) * coarse - p2) / d2); else if (d2 < 0) t_exit = fminf(t_exit, (b2 * coarse - p2) / d2);
                if (t_exit < FLT_MAX) step = fmaxf(fine, t_exit + 0.01f * res);
            }
            t += step;
        }
    }
    depth_out[pix] = hit;
    float n[3] = {0, 0, 0}, c[3] = {0, 0, 0};
    if (hit > 0 && (normals_out || colors_out)) {
        const float x = o0 + hit * d0, y = o1 + hit * d1, z = o2 + hit * d2, h = 0.5f * res;
        float s0;
        if (!rc_sample(V, bc, res, x, y, z, &s0, c)) { c[0] = c[1] = c[2] = 0; }
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float sp = 0, sm = 0;
            if (ok) ok = rc_sample(V, bc, res, x + (a == 0 ? h : 0.0f), y + (a == 1 ? h : 0.0f), z + (a == 2 ? h : 0.0f), &sp, nullptr) &&
                         rc_sample(V, bc, res, x - (a == 0 ? h : 0.0f), y - (a == 1 ? h : 0.0f), z - (a == 2 ? h : 0.0f), &sm, nullptr);
            n[a] = sp - sm;
        }
        const float l2 = sum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
        if (ok && l2 > 0) { const float l = sqrtf(l2); n[0] /= l; n[1] /= l; n[2] /= l; } else { n[0] = n[1] = n[2] = 0; }
    }
    if (normals_out) { normals_out[3 * pix] = n[0]; normals_out[3 * pix + 1] = n[1]; normals_out[3 * pix + 2] = n[2]; }
    if (colors_out) { colors_out[3 * pix] = c[0]; colors_out[3 * pix + 1] = c[1]; colors_out[3 * pix + 2] = c[2]; }
}

__global__ void k_has_cube(VolView V, int x, int y, int z, int* out) { *out = table_find(V, x, y, z) >= 0 ? 1 : 0; }

unsigned next_pow2(unsigned long long v) {
    unsigned long long p = 1;
    while (p < v) p <<= 1;
    return (unsigned)p;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// op_volume: host object
// ---------------------------------------------------------------------------------------------
struct op_volume {
    int device = 0;
    hipStream_t stream = nullptr;
    op_camera cam{};
    float res = 0.01f, trunc = 0.1f, far_d = 5.0f, near_d = 0.5f;
    unsigned max_blocks = 0;
    unsigned table_size = 0;
    // device memory
    unsigned long long* tkeys = nullptr;
    int* tvals = nullptr;
    int* keys = nullptr;
    float* pool = nullptr;
    unsigned* n_blocks = nullptr;
    bmask_t* bmask = nullptr;
    int* blist = nullptr;
    int* sel_list = nullptr;
    unsigned long long* sel_cand = nullptr;
    State* state = nullptr;
    float* partial = nullptr;   // kMaxBatch x ka_grid x 8
    uint2* pimg = nullptr;      // kMaxBatch x W*H packed {depth, rgba}
    unsigned long long* sbits = nullptr; // kMaxBatch x kVoteCap words: a frame's selections per super-block of its range (k_select_vote -> k_select_merge)
    float2* ptile = nullptr;    // kMaxBatch x tiles: {min, max} valid depth of every 16 x 16 pixel tile (KA -> KB)
    size_t pimg_px = 0;
    int pimg_w = 0, pimg_h = 0;
    unsigned long long* upd_partial = nullptr;
    unsigned long long* sel_partial = nullptr;
    unsigned long long* chg_partial = nullptr; // [0, grid): voxels written, [grid, 2 grid): blocks read, summed over launches
    // optional HIP-event timing (op_volume_profile_*): every `prof_every`-th batch gets four events
    // on the volume's stream (before KA, after KA, after KB, after KC); prof_frames = frames per sample
    int prof_every = 0;
    uint64_t prof_batch = 0;
    std::vector<hipEvent_t> prof_events; // 4 per sampled batch
    std::vector<int> prof_frames;
    // scratch for the single-frame synchronous calls that take host images (ComputeBounding / PrepareCubes)
    void* img_depth = nullptr;
    unsigned char* img_rgb = nullptr;
    size_t img_cap_px = 0;
    // Staging ring for host images handed to op_volume_integrate: kRing batch slots, each with pinned host buffers and
    // device buffers for kMaxBatch frames.  A frame is copied into the pinned slot by the caller's thread (+ helper
    // threads), DMA'd on `copy_stream` while the caller fills the next frame, and the batch's kernels wait for the
    // slot's `copied` event -- so the H2D of batch b+1 overlaps the kernels of batch b.  A slot is reused only after the
    // batch that used it is CONFIRMED complete (its device images are what a replay after pool growth reads).
    struct RingSlot {
        void* d_depth = nullptr; unsigned char* d_rgb = nullptr;
        void* h_depth = nullptr; unsigned char* h_rgb = nullptr;
        hipEvent_t copied = nullptr;
        uint64_t busy_seq = 0; // sequence number of the batch staged here, 0 = free
        int dma_lo = 0, dma_hi = 0; // positions [dma_lo, dma_hi) are staged in pinned memory and not yet on their way to the device
        size_t dma_dbytes = 0;      // depth bytes per frame of those positions
    };
    static constexpr int kRing = 3;
    RingSlot ring[kRing];
    size_t ring_px = 0;
    int ring_cur = -1;          // slot of the batch being assembled (-1: none acquired yet)
    unsigned ring_next = 0;
    hipStream_t copy_stream = nullptr;
    // Growth / replay.  Every launched batch is logged until it is confirmed complete; if a batch exhausts the pool or the
    // hash table the stream is poisoned on the device (nothing is fused from that batch on), and the host -- at its next
    // look -- grows the volume and replays the log from the failing batch.  No frame is lost or partially applied.
    struct BatchRec { uint64_t seq; BatchFwd F; BatchInv I; BatchPtrs P; int nf, fmt, ring_slot; };
    std::deque<BatchRec> log;
    uint64_t seq = 0;            // sequence number of the last launched batch
    unsigned* hstat = nullptr;   // pinned + mapped: [0] = last batch known complete, [1] = n_blocks at that time
    unsigned* hstat_dev = nullptr;
    bool recovering = false;     // vol_recover is replaying: no nested growth checks
    // true while every voxel was written by k_integrate only since create / clear (see k_integrate<., PLAIN>): any other
    // writer (upload, merge, sum-form unpack, resampling result, file) clears it and fusion takes the general update
    bool plain = true;
    int select_mode = 0;         // OP_VOLUME_OPT_SELECT: OP_VOLUME_SELECT_AUTO, OP_VOLUME_SELECT_DIRECT, or the largest range (in super-blocks) a frame may vote with
    int update_mode = 0;         // OP_VOLUME_OPT_UPDATE: OP_VOLUME_UPDATE_EXACT (the reference's frame-by-frame running mean, bit for bit) or _SUM_FORM
    unsigned plain_from = 0;     // with !plain: pool slots below this bound may hold foreign data (general update); later blocks are k_integrate's own
    int* unpack_slots = nullptr; // table slots of the union keys between op_volume_unpack_sum_begin and its chunks
    size_t unpack_n = 0;
    uint64_t generation = 0, unpack_gen = 0; // bumped by whatever moves or drops table slots (growth, clear) or fuses frames; _chunk checks it
    uint64_t n_grows = 0, n_replayed = 0; // pool growths and batches launched again after one (op_volume_growth_stats)
    uint64_t frames_accepted = 0; // op_volume_progress: frames handed to the integrate calls so far
    bool grow_refused = false;   // an early growth could not get memory: stop asking before every batch (a real overflow still tries)
    // frames accepted by op_volume_integrate but not launched yet: single-frame calls are queued
    // and fused in batches of kMaxBatch (every accessor flushes first, so this is unobservable)
    int pend_n = 0, pend_fmt = 0;
    BatchFwd pend_F;
    BatchInv pend_I;
    BatchPtrs pend_P;

    VolView view() const {
        VolView V;
        V.tkeys = tkeys; V.tvals = tvals; V.table_mask = table_size - 1; V.keys = keys; V.pool = pool;
        V.max_blocks = max_blocks; V.n_blocks = n_blocks; V.bmask = bmask; V.blist = blist;
        V.sel_list = sel_list; V.sel_cand = sel_cand;
        return V;
    }
};

namespace {

int vol_flush(op_volume* v); // launches the frames queued by op_volume_integrate
int vol_ring_send(op_volume* v, op_volume::RingSlot& r); // the DMA of the staged host frames that have not been sent yet

// Every block with a pool slot below `bound` may hold data k_integrate did not write (see its PLAIN comment).
void vol_mark_foreign(op_volume* v, unsigned long long bound) {
    v->plain = false;
    const unsigned b = bound > 0xffffffffull ? 0xffffffffu : (unsigned)bound;
    if (b > v->plain_from) v->plain_from = b;
}

int vol_reset(op_volume* v) {
    ++v->generation;
    hipLaunchKernelGGL(k_clear_table, dim3(1024), dim3(256), 0, v->stream, v->tkeys, v->tvals, (size_t)v->table_size);
    OP_HIP(hipMemsetAsync(v->n_blocks, 0, sizeof(unsigned), v->stream));
    OP_HIP(hipMemsetAsync(v->bmask, 0, sizeof(bmask_t) * (size_t)v->table_size, v->stream));
    OP_HIP(hipMemsetAsync(v->state, 0, sizeof(State), v->stream));
    OP_HIP(hipMemsetAsync(v->upd_partial, 0, sizeof(unsigned long long) * kPartialGrid, v->stream));
    OP_HIP(hipMemsetAsync(v->sel_partial, 0, sizeof(unsigned long long) * kPartialGrid, v->stream));
    OP_HIP(hipMemsetAsync(v->chg_partial, 0, sizeof(unsigned long long) * 2 * kPartialGrid, v->stream));
    OP_HIP(hipGetLastError());
    return OP_OK;
}

constexpr unsigned kHardMaxBlocks = 1u << 24; // 172 GB of pool: what one 288 GB MI355X can hold next to its inputs

int vol_enqueue_batch(op_volume* v, const BatchFwd& F, const BatchInv& I, const BatchPtrs& Q, int nf, int depth_fmt, bool select_only, bool record,
                      const int* cube_keys = nullptr, unsigned n_cubes = 0);

// Grows the pool to new_max blocks (and the hash table to twice that), keeping the first n_valid blocks.  The stream
// must be idle.  The new buffers are allocated before the old ones are released, so a failed allocation leaves the
// volume intact (OP_ERR_CAPACITY).  CubeMap growth in the reference is std::unordered_map's (CubeHandler.h:22).
int vol_grow(op_volume* v, unsigned long long want, unsigned n_valid) {
    if (want > kHardMaxBlocks) want = kHardMaxBlocks;
    if (want <= v->max_blocks) return fail(OP_ERR_CAPACITY, "volume cannot grow beyond %u blocks", v->max_blocks);
    const unsigned new_max = (unsigned)want, new_table = next_pow2(2ull * new_max);
    float* pool = nullptr;
    int *keys = nullptr, *blist = nullptr, *sel_list = nullptr, *tvals = nullptr;
    unsigned long long *sel_cand = nullptr, *tkeys = nullptr;
    bmask_t* bmask = nullptr;
    hipError_t e = op::cached_malloc((void**)&pool, sizeof(float) * kBlockFloats * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&keys, sizeof(int) * 3 * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&blist, sizeof(int) * (size_t)(KC_BANDS ? kBands : 1) * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&sel_list, sizeof(int) * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&sel_cand, sizeof(unsigned long long) * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&tkeys, sizeof(unsigned long long) * (size_t)new_table);
    if (e == hipSuccess) e = op::cached_malloc((void**)&tvals, sizeof(int) * (size_t)new_table);
    if (e == hipSuccess) e = op::cached_malloc((void**)&bmask, sizeof(bmask_t) * (size_t)new_table);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        void* got[] = {pool, keys, blist, sel_list, sel_cand, tkeys, tvals, bmask};
        for (void* q : got)
            if (q) op::cached_free(q);
        return fail(OP_ERR_CAPACITY, "cannot grow the volume to %u blocks: %s", new_max, hipGetErrorString(e));
    }
    if (n_valid) {
        OP_HIP(hipMemcpyAsync(pool, v->pool, sizeof(float) * kBlockFloats * (size_t)n_valid, hipMemcpyDeviceToDevice, v->stream));
        OP_HIP(hipMemcpyAsync(keys, v->keys, sizeof(int) * 3 * (size_t)n_valid, hipMemcpyDeviceToDevice, v->stream));
    }
    hipLaunchKernelGGL(k_fill_pool, dim3(4096), dim3(256), 0, v->stream, pool, (size_t)n_valid, (size_t)(new_max - n_valid));
    hipLaunchKernelGGL(k_clear_table, dim3(1024), dim3(256), 0, v->stream, tkeys, tvals, (size_t)new_table);
    OP_HIP(hipMemsetAsync(bmask, 0, sizeof(bmask_t) * (size_t)new_table, v->stream));
    if (n_valid) hipLaunchKernelGGL(k_rehash, dim3((n_valid + 255) / 256), dim3(256), 0, v->stream, tkeys, tvals, new_table - 1, (const int*)keys, n_valid);
    OP_HIP(hipMemcpyAsync(v->n_blocks, &n_valid, sizeof(unsigned), hipMemcpyHostToDevice, v->stream));
    OP_HIP(hipGetLastError());
    OP_HIP(hipStreamSynchronize(v->stream)); // n_valid is a stack variable; the old buffers are released next
    void* old[] = {v->pool, v->keys, v->blist, v->sel_list, v->sel_cand, v->tkeys, v->tvals, v->bmask};
    for (void* q : old)
        if (q) op::cached_free(q);
    v->pool = pool; v->keys = keys; v->blist = blist; v->sel_list = sel_list; v->sel_cand = sel_cand;
    v->tkeys = tkeys; v->tvals = tvals; v->bmask = bmask;
    v->max_blocks = new_max; v->table_size = new_table;
    ++v->n_grows; ++v->generation;
    return OP_OK;
}

// Makes room for `need` blocks in total (upload / merge / unpack know their demand up front).  Synchronises when it grows.
int vol_reserve(op_volume* v, unsigned long long need) {
    if (need <= v->max_blocks) return OP_OK;
    if (need > kHardMaxBlocks) return fail(OP_ERR_CAPACITY, "%llu blocks exceed the limit of %u blocks per volume", need, kHardMaxBlocks);
    OP_HIP(hipStreamSynchronize(v->stream));
    unsigned n = 0;
    OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
    if (n > v->max_blocks) n = v->max_blocks;
    return vol_grow(v, std::max<unsigned long long>(next_pow2(need), 2ull * v->max_blocks), n);
}

// Retires log entries (and their staging slots) of batches the device has reported complete.  Never blocks.
void vol_retire(op_volume* v) {
    if (!v->hstat) return;
    const unsigned lo = __atomic_load_n(&v->hstat[0], __ATOMIC_ACQUIRE);
    // widen the 32-bit report: it can only lie at or behind the last launched batch, and less than 2^31 behind
    uint64_t done = (v->seq & ~0xffffffffull) | lo;
    if (done > v->seq) done -= 1ull << 32;
    while (!v->log.empty() && v->log.front().seq <= done) {
        const int rs = v->log.front().ring_slot;
        if (rs >= 0 && v->ring[rs].busy_seq == v->log.front().seq) v->ring[rs].busy_seq = 0;
        v->log.pop_front();
    }
}

// The stream is idle.  If a batch exhausted the pool or the table: grow, then replay everything from that batch on (the
// failing batch's KC and all later batches did nothing).  Loops until the log has gone through.  Other overflow bits
// (bad frames) are left for the caller to report.
int vol_recover(op_volume* v, unsigned* flags_out) {
    for (;;) {
        StateHead st; // the head only: State is ~20 KB, and a pageable device-to-host copy of 16 KB or more takes the runtime's pinned-staging path (ms)
        OP_HIP(hipMemcpy(&st, v->state, sizeof(st), hipMemcpyDeviceToHost));
        if (flags_out) *flags_out = st.overflow;
        if ((st.overflow & 3u) == 0u) {
            for (auto& r : v->log)
                if (r.ring_slot >= 0 && v->ring[r.ring_slot].busy_seq == r.seq) v->ring[r.ring_slot].busy_seq = 0;
            v->log.clear();
            return OP_OK;
        }
        unsigned n = 0;
        OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
        if (n > v->max_blocks) n = v->max_blocks; // claims beyond the pool were marked dead; the blocks below it are real
        // first failing batch: widen the device's 32-bit sequence number like vol_retire does
        uint64_t fail_seq = (v->seq & ~0xffffffffull) | st.fail_seq;
        if (fail_seq > v->seq) fail_seq -= 1ull << 32;
        int rc = vol_grow(v, 2ull * v->max_blocks, n);
        if (rc != OP_OK) { // cannot grow: report, drop the frames that cannot be fused
            v->log.clear();
            for (auto& r : v->ring) r.busy_seq = 0;
            const unsigned keep = st.overflow & ~3u;
            OP_HIP(hipMemcpy(&v->state->overflow, &keep, sizeof(keep), hipMemcpyHostToDevice));
            return rc;
        }
        const unsigned keep = st.overflow & ~3u;
        OP_HIP(hipMemcpy(&v->state->overflow, &keep, sizeof(keep), hipMemcpyHostToDevice));
        std::deque<op_volume::BatchRec> replay;
        replay.swap(v->log);
        while (!replay.empty() && replay.front().seq < fail_seq) replay.pop_front(); // those completed
        v->recovering = true;
        int rrc = OP_OK;
        for (auto& r : replay) {
            rrc = vol_enqueue_batch(v, r.F, r.I, r.P, r.nf, r.fmt, false, false); // assigns a new sequence number
            ++v->n_replayed;
            if (rrc != OP_OK) break;
            if (r.ring_slot >= 0) v->ring[r.ring_slot].busy_seq = v->seq;
            v->log.back().ring_slot = r.ring_slot;
        }
        v->recovering = false;
        OP_TRY(rrc);
        OP_HIP(hipStreamSynchronize(v->stream));
    }
}

// Flushes, synchronises, grows + replays if needed, and reports frames that cannot be fused at all.
int vol_check(op_volume* v) {
    OP_TRY(vol_flush(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    unsigned of = 0;
    OP_TRY(vol_recover(v, &of));
    if (of & 12u) { // report once, then clear: the offending frames selected nothing, everything else was fused
        const unsigned zero = 0;
        OP_HIP(hipMemcpy(&v->state->overflow, &zero, sizeof(zero), hipMemcpyHostToDevice));
    }
    if (of & 4u) return fail(OP_ERR_INVALID, "frame bounding box spans more than 4096 blocks on an axis");
    if (of & 8u) return fail(OP_ERR_INVALID, "block coordinate outside +-2^20 (not representable in the device hash key)");
    return OP_OK;
}

int vol_block_count(op_volume* v, unsigned* n) {
    OP_TRY(vol_check(v));
    OP_HIP(hipMemcpy(n, v->n_blocks, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (*n > v->max_blocks) *n = v->max_blocks;
    return OP_OK;
}

// Single-frame scratch of the synchronous calls (ComputeBounding / PrepareCubes): host images are copied to the device
// stream-ordered; device images are used in place.  (op_volume_integrate stages through the pinned ring below.)
int vol_stage_images(op_volume* v, const void** depth, int depth_fmt, const unsigned char** rgb, int mem, int slot = 0) {
    if (mem == OP_MEM_DEVICE) return OP_OK;
    const size_t npx = (size_t)v->cam.width * v->cam.height;
    if (npx > v->img_cap_px) {
        OP_TRY(vol_flush(v));
        OP_HIP(hipStreamSynchronize(v->stream));
        if (v->img_depth) op::cached_free(v->img_depth);
        if (v->img_rgb) op::cached_free(v->img_rgb);
        v->img_depth = nullptr; v->img_rgb = nullptr; v->img_cap_px = 0;
        OP_HIP(op::cached_malloc(&v->img_depth, npx * 4));
        OP_HIP(op::cached_malloc((void**)&v->img_rgb, npx * 3));
        v->img_cap_px = npx;
    }
    void* d = (char*)v->img_depth + (size_t)slot * npx * 4;
    OP_HIP(hipMemcpyAsync(d, *depth, npx * (depth_fmt == OP_DEPTH_U16 ? 2 : 4), hipMemcpyHostToDevice, v->stream));
    *depth = d;
    if (rgb && *rgb) {
        unsigned char* c = v->img_rgb + (size_t)slot * npx * 3;
        OP_HIP(hipMemcpyAsync(c, *rgb, npx * 3, hipMemcpyHostToDevice, v->stream));
        *rgb = c;
    }
    return OP_OK;
}

CamParams cam_params(const op_volume* v, int depth_fmt) {
    CamParams C;
    C.fx = v->cam.fx; C.fy = v->cam.fy; C.cx = v->cam.cx; C.cy = v->cam.cy;
    C.depth_scale = v->cam.depth_scale; C.res = v->res; C.trunc = v->trunc;
    C.width = v->cam.width; C.height = v->cam.height; C.depth_u16 = depth_fmt == OP_DEPTH_U16;
    C.ax = px_axis(C.cx, C.width); C.ay = px_axis(C.cy, C.height);
    C.fast_px = C.ax.exact && C.ay.exact;
    return C;
}

// per-frame host work of the path: frustum planes (Frustum.cpp:7-46) and pose^-1 (Integrator.cpp:48)
void frame_params(const op_volume* v, const float pose[16], const float* pose_inv, PoseFwd* fwd, PoseInv* inv) {
    std::memcpy(fwd->pose, pose, sizeof(fwd->pose));
    op_host::CameraPOD c{v->cam.fx, v->cam.fy, v->cam.cx, v->cam.cy, v->cam.width, v->cam.height, v->cam.depth_scale};
    op_host::frustum_planes(c, pose, v->far_d, v->near_d, fwd->planes);
    float full[16];
    if (pose_inv) std::memcpy(full, pose_inv, sizeof(full));
    else op_host::mat4_inverse(pose, full);
    std::memcpy(inv->m, full, sizeof(inv->m));
}

int vol_ensure_frame_buffers(op_volume* v) {
    if (!v->sbits) OP_HIP(op::cached_malloc((void**)&v->sbits, (size_t)kMaxBatch * kVoteCap * sizeof(unsigned long long)));
    const size_t npx = (size_t)v->cam.width * v->cam.height;
    if (npx <= v->pimg_px && v->cam.width == v->pimg_w && v->cam.height == v->pimg_h) return OP_OK; // (KA's grid and the tile grid depend on both)
    OP_HIP(hipStreamSynchronize(v->stream)); // released buffers go back to a cache and may be handed out at once
    if (v->pimg) op::cached_free(v->pimg);
    if (v->partial) op::cached_free(v->partial);
    if (v->ptile) op::cached_free(v->ptile);
    v->pimg = nullptr; v->partial = nullptr; v->ptile = nullptr; v->pimg_px = 0; v->pimg_w = v->pimg_h = 0;
    const size_t g1 = (size_t)ka_grid(v->cam.width, v->cam.height);
    OP_HIP(op::cached_malloc((void**)&v->pimg, (size_t)kMaxBatch * npx * sizeof(uint2)));
    OP_HIP(op::cached_malloc((void**)&v->partial, (size_t)kMaxBatch * g1 * 8 * sizeof(float)));
    OP_HIP(op::cached_malloc((void**)&v->ptile, (size_t)kMaxBatch * tiles_w(v->cam.width) * tiles_h(v->cam.height) * sizeof(float2)));
    v->pimg_px = npx; v->pimg_w = v->cam.width; v->pimg_h = v->cam.height;
    return OP_OK;
}

// Enqueue one batch (1..kMaxBatch frames whose images are on the device): KA, KB and, unless
// select_only, KC.  No host synchronisation.
// With cube_keys (device array of n_cubes ids, nf == 1): KB is replaced by k_mark_cubes -- the frame is fused into exactly
// those cubes (op_volume_integrate_cubes); the caller has reserved the room and synchronises, so the batch is not logged.
int vol_enqueue_batch(op_volume* v, const BatchFwd& F, const BatchInv& I, const BatchPtrs& Q, int nf, int depth_fmt, bool select_only, bool record,
                      const int* cube_keys, unsigned n_cubes) {
    OP_TRY(vol_ensure_frame_buffers(v));
    // Look at the device's lagging progress report (no synchronisation): retire confirmed batches and, when the pool is
    // about to run full, grow it NOW -- between batches -- instead of paying for a replay later.
    vol_retire(v);
    if (v->hstat && !v->recovering && !v->grow_refused && (unsigned long long)v->hstat[1] + v->max_blocks / 8u > v->max_blocks && v->max_blocks < kHardMaxBlocks) {
        OP_HIP(hipStreamSynchronize(v->stream));
        unsigned of = 0;
        OP_TRY(vol_recover(v, &of)); // also handles an overflow that has already happened
        unsigned n = 0;
        OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
        if ((unsigned long long)n + v->max_blocks / 8u > v->max_blocks) {
            const int rc = vol_grow(v, 2ull * v->max_blocks, n < v->max_blocks ? n : v->max_blocks);
            if (rc != OP_OK && rc != OP_ERR_CAPACITY) return rc; // out of memory is not fatal yet: the pool may still suffice
            if (rc == OP_ERR_CAPACITY) v->grow_refused = true;
        }
        v->hstat[1] = n;
    }
    const unsigned seq = (unsigned)(++v->seq);
    ++v->generation;
    if (!select_only && !cube_keys) v->log.push_back(op_volume::BatchRec{v->seq, F, I, Q, nf, depth_fmt, -1});
    const CamParams C = cam_params(v, depth_fmt);
    const int g1 = ka_grid(C.width, C.height);
    const VolView V = v->view();
    const bool sample = !select_only && v->prof_every > 0 && (v->prof_batch++ % (uint64_t)v->prof_every) == 0 &&
                        v->prof_events.size() < 4 * 65536;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (sample) {
        for (auto& e : ev) OP_HIP(hipEventCreate(&e));
        OP_HIP(hipEventRecord(ev[0], v->stream));
    }
    for (int f0 = 0; f0 < nf; f0 += kKaFrames) {
        KaFwd A;
        const int na = nf - f0 < kKaFrames ? nf - f0 : kKaFrames;
        std::memcpy(A.f, F.f + f0, sizeof(PoseFwd) * (size_t)na);
        hipLaunchKernelGGL(k_prepare_frames, dim3(g1, na), dim3(256), 0, v->stream, A, f0, C, Q, v->pimg, v->ptile, v->partial, v->state, seq,
                           (const unsigned*)v->n_blocks, v->hstat_dev);
    }
    if (sample) OP_HIP(hipEventRecord(ev[1], v->stream));
    if (cube_keys)
        hipLaunchKernelGGL(k_mark_cubes, dim3((n_cubes + 255u) / 256u), dim3(256), 0, v->stream, V, v->state, cube_keys, n_cubes);
    else if (KB_VOTE && KC_BANDS == 0 && !record && nf >= (v->select_mode > 0 ? 2 : KB_VOTE_MIN_FRAMES) && v->select_mode != OP_VOLUME_SELECT_DIRECT) { // (an explicit limit: every batch of >= 2 frames) // several frames: they record their selections, one pass claims every block once
        const unsigned vote_cap = v->select_mode > 0 ? (unsigned)v->select_mode : kVoteCap;
        const int per_frame = std::max(8, std::min(kSelectGrid, (KB_VOTE_WGS / nf + 7) / 8 * 8)); // a multiple of 8: whole frames per XCD
        if (C.fast_px)
            hipLaunchKernelGGL(k_select_vote<true>, dim3(per_frame, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile, v->state, v->sbits, vote_cap);
        else
            hipLaunchKernelGGL(k_select_vote<false>, dim3(per_frame, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile, v->state, v->sbits, vote_cap);
        hipLaunchKernelGGL(k_select_merge, dim3(KB_MERGE_GRID), dim3(64 * kMergeWords), 0, v->stream, V, v->state, (const unsigned long long*)v->sbits, nf);
    } else if (C.fast_px)
        hipLaunchKernelGGL(k_select<true>, dim3(kSelectGrid, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile,
                           v->state, record ? 1 : 0);
    else
        hipLaunchKernelGGL(k_select<false>, dim3(kSelectGrid, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile,
                           v->state, record ? 1 : 0);
    if (sample) OP_HIP(hipEventRecord(ev[2], v->stream));
    if (select_only)
        hipLaunchKernelGGL(k_finish_select, dim3(256), dim3(256), 0, v->stream, V, v->state);
    else {
#define OP_KC(FASTPX, PLAINV, SUMFV) hipLaunchKernelGGL((k_integrate<FASTPX, PLAINV, (SUMFV ? KC_ZT_SUM : KC_ZT), SUMFV>), dim3(SUMFV ? kColGridSum : kColGrid), dim3(512 / (SUMFV ? KC_ZT_SUM : KC_ZT)), 0, v->stream, I, C, V, (const uint2*)v->pimg, \
                                                 v->state, nf, v->upd_partial, v->sel_partial, v->chg_partial, v->plain_from)
        if (v->update_mode == OP_VOLUME_UPDATE_SUM_FORM) { if (C.fast_px) OP_KC(true, true, true); else OP_KC(false, true, true); }
        else if (C.fast_px) { if (v->plain) OP_KC(true, true, false); else OP_KC(true, false, false); }
        else { if (v->plain) OP_KC(false, true, false); else OP_KC(false, false, false); }
#undef OP_KC
    }
    if (sample) {
        OP_HIP(hipEventRecord(ev[3], v->stream));
        for (auto e : ev) v->prof_events.push_back(e);
        v->prof_frames.push_back(nf);
    }
    OP_HIP(hipGetLastError());
    return OP_OK;
}

// launch the frames queued by op_volume_integrate
int vol_flush(op_volume* v) {
    if (v->pend_n == 0) return OP_OK;
    const int nf = v->pend_n, rs = v->ring_cur;
    v->pend_n = 0;
    v->ring_cur = -1;
    if (rs >= 0) { // the batch's kernels start when its host images have arrived
        OP_TRY(vol_ring_send(v, v->ring[rs]));
        OP_HIP(hipEventRecord(v->ring[rs].copied, v->copy_stream));
        OP_HIP(hipStreamWaitEvent(v->stream, v->ring[rs].copied, 0));
    }
    OP_TRY(vol_enqueue_batch(v, v->pend_F, v->pend_I, v->pend_P, nf, v->pend_fmt, false, false));
    if (rs >= 0) { v->ring[rs].busy_seq = v->seq; v->log.back().ring_slot = rs; }
    return OP_OK;
}

// ---- host-image staging ring (op_volume_integrate with OP_MEM_HOST) ------------------------------------------------
// Helper threads for the pageable -> pinned copy: one host core moves ~10-15 GB/s, i.e. 2.1 MB frames at 5-7 k frames/s;
// the caller's thread plus two helpers share the chunks of a frame.  The helpers spin briefly after work (calls arrive
// every < 100 us while a sequence is being fused) and sleep on a condition variable when idle.
class CopyPool {
  public:
    static CopyPool& get() { static CopyPool p; return p; }
    void copy2(void* d0, const void* s0, size_t n0, void* d1, const void* s1, size_t n1) {
        if (helpers_.empty() || n0 + n1 < (1u << 18)) { std::memcpy(d0, s0, n0); if (n1) std::memcpy(d1, s1, n1); return; }
        std::lock_guard<std::mutex> call(call_mutex_); // one copy at a time (volumes on several threads share the pool)
        Job job[2] = {{(char*)d0, (const char*)s0, n0}, {(char*)d1, (const char*)s1, n1}};
        const size_t chunks = (n0 + kChunk - 1) / kChunk + (n1 + kChunk - 1) / kChunk;
        uint64_t epoch;
        {   // publish the job under the lock the helpers take their snapshot under: a helper sees all of it or none
            std::lock_guard<std::mutex> lk(m_);
            job_[0] = job[0]; job_[1] = job[1];
            total_ = chunks;
            epoch = ++epoch_;
            done_.store(0, std::memory_order_relaxed);
            next_.store((epoch & 0xffffffffull) << 32, std::memory_order_release); // chunk counter tagged with the epoch
            epoch_hint_.store(epoch_, std::memory_order_release);
        }
        cv_.notify_all();
        work(epoch, job, chunks);
        // Every chunk is DRAWN with a compare-exchange on the epoch-tagged counter and counted in done_ after its memcpy, so
        // done_ == chunks means nobody is still copying.  A helper that wakes up late holds a snapshot of THIS job and this
        // epoch: its first draw fails against the next job's tag and it leaves without touching anything.
        while (done_.load(std::memory_order_acquire) < chunks) __builtin_ia32_pause();
    }
  private:
    static constexpr size_t kChunk = 1u << 17;
    struct Job { char* d; const char* s; size_t n; };
    CopyPool() {
        int n = 2;
        if (const char* e = std::getenv("ONEPIECE_HIP_COPY_THREADS")) n = std::atoi(e);
        if ((int)std::thread::hardware_concurrency() <= 2) n = 0;
        for (int i = 0; i < n && i < 8; ++i) helpers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++epoch_; epoch_hint_.store(epoch_, std::memory_order_release); }
        cv_.notify_all();
        for (auto& t : helpers_) t.join();
    }
    // copies chunks of the job of `epoch` (the caller's own copy of it) until none is left or the shared counter has moved on
    void work(uint64_t epoch, const Job job[2], size_t total) {
        const size_t c0 = (job[0].n + kChunk - 1) / kChunk;
        const uint64_t tag = (epoch & 0xffffffffull) << 32;
        for (;;) {
            uint64_t v = next_.load(std::memory_order_acquire);
            for (;;) {
                if ((v & 0xffffffff00000000ull) != tag || (v & 0xffffffffull) >= total) return;
                if (next_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) break;
            }
            const size_t c = (size_t)(v & 0xffffffffull);
            const Job& j = c < c0 ? job[0] : job[1];
            const size_t off = (c < c0 ? c : c - c0) * kChunk;
            std::memcpy(j.d + off, j.s + off, std::min(kChunk, j.n - off));
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    // takes a snapshot of a new epoch's job under the lock, or reports that there is none / that it is time to stop
    int poll(uint64_t& seen, Job job[2], size_t& total) {
        if (epoch_hint_.load(std::memory_order_acquire) == seen) return 0; // nothing new: do not touch the lock while spinning
        std::lock_guard<std::mutex> lk(m_);
        if (stop_) return -1;
        if (epoch_ == seen) return 0;
        seen = epoch_;
        job[0] = job_[0]; job[1] = job_[1]; total = total_;
        return 1;
    }
    void loop() {
        uint64_t seen = 0;
        Job job[2] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
        size_t total = 0;
        for (;;) {
            // spin for a while (a frame arrives every < 100 us during a sequence), then sleep
            int got = 0;
            for (int spin = 0; spin < 20000 && got == 0; ++spin) {
                got = poll(seen, job, total);
                if (got == 0) __builtin_ia32_pause();
            }
            if (got == 0) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                continue; // poll() takes the snapshot under the lock
            }
            if (got < 0) return;
            work(seen, job, total);
        }
    }
    std::vector<std::thread> helpers_;
    std::mutex m_, call_mutex_;
    std::condition_variable cv_;
    uint64_t epoch_ = 0;
    bool stop_ = false;
    Job job_[2];
    size_t total_ = 0;
    std::atomic<uint64_t> next_{0};       // (epoch & 0xffffffff) << 32 | next chunk index
    std::atomic<size_t> done_{0};
    std::atomic<uint64_t> epoch_hint_{0}; // lock-free mirror of epoch_ for the spinning helpers
};

// op_device_write: pageable host memory -> device memory.  A plain hipMemcpy from pageable memory makes the runtime pin the caller's pages for
// the transfer (0.5-0.7 ms for a 0.9 MB image); like the volume's staging ring this goes through a pinned buffer instead -- the parts are
// copied next to each other by the caller's thread and the CopyPool helpers, then ONE DMA moves the span they cover.
int write_staged(void* dst, size_t n_parts, const void* const* parts, const size_t* bytes, const size_t* offsets, int device) {
    struct Stage { std::mutex mu; void* pinned = nullptr; size_t cap = 0; hipStream_t stream = nullptr; };
    static Stage stage[16];
    Stage& S = stage[device & 15];
    size_t lo = (size_t)-1, hi = 0;
    for (size_t i = 0; i < n_parts; ++i) {
        if (!parts[i] || !bytes[i]) return fail(OP_ERR_INVALID, "op_device_write: empty part");
        lo = std::min(lo, offsets[i]); hi = std::max(hi, offsets[i] + bytes[i]);
    }
    if (n_parts == 0 || n_parts > 2) return fail(OP_ERR_INVALID, "op_device_write: one or two parts per call");
    const size_t span = hi - lo;
    std::lock_guard<std::mutex> lock(S.mu);
    hipError_t e = hipSuccess;
    if (S.cap < span) {
        if (S.pinned) (void)hipHostFree(S.pinned);
        S.pinned = nullptr; S.cap = 0;
        const size_t want = span < (4u << 20) ? (4u << 20) : span;
        e = hipHostMalloc(&S.pinned, want, hipHostMallocDefault);
        if (e == hipSuccess) S.cap = want;
    }
    if (e == hipSuccess && !S.stream) e = hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        char* h = (char*)S.pinned;
        CopyPool::get().copy2(h + (offsets[0] - lo), parts[0], bytes[0], n_parts > 1 ? h + (offsets[1] - lo) : nullptr, n_parts > 1 ? parts[1] : nullptr,
                              n_parts > 1 ? bytes[1] : 0);
        e = hipMemcpyAsync((char*)dst + lo, S.pinned, span, hipMemcpyHostToDevice, S.stream); // (a gap between the parts carries staging leftovers: it is the caller's padding)
        if (e == hipSuccess) e = hipStreamSynchronize(S.stream); // complete on return: the staging buffer is free again, the data is visible to every stream
    }
    if (e != hipSuccess) return fail(OP_ERR_HIP, "op_device_write: %s", hipGetErrorString(e));
    return OP_OK;
}

int vol_ring_alloc(op_volume* v) {
    const size_t npx = (size_t)v->cam.width * v->cam.height;
    if (v->ring_px >= npx && v->copy_stream) return OP_OK;
    OP_TRY(vol_check(v)); // nothing in flight may still read the old slots
    if (!v->copy_stream) OP_HIP(hipStreamCreateWithFlags(&v->copy_stream, hipStreamNonBlocking));
    for (auto& r : v->ring) {
        if (r.d_depth) op::cached_free(r.d_depth);
        if (r.d_rgb) op::cached_free(r.d_rgb);
        if (r.h_depth) op::cached_free(r.h_depth);
        if (r.h_rgb) op::cached_free(r.h_rgb);
        r.d_depth = r.h_depth = nullptr; r.d_rgb = r.h_rgb = nullptr; r.busy_seq = 0;
        OP_HIP(op::cached_malloc(&r.d_depth, (size_t)kMaxBatch * npx * 4));
        OP_HIP(op::cached_malloc((void**)&r.d_rgb, (size_t)kMaxBatch * npx * 3));
        OP_HIP(op::cached_host_malloc(&r.h_depth, (size_t)kMaxBatch * npx * 4));
        OP_HIP(op::cached_host_malloc((void**)&r.h_rgb, (size_t)kMaxBatch * npx * 3));
        if (!r.copied) OP_HIP(hipEventCreateWithFlags(&r.copied, hipEventDisableTiming));
    }
    v->ring_px = npx;
    v->ring_cur = -1;
    return OP_OK;
}

// The DMA of the staged positions that have not been sent: consecutive frames go as ONE span per image kind -- a copy of ~1 MB runs at 33 GB/s over
// this PCIe link, one of >= 4 MB at 53 GB/s (tests/tools/pcie_probe.py).
#ifndef OP_RING_DMA_FRAMES
#define OP_RING_DMA_FRAMES 8
#endif
int vol_ring_send(op_volume* v, op_volume::RingSlot& r) {
    if (r.dma_hi <= r.dma_lo) return OP_OK;
    const size_t npx = (size_t)v->cam.width * v->cam.height, cbytes = npx * 3, n = (size_t)(r.dma_hi - r.dma_lo);
    OP_HIP(hipMemcpyAsync((char*)r.d_depth + (size_t)r.dma_lo * r.dma_dbytes, (const char*)r.h_depth + (size_t)r.dma_lo * r.dma_dbytes, n * r.dma_dbytes, hipMemcpyHostToDevice, v->copy_stream));
    OP_HIP(hipMemcpyAsync(r.d_rgb + (size_t)r.dma_lo * cbytes, r.h_rgb + (size_t)r.dma_lo * cbytes, n * cbytes, hipMemcpyHostToDevice, v->copy_stream));
    r.dma_lo = r.dma_hi;
    return OP_OK;
}

// Stages one host frame into position `pos` of the batch being assembled: pageable -> pinned on the host (parallel); the
// asynchronous DMA on the copy stream follows when OP_RING_DMA_FRAMES consecutive frames are waiting, or when the batch is launched.
// The caller's buffers are free again when this returns.  (A batch has one depth format: frames lie dbytes apart.)
int vol_ring_stage(op_volume* v, const void** depth, int depth_fmt, const unsigned char** rgb, int pos) {
    OP_TRY(vol_ring_alloc(v));
    if (v->ring_cur < 0) { // first host frame of this batch: take the next slot
        const int rs = (int)(v->ring_next++ % (unsigned)op_volume::kRing);
        vol_retire(v);
        if (v->ring[rs].busy_seq != 0) { // its batch is not confirmed yet (only when the GPU is the bottleneck)
            const int keep_n = v->pend_n;
            v->pend_n = 0;                 // vol_check must not flush the half-assembled batch
            const int rc = vol_check(v);
            v->pend_n = keep_n;
            OP_TRY(rc);
        }
        v->ring_cur = rs;
        v->ring[rs].dma_lo = v->ring[rs].dma_hi = 0;
    }
    op_volume::RingSlot& r = v->ring[v->ring_cur];
    const size_t npx = (size_t)v->cam.width * v->cam.height, dbytes = npx * (depth_fmt == OP_DEPTH_U16 ? 2 : 4), cbytes = npx * 3;
    if (r.dma_hi > r.dma_lo && (pos != r.dma_hi || dbytes != r.dma_dbytes)) OP_TRY(vol_ring_send(v, r)); // not adjacent to what is waiting (device frames in between)
    char* hd = (char*)r.h_depth + (size_t)pos * dbytes;
    unsigned char* hc = r.h_rgb + (size_t)pos * cbytes;
    CopyPool::get().copy2(hd, *depth, dbytes, hc, *rgb, cbytes);
    if (r.dma_hi == r.dma_lo) r.dma_lo = pos;
    r.dma_hi = pos + 1;
    r.dma_dbytes = dbytes;
    if (r.dma_hi - r.dma_lo >= OP_RING_DMA_FRAMES) OP_TRY(vol_ring_send(v, r));
    *depth = (char*)r.d_depth + (size_t)pos * dbytes;
    *rgb = r.d_rgb + (size_t)pos * cbytes;
    return OP_OK;
}

// The fusion kernels address a batch of kMaxBatch packed frames (8 B per pixel) with 32-bit byte offsets -- frame f starts at f * npix * 8
// (k_integrate's buffer resource) -- and multiply image rows with 24-bit integer multiplies: 2^24 pixels (4096 x 4096) with batches of
// up to 32 frames, 2^23 with the 64-frame build (-DOP_MAX_BATCH=64).
constexpr long long kMaxPixels = (1LL << 29) / kMaxBatch < (1LL << 24) ? (1LL << 29) / kMaxBatch : (1LL << 24);
static_assert((unsigned long long)kMaxBatch * (unsigned long long)kMaxPixels * 8ull <= (1ull << 32), "the last frame of a batch must start and end below 2^32 bytes");
int check_cam(const op_camera* cam) {
    if (!cam || cam->width <= 0 || cam->height <= 0 || (long long)cam->width * cam->height > kMaxPixels)
        return fail(OP_ERR_INVALID, "invalid camera (images of up to %lld pixels are supported)", kMaxPixels);
    return OP_OK;
}

// Host-side helper of the .map stream code: f(block_begin, block_end) on up to 16 host threads (the formatting loops are
// per-block independent once the per-block offsets are known; one thread formats ~0.1 GB/s of this stream).
template <class F>
void for_block_ranges(size_t n, F f) {
    size_t nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
    if (n < 1024 || nt == 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        const size_t lo = t * per, hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back([=] { f(lo, hi); });
    }
    for (auto& x : th) x.join();
}

} // namespace

extern "C" {

int op_abi_version(void) { return OP_ABI_VERSION; }

int op_device_alloc(size_t bytes, int device, void** device_ptr) {
    if (!device_ptr || bytes == 0) return fail(OP_ERR_INVALID, "op_device_alloc: null argument");
    *device_ptr = nullptr;
    OP_TRY(op::use_device(device));
    OP_HIP(op::cached_malloc(device_ptr, bytes));
    return OP_OK;
}

int op_device_write(void* device_ptr, size_t n_parts, const void* const* parts, const size_t* bytes, const size_t* offsets, int device) {
    if (!device_ptr || !parts || !bytes || !offsets) return fail(OP_ERR_INVALID, "op_device_write: null argument");
    OP_TRY(op::use_device(device));
    return write_staged(device_ptr, n_parts, parts, bytes, offsets, device);
}

int op_device_upload(const void* host, size_t bytes, int device, void** device_ptr) {
    if (!host || !device_ptr || bytes == 0) return fail(OP_ERR_INVALID, "op_device_upload: null argument");
    OP_TRY(op_device_alloc(bytes, device, device_ptr));
    const size_t zero = 0;
    const int rc = write_staged(*device_ptr, 1, &host, &bytes, &zero, device);
    if (rc != OP_OK) { op::cached_free(*device_ptr); *device_ptr = nullptr; }
    return rc;
}

int op_device_release(void* device_ptr, int device) {
    if (!device_ptr) return OP_OK;
    OP_TRY(op::use_device(device));
    op::cached_free(device_ptr);
    return OP_OK;
}

int op_runtime_hw_queues(int* requested) {
    if (!requested) return fail(OP_ERR_INVALID, "null argument");
    const char* e = std::getenv("GPU_MAX_HW_QUEUES");
    *requested = e ? std::atoi(e) : 4; // 4 = the runtime's own default
    return OP_OK;
}
const char* op_last_error(void) { return op::g_last_error; }

int op_device_count(int* count) {
    if (!count) return fail(OP_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = e == hipSuccess ? n : 0;
    return OP_OK;
}

int op_camera_preset(int type, op_camera* out) {
    if (!out) return fail(OP_ERR_INVALID, "null camera");
    if (type == 0) *out = op_camera{517.3f, 516.5f, 318.6f, 255.3f, 640, 480, 5000.0f};          // Camera.h:78-92
    else if (type == 1) *out = op_camera{514.817f, 515.375f, 318.771f, 238.447f, 640, 480, 1000.0f}; // Camera.h:94-104
    else return fail(OP_ERR_INVALID, "unknown camera preset %d", type);
    return OP_OK;
}

int op_mat4_inverse(const float m[16], float out[16]) {
    if (!m || !out) return fail(OP_ERR_INVALID, "null matrix");
    op_host::mat4_inverse(m, out);
    return OP_OK;
}

uint64_t op_hash_key(int32_t x, int32_t y, int32_t z) { return op_host::hash_key(x, y, z); }

int op_frustum_planes(const op_camera* cam, const float pose[16], float far_dist, float near_dist, float planes[24]) {
    OP_TRY(check_cam(cam));
    if (!pose || !planes) return fail(OP_ERR_INVALID, "null argument");
    op_host::CameraPOD c{cam->fx, cam->fy, cam->cx, cam->cy, cam->width, cam->height, cam->depth_scale};
    op_host::frustum_planes(c, pose, far_dist, near_dist, planes);
    return OP_OK;
}

int op_frustum_from_camera(const op_camera* cam, const float pose[16], float far_dist, float near_dist, float planes[24], float corners[24]) {
    OP_TRY(check_cam(cam));
    if (!pose || !planes) return fail(OP_ERR_INVALID, "null argument");
    op_host::CameraPOD c{cam->fx, cam->fy, cam->cx, cam->cy, cam->width, cam->height, cam->depth_scale};
    op_host::frustum_planes(c, pose, far_dist, near_dist, planes, corners);
    return OP_OK;
}

int op_frustum_from_vectors(const float forward[3], const float position[3], const float right[3], const float up[3], float far_dist, float near_dist,
                            float fov, float aspect, float planes[24], float corners[24]) {
    if (!forward || !position || !right || !up || !planes) return fail(OP_ERR_INVALID, "null argument");
    op_host::frustum_from_vectors(forward, position, right, up, far_dist, near_dist, fov, aspect, planes, corners);
    return OP_OK;
}

int op_get_sdf(const op_camera* cam, const float point[3], const float pose[16], const float* pose_inv, const void* depth, int depth_fmt, float* sdf) {
    OP_TRY(check_cam(cam));
    if (!point || !pose || !depth || !sdf) return fail(OP_ERR_INVALID, "null argument");
    float inv[16];
    if (pose_inv) std::memcpy(inv, pose_inv, sizeof(inv));
    else op_host::mat4_inverse(pose, inv);                          // Integrator.cpp:18
    // (pose_inv * (p, 1)).head<3>() accumulated column by column (:19)
    float q[3];
    for (int r = 0; r < 3; ++r) q[r] = ((inv[4 * r] * point[0] + inv[4 * r + 1] * point[1]) + inv[4 * r + 2] * point[2]) + inv[4 * r + 3] * 1.0f;
    const int u = px_round_dp((cam->fx * q[0]) / q[2], cam->cx);    // :20-21: float product and quotient, the sum in double, truncated
    const int w = px_round_dp((cam->fy * q[1]) / q[2], cam->cy);
    *sdf = 999.0f;
    if (w < 0 || w >= cam->height || u < 0 || u >= cam->width) return OP_OK;   // :23-24
    const size_t at = (size_t)w * cam->width + u;
    const float d = depth_fmt == OP_DEPTH_U16 ? (float)((const unsigned short*)depth)[at] / cam->depth_scale : ((const float*)depth)[at]; // :26-29
    if (d <= 0) return OP_OK;                                        // :30
    *sdf = d - q[2];
    return OP_OK;
}

int op_debug_project_px(float a, float c, int fast) {
    return fast ? px_round_sp(a, px_split(c)) : px_round_dp(a, c);
}

int op_debug_project_uv(float fx, float fy, float cx, float cy, const float* X, const float* Y, const float* Z, size_t n, int device, int32_t* out) {
    if (!X || !Y || !Z || !out) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(op::use_device(device));
    if (n == 0) return OP_OK;
    float* d_in = nullptr;
    int* d_out = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_in, 3 * n * sizeof(float)));
    hipError_t e = op::cached_malloc((void**)&d_out, 4 * n * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_in, X, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + n, Y, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + 2 * n, Z, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        CamParams C{};
        C.fx = fx; C.fy = fy; C.cx = cx; C.cy = cy; C.width = 640; C.height = 480;
        C.ax = px_axis(cx, C.width); C.ay = px_axis(cy, C.height);
        C.fast_px = C.ax.exact && C.ay.exact;
        hipLaunchKernelGGL(k_debug_project_uv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, C, (const float*)d_in,
                           (const float*)(d_in + n), (const float*)(d_in + 2 * n), n, d_out);
        e = hipMemcpy(out, d_out, 4 * n * sizeof(int), hipMemcpyDeviceToHost);
    }
    op::cached_free(d_in);
    if (d_out) op::cached_free(d_out);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "op_debug_project_uv failed: %s", hipGetErrorString(e));
    return OP_OK;
}

int op_se3_exp(const float x[6], float T[16]) {
    if (!x || !T) return fail(OP_ERR_INVALID, "null argument");
    op_host::se3_exp(x, T);
    return OP_OK;
}

int op_volume_create(const op_camera* cam, float voxel_res, float truncation, float far_dist, float near_dist, int device,
                     uint64_t max_blocks, op_volume** out) {
    if (!out) return fail(OP_ERR_INVALID, "null out");
    *out = nullptr;
    OP_TRY(check_cam(cam));
    if (!(voxel_res > 0) || !(truncation > 0)) return fail(OP_ERR_INVALID, "voxel_res and truncation must be > 0");
    OP_TRY(op::use_device(device));
    if (max_blocks == 0) max_blocks = 1u << 18; // initial capacity; the pool and the table grow on demand (vol_grow)
    if (max_blocks > kHardMaxBlocks) return fail(OP_ERR_INVALID, "max_blocks too large (limit %u)", kHardMaxBlocks);
    op_volume* v = new op_volume();
    v->device = device; v->cam = *cam; v->res = voxel_res; v->trunc = truncation; v->far_d = far_dist; v->near_d = near_dist;
    v->max_blocks = (unsigned)max_blocks;
    v->table_size = next_pow2(2ull * max_blocks);
    auto cleanup = [&](int rc) { op_volume_destroy(v); return rc; };
#define OP_HIP_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cleanup(fail(OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
    OP_HIP_C(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    OP_HIP_C(op::cached_malloc((void**)&v->tkeys, sizeof(unsigned long long) * (size_t)v->table_size));
    OP_HIP_C(op::cached_malloc((void**)&v->tvals, sizeof(int) * (size_t)v->table_size));
    OP_HIP_C(op::cached_malloc((void**)&v->bmask, sizeof(bmask_t) * (size_t)v->table_size));
    OP_HIP_C(op::cached_malloc((void**)&v->blist, sizeof(int) * (size_t)(KC_BANDS ? kBands : 1) * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->sel_partial, sizeof(unsigned long long) * kPartialGrid));
    OP_HIP_C(op::cached_malloc((void**)&v->keys, sizeof(int) * 3 * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->pool, sizeof(float) * kBlockFloats * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->n_blocks, sizeof(unsigned)));
    OP_HIP_C(op::cached_malloc((void**)&v->sel_list, sizeof(int) * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->sel_cand, sizeof(unsigned long long) * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->state, sizeof(State)));
    OP_HIP_C(op::cached_malloc((void**)&v->upd_partial, sizeof(unsigned long long) * kPartialGrid));
    OP_HIP_C(op::cached_malloc((void**)&v->chg_partial, sizeof(unsigned long long) * 2 * kPartialGrid));
    OP_HIP_C(op::cached_host_malloc((void**)&v->hstat, 2 * sizeof(unsigned)));
    v->hstat[0] = 0; v->hstat[1] = 0;
    OP_HIP_C(hipHostGetDevicePointer((void**)&v->hstat_dev, v->hstat, 0));
#undef OP_HIP_C
    hipLaunchKernelGGL(k_fill_pool, dim3(4096), dim3(256), 0, v->stream, v->pool, (size_t)0, (size_t)v->max_blocks);
    int rc = vol_reset(v);
    if (rc != OP_OK) return cleanup(rc);
    if (hipStreamSynchronize(v->stream) != hipSuccess) return cleanup(fail(OP_ERR_HIP, "volume initialisation failed"));
    *out = v;
    return OP_OK;
}

int op_volume_destroy(op_volume* v) {
    if (!v) return OP_OK;
#ifdef KC_TRACE
    if (hipSetDevice(v->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
        const int nw = kColGrid * (8 / KC_ZT);
        std::vector<unsigned long long> t((size_t)4096 * 4 * 8);
        if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_kc_trace), t.size() * 8) == hipSuccess) {
            double sum[8] = {0}, mx[8] = {0};
            for (int w = 0; w < nw; ++w)
                for (int k = 0; k < 8; ++k) { const double d = (double)t[(size_t)w * 8 + k]; sum[k] += d; mx[k] = std::max(mx[k], d); }
            const double n = nw;
            fprintf(stderr, "kc trace (shader cycles per wave of the last launch, %d waves; mean/max): draw+list %.0f/%.0f metadata+voxels %.0f/%.0f frames %.0f/%.0f stores %.0f/%.0f barrier %.0f/%.0f tail %.0f/%.0f | blocks %.1f/%.0f frames applied %.0f/%.0f\n",
                    nw, sum[0] / n, mx[0], sum[1] / n, mx[1], sum[2] / n, mx[2], sum[3] / n, mx[3], sum[4] / n, mx[4], sum[5] / n, mx[5], sum[6] / n, mx[6], sum[7] / n, mx[7]);
        }
    }
#endif
#ifdef KB_TRACE
    if (hipSetDevice(v->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
        const int nw = std::max(8, std::min(kSelectGrid, (KB_VOTE_WGS / kMaxBatch + 7) / 8 * 8)) * kMaxBatch * 4; // waves of a full batch's launch
        std::vector<unsigned long long> t((size_t)nw * 8);
        if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_kb_trace), t.size() * 8) == hipSuccess) {
            double sum[8] = {0}, mx[8] = {0}, tot_max = 0, tot_sum = 0;
            std::vector<double> tots;
            for (int w = 0; w < nw; ++w) {
                double tot = 0;
                for (int k = 0; k < 8; ++k) { const double d = (double)t[(size_t)w * 8 + k]; sum[k] += d; mx[k] = std::max(mx[k], d); if (k < 4) tot += d; }
                tot_max = std::max(tot_max, tot); tot_sum += tot; tots.push_back(tot);
            }
            std::sort(tots.begin(), tots.end());
            const double n = nw;
            fprintf(stderr, "kb trace (shader cycles, %d waves; mean/max): total %.0f/%.0f (median %.0f, 90%% %.0f, 99%% %.0f) setup %.0f/%.0f coarse %.0f/%.0f exact %.0f/%.0f barrier %.0f/%.0f | rounds %.2f/%.0f exact tests %.2f/%.0f\n",
                    nw, tot_sum / n, tot_max, tots[tots.size() / 2], tots[tots.size() * 9 / 10], tots[tots.size() * 99 / 100], sum[0] / n, mx[0], sum[1] / n, mx[1], sum[2] / n, mx[2], sum[3] / n, mx[3], sum[4] / n, mx[4], sum[5] / n, mx[5]);
        }
    }
#endif
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    for (auto e : v->prof_events) (void)hipEventDestroy(e);
    void* ptrs[] = {v->tkeys, v->tvals, v->keys, v->pool, v->n_blocks, v->bmask, v->blist, v->sel_list, v->sel_cand, v->state,
                    v->partial, v->pimg, v->ptile, v->sbits, v->upd_partial, v->sel_partial, v->chg_partial, v->img_depth, v->img_rgb, v->unpack_slots};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (v->copy_stream) (void)hipStreamSynchronize(v->copy_stream);
    for (auto& r : v->ring) {
        if (r.d_depth) op::cached_free(r.d_depth);
        if (r.d_rgb) op::cached_free(r.d_rgb);
        if (r.h_depth) op::cached_free(r.h_depth);
        if (r.h_rgb) op::cached_free(r.h_rgb);
        if (r.copied) (void)hipEventDestroy(r.copied);
    }
    if (v->hstat) op::cached_free(v->hstat);
    if (v->copy_stream) (void)hipStreamDestroy(v->copy_stream);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
    return OP_OK;
}

#define OP_VOL(v)                                              \
    if (!(v)) return fail(OP_ERR_INVALID, "null volume");      \
    OP_HIP(hipSetDevice((v)->device))

// Queued and in-flight frames were accepted under the old setting: a CHANGE of resolution, truncation or camera waits until
// they are fused for good (vol_check: flush, synchronise, grow + replay if a batch ran out of pool).  A replay rebuilds its
// kernel parameters from the volume's current settings (BatchRec keeps poses, frustum planes and image pointers only), so no
// batch may still be replayable when a setting changes.
int op_volume_set_resolution(op_volume* v, float voxel_res) {
    OP_VOL(v);
    if (!(voxel_res > 0)) return fail(OP_ERR_INVALID, "voxel_res must be > 0");
    if (voxel_res == v->res) return OP_OK;
    OP_TRY(vol_check(v));
    v->res = voxel_res;
    return OP_OK;
}
int op_volume_set_truncation(op_volume* v, float truncation) {
    OP_VOL(v);
    if (truncation == v->trunc) return OP_OK;
    OP_TRY(vol_check(v));
    v->trunc = truncation;
    return OP_OK;
}
int op_volume_set_camera(op_volume* v, const op_camera* cam) {
    OP_VOL(v);
    OP_TRY(check_cam(cam));
    if (std::memcmp(cam, &v->cam, sizeof(op_camera)) == 0) return OP_OK;
    OP_TRY(vol_check(v));
    v->cam = *cam;
    return OP_OK;
}
int op_volume_set_near_far(op_volume* v, float near_dist, float far_dist) {
    OP_VOL(v);
    OP_TRY(vol_flush(v)); // queued frames were accepted under the old setting; launched batches carry their frustum planes with them
    v->near_d = near_dist; v->far_d = far_dist;
    return OP_OK;
}

int op_volume_set_option(op_volume* v, int option, int value) {
    OP_VOL(v);
    if (option == OP_VOLUME_OPT_SELECT) { // which form of the selection step batches take (results are identical; a tuning / test knob)
        if (value != OP_VOLUME_SELECT_AUTO && value != OP_VOLUME_SELECT_DIRECT && (value < 1 || value > (int)kVoteCap))
            return fail(OP_ERR_INVALID, "op_volume_set_option: OP_VOLUME_OPT_SELECT takes OP_VOLUME_SELECT_AUTO, OP_VOLUME_SELECT_DIRECT or 1..%u super-blocks per frame", kVoteCap);
        v->select_mode = value;
        return OP_OK;
    }
    if (option != OP_VOLUME_OPT_UPDATE) return fail(OP_ERR_INVALID, "op_volume_set_option: unknown option %d", option);
    if (value != OP_VOLUME_UPDATE_EXACT && value != OP_VOLUME_UPDATE_SUM_FORM) return fail(OP_ERR_INVALID, "op_volume_set_option: bad value %d", value);
    if (value == v->update_mode) return OP_OK;
    OP_TRY(vol_check(v)); // queued and replayable batches were accepted under the old setting (see op_volume_set_resolution)
    v->update_mode = value;
    return OP_OK;
}

int op_volume_clear(op_volume* v) {
    OP_VOL(v);
    v->pend_n = 0; // queued frames would be wiped anyway
    v->ring_cur = -1;
    unsigned n = 0;
    if (v->copy_stream) OP_HIP(hipStreamSynchronize(v->copy_stream));
    OP_HIP(hipStreamSynchronize(v->stream));
    v->log.clear(); // whatever was in flight (fused or poisoned) is wiped with the volume
    v->plain = true; v->plain_from = 0;
    for (auto& r : v->ring) r.busy_seq = 0;
    if (v->hstat) v->hstat[1] = 0;
    OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
    if (n > v->max_blocks) n = v->max_blocks;
    if (n) hipLaunchKernelGGL(k_fill_pool, dim3(4096), dim3(256), 0, v->stream, v->pool, (size_t)0, (size_t)n);
    OP_TRY(vol_reset(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    return OP_OK;
}

int op_volume_sync(op_volume* v) {
    OP_VOL(v);
    return vol_check(v);
}

int op_volume_flush(op_volume* v) {
    OP_VOL(v);
    return vol_flush(v);
}

int op_volume_stream(op_volume* v, void** stream) {
    OP_VOL(v);
    if (!stream) return fail(OP_ERR_INVALID, "null stream out");
    *stream = (void*)v->stream;
    return OP_OK;
}

int op_volume_block_count(op_volume* v, size_t* n) {
    OP_VOL(v);
    if (!n) return fail(OP_ERR_INVALID, "null n");
    unsigned c = 0;
    OP_TRY(vol_block_count(v, &c));
    *n = c;
    return OP_OK;
}

int op_volume_compute_bounding(op_volume* v, const void* depth, int depth_fmt, int mem, const float pose[16], float max_pos[3],
                               float min_pos[3], size_t* n_inside) {
    OP_VOL(v);
    if (!depth || !pose) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_flush(v));
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, nullptr, mem));
    OP_TRY(vol_ensure_frame_buffers(v));
    BatchFwd F;
    BatchInv I;
    BatchPtrs Q{};
    frame_params(v, pose, nullptr, &F.f[0], &I.f[0]);
    Q.depth[0] = depth;
    const CamParams C = cam_params(v, depth_fmt);
    const int g1 = ka_grid(C.width, C.height);
    KaFwd A;
    A.f[0] = F.f[0];
    hipLaunchKernelGGL(k_prepare_frames, dim3(g1, 1), dim3(256), 0, v->stream, A, 0, C, Q, v->pimg, v->ptile, v->partial, v->state, (unsigned)(++v->seq),
                       (const unsigned*)v->n_blocks, v->hstat_dev);
    OP_HIP(hipGetLastError());
    // no KB / KC follows to consume and zero the frame's bounding accumulators: do it here (the rows are read below)
    OP_HIP(hipMemsetAsync(reinterpret_cast<char*>(v->state) + offsetof(State, acc), 0, sizeof(State::acc), v->stream));
    OP_HIP(hipStreamSynchronize(v->stream));
    std::vector<float> part((size_t)g1 * 8);
    OP_HIP(hipMemcpy(part.data(), v->partial, part.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    size_t inside = 0;
    for (int g = 0; g < g1; ++g) {
        for (int c = 0; c < 3; ++c) {
            mx[c] = std::max(mx[c], part[g * 8 + c]);
            mn[c] = std::min(mn[c], part[g * 8 + 3 + c]);
        }
        unsigned cnt;
        std::memcpy(&cnt, &part[g * 8 + 6], 4);
        inside += cnt;
    }
    if (max_pos) std::memcpy(max_pos, mx, sizeof(mx));
    if (min_pos) std::memcpy(min_pos, mn, sizeof(mn));
    if (n_inside) *n_inside = inside;
    return OP_OK;
}

int op_volume_prepare_cubes(op_volume* v, const void* depth, int depth_fmt, int mem, const float pose[16], const float* pose_inv,
                            int32_t* ids_xyz, size_t cap, size_t* n, size_t* n_candidates) {
    OP_VOL(v);
    if (!depth || !pose) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_flush(v));
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, nullptr, mem));
    BatchFwd F;
    BatchInv I;
    BatchPtrs Q{};
    frame_params(v, pose, pose_inv, &F.f[0], &I.f[0]);
    Q.depth[0] = depth;
    for (;;) { // a selection that outgrows the pool grows it (vol_check) and is then simply run again
        const unsigned cap_before = v->max_blocks;
        OP_TRY(vol_enqueue_batch(v, F, I, Q, 1, depth_fmt, /*select_only=*/true, /*record=*/true));
        OP_TRY(vol_check(v));
        if (v->max_blocks == cap_before) break;
    }
    StateHead st;
    OP_HIP(hipMemcpy(&st, v->state, sizeof(st), hipMemcpyDeviceToHost));
    unsigned long long n_cand0 = 0;
    OP_HIP(hipMemcpy(&n_cand0, reinterpret_cast<const char*>(v->state) + offsetof(State, n_cand), sizeof(n_cand0), hipMemcpyDeviceToHost));
    const size_t ns = std::min((size_t)st.n_rec, (size_t)v->max_blocks);
    if (n) *n = ns;
    if (n_candidates) *n_candidates = (size_t)n_cand0;
    if (ids_xyz && ns) {
        std::vector<int> list(ns);
        std::vector<unsigned long long> cand(ns);
        OP_HIP(hipMemcpy(list.data(), v->sel_list, ns * sizeof(int), hipMemcpyDeviceToHost));
        OP_HIP(hipMemcpy(cand.data(), v->sel_cand, ns * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned nb = 0;
        OP_HIP(hipMemcpy(&nb, v->n_blocks, sizeof(nb), hipMemcpyDeviceToHost));
        std::vector<int> keys((size_t)nb * 3);
        OP_HIP(hipMemcpy(keys.data(), v->keys, keys.size() * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<size_t> order(ns);
        std::iota(order.begin(), order.end(), (size_t)0);
        // candidate rank == position in the reference's i,j,k loop nest -> cube_id_list order
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cand[a] < cand[b]; });
        for (size_t i = 0; i < ns && i < cap; ++i) {
            const int idx = list[order[i]];
            for (int c = 0; c < 3; ++c) ids_xyz[3 * i + c] = keys[(size_t)idx * 3 + c];
        }
    }
    return OP_OK;
}

int op_volume_integrate(op_volume* v, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, const float pose[16],
                        const float* pose_inv) {
    OP_VOL(v);
    if (!depth || !rgb || !pose) return fail(OP_ERR_INVALID, "null argument");
    // Single frames are queued and fused kMaxBatch at a time (one launch group per batch instead of
    // per frame); every accessor, setter and synchronising call flushes the queue first, so the
    // deferral cannot be observed.  Host images are copied to the staging ring right here (the
    // caller's buffers are only borrowed for the call); device images are used in place at flush.
    if (v->pend_n > 0 && v->pend_fmt != depth_fmt) OP_TRY(vol_flush(v));
    const unsigned char* c = rgb;
    // The first host image of a volume (or a larger one) allocates the staging ring, and that launches whatever is queued
    // (vol_ring_alloc -> vol_check -> vol_flush): the queue position is only valid afterwards.
    if (mem == OP_MEM_HOST) OP_TRY(vol_ring_alloc(v));
    const int slot = v->pend_n;
    if (mem == OP_MEM_HOST) OP_TRY(vol_ring_stage(v, &depth, depth_fmt, &c, slot));
    frame_params(v, pose, pose_inv, &v->pend_F.f[slot], &v->pend_I.f[slot]);
    v->pend_P.depth[slot] = depth;
    v->pend_P.rgb[slot] = c;
    v->pend_fmt = depth_fmt;
    ++v->frames_accepted;
    if (++v->pend_n == kMaxBatch) return vol_flush(v);
    return OP_OK;
}

int op_volume_integrate_cubes(op_volume* v, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, const float pose[16], const float* pose_inv,
                              const int32_t* keys_xyz, size_t n) {
    OP_VOL(v);
    if (!depth || !rgb || !pose || (n && !keys_xyz)) return fail(OP_ERR_INVALID, "null argument");
    if (n == 0) return OP_OK;
    if (n > kHardMaxBlocks) return fail(OP_ERR_CAPACITY, "%zu cubes exceed the limit of %u blocks per volume", n, kHardMaxBlocks);
    unsigned have = 0;
    OP_TRY(vol_block_count(v, &have)); // flushes, synchronises, recovers
    OP_TRY(vol_reserve(v, (unsigned long long)have + n));
    const unsigned char* c = rgb;
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, &c, mem));
    int* d_keys = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_keys, n * 3 * sizeof(int)));
    hipError_t e = hipMemcpyAsync(d_keys, keys_xyz, n * 3 * sizeof(int), hipMemcpyHostToDevice, v->stream);
    int rc = OP_OK;
    if (e == hipSuccess) {
        BatchFwd F;
        BatchInv I;
        BatchPtrs Q{};
        frame_params(v, pose, pose_inv, &F.f[0], &I.f[0]);
        Q.depth[0] = depth; Q.rgb[0] = c;
        rc = vol_enqueue_batch(v, F, I, Q, 1, depth_fmt, false, false, d_keys, (unsigned)n);
        if (rc == OP_OK) rc = vol_check(v);
    }
    (void)hipStreamSynchronize(v->stream);
    op::cached_free(d_keys);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "copying the cube list failed: %s", hipGetErrorString(e));
    return rc;
}

int op_volume_integrate_sequence(op_volume* v, const void* depth, size_t depth_stride_bytes, int depth_fmt, const uint8_t* rgb,
                                 size_t rgb_stride_bytes, const float* poses, size_t n_frames) {
    OP_VOL(v);
    if (!depth || !rgb || !poses) return fail(OP_ERR_INVALID, "null argument");
    // The frames join the queue op_volume_integrate fills: full batches are launched as they complete, the remainder waits for the next
    // frames (of this or the next call) or for the first call that flushes -- a stream of calls fuses kMaxBatch frames per launch
    // whatever the calls' lengths are.  (The device images must stay valid until the next synchronising call, as for op_volume_integrate.)
    if (v->pend_n > 0 && v->pend_fmt != depth_fmt) OP_TRY(vol_flush(v));
    for (size_t f = 0; f < n_frames; ++f) {
        const int slot = v->pend_n;
        frame_params(v, poses + 16 * f, nullptr, &v->pend_F.f[slot], &v->pend_I.f[slot]);
        v->pend_P.depth[slot] = (const char*)depth + f * depth_stride_bytes;
        v->pend_P.rgb[slot] = rgb + f * rgb_stride_bytes;
        v->pend_fmt = depth_fmt;
        ++v->frames_accepted;
        if (++v->pend_n == kMaxBatch) OP_TRY(vol_flush(v));
    }
    return OP_OK;
}

int op_volume_progress(op_volume* v, uint64_t* frames_accepted, uint64_t* frames_done) {
    OP_VOL(v);
    vol_retire(v); // looks at the device's progress report; never blocks
    uint64_t open = (uint64_t)v->pend_n;
    for (const auto& r : v->log) open += (uint64_t)r.nf;
    if (frames_accepted) *frames_accepted = v->frames_accepted;
    if (frames_done) *frames_done = v->frames_accepted >= open ? v->frames_accepted - open : 0;
    return OP_OK;
}

int op_volume_stats(op_volume* v, uint64_t* frames, uint64_t* blocks_selected, uint64_t* voxels_visited, uint64_t* voxels_updated) {
    OP_VOL(v);
    OP_TRY(vol_check(v));
    unsigned long long stat_frames = 0;
    OP_HIP(hipMemcpy(&stat_frames, reinterpret_cast<const char*>(v->state) + offsetof(State, stat_frames), sizeof(stat_frames), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> part(2 * kPartialGrid);
    OP_HIP(hipMemcpy(part.data(), v->upd_partial, kPartialGrid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    OP_HIP(hipMemcpy(part.data() + kPartialGrid, v->sel_partial, kPartialGrid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long upd = 0, sel = 0;
    for (int i = 0; i < kPartialGrid; ++i) { upd += part[i]; sel += part[kPartialGrid + i]; }
    if (frames) *frames = stat_frames;
    if (blocks_selected) *blocks_selected = sel;
    if (voxels_visited) *voxels_visited = sel * (uint64_t)kVox;
    if (voxels_updated) *voxels_updated = upd;
    return OP_OK;
}

int op_volume_stats_launches(op_volume* v, uint64_t* launches, uint64_t* blocks_read, uint64_t* voxels_written, uint64_t* shader_cycles) {
    OP_VOL(v);
    OP_TRY(vol_check(v));
    struct { unsigned long long stat_frames, stat_launches, kc_t[kKcTSlots], stat_kc_ticks; } st; // that stretch of State (2 KB), not all 20 KB of it
    static_assert(offsetof(State, stat_kc_ticks) - offsetof(State, stat_frames) + 8 == sizeof(st), "contiguous statistics fields of State");
    OP_HIP(hipMemcpy(&st, reinterpret_cast<const char*>(v->state) + offsetof(State, stat_frames), sizeof(st), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> part(2 * kPartialGrid);
    OP_HIP(hipMemcpy(part.data(), v->chg_partial, 2 * kPartialGrid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long chg = 0, blk = 0;
    for (int i = 0; i < kPartialGrid; ++i) { chg += part[i]; blk += part[kPartialGrid + i]; }
    if (shader_cycles) { // the last launch has not been folded by a following batch yet
        unsigned long long kc = 0;
        for (int x = 0; x < kKcTSlots; ++x)
            if (st.kc_t[x] > kc) kc = st.kc_t[x];
        *shader_cycles = st.stat_kc_ticks + kc;
    }
    if (launches) *launches = st.stat_launches;
    if (blocks_read) *blocks_read = blk;
    if (voxels_written) *voxels_written = chg;
    return OP_OK;
}

int op_volume_growth_stats(op_volume* v, uint64_t* max_blocks, uint64_t* grows, uint64_t* replayed_batches) {
    OP_VOL(v);
    if (max_blocks) *max_blocks = v->max_blocks;
    if (grows) *grows = v->n_grows;
    if (replayed_batches) *replayed_batches = v->n_replayed;
    return OP_OK;
}

int op_volume_profile_enable(op_volume* v, int sample_every) {
    OP_VOL(v);
    OP_TRY(vol_flush(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    for (auto e : v->prof_events) (void)hipEventDestroy(e);
    v->prof_events.clear();
    v->prof_frames.clear();
    v->prof_every = sample_every > 0 ? sample_every : 0;
    v->prof_batch = 0;
    return OP_OK;
}

int op_volume_profile_read(op_volume* v, double ms_sum[3], uint64_t* n_launches, uint64_t* n_frames) {
    OP_VOL(v);
    if (!ms_sum || !n_launches || !n_frames) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_flush(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    ms_sum[0] = ms_sum[1] = ms_sum[2] = 0.0;
    const size_t n = v->prof_events.size() / 4;
    uint64_t frames = 0;
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            float ms = 0.0f;
            OP_HIP(hipEventElapsedTime(&ms, v->prof_events[4 * i + k], v->prof_events[4 * i + k + 1]));
            ms_sum[k] += ms;
        }
        frames += (uint64_t)v->prof_frames[i];
    }
    *n_launches = n;
    *n_frames = frames;
    return OP_OK;
}

int op_volume_has_cube(op_volume* v, int32_t x, int32_t y, int32_t z, int* present) {
    OP_VOL(v);
    if (!present) return fail(OP_ERR_INVALID, "null present");
    OP_TRY(vol_check(v));
    int* d_out = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_out, sizeof(int)));
    hipLaunchKernelGGL(k_has_cube, dim3(1), dim3(1), 0, v->stream, v->view(), x, y, z, d_out); // cube_map.find (CubeHandler.h:129-132)
    hipError_t e = hipMemcpyAsync(present, d_out, sizeof(int), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    op::cached_free(d_out);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "has_cube failed: %s", hipGetErrorString(e));
    return OP_OK;
}

int op_volume_download(op_volume* v, int32_t* keys_xyz, float* voxels_aos, size_t cap, size_t* n) {
    OP_VOL(v);
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    if (n) *n = nb;
    const size_t take = std::min((size_t)nb, cap);
    if (keys_xyz && take) OP_HIP(hipMemcpy(keys_xyz, v->keys, take * 3 * sizeof(int), hipMemcpyDeviceToHost));
    if (voxels_aos && take) {
        const size_t chunk = 8192; // 80 MiB of staging
        float* stage = nullptr;
        OP_HIP(op::cached_malloc((void**)&stage, std::min(chunk, take) * kBlockFloats * sizeof(float)));
        for (size_t first = 0; first < take; first += chunk) {
            const size_t cnt = std::min(chunk, take - first);
            hipLaunchKernelGGL(k_export_aos, dim3((unsigned)cnt), dim3(512), 0, v->stream, (const float*)v->pool, first, stage);
            hipError_t e = hipStreamSynchronize(v->stream);
            if (e == hipSuccess)
                e = hipMemcpy(voxels_aos + first * kBlockFloats, stage, cnt * kBlockFloats * sizeof(float), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { op::cached_free(stage); return fail(OP_ERR_HIP, "download failed: %s", hipGetErrorString(e)); }
        }
        op::cached_free(stage);
    }
    return OP_OK;
}

int op_volume_upload(op_volume* v, const int32_t* keys_xyz, const float* voxels_aos, size_t n) {
    OP_VOL(v);
    if (n == 0) return OP_OK;
    if (!keys_xyz || !voxels_aos) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_flush(v)); // frames queued by op_volume_integrate come BEFORE the upload, as the caller issued them
    // later duplicates override earlier ones, like repeated map assignment; the device insert needs distinct keys
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return std::lexicographical_compare(keys_xyz + 3 * a, keys_xyz + 3 * a + 3, keys_xyz + 3 * b, keys_xyz + 3 * b + 3);
    });
    std::vector<size_t> uniq;
    for (size_t i = 0; i < n; ++i) {
        const bool last = i + 1 == n || !std::equal(keys_xyz + 3 * order[i], keys_xyz + 3 * order[i] + 3, keys_xyz + 3 * order[i + 1]);
        if (last) uniq.push_back(order[i]);
    }
    {   // room for every new block up front (the pool grows; nothing can overflow below)
        unsigned nb = 0;
        OP_TRY(vol_block_count(v, &nb));
        OP_TRY(vol_reserve(v, (unsigned long long)nb + uniq.size()));
    }
    { unsigned nb = 0; OP_TRY(vol_block_count(v, &nb)); vol_mark_foreign(v, (unsigned long long)nb + uniq.size()); } // caller-supplied voxel data in every block that exists after this call
    const size_t chunk = 8192;
    int *d_keys = nullptr, *d_slots = nullptr;
    float* d_vox = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_keys, chunk * 3 * sizeof(int)));
    OP_HIP(op::cached_malloc((void**)&d_slots, chunk * sizeof(int)));
    OP_HIP(op::cached_malloc((void**)&d_vox, chunk * kBlockFloats * sizeof(float)));
    std::vector<int> hk(chunk * 3);
    std::vector<float> hv(chunk * kBlockFloats);
    int rc = OP_OK;
    for (size_t first = 0; first < uniq.size() && rc == OP_OK; first += chunk) {
        const size_t cnt = std::min(chunk, uniq.size() - first);
        for (size_t i = 0; i < cnt; ++i) {
            std::memcpy(&hk[3 * i], keys_xyz + 3 * uniq[first + i], 3 * sizeof(int));
            std::memcpy(&hv[i * kBlockFloats], voxels_aos + uniq[first + i] * kBlockFloats, kBlockFloats * sizeof(float));
        }
        hipError_t e = hipMemcpy(d_keys, hk.data(), cnt * 3 * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_vox, hv.data(), cnt * kBlockFloats * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_insert_keys, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, v->stream, v->view(), (const int*)d_keys, cnt, d_slots, v->state);
            hipLaunchKernelGGL(k_import_aos, dim3((unsigned)cnt), dim3(512), 0, v->stream, v->pool, (const int*)d_slots, (const int*)v->tvals, (const float*)d_vox);
            e = hipStreamSynchronize(v->stream);
        }
        if (e != hipSuccess) rc = fail(OP_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
    }
    op::cached_free(d_keys); op::cached_free(d_slots); op::cached_free(d_vox);
    if (rc != OP_OK) return rc;
    return vol_check(v);
}

int op_volume_merge(op_volume* dst, op_volume* src) {
    OP_VOL(dst);
    if (!src) return fail(OP_ERR_INVALID, "null src");
    if (dst->device != src->device) return fail(OP_ERR_INVALID, "op_volume_merge needs both volumes on one device; use pack_sum/unpack_sum across devices");
    if (dst->res != src->res) // CubeHandler.h:147-151: warn and leave dst untouched
        return fail(OP_ERR_MISMATCH, "[Warning]::[MergeVoxelHash]::Voxel resolution is not identical.");
    if (dst == src) return fail(OP_ERR_INVALID, "cannot merge a volume into itself");
    unsigned ns = 0;
    OP_TRY(vol_block_count(src, &ns));
    unsigned nd = 0;
    OP_TRY(vol_block_count(dst, &nd));
    if (!ns) return OP_OK;
    OP_TRY(vol_reserve(dst, (unsigned long long)nd + ns)); // worst case: no block in common
    vol_mark_foreign(dst, (unsigned long long)nd + ns); // merged means: general weights in the blocks that exist after this call
    int* d_slots = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_slots, (size_t)ns * sizeof(int)));
    hipLaunchKernelGGL(k_insert_keys, dim3((ns + 255) / 256), dim3(256), 0, dst->stream, dst->view(), (const int*)src->keys, (size_t)ns, d_slots, dst->state);
    hipLaunchKernelGGL(k_merge_blocks, dim3(ns), dim3(512), 0, dst->stream, dst->pool, (const float*)src->pool, (const int*)d_slots, (const int*)dst->tvals);
    hipError_t e = hipStreamSynchronize(dst->stream);
    op::cached_free(d_slots);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "merge failed: %s", hipGetErrorString(e));
    return vol_check(dst);
}

int op_volume_keys_device(op_volume* v, int32_t* d_keys, size_t cap, size_t* n) {
    OP_VOL(v);
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    if (n) *n = nb;
    const size_t take = std::min((size_t)nb, cap);
    if (d_keys && take) { // on the volume's stream and complete on return (a device-to-device hipMemcpy does not block the host)
        OP_HIP(hipMemcpyAsync(d_keys, v->keys, take * 3 * sizeof(int), hipMemcpyDeviceToDevice, v->stream));
        OP_HIP(hipStreamSynchronize(v->stream));
    }
    return OP_OK;
}

int op_volume_pack_sum(op_volume* v, const int32_t* d_union_keys, size_t n_union, float* d_out) {
    OP_VOL(v);
    if (n_union == 0) return OP_OK;
    if (!d_union_keys || !d_out) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_check(v));
    hipLaunchKernelGGL(k_pack_sum, dim3((unsigned)n_union), dim3(512), 0, v->stream, v->view(), (const int*)d_union_keys, d_out);
    OP_HIP(hipGetLastError());
    OP_HIP(hipStreamSynchronize(v->stream));
    return OP_OK;
}

// The root's side of the merge in two steps, so that a caller can normalise slices of the union while later slices are
// still in the reduce: _begin validates, makes room (the pool grows if the union needs it), drops the volume's own
// content and enters all union keys; _chunk writes the normalised voxels of union blocks [first, first + count).
int op_volume_unpack_sum_begin(op_volume* v, const int32_t* d_union_keys, size_t n_union) {
    OP_VOL(v);
    // validate BEFORE the volume's own content is dropped: a refused unpack must leave the locally fused volume intact
    if (n_union && !d_union_keys) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_check(v));
    OP_TRY(vol_reserve(v, n_union)); // grows the root's pool if the union needs it; a refusal leaves the volume as it was
    OP_TRY(op_volume_clear(v));
    if (v->unpack_slots) { op::cached_free(v->unpack_slots); v->unpack_slots = nullptr; }
    v->unpack_n = n_union;
    if (n_union == 0) return OP_OK;
    vol_mark_foreign(v, n_union); // normalised sums of several ranks
    OP_HIP(op::cached_malloc((void**)&v->unpack_slots, n_union * sizeof(int)));
    hipLaunchKernelGGL(k_insert_keys, dim3((unsigned)((n_union + 255) / 256)), dim3(256), 0, v->stream, v->view(), (const int*)d_union_keys, n_union, v->unpack_slots, v->state);
    OP_HIP(hipGetLastError());
    OP_TRY(vol_check(v));
    v->unpack_gen = v->generation;
    return OP_OK;
}

int op_volume_unpack_sum_chunk(op_volume* v, size_t first, size_t count, const float* d_sum_chunk) {
    OP_VOL(v);
    if (count == 0) return OP_OK;
    if (!d_sum_chunk) return fail(OP_ERR_INVALID, "null argument");
    if (!v->unpack_slots || first + count > v->unpack_n) return fail(OP_ERR_INVALID, "op_volume_unpack_sum_chunk: range outside the union given to _begin");
    if (v->unpack_gen != v->generation) // growth re-hashes the table, clear drops it, fusion may do either: the slots of _begin are stale
        return fail(OP_ERR_INVALID, "op_volume_unpack_sum_chunk: the volume was cleared, grown or fused into since op_volume_unpack_sum_begin");
    hipLaunchKernelGGL(k_unpack_sum, dim3((unsigned)count), dim3(512), 0, v->stream, v->pool, (const int*)(v->unpack_slots + first), (const int*)v->tvals, d_sum_chunk);
    OP_HIP(hipGetLastError());
    OP_HIP(hipStreamSynchronize(v->stream));
    return OP_OK;
}

int op_volume_unpack_sum(op_volume* v, const int32_t* d_union_keys, size_t n_union, const float* d_sum) {
    if (v && n_union && !d_sum) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(op_volume_unpack_sum_begin(v, d_union_keys, n_union));
    return op_volume_unpack_sum_chunk(v, 0, n_union, d_sum);
}

int op_volume_transform(op_volume* src, const float T[16], const float* T_inv, int nearest, uint64_t max_blocks, op_volume** out) {
    OP_VOL(src);
    if (!T || !out) return fail(OP_ERR_INVALID, "null argument");
    *out = nullptr;
    unsigned ns = 0;
    OP_TRY(vol_block_count(src, &ns));
    // Transform copies c_para into the result (CubeHandler.h:249); TransformNearest does not
    // (CubeHandler.h:301-305), so its result keeps the default resolution 0.01 (VoxelCube.h:27)
    const float dst_res = nearest ? 0.01f : src->res;
    if (max_blocks == 0) max_blocks = std::max<uint64_t>(8ull * ns + 4096ull, 1ull << 14);
    op_volume* dst = nullptr;
    OP_TRY(op_volume_create(&src->cam, dst_res, src->trunc, src->far_d, src->near_d, src->device, max_blocks, &dst));
    vol_mark_foreign(dst, max_blocks); // resampled values (the reference's own divisions may even leave NaN / inf in them); the bound is tightened below
    Mat4 M, Mi;
    std::memcpy(M.m, T, sizeof(M.m));
    if (T_inv) std::memcpy(Mi.m, T_inv, sizeof(Mi.m));
    else op_host::mat4_inverse(T, Mi.m); // trans.inverse() (CubeHandler.h:265,320)
    int rc = OP_OK;
    if (ns) {
        unsigned nd = 0;
        for (;;) { // if the result outgrows its pool, vol_block_count grows it and the (idempotent) allocation pass runs again
            const unsigned cap_before = dst->max_blocks;
            if (nearest) hipLaunchKernelGGL(k_transform_alloc<true>, dim3(ns), dim3(512), 0, dst->stream, src->view(), dst->view(), dst->state, M, dst_res);
            else hipLaunchKernelGGL(k_transform_alloc<false>, dim3(ns), dim3(512), 0, dst->stream, src->view(), dst->view(), dst->state, M, dst_res);
            rc = vol_block_count(dst, &nd);
            if (rc != OP_OK || dst->max_blocks == cap_before) break;
        }
        if (rc == OP_OK && nd) {
            if (nearest) hipLaunchKernelGGL(k_transform_fill<true>, dim3(nd), dim3(512), 0, dst->stream, src->view(), dst->view(), Mi, src->res);
            else hipLaunchKernelGGL(k_transform_fill<false>, dim3(nd), dim3(512), 0, dst->stream, src->view(), dst->view(), Mi, src->res);
            rc = vol_check(dst);
        }
    }
    if (rc != OP_OK) { op_volume_destroy(dst); return rc; }
    { unsigned nd = 0; if (vol_block_count(dst, &nd) == OP_OK) dst->plain_from = nd; } // exactly the resampled blocks
    *out = dst;
    return OP_OK;
}

int op_volume_resolution(op_volume* v, float* voxel_res) {
    OP_VOL(v);
    if (!voxel_res) return fail(OP_ERR_INVALID, "null argument");
    *voxel_res = v->res;
    return OP_OK;
}

int op_volume_point_cloud(op_volume* v, float* xyz, float* colors, size_t cap, size_t* n) {
    OP_VOL(v);
    if (!n) return fail(OP_ERR_INVALID, "null n");
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    *n = 0;
    if (!nb) return OP_OK;
    unsigned *d_counts = nullptr, *d_offsets = nullptr;
    float *d_xyz = nullptr, *d_col = nullptr;
    int rc = OP_OK;
    hipError_t e = op::cached_malloc((void**)&d_counts, nb * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_offsets, nb * sizeof(unsigned));
    std::vector<unsigned> cnt(nb), off(nb);
    size_t total = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_point_cloud, dim3(nb), dim3(512), 0, v->stream, v->view(), v->res, v->trunc, d_counts, (const unsigned*)nullptr,
                           (float*)nullptr, (float*)nullptr);
        e = hipStreamSynchronize(v->stream);
        if (e == hipSuccess) e = hipMemcpy(cnt.data(), d_counts, nb * sizeof(unsigned), hipMemcpyDeviceToHost);
        for (unsigned b = 0; b < nb; ++b) { off[b] = (unsigned)total; total += cnt[b]; }
    }
    *n = total;
    if (e == hipSuccess && xyz && colors && total) {
        if (total > cap) rc = fail(OP_ERR_CAPACITY, "point cloud has %zu points, buffer holds %zu", total, cap);
        else {
            e = hipMemcpy(d_offsets, off.data(), nb * sizeof(unsigned), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_xyz, total * 12);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_col, total * 12);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_point_cloud, dim3(nb), dim3(512), 0, v->stream, v->view(), v->res, v->trunc, (unsigned*)nullptr,
                                   (const unsigned*)d_offsets, d_xyz, d_col);
                e = hipStreamSynchronize(v->stream);
            }
            if (e == hipSuccess) e = hipMemcpy(xyz, d_xyz, total * 12, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(colors, d_col, total * 12, hipMemcpyDeviceToHost);
        }
    }
    void* ptrs[] = {d_counts, d_offsets, d_xyz, d_col};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "point cloud failed: %s", hipGetErrorString(e));
    return rc;
}


int op_volume_extract_mesh(op_volume* v, const int32_t* tri_table, const int32_t* edge_pairs, const int32_t* only_block, float* points,
                           float* colors, size_t cap_vertices, size_t* n_vertices) {
    OP_VOL(v);
    if (!tri_table || !edge_pairs || !n_vertices) return fail(OP_ERR_INVALID, "null argument");
    for (int c = 0; c < 256; ++c)
        for (int i = 0; i < 16; ++i) {
            const int e = tri_table[16 * c + i];
            if (e < -1 || e > 11) return fail(OP_ERR_INVALID, "tri_table[%d][%d] = %d is not an edge id or -1", c, i, e);
            if (i == 15 && e != -1) return fail(OP_ERR_INVALID, "tri_table row %d is not -1 terminated", c);
        }
    for (int i = 0; i < 24; ++i)
        if (edge_pairs[i] < 0 || edge_pairs[i] > 7) return fail(OP_ERR_INVALID, "edge_pairs[%d] = %d is not a corner id", i, edge_pairs[i]);
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    *n_vertices = 0;
    if (!nb) return OP_OK;
    // block list: every block in pool order, or the one requested (GenerateMeshByCube)
    std::vector<unsigned> list;
    if (only_block) {
        std::vector<int> keys((size_t)nb * 3);
        OP_HIP(hipMemcpy(keys.data(), v->keys, keys.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (unsigned b = 0; b < nb; ++b)
            if (keys[3 * b] == only_block[0] && keys[3 * b + 1] == only_block[1] && keys[3 * b + 2] == only_block[2]) list.push_back(b);
        if (list.empty()) return OP_OK;
    } else {
        list.resize(nb);
        for (unsigned b = 0; b < nb; ++b) list[b] = b;
    }
    const unsigned nl = (unsigned)list.size();
    unsigned *d_list = nullptr, *d_counts = nullptr, *d_offsets = nullptr;
    int *d_tri = nullptr, *d_edge = nullptr;
    float *d_pts = nullptr, *d_col = nullptr;
    int rc = OP_OK;
    hipError_t e = op::cached_malloc((void**)&d_list, nl * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_counts, nl * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_offsets, nl * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_tri, 256 * 16 * sizeof(int));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_edge, 24 * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_list, list.data(), nl * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_tri, tri_table, 256 * 16 * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_edge, edge_pairs, 24 * sizeof(int), hipMemcpyHostToDevice);
    std::vector<unsigned> cnt(nl), off(nl);
    size_t total_tri = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_mesh, dim3(nl), dim3(512), 0, v->stream, v->view(), v->res, (const int*)d_tri, (const int*)d_edge, (const unsigned*)d_list,
                           d_counts, (const unsigned*)nullptr, (float*)nullptr, (float*)nullptr);
        e = hipStreamSynchronize(v->stream);
        if (e == hipSuccess) e = hipMemcpy(cnt.data(), d_counts, nl * sizeof(unsigned), hipMemcpyDeviceToHost);
        for (unsigned b = 0; b < nl; ++b) { off[b] = (unsigned)total_tri; total_tri += cnt[b]; }
    }
    const size_t total = total_tri * 3;
    *n_vertices = total;
    if (e == hipSuccess && points && colors && total) {
        if (total > cap_vertices) rc = fail(OP_ERR_CAPACITY, "mesh has %zu vertices, buffer holds %zu", total, cap_vertices);
        else if (total_tri > 0xffffffffull / 3) rc = fail(OP_ERR_CAPACITY, "mesh too large");
        else {
            e = hipMemcpy(d_offsets, off.data(), nl * sizeof(unsigned), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_pts, total * 12);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_col, total * 12);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_mesh, dim3(nl), dim3(512), 0, v->stream, v->view(), v->res, (const int*)d_tri, (const int*)d_edge,
                                   (const unsigned*)d_list, (unsigned*)nullptr, (const unsigned*)d_offsets, d_pts, d_col);
                e = hipStreamSynchronize(v->stream);
            }
            if (e == hipSuccess) e = hipMemcpy(points, d_pts, total * 12, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(colors, d_col, total * 12, hipMemcpyDeviceToHost);
        }
    }
    void* ptrs[] = {d_list, d_counts, d_offsets, d_tri, d_edge, d_pts, d_col};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "mesh extraction failed: %s", hipGetErrorString(e));
    return rc;
}

int op_volume_write_file(op_volume* v, const char* path) {
    OP_VOL(v);
    if (!path) return fail(OP_ERR_INVALID, "null path");
    size_t n = 0;
    OP_TRY(op_volume_block_count(v, &n));
    std::vector<int32_t> keys(3 * n);
    std::unique_ptr<float[]> vox(new float[std::max<size_t>(n, 1) * (size_t)kBlockFloats]);
    if (n) OP_TRY(op_volume_download(v, keys.data(), vox.get(), n, &n));
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(OP_ERR_INVALID, "cannot open %s for writing", path);
    // CubeHandler::WriteToFile (CubeHandler.h:113-128): the block count's raw bits sit in a float slot; then per block
    // VoxelCube::WriteToBuffer (VoxelCube.h:128-148): id, {i, sdf, w, c0, c1, c2} of every voxel with |sdf| < 1 and
    // w != 0, terminator -2.  Two passes: per-block record counts -> offsets, then the blocks are formatted in parallel.
    const float* vx = vox.get();
    std::vector<size_t> off(n + 1, 0);
    for_block_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b) {
            size_t c = 0;
            for (int i = 0; i < kVox; ++i) {
                const float* t = &vx[(b * kVox + i) * 5];
                c += (std::fabs(t[0]) < 1 && t[1] != 0) ? 1 : 0;
            }
            off[b + 1] = 4 + 6 * c;
        }
    });
    off[0] = 1;
    for (size_t b = 0; b < n; ++b) off[b + 1] += off[b];
    const size_t total = off[n];
    std::unique_ptr<float[]> buffer(new float[total]);
    const unsigned int size = (unsigned int)n;
    std::memcpy(&buffer[0], &size, 4);
    float* out = buffer.get();
    for_block_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b) {
            float* o = out + off[b];
            for (int c = 0; c < 3; ++c) *o++ = (float)keys[3 * b + c];
            for (int i = 0; i < kVox; ++i) {
                const float* t = &vx[(b * kVox + i) * 5];
                if (std::fabs(t[0]) < 1 && t[1] != 0) {
                    *o++ = (float)i;
                    for (int k = 0; k < 5; ++k) *o++ = t[k];
                }
            }
            *o++ = -2.0f;
        }
    });
    const bool ok = std::fwrite(buffer.get(), sizeof(float), total, f) == total;
    std::fclose(f);
    return ok ? OP_OK : fail(OP_ERR_INVALID, "short write to %s", path);
}

int op_volume_read_file(op_volume* v, const char* path, int legacy_float_format) {
    OP_VOL(v);
    if (!path) return fail(OP_ERR_INVALID, "null path");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(OP_ERR_INVALID, "cannot open %s", path);
    std::fseek(f, 0, SEEK_END);
    const long len = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const size_t nfl = (size_t)len / sizeof(float);
    std::unique_ptr<float[]> buffer_mem(new float[nfl + 1]);
    float* buffer = buffer_mem.get();
    buffer[nfl] = 0.0f;
    const bool ok = std::fread(buffer, sizeof(float), nfl, f) == nfl;
    std::fclose(f);
    if (!ok || nfl < 2) return fail(OP_ERR_INVALID, "cannot read %s", path);
    unsigned int count = 0;
    size_t ptr = 0;
    if (legacy_float_format) { count = (unsigned int)buffer[1]; ptr = 2; } // CubeHandler.h:91-94
    else { std::memcpy(&count, &buffer[0], 4); ptr = 1; }                   // CubeHandler.h:51-53
    // pass 1 (sequential, cheap): where every block's record starts -- the stream is only delimited by its terminators
    std::vector<size_t> start;
    start.reserve((size_t)count + 1);
    for (unsigned int c = 0; c < count && ptr + 3 <= nfl; ++c) {
        start.push_back(ptr);
        ptr += 3;
        if (!legacy_float_format) {                  // VoxelCube::ReadFromBuffer (VoxelCube.h:153-166): {i, 5 floats}* -2
            while (ptr < nfl && buffer[ptr] != -2.0f) ptr += 6;
            ptr++;
        } else {                                     // VoxelCube::ReadFromBufferFloat (VoxelCube.h:168-193)
            ptr++;
            while (ptr < nfl && buffer[ptr] != -2.0f) ptr += 3;
            ptr++;
            const size_t cnt = ptr < nfl ? (size_t)buffer[ptr++] : 0;
            ptr += 5 * cnt;
        }
        if (ptr > nfl + 1) return fail(OP_ERR_INVALID, "corrupt .map file %s", path);
    }
    const size_t nb = start.size();
    start.push_back(ptr < nfl ? ptr : nfl);
    // pass 2 (parallel over blocks): cube_map[cube_id] = VoxelCube(cube_id) (default voxels), then the stored voxels
    std::vector<int32_t> keys(3 * nb);
    std::unique_ptr<float[]> vox_mem(new float[std::max<size_t>(nb, 1) * (size_t)kBlockFloats]);
    float* vox = vox_mem.get();
    std::atomic<int> bad{0};
    for_block_ranges(nb, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b) {
            size_t q = start[b];
            const size_t end = start[b + 1];
            for (int c = 0; c < 3; ++c) keys[3 * b + c] = (int32_t)buffer[q + c];
            q += 3;
            float* blk = vox + b * (size_t)kBlockFloats;
            for (int i = 0; i < kVox; ++i) { blk[5 * i] = 999.0f; blk[5 * i + 1] = 0.0f; blk[5 * i + 2] = blk[5 * i + 3] = blk[5 * i + 4] = -1.0f; }
            if (!legacy_float_format) {
                while (q < end && buffer[q] != -2.0f) {
                    const int i = (int)buffer[q++];
                    if (i < 0 || i >= kVox || q + 5 > nfl) { bad = 1; break; }
                    for (int k = 0; k < 5; ++k) blk[5 * i + k] = buffer[q++];
                }
            } else {
                q++;
                while (q < end && buffer[q] != -2.0f) {
                    const int i = (int)buffer[q++];
                    if (i < 0 || i >= kVox || q + 2 > nfl) { bad = 1; break; }
                    blk[5 * i] = buffer[q++]; blk[5 * i + 1] = buffer[q++];
                }
                q++;
                const size_t cnt = q < nfl ? (size_t)buffer[q++] : 0;
                for (size_t k = 0; k < cnt && q + 5 <= nfl; ++k) {
                    const int i = (int)buffer[q++];
                    if (i < 0 || i >= kVox) { bad = 1; break; }
                    float* t = &blk[5 * i];
                    t[2] = (float)(buffer[q++] / 255.0); t[3] = (float)(buffer[q++] / 255.0); t[4] = (float)(buffer[q++] / 255.0);
                    const float cw = buffer[q++];
                    t[2] = t[2] / cw; t[3] = t[3] / cw; t[4] = t[4] / cw;
                }
            }
        }
    });
    if (bad) return fail(OP_ERR_INVALID, "corrupt .map file %s", path);
    OP_TRY(op_volume_clear(v)); // cube_map.clear() (CubeHandler.h:42)
    return op_volume_upload(v, keys.data(), vox, nb);
}

int op_volume_raycast(op_volume* v, const op_camera* cam, const float pose[16], float* depth_out, float* normals_out, float* colors_out, int mem) {
    OP_VOL(v);
    if (!pose || !depth_out) return fail(OP_ERR_INVALID, "null argument");
    const op_camera c = cam ? *cam : v->cam;
    OP_TRY(check_cam(&c));
    OP_TRY(vol_check(v));
    const size_t npx = (size_t)c.width * c.height;
    float *d_depth = depth_out, *d_nrm = normals_out, *d_col = colors_out;
    if (mem == OP_MEM_HOST) {
        d_depth = d_nrm = d_col = nullptr;
        OP_HIP(op::cached_malloc((void**)&d_depth, npx * 4));
        if (normals_out) OP_HIP(op::cached_malloc((void**)&d_nrm, npx * 12));
        if (colors_out) OP_HIP(op::cached_malloc((void**)&d_col, npx * 12));
    }
    Mat4 P;
    std::memcpy(P.m, pose, sizeof(P.m));
    hipLaunchKernelGGL(k_raycast, dim3((c.width + 15) / 16, (c.height + 15) / 16), dim3(256), 0, v->stream, v->view(), c, P, v->res, v->near_d,
                       v->far_d, d_depth, d_nrm, d_col);
    hipError_t e = hipStreamSynchronize(v->stream);
    if (mem == OP_MEM_HOST) {
        if (e == hipSuccess) e = hipMemcpy(depth_out, d_depth, npx * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && normals_out) e = hipMemcpy(normals_out, d_nrm, npx * 12, hipMemcpyDeviceToHost);
        if (e == hipSuccess && colors_out) e = hipMemcpy(colors_out, d_col, npx * 12, hipMemcpyDeviceToHost);
        op::cached_free(d_depth);
        if (d_nrm) op::cached_free(d_nrm);
        if (d_col) op::cached_free(d_col);
    }
    if (e != hipSuccess) return fail(OP_ERR_HIP, "raycast failed: %s", hipGetErrorString(e));
    return OP_OK;
}

} // extern "C"
