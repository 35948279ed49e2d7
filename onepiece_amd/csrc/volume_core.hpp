// volume_core.hpp -- what the translation units of the voxel-block-hashed TSDF volume share: constants, the device views of a volume (State, VolView),
// the hash-table device helpers, the host object op_volume and the host-side functions / kernel launchers that cross translation units.
//   volume.hip      the host object: create / grow / replay, the staging ring, the integrate entry points (C-ABI), pool and table maintenance kernels
//   select.hip      KA k_prepare_frames (ComputeBounding + frame packing), KB k_select / k_select_vote / k_select_merge (PrepareCubes)
//   integrate.hip   KC k_integrate (Integrator::IntegrateImage)
//   volume_ops.hip  GetCubeMap / SetCubeMap / Merge / sum-form pack + unpack / Transform / GetPointCloud / ExtractTriangleMesh / .map files
//   raycast.hip     the raycaster
// Data layout in HBM: see volume.hip (and DESIGN.md section 2).
#pragma once
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

namespace opv {

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

__device__ __forceinline__ unsigned ord_enc(float f) { // order-preserving float -> unsigned
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

__host__ __device__ inline int ka_grid(int w, int h) { return ((w + kKaW - 1) / kKaW) * ((h + kKaH - 1) / kKaH); }
__host__ __device__ inline int tiles_w(int w) { return (w + kTile - 1) / kTile; }
__host__ __device__ inline int tiles_h(int h) { return (h + kTile - 1) / kTile; }

constexpr unsigned kVoteCap = 1u << 18; // super-blocks per frame in sbits (2 MB per frame; 16.8 M blocks = 1000 m^3 at 5 mm voxels)

struct Mat4 { float m[16]; };
struct Vox5 { float s, w, c0, c1, c2; };

inline unsigned next_pow2(unsigned long long v) {
    unsigned long long p = 1;
    while (p < v) p <<= 1;
    return (unsigned)p;
}

constexpr unsigned kHardMaxBlocks = 1u << 24; // 172 GB of pool: what one 288 GB MI355X can hold next to its inputs

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

} // namespace opv
using namespace opv; // (an internal header: every includer is one of the volume's translation units)

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
    void* rc_list = nullptr;     // raycast.hip: the visible-block list of the view being cast (one entry per pool block at most), its capacity in blocks
    unsigned rc_cap = 0;
    unsigned* rc_count = nullptr; // ... and its length
    unsigned* rc_order = nullptr; // the order k_rc_march takes the list in: every XCD's contiguous eighth sorted front to back (k_rc_order)
    unsigned* rc_sum = nullptr;   // per pool slot: what the march learnt about the block's own voxels ((stamp << 2) | has sdf > 0 << 1 | has sdf <= 0), valid while stamp == content_gen (written by k_rc_march from its tile and by k_integrate from the block it has just updated)
    uint64_t rc_sum_epoch = 0;    // content_gen >> 30 the summaries were last wiped for
    int rc_prune = 1;             // OP_VOLUME_OPT_RAYCAST_PRUNE
    uint64_t content_gen = 1;     // bumped by everything that can change a voxel or a pool slot's meaning without restating the summaries (clear, growth, every foreign writer, sum-form fusion; exact fusion only while no summaries exist)
    unsigned char* rc_hit = nullptr; // one byte per pool slot: the block holds hit points of the view being cast (zero between calls)
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

#define OP_VOL(v)                                              \
    if (!(v)) return fail(OP_ERR_INVALID, "null volume");      \
    OP_HIP(hipSetDevice((v)->device))

namespace opv {
// ---- host side, defined in volume.hip
int vol_flush(op_volume* v);                 // launches the frames queued by op_volume_integrate
int vol_check(op_volume* v);                 // flush, synchronise, grow + replay if needed, report frames that cannot be fused
int vol_reset(op_volume* v);
int vol_reserve(op_volume* v, unsigned long long need);
int vol_block_count(op_volume* v, unsigned* n);
void vol_mark_foreign(op_volume* v, unsigned long long bound);
int check_cam(const op_camera* cam);
// ---- kernel launchers (a kernel is launched from the translation unit that defines it)
// select.hip: KA over the batch's frames (kKaFrames per launch); KB in the form the batch takes (cube_keys: k_mark_cubes instead); k_finish_select
void launch_prepare_frames(op_volume* v, const BatchFwd& F, int nf, const CamParams& C, const BatchPtrs& Q, unsigned seq);
int launch_select(op_volume* v, const BatchInv& I, const CamParams& C, int nf, bool record, const int* cube_keys, unsigned n_cubes); // (may allocate the voting words: can fail)
void launch_finish_select(op_volume* v);
void kb_trace_dump(op_volume* v);            // -DKB_TRACE builds only
// integrate.hip: KC
void launch_integrate(op_volume* v, const BatchInv& I, const CamParams& C, int nf);
bool vol_fusion_keeps_summaries(const op_volume* v); // integrate.hip: this batch's k_integrate restates the raycaster's summaries of the blocks it changes
void kc_trace_dump(op_volume* v);            // -DKC_TRACE builds only
} // namespace opv
