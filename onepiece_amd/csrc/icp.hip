// icp.hip -- point-to-plane / point-to-point ICP for gfx950 (MI355X) and the C-ABI entry points
// op_icp_* / op_points_from_depth declared in include/onepiece_hip.h.
//
// What it replaces (file:line under /root/reference/src):
//   registration::PointToPlane                       Registration/ICP.cpp:146-224
//   registration::PointToPoint                       Registration/ICP.cpp:31-107
//   geometry::TransformPoints + KDTree 1-NN          Registration/ICP.cpp:182-192, Geometry/KDTree.h:167-196
//   CountInliers                                     Registration/ICP.cpp:9-30
//   EstimateRigidTransformationPointToPlane (sums)   Registration/ICP.cpp:121-136
//   geometry::EstimateRigidTransformation (sums)     Geometry/Geometry.cpp:122-133
//   PointCloud::LoadFromDepth                        Geometry/PointCloud.cpp:72-100
//
// Design (DESIGN.md "ICP"): the reference's exact 1-NN is only ever consumed through
// CountInliers, which discards correspondences farther than `threshold`; a uniform grid over the
// target with cell >= threshold and a 27-cell scan therefore yields the identical inlier set.  The
// target is counting-sorted by cell into float4 records (xyz + original index) so candidate reads
// are contiguous 16-byte loads.  One kernel per iteration fuses transform + NN + inlier test + the
// normal-equation contributions; the 27 (plane) / 15 (point) sums are reduced in fp64 with
// wave64 shuffles, then LDS across the 4 waves of a workgroup, then across workgroups by the last ones to arrive.
// The 6x6 solve / SE3 exp / Kabsch stay on the host (host_math.hpp) exactly as north_star asks;
// this accumulation is 2*27*N flops -- not a dense contraction, so no MFMA.
#include <cfloat>
#include <climits>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "host_math.hpp"
#include "nn_tree.hpp"
#include "seq_sums.hpp"

namespace {

using op::fail;

constexpr int kNSums = 32;      // doubles per partial: sums[0..26], [27] = sum_sq_err, [28] = inlier count
#ifndef ICP_THREADS
#define ICP_THREADS 256
#endif
constexpr int kIterThreads = ICP_THREADS;
#ifndef ICP_SCAN_C
#define ICP_SCAN_C 8
#endif
#ifndef ICP_SCAN_R
#define ICP_SCAN_R 8
#endif
constexpr int kScanC = ICP_SCAN_C; // candidates fetched per trip of the neighbour scan: centre row,
constexpr int kScan = ICP_SCAN_R;  // the other rows
constexpr unsigned long long kMaxCells = 1ull << 26;
constexpr size_t kMaxPoints = (size_t)1 << 28; // 16-byte records and 12-byte points are addressed with 32-bit byte offsets

struct Grid {
    float ox, oy, oz, inv_cell; // origin and 1/cell
    int gx, gy, gz;
};

struct Mat4 { float m[16]; };

__device__ __forceinline__ float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }

__device__ __forceinline__ unsigned enc_f(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float dec_f(unsigned e) {
    const unsigned b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int cell_coord(float p, float o, float inv, int g) {
    int c = (int)floorf((p - o) * inv);
    return c < 0 ? 0 : (c >= g ? g - 1 : c);
}

// ---- target grid build ----------------------------------------------------------------------
__global__ void k_bbox(const float* __restrict__ xyz, size_t m, unsigned* __restrict__ box /*max3,min3*/) {
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) {
            const float v = xyz[3 * i + c];
            if (fabsf(v) <= FLT_MAX) { mx[c] = fmaxf(mx[c], v); mn[c] = fminf(mn[c], v); } // NaN and +-inf never enter the grid
        }
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
        }
    }
    // one atomic pair per workgroup and component (six hot addresses: per-wave atomics serialise badly)
    __shared__ float s_mx[4][3], s_mn[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) { s_mx[wave][c] = mx[c]; s_mn[wave][c] = mn[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        float a = s_mx[0][c], b = s_mn[0][c];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { a = fmaxf(a, s_mx[w][c]); b = fminf(b, s_mn[w][c]); }
        atomicMax(&box[c], enc_f(a));
        atomicMin(&box[3 + c], enc_f(b));
    }
}

__global__ void k_cell_count(const float* __restrict__ xyz, size_t m, Grid g, unsigned* __restrict__ count) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX)) return;
    const int cx = cell_coord(x, g.ox, g.inv_cell, g.gx), cy = cell_coord(y, g.oy, g.inv_cell, g.gy),
              cz = cell_coord(z, g.oz, g.inv_cell, g.gz);
    atomicAdd(&count[((size_t)cz * g.gy + cy) * g.gx + cx], 1u);
}

// Exclusive scan of the cell counts in cell order, so that x-adjacent cells own adjacent ranges of
// the sorted target (the 27-cell scan then touches 9 contiguous runs).  Three small kernels:
// per-workgroup totals -> scan of totals (one workgroup) -> per-element offsets.
constexpr int kScanWg = 1024; // elements per workgroup (256 threads x 4)
__global__ __launch_bounds__(256) void k_scan_totals(const unsigned* __restrict__ count, size_t n, unsigned* __restrict__ totals) {
    __shared__ unsigned s[4];
    const size_t base = (size_t)blockIdx.x * kScanWg + threadIdx.x * 4;
    unsigned t = 0;
    for (int k = 0; k < 4; ++k) t += base + k < n ? count[base + k] : 0u;
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) totals[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(1024) void k_scan_of_totals(unsigned* __restrict__ totals, size_t n) {
    // single workgroup, sequential over tiles of 1024
    __shared__ unsigned s[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < n; base += 1024) {
        const size_t i = base + threadIdx.x;
        const unsigned v = i < n ? totals[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
            const unsigned add = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n) totals[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_apply(const unsigned* __restrict__ count, size_t n, const unsigned* __restrict__ totals,
                                                    unsigned* __restrict__ start) {
    __shared__ unsigned s[4];
    const size_t base = (size_t)blockIdx.x * kScanWg + threadIdx.x * 4;
    unsigned c[4], t = 0;
    for (int k = 0; k < 4; ++k) { c[k] = base + k < n ? count[base + k] : 0u; t += c[k]; }
    // exclusive scan of t across the workgroup
    unsigned incl = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s[wave] = incl;
    __syncthreads();
    unsigned off = totals[blockIdx.x] + incl - t;
    for (int w = 0; w < wave; ++w) off += s[w];
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) start[base + k] = off;
        off += c[k];
    }
}

__global__ void k_cell_scatter(const float* __restrict__ xyz, size_t m, Grid g,
                               const unsigned* __restrict__ start, unsigned* __restrict__ left, float4* __restrict__ sorted) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX)) return;
    const int cx = cell_coord(x, g.ox, g.inv_cell, g.gx), cy = cell_coord(y, g.oy, g.inv_cell, g.gy),
              cz = cell_coord(z, g.oz, g.inv_cell, g.gz);
    const size_t c = ((size_t)cz * g.gy + cy) * g.gx + cx;
    // `left` is the cell's count from k_cell_count, counted down: the order inside a cell is arbitrary either way (the search
    // orders candidates by (distance, original index)), and no second per-cell array has to be allocated and zeroed
    const unsigned pos = start[c] + (atomicSub(&left[c], 1u) - 1u);
    sorted[pos] = make_float4(x, y, z, __int_as_float((int)i));
}

// ---- per-iteration kernel ----------------------------------------------------------------------
// MODE 1 (plane): sums[0..20] = upper triangle of JTJ (row-major), [21..26] = JTr.
// MODE 0 (point): sums[0..2] = sum s', [3..5] = sum t, [6..14] = sum s' t^T.
// MODE 2 (final): like MODE 0 but over the ORIGINAL source points and the stored nn[] (no search).
// MODE 3 / 4: MODE 0 / 1 with the stored nn[] instead of the search -- the second pass of an iteration whose exactly equidistant
//   candidates were re-decided on the host (OP_ICP_TIES_REFERENCE, below).
// DETECT: the search also reports the queries whose nearest distance is shared by more than one target (TieRec records in host-mapped memory: transformed query, source point,
//   its index; sums[29] = how many): an extra compare and select per candidate.
//
// The kernel also finishes the reduction itself (no second-pass kernels on the per-iteration critical path): every
// workgroup writes its row of partial sums, the LAST workgroup of each group of `per_group` rows to arrive folds that
// group into one stage row, and the last group to finish folds the stage rows and writes the totals.  In the
// host-solve loop (host_out != nullptr) the chain is cut short: each group's row is written to host-mapped pinned
// memory with a sequence number, and the host, which needs the totals anyway, adds the (at most 32) rows in group
// order -- three device-memory round trips less on the critical path of every iteration.  Who does the folding depends on timing, what is added in which order does not, so the sums are
// reproducible bit for bit.  sync[0..kGroups-1] count the arrivals per group, sync[kGroups] the finished groups; the
// workgroup that completes a count resets it for the next launch.
// What the search reports about a query whose nearest distance more than one target has (OP_ICP_TIES_REFERENCE).  The records live in
// host-mapped pinned memory: a pass has a handful at most on depth-derived clouds, and the host needs them right after the sums.
struct TieRec {
    float tp[3];        // the transformed query (what the reference hands to its kd-tree)
    int src;            // source index
    float s[3];         // the source point itself (CountInliers transforms it again, in its own operand order)
    int best;           // the target the search picked: the smallest index among the equidistant ones
    unsigned stamp;     // the launch's tie stamp, stored after everything else has been acknowledged
    unsigned pad[3];
};
// What the FINAL pass (MODE 2) needs to tell which stored correspondences it may not trust.  The 27-cell search returns the true nearest target of
// every query that has one within `reach` (< one cell edge) -- all CountInliers ever looks at while search and count share one pose.  The final
// CountInliers (ICP.cpp:206) does not: it measures the LAST search's pairs with the pose the last solve produced, so a point whose nearest target lay
// beyond `reach` under the old pose (nn = the nearest the 27 cells happened to hold, or none) can come within the threshold under the new one once the
// last step moved it by more than reach - threshold -- never in a converged registration (the margin is 0.05 % of the threshold and the last step is
// orders below it), routinely when the loop is stopped early.  The final pass therefore recomputes the old query of every point, and reports (sums[30],
// list) those whose stored partner lies beyond `reach` there AND whose displacement could bridge the gap; the host re-decides exactly these in the
// tree the reference would search (nn_tree.hpp), patches nn[] and repeats the pass.  Handed to the kernel through the tie_rec argument (unused in MODE 2).
struct FinalAux {
    float T_old[16];    // the pose of the last search
    float reach, reach2, thr;
    unsigned count;     // entries of list (grows by atomicAdd)
    unsigned* list;     // source indices to re-decide, n entries
};
constexpr int kGroups = 32;
constexpr unsigned long long kNoKey = 0x7f7fffff00000000ull; // (FLT_MAX, index 0): no candidate compares below it

template <class V>
__device__ __forceinline__ V ld_off(const void* base, unsigned byte_off) { // base + zero-extended 32-bit offset (SGPR base + VGPR offset form)
    return *reinterpret_cast<const V*>(static_cast<const char*>(base) + byte_off);
}
struct __attribute__((packed, aligned(4))) U4 { unsigned a, b, c, d; };
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };

// Rows exchanged between workgroups of ONE launch live behind different L2s (one per XCD).  A release fence at agent
// scope would write back the XCD's whole L2 (measured: 1200 of them cost 90 us per launch); instead the few values
// that cross are stored and loaded with agent-scope accesses (write-through / L2-bypassing), the writer waits for its
// stores to be acknowledged (s_waitcnt 0) before the barrier that precedes the arrival count, and the arrival count is
// a relaxed agent-scope atomic.
// v = set-lanes ? if_set : v, with the lane mask in an SGPR pair (VOP3 encoding).  The compiler's own select after a
// 64-bit compare is two VOP2 v_cndmask_b32 reading VCC back to back, which issue at ~11 cycles each on gfx950
// (tools/valu_ubench.hip) -- for the neighbour scan that was more than the distance computation itself.
__device__ __forceinline__ unsigned select_lanes(unsigned long long lane_mask, unsigned if_clear, unsigned if_set) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(lane_mask));
    return r;
}

__device__ __forceinline__ double ld_coherent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coherent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_stores_then_barrier() {
    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0): every store of this wave has been acknowledged
    __syncthreads();
}

#ifdef ICP_TRACE // development aid (make EXTRA=-DICP_TRACE): per-wave timestamps of the phases of the last launch, dumped by op_icp_destroy
__device__ unsigned long long g_icp_trace[8 * 8192];
#define ICP_STAMP(K) do { __builtin_amdgcn_s_waitcnt(0); if ((threadIdx.x & 63) == 0 && MODE == 1) g_icp_trace[(blockIdx.x * (kIterThreads / 64) + (threadIdx.x >> 6)) * 8 + (K)] = wall_clock64(); } while (0)
#else
#define ICP_STAMP(K) do { } while (0)
#endif

template <int MODE, bool DETECT = false>
__global__ __launch_bounds__(kIterThreads) void k_icp_iter(const float* __restrict__ T, Mat4 T_arg, const float* __restrict__ src, unsigned n, Grid g,
                                                           const unsigned* __restrict__ cell_start, const float4* __restrict__ tgt, unsigned dummy,
                                                           const float* __restrict__ tgt_orig, const float* __restrict__ nrm_orig, double thr2,
                                                           int* __restrict__ nn, int* __restrict__ inl, double* __restrict__ partials,
                                                           double* __restrict__ stage, unsigned* __restrict__ sync, unsigned per_group,
                                                           double* __restrict__ out, double* __restrict__ host_out, double seq,
                                                           unsigned* __restrict__ tie_count, unsigned tie_base, TieRec* __restrict__ tie_rec, unsigned tie_stamp) {
    constexpr bool kPlane = MODE == 1 || MODE == 4;
    bool tied = false; // DETECT: more than one target at this point's nearest distance
    bool unsure = false; // MODE 2 with a FinalAux: the stored partner of this point may not be its nearest target (see FinalAux)
    __shared__ double s_red[kIterThreads / 64][kNSums];
    __shared__ double s_fin[kIterThreads / 32][kNSums];
    __shared__ uint2 s_runs[8][kIterThreads]; // per lane: the [begin, end) runs of the rows it still has to scan
    __shared__ int s_last;
    // what the point contributes to the sums; the 29 fp64 accumulators themselves are only formed after the search
    bool inlier = false;
    double e = 0.0;
    float a0 = 0, a1 = 0, a2 = 0, t0 = 0, t1 = 0, t2 = 0, n0 = 0, n1 = 0, n2 = 0;

    // exactly one source point per thread (grid = ceil(n / 256)): the 29 fp64 accumulators are then
    // not live across the neighbour search, which keeps the kernel at ~80 VGPRs instead of 150
    // XCD-aware: the source is in image order, so a contiguous slab of it meets a contiguous part of the cell-sorted
    // target; with the plain order every XCD's L2 would see all of target + normals + cell tables (> 4 MiB)
    const unsigned wg = op::xcd_slab_index(blockIdx.x, gridDim.x);
    const unsigned i = wg * (unsigned)kIterThreads + threadIdx.x;
    ICP_STAMP(0);
    if (i < n) {
        const F3 sp = ld_off<F3>(src, 12u * i);
        const float s0 = sp.x, s1 = sp.y, s2 = sp.z;
        // start_T: device memory when the update step runs on the device (T != nullptr), a by-value kernel
        // argument when the host does the solve (saves the per-iteration host-to-device copy)
        float M[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) M[k] = T ? T[k] : T_arg.m[k];
        float tp0 = 0, tp1 = 0, tp2 = 0;
        int best = -1;
        if (MODE != 2) {
            // TransformPoints (Geometry.cpp:19-27): 4x4 * (s,1), then divide by w
            const float q0 = ((M[0] * s0 + M[1] * s1) + M[2] * s2) + M[3] * 1.0f;
            const float q1 = ((M[4] * s0 + M[5] * s1) + M[6] * s2) + M[7] * 1.0f;
            const float q2 = ((M[8] * s0 + M[9] * s1) + M[10] * s2) + M[11] * 1.0f;
            const float q3 = ((M[12] * s0 + M[13] * s1) + M[14] * s2) + M[15] * 1.0f;
            tp0 = q0 / q3; tp1 = q1 / q3; tp2 = q2 / q3;
          if (MODE >= 3) {
            best = nn[i]; // decided by an earlier pass (and, for tied queries, by the host)
          } else {
            // exact 1-NN restricted to the 27 cells around the query (see header comment).  The running best is ONE
            // 64-bit key (bits of the squared distance, original index): the distance is never negative, so its bit
            // pattern orders like the value, and "nearer, ties to the smaller original index" is an unsigned minimum --
            // the visiting order does not matter and a candidate costs one 64-bit compare and two selects.
            unsigned long long best_key = kNoKey;
            unsigned tie_d = 0xffffffffu; // DETECT: bits of the distance at which a second candidate last equalled the running best
#ifdef ICP_REPEAT // measurement aid (make EXTRA=-DICP_REPEAT=2): the search runs ICP_REPEAT times in ONE launch, the later passes with this launch's L2 content
            for (int rep_ = 0; rep_ < ICP_REPEAT; ++rep_) {
            if (rep_ > 0) { tp0 += best_key == 0x0123456789abcdefull ? 1.0f : 0.0f; best_key = kNoKey; ICP_STAMP(7); } // (depends on the pass before; never true)
#endif
            if (fabsf(tp0) <= FLT_MAX && fabsf(tp1) <= FLT_MAX && fabsf(tp2) <= FLT_MAX) { // NaN / inf queries match nothing
                // cell of the query, clamped to two cells outside the grid (beyond that nothing can be within a cell of it;
                // keeps the int conversion and the +-1 neighbourhood arithmetic in range for far-away points)
                const int cx = (int)fminf(fmaxf(floorf((tp0 - g.ox) * g.inv_cell), -2.0f), (float)g.gx + 1.0f),
                          cy = (int)fminf(fmaxf(floorf((tp1 - g.oy) * g.inv_cell), -2.0f), (float)g.gy + 1.0f),
                          cz = (int)fminf(fmaxf(floorf((tp2 - g.oz) * g.inv_cell), -2.0f), (float)g.gz + 1.0f);
                const int x_lo = max(cx - 1, 0), x_hi = min(cx + 1, g.gx - 1);
                // distance from the query to the near face of the neighbouring rows of cells: every point of
                // row (cy+dy, cz+dz) is at least sqrt(gy[dy]^2 + gz[dz]^2) away, so once a candidate nearer
                // than that bound (with 1 % slack for the rounding of the cell assignment) is known the row
                // cannot contain the nearest neighbour.  The centre row is scanned first.
                const float cell = 1.0f / g.inv_cell;
                const float fy = (tp1 - g.oy) - (float)cy * cell, fz = (tp2 - g.oz) - (float)cz * cell;
                const float gy[3] = {fmaxf(fy, 0.0f), 0.0f, fmaxf(cell - fy, 0.0f)};
                const float gz[3] = {fmaxf(fz, 0.0f), 0.0f, fmaxf(cell - fz, 0.0f)};
                if (x_lo <= x_hi) {
                    // the [begin, end) runs of all nine rows are fetched first (independent loads, one round trip) instead of
                    // one dependent round trip per visited row; the centre row's stays in registers, the other eight are
                    // parked in the lane's LDS column (slot = q, skipping the centre) until the centre row has been scanned.
                    // cell_start is the exclusive scan over ALL cells (+4 entries of padding), so cells x_lo..x_hi own
                    // [cell_start[x_lo], cell_start[x_hi + 1]) and one 16-byte load returns both ends.
                    const int w = x_hi - x_lo; // 0..2
                    unsigned cb = 0u, ce = 0u;
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int dy = q % 3 - 1, dz = q / 3 - 1;
                        const int z = cz + dz, y = cy + dy;
                        unsigned rb = 0u, re = 0u;
                        if (!(z < 0 || z >= g.gz || y < 0 || y >= g.gy)) {
                            const unsigned first = ((unsigned)z * (unsigned)g.gy + (unsigned)y) * (unsigned)g.gx + (unsigned)x_lo;
                            const U4 u = ld_off<U4>(cell_start, 4u * first);
                            rb = u.a;
                            re = w == 0 ? u.b : (w == 1 ? u.c : u.d);
                        }
                        if (q == 4) { cb = rb; ce = re; }
                        else s_runs[q < 4 ? q : q - 1][threadIdx.x] = make_uint2(rb, re);
                    }
                    ICP_STAMP(1);
                    // one candidate.  Slots past the end of a lane's candidates read the dummy record tgt[dummy] (+inf
                    // coordinates: its distance is +inf, above FLT_MAX, so it never wins), which keeps the scan free of
                    // per-candidate branches.
                    auto visit = [&](const float4& c) {
                        const float dx = tp0 - c.x, dyy = tp1 - c.y, dzz = tp2 - c.z;
                        const float d = dx * dx + dyy * dyy + dzz * dzz;
                        const unsigned kd = __float_as_uint(d), ki = __float_as_uint(c.w);
                        const unsigned long long key = ((unsigned long long)kd << 32) | (unsigned long long)ki;
                        if (DETECT) // every target is visited once, so an equal distance is another target's (the running best only falls: the last such event is the one
                            tie_d = select_lanes(__builtin_amdgcn_ballot_w64(kd == (unsigned)(best_key >> 32)), tie_d, kd); // at the final distance, if there is one).
                        // (Measured and not kept: the mark as one bit per lane in a scalar register pair -- one VALU compare, scalar bookkeeping: +6 % instead of
                        //  +2 %; the mark in bit 31 of the running best's index -- no register of its own, three VALU: +6 %.)
                        const unsigned long long nearer = __builtin_amdgcn_ballot_w64(key < best_key);
                        best_key = ((unsigned long long)select_lanes(nearer, (unsigned)(best_key >> 32), kd) << 32) |
                                   (unsigned long long)select_lanes(nearer, (unsigned)best_key, ki);
                    };
                    // 1. the centre row (dy,dz) = (0,0), kScan candidates per trip: the loads are independent, so their
                    //    L2 round trips overlap (the scan is a latency chain otherwise)
                    for (unsigned p = cb; p < ce; p += kScanC) {
                        float4 c[kScanC];
#pragma unroll
                        for (int k = 0; k < kScanC; ++k) c[k] = ld_off<float4>(tgt, 16u * (p + k < ce ? p + k : dummy));
#pragma unroll
                        for (int k = 0; k < kScanC; ++k) visit(c[k]);
                    }
                    // 2. the other 8 rows: those that can still hold the nearest neighbour are decided NOW, with the centre
                    //    row's best distance, and their runs are walked as ONE flattened candidate stream.  A wave then
                    //    makes max-over-lanes ceil(candidates / kScan) trips instead of one or two trips for every row that
                    //    ANY of its lanes still needs (the union over 64 lanes is almost always all 8 rows).  The runs of a
                    //    lane sit in its private LDS column, which a dynamic index reaches without scratch memory; the
                    //    survivors are compacted in place (nr never overtakes the slot being read).
                    ICP_STAMP(2);
                    const float best_d = __uint_as_float((unsigned)(best_key >> 32));
                    int nr = 0;
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        if (q == 4) continue;
                        const int dy = q % 3 - 1, dz = q / 3 - 1;
                        const float bound = gy[dy + 1] * gy[dy + 1] + gz[dz + 1] * gz[dz + 1];
                        const uint2 run = s_runs[q < 4 ? q : q - 1][threadIdx.x];
                        if (run.x < run.y && !(0.99f * bound > best_d)) { s_runs[nr][threadIdx.x] = run; ++nr; }
                    }
                    unsigned p = 0, e = 0;
                    int ri = 0;
                    while (p < e || ri < nr) {
                        unsigned idx[kScan];
#pragma unroll
                        for (int k = 0; k < kScan; ++k) {
                            if (p == e && ri < nr) { const uint2 run = s_runs[ri][threadIdx.x]; p = run.x; e = run.y; ++ri; } // runs are non-empty
                            idx[k] = p < e ? p++ : dummy;
                        }
                        float4 c[kScan];
#pragma unroll
                        for (int k = 0; k < kScan; ++k) c[k] = ld_off<float4>(tgt, 16u * idx[k]);
#pragma unroll
                        for (int k = 0; k < kScan; ++k) visit(c[k]);
                    }
                }
            }
#ifdef ICP_REPEAT
            }
#endif
            ICP_STAMP(3);
            best = best_key != kNoKey ? (int)(unsigned)best_key : -1;
            nn[i] = best;
            if (DETECT && best >= 0 && tie_d == (unsigned)(best_key >> 32)) {
                // tie_count only ever grows (no reset between launches: the host keeps the running total, which it learns from sums[29])
                tied = true;
                TieRec* rec = tie_rec + (atomicAdd(tie_count, 1u) - tie_base); // at most n records per launch
                auto put = [](void* p, unsigned v) { __hip_atomic_store(static_cast<unsigned*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
                put(&rec->tp[0], __float_as_uint(tp0)); put(&rec->tp[1], __float_as_uint(tp1)); put(&rec->tp[2], __float_as_uint(tp2)); put(&rec->src, i);
                put(&rec->s[0], __float_as_uint(s0)); put(&rec->s[1], __float_as_uint(s1)); put(&rec->s[2], __float_as_uint(s2)); put(&rec->best, (unsigned)best);
                __builtin_amdgcn_s_waitcnt(0); // the record has arrived before its stamp says so (no fence: that would write the whole L2 back)
                put(&rec->stamp, tie_stamp);
            }
          }
        } else {
            best = nn[i];
        }
        if (best >= 0) {
            const F3 tv = ld_off<F3>(tgt_orig, 12u * (unsigned)best);
            t0 = tv.x; t1 = tv.y; t2 = tv.z;
            if (kPlane) { const F3 nv = ld_off<F3>(nrm_orig, 12u * (unsigned)best); n0 = nv.x; n1 = nv.y; n2 = nv.z; }
            // CountInliers (ICP.cpp:15-23): ||(R s + t) - target||^2 in float, compared in double
            const float d0 = (sum3(M[0] * s0, M[1] * s1, M[2] * s2) + M[3]) - t0;
            const float d1 = (sum3(M[4] * s0, M[5] * s1, M[6] * s2) + M[7]) - t1;
            const float d2 = (sum3(M[8] * s0, M[9] * s1, M[10] * s2) + M[11]) - t2;
            e = (double)sum3(d0 * d0, d1 * d1, d2 * d2);
            inlier = e < thr2;
        }
        // the point the sums are taken over: the transformed point, except for the final pass of PointToPoint (MODE 2)
        a0 = MODE == 2 ? s0 : tp0; a1 = MODE == 2 ? s1 : tp1; a2 = MODE == 2 ? s2 : tp2;
        if (inl) inl[i] = inlier ? best : -1;
        if (MODE == 2 && tie_rec) { // (FinalAux: see there)
            FinalAux* ax = reinterpret_cast<FinalAux*>(tie_rec);
            const float* O = ax->T_old;
            const float o0 = ((O[0] * s0 + O[1] * s1) + O[2] * s2) + O[3] * 1.0f, o1 = ((O[4] * s0 + O[5] * s1) + O[6] * s2) + O[7] * 1.0f;
            const float o2 = ((O[8] * s0 + O[9] * s1) + O[10] * s2) + O[11] * 1.0f, o3 = ((O[12] * s0 + O[13] * s1) + O[14] * s2) + O[15] * 1.0f;
            const float p0 = o0 / o3, p1 = o1 / o3, p2 = o2 / o3; // the query the last search ran
            bool beyond = best < 0;
            if (!beyond) { const float dx = p0 - t0, dy = p1 - t1, dz = p2 - t2; beyond = !(dx * dx + dy * dy + dz * dz <= ax->reach2); }
            const float n3 = ((M[12] * s0 + M[13] * s1) + M[14] * s2) + M[15] * 1.0f;
            const float m0 = (((M[0] * s0 + M[1] * s1) + M[2] * s2) + M[3] * 1.0f) / n3 - p0, m1 = (((M[4] * s0 + M[5] * s1) + M[6] * s2) + M[7] * 1.0f) / n3 - p1,
                        m2 = (((M[8] * s0 + M[9] * s1) + M[10] * s2) + M[11] * 1.0f) / n3 - p2;
            const float moved = sqrtf(m0 * m0 + m1 * m1 + m2 * m2);
            // (NaN anywhere: the comparisons are false -- such a point is no inlier in the reference either)
            if (beyond && (moved + ax->thr) * 1.0001f >= ax->reach) {
                unsure = true;
                ax->list[atomicAdd(&ax->count, 1u)] = i;
            }
        }
    }
    ICP_STAMP(4);
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    if (DETECT && tied) acc[29] = 1.0; // the number of reported queries travels with the sums
    if (MODE == 2 && unsure) acc[30] = 1.0; // likewise the final pass's points to re-decide
    if (inlier) {
        acc[27] = e;
        acc[28] = 1.0;
        if (kPlane) {
            // ICP.cpp:121-136: row = [n ; s' x n], r = n.s' - n.t
            const float r = sum3(n0 * a0, n1 * a1, n2 * a2) - sum3(n0 * t0, n1 * t1, n2 * t2);
            const float row[6] = {n0, n1, n2, a1 * n2 - a2 * n1, a2 * n0 - a0 * n2, a0 * n1 - a1 * n0};
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[k++] = (double)(row[a] * row[b]);
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] = (double)(r * row[a]);
        } else {
            acc[0] = a0; acc[1] = a1; acc[2] = a2;
            acc[3] = t0; acc[4] = t1; acc[5] = t2;
            acc[6] = (double)a0 * t0; acc[7] = (double)a0 * t1; acc[8] = (double)a0 * t2;
            acc[9] = (double)a1 * t0; acc[10] = (double)a1 * t1; acc[11] = (double)a1 * t2;
            acc[12] = (double)a2 * t0; acc[13] = (double)a2 * t1; acc[14] = (double)a2 * t2;
        }
    }
    // wave64 reduce-scatter, then LDS across the workgroup's waves, one partial per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    op::wave_reduce_scatter32(acc);
    if ((lane & 1) == 0) s_red[wave][lane >> 1] = acc[0];
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double v = 0;
        for (int w = 0; w < kIterThreads / 64; ++w) v += s_red[w][threadIdx.x];
        st_coherent(partials + (size_t)wg * kNSums + threadIdx.x, v); // logical order: the folds below sum in source order
    }

    ICP_STAMP(5);
    // ---- cross-workgroup finish ----
    constexpr int kRows = kIterThreads / 32;      // row lanes of the folds below
    const int fk = threadIdx.x & 31, fr = threadIdx.x >> 5;
    const unsigned grp = wg / per_group, n_groups = (gridDim.x + per_group - 1) / per_group;
    wait_stores_then_barrier(); // the partial row has reached memory before the arrival is counted
    if (threadIdx.x == 0) {
        const unsigned members = min(per_group, gridDim.x - grp * per_group);
        const unsigned prev = __hip_atomic_fetch_add(&sync[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev + 1u == members;
        if (s_last) __hip_atomic_store(&sync[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    ICP_STAMP(6);
    if (!s_last) return;
    {
        const unsigned lo = grp * per_group, hi = min(lo + per_group, gridDim.x);
        double v0 = 0, v1 = 0;
        unsigned p = lo + fr;
        for (; p + kRows < hi; p += 2 * kRows) { v0 += ld_coherent(partials + (size_t)p * kNSums + fk); v1 += ld_coherent(partials + (size_t)(p + kRows) * kNSums + fk); }
        for (; p < hi; p += kRows) v0 += ld_coherent(partials + (size_t)p * kNSums + fk);
        s_fin[fr][fk] = v0 + v1;
        __syncthreads();
        if (threadIdx.x < kNSums) {
            double t = 0;
            for (int r = 0; r < kRows; ++r) t += s_fin[r][threadIdx.x];
            if (host_out) { // host-solve loop: the group's row goes straight to host-mapped pinned memory, the host folds the rows
                if (threadIdx.x < kNSums - 1) __hip_atomic_store(&host_out[(size_t)grp * kNSums + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                st_coherent(stage + (size_t)grp * kNSums + threadIdx.x, t);
            }
        }
    }
    wait_stores_then_barrier();
    if (host_out) { // publish the row: the host spins on this sequence number (one per group)
        if (threadIdx.x == 0) __hip_atomic_store(&host_out[(size_t)grp * kNSums + kNSums - 1], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&sync[kGroups], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev + 1u == n_groups;
        if (s_last) __hip_atomic_store(&sync[kGroups], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    {
        double v = 0;
        for (unsigned p = fr; p < n_groups; p += kRows) v += ld_coherent(stage + (size_t)p * kNSums + fk);
        s_fin[fr][fk] = v;
        __syncthreads();
        if (threadIdx.x < kNSums) {
            double t = 0;
            for (int r = 0; r < kRows; ++r) t += s_fin[r][threadIdx.x];
            out[threadIdx.x] = t;
        }
    }
}

// Sums over an EXPLICIT correspondence list -- registration::EstimateRigidTransformationPointToPlane
// (ICP.cpp:108-144; MODE 1: rows [n ; s x n], r = n.s - n.t over inliers (source id, target id)) and
// geometry::EstimateRigidTransformation (Geometry.cpp:107-151; MODE 0: sum s, sum t, sum s t^T over point
// pairs given as 6 floats each).  Same accumulation and reduction as k_icp_iter.
template <int MODE>
__global__ __launch_bounds__(kIterThreads) void k_pair_sums(const float* __restrict__ src, const float* __restrict__ tgt, const float* __restrict__ nrm,
                                                            const int* __restrict__ inliers, size_t n, double* __restrict__ partials) {
    __shared__ double s_red[kIterThreads / 64][kNSums];
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (MODE == 1) {
            const size_t a = (size_t)inliers[2 * i], b = (size_t)inliers[2 * i + 1];
            const float s0 = src[3 * a], s1 = src[3 * a + 1], s2 = src[3 * a + 2];
            const float t0 = tgt[3 * b], t1 = tgt[3 * b + 1], t2 = tgt[3 * b + 2];
            const float n0 = nrm[3 * b], n1 = nrm[3 * b + 1], n2 = nrm[3 * b + 2];
            const float r = sum3(n0 * s0, n1 * s1, n2 * s2) - sum3(n0 * t0, n1 * t1, n2 * t2);
            const float row[6] = {n0, n1, n2, s1 * n2 - s2 * n1, s2 * n0 - s0 * n2, s0 * n1 - s1 * n0};
            int k = 0;
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int q = p; q < 6; ++q) acc[k++] += (double)(row[p] * row[q]);
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[21 + p] += (double)(r * row[p]);
        } else {
            const float a0 = src[6 * i], a1 = src[6 * i + 1], a2 = src[6 * i + 2], t0 = src[6 * i + 3], t1 = src[6 * i + 4], t2 = src[6 * i + 5];
            acc[0] += a0; acc[1] += a1; acc[2] += a2;
            acc[3] += t0; acc[4] += t1; acc[5] += t2;
            acc[6] += (double)a0 * t0; acc[7] += (double)a0 * t1; acc[8] += (double)a0 * t2;
            acc[9] += (double)a1 * t0; acc[10] += (double)a1 * t1; acc[11] += (double)a1 * t2;
            acc[12] += (double)a2 * t0; acc[13] += (double)a2 * t1; acc[14] += (double)a2 * t2;
        }
        acc[28] += 1.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    op::wave_reduce_scatter32(acc);
    if ((lane & 1) == 0) s_red[wave][lane >> 1] = acc[0];
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double v = 0;
        for (int w = 0; w < kIterThreads / 64; ++w) v += s_red[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kNSums + threadIdx.x] = v;
    }
}

// ---- ordered emission of the inlier rows ----------------------------------------------------------
// The reference's own accumulations are sequential float32 loops over the inliers in ascending source index
// (Geometry.cpp:117-133 for the returned T, ICP.cpp:121-136 for JTJ/JTr).  To reproduce their rounding the
// inlier rows are compacted in that order on the device (flag -> scan -> scatter) and summed by ONE host thread.
// KIND 0: original source point, target point (6 floats; the correspondence_set of ICP.cpp:215-221)
// KIND 1: transformed source point, target point, target normal (9 floats; ICP.cpp:195-196)
// KIND 2: transformed source point, target point (6 floats; ICP.cpp:76-79)
__global__ __launch_bounds__(256) void k_inl_flag(const int* __restrict__ inl, size_t n, unsigned* __restrict__ count) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) count[i] = inl[i] >= 0 ? 1u : 0u;
}
template <int KIND>
__global__ __launch_bounds__(256) void k_emit_rows(const float* __restrict__ T, const float* __restrict__ src, const float* __restrict__ tgt_orig,
                                                   const float* __restrict__ nrm_orig, const int* __restrict__ inl,
                                                   const unsigned* __restrict__ start, size_t n, float* __restrict__ rows) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = inl[i];
    if (b < 0) return;
    float s0 = src[3 * i], s1 = src[3 * i + 1], s2 = src[3 * i + 2];
    if (KIND != 0) { // TransformPoints (Geometry.cpp:19-27), the same operations as k_icp_iter
        float M[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) M[k] = T[k];
        const float q0 = ((M[0] * s0 + M[1] * s1) + M[2] * s2) + M[3] * 1.0f;
        const float q1 = ((M[4] * s0 + M[5] * s1) + M[6] * s2) + M[7] * 1.0f;
        const float q2 = ((M[8] * s0 + M[9] * s1) + M[10] * s2) + M[11] * 1.0f;
        const float q3 = ((M[12] * s0 + M[13] * s1) + M[14] * s2) + M[15] * 1.0f;
        s0 = q0 / q3; s1 = q1 / q3; s2 = q2 / q3;
    }
    if (KIND == 3) { // the Jacobian row and residual of the pair as ICP.cpp:121-136 forms them (op_host::plane_sums_reference_order): {n, s x n, n.s - n.t}
        const float t0 = tgt_orig[3 * (size_t)b], t1 = tgt_orig[3 * (size_t)b + 1], t2 = tgt_orig[3 * (size_t)b + 2];
        const float n0 = nrm_orig[3 * (size_t)b], n1 = nrm_orig[3 * (size_t)b + 1], n2 = nrm_orig[3 * (size_t)b + 2];
        const float ns = n0 * s0 + (n1 * s1 + n2 * s2), nt = n0 * t0 + (n1 * t1 + n2 * t2); // dot3 = Eigen's a0 + (a1 + a2)
        float* o7 = rows + (size_t)start[i] * 7;
        o7[0] = n0; o7[1] = n1; o7[2] = n2;
        o7[3] = s1 * n2 - s2 * n1; o7[4] = s2 * n0 - s0 * n2; o7[5] = s0 * n1 - s1 * n0;
        o7[6] = ns - nt;
        return;
    }
    constexpr int W = KIND == 1 ? 9 : 6;
    float* o = rows + (size_t)start[i] * W;
    o[0] = s0; o[1] = s1; o[2] = s2;
    o[3] = tgt_orig[3 * (size_t)b]; o[4] = tgt_orig[3 * (size_t)b + 1]; o[5] = tgt_orig[3 * (size_t)b + 2];
    if (KIND == 1) { o[6] = nrm_orig[3 * (size_t)b]; o[7] = nrm_orig[3 * (size_t)b + 1]; o[8] = nrm_orig[3 * (size_t)b + 2]; }
}

// Second pass of the reduction for the stand-alone estimators (k_pair_sums): one workgroup folds the per-workgroup rows
// in a fixed order.  (The ICP iteration kernel folds its own rows, see k_icp_iter.)
__global__ __launch_bounds__(1024) void k_reduce_rows(const double* __restrict__ partials, int n_partials, double* __restrict__ out) {
    __shared__ double s[32][kNSums];
    const int k = threadIdx.x & 31, grp = threadIdx.x >> 5; // 32 groups x 32 sums
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;                  // independent chains: loads stay in flight
    int p = grp;
    for (; p + 96 < n_partials; p += 128) {
        v0 += partials[(size_t)p * kNSums + k];
        v1 += partials[(size_t)(p + 32) * kNSums + k];
        v2 += partials[(size_t)(p + 64) * kNSums + k];
        v3 += partials[(size_t)(p + 96) * kNSums + k];
    }
    for (; p < n_partials; p += 32) v0 += partials[(size_t)p * kNSums + k];
    s[grp][k] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double t = 0;
        for (int g = 0; g < 32; ++g) t += s[g][threadIdx.x];
        out[threadIdx.x] = t;
    }
}

// ---- EstimateNormals: exact k-NN over the cell grid + PCA plane fit ----------------------------
// PointCloud::EstimateNormals (PointCloud.cpp:102-144): knn nearest points (nanoflann order:
// ascending squared distance), the prefix with SQUARED distance <= radius (KDTree.h:245-251),
// geometry::FitPlane (Geometry.cpp:172-218).  One thread per (cell-sorted) point; the k best are
// kept sorted in LDS (one column per thread); cells are scanned in growing Chebyshev rings until
// the k-th distance is provably final: every unscanned point is farther than ring * cell.
constexpr int kNrmThreads = 128;
constexpr int kNrmMaxK = 32;

__device__ __forceinline__ void sym3_smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double v[3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off < 1e-300 || off <= 1e-34 * diag) break; // converged to the limit of double
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), sn = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double kp = A[k][p], kq = A[k][q]; A[k][p] = c * kp - sn * kq; A[k][q] = sn * kp + c * kq; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double pk = A[p][k], qk = A[q][k]; A[p][k] = c * pk - sn * qk; A[q][k] = sn * pk + c * qk; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double kp = V[k][p], kq = V[k][q]; V[k][p] = c * kp - sn * kq; V[k][q] = sn * kp + c * kq; }
            }
    }
    // column of the smallest eigenvalue, selected without dynamic indexing (keeps A, V in registers)
    const bool m1 = A[1][1] < A[0][0];
    const double e01 = m1 ? A[1][1] : A[0][0];
    const bool m2 = A[2][2] < e01;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = m2 ? V[k][2] : (m1 ? V[k][1] : V[k][0]);
}

__global__ __launch_bounds__(kNrmThreads) void k_estimate_normals(Grid g, const unsigned* __restrict__ cell_start, const float4* __restrict__ pts,
                                                                  size_t m, int knn, float radius, float cell, float* __restrict__ normals) {
    // The k best candidates of a lane live in its LDS column, UNSORTED while the search runs: a candidate is compared with
    // the worst one kept (its key and slot are in registers) and, if better, overwrites it, after which the column is
    // rescanned for the new worst -- k reads.  Keeping the column sorted instead costs a shift loop per accepted
    // candidate whose trip count is the maximum over the 64 lanes, and some lane accepts almost every candidate: 72 k
    // LDS operations per wave against 11 k here (1.03 -> 0.4 ms at 307 200 points).  Keys are (bits of the squared
    // distance, original index): non-negative floats order like their bit patterns, so "nearer, ties to the smaller
    // index" (nanoflann's order, KDTree.h:245-251) is one unsigned 64-bit compare.  The column is sorted once at the end.
    __shared__ unsigned long long s_key[kNrmMaxK][kNrmThreads];
    __shared__ int s_p[kNrmMaxK][kNrmThreads]; // sorted position of the neighbour (its record is pts[pos])
    const int tid = threadIdx.x;
    const size_t q = blockIdx.x * (size_t)blockDim.x + tid;
    if (q >= m) return;
    const float4 me = pts[q];
    const int cx = cell_coord(me.x, g.ox, g.inv_cell, g.gx), cy = cell_coord(me.y, g.oy, g.inv_cell, g.gy),
              cz = cell_coord(me.z, g.oz, g.inv_cell, g.gz);
    int cnt = 0, worst_slot = 0;
    unsigned long long worst = 0ull;
    const int max_ring = max(g.gx, max(g.gy, g.gz));
    auto offer = [&](const float4& c, unsigned p) __attribute__((always_inline)) {
        const float dx = me.x - c.x, dy = me.y - c.y, dz = me.z - c.z;
        const float d = dx * dx + dy * dy + dz * dz;
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(c.w);
        if (cnt < knn) {
            s_key[cnt][tid] = key; s_p[cnt][tid] = (int)p;
            if (cnt == 0 || key > worst) { worst = key; worst_slot = cnt; }
            ++cnt;
        } else if (key < worst) {
            s_key[worst_slot][tid] = key; s_p[worst_slot][tid] = (int)p;
            worst = 0ull;
            for (int k = 0; k < knn; ++k) {
                const unsigned long long kk = s_key[k][tid];
                if (kk >= worst) { worst = kk; worst_slot = k; }
            }
        }
    };
    for (int ring = 0; ring <= max_ring; ++ring) {
        for (int z = cz - ring; z <= cz + ring; ++z) {
            if (z < 0 || z >= g.gz) continue;
            for (int y = cy - ring; y <= cy + ring; ++y) {
                if (y < 0 || y >= g.gy) continue;
                const bool shell_row = (z == cz - ring || z == cz + ring || y == cy - ring || y == cy + ring);
                // on a shell row scan the whole x run, otherwise only the two x end cells of the ring
                for (int part = 0; part < (shell_row ? 1 : 2); ++part) {
                    int x_lo, x_hi;
                    if (shell_row) { x_lo = cx - ring; x_hi = cx + ring; }
                    else { x_lo = x_hi = part == 0 ? cx - ring : cx + ring; if (ring == 0 && part == 1) continue; }
                    x_lo = max(x_lo, 0); x_hi = min(x_hi, g.gx - 1);
                    if (x_lo > x_hi) continue;
                    const size_t row = ((size_t)z * g.gy + y) * g.gx;
                    const unsigned beg = cell_start[row + x_lo], end = cell_start[row + x_hi + 1]; // exclusive scan incl. the total
                    for (unsigned p = beg; p < end; p += 4) { // four candidates in flight per trip
                        float4 c[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) c[k] = pts[min(p + k, end - 1)];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (p + k < end) offer(c[k], p + k);
                    }
                }
            }
        }
        // every point outside the scanned cube is farther than ring * cell from the query
        const float reach = (float)ring * cell;
        if (cnt == knn && __uint_as_float((unsigned)(worst >> 32)) <= reach * reach) break;
        if (cnt == (int)min((size_t)knn, m) && ring >= max_ring) break;
    }
    // ascending (distance, index): insertion sort of the lane's column
    for (int i = 1; i < cnt; ++i) {
        const unsigned long long key = s_key[i][tid];
        const int pp = s_p[i][tid];
        int j = i;
        while (j > 0) {
            const unsigned long long prev = s_key[j - 1][tid];
            if (!(key < prev)) break;
            s_key[j][tid] = prev; s_p[j][tid] = s_p[j - 1][tid];
            --j;
        }
        s_key[j][tid] = key; s_p[j][tid] = pp;
    }
    int used = 0;
    while (used < cnt && !(__uint_as_float((unsigned)(s_key[used][tid] >> 32)) > radius)) ++used; // squared distance vs radius, as the reference does
    float nx = 0, ny = 0, nz = 0;
    if (used >= 3) {
        float s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < used; ++k) { const float4 c = pts[s_p[k][tid]]; s0 += c.x; s1 += c.y; s2 += c.z; }
        const float m0 = s0 / (float)used, m1 = s1 / (float)used, m2 = s2 / (float)used;
        float w00 = 0, w01 = 0, w02 = 0, w11 = 0, w12 = 0, w22 = 0, w10 = 0, w20 = 0, w21 = 0;
        for (int k = 0; k < used; ++k) {
            const float4 c = pts[s_p[k][tid]];
            const float d0 = c.x - m0, d1 = c.y - m1, d2 = c.z - m2;
            w00 += d0 * d0; w01 += d0 * d1; w02 += d0 * d2; w10 += d1 * d0; w11 += d1 * d1; w12 += d1 * d2;
            w20 += d2 * d0; w21 += d2 * d1; w22 += d2 * d2;
        }
        const float fn = (float)used;
        double v[3];
        sym3_smallest_eigvec((double)(w00 / fn), 0.5 * ((double)(w01 / fn) + (double)(w10 / fn)), 0.5 * ((double)(w02 / fn) + (double)(w20 / fn)),
                             (double)(w11 / fn), 0.5 * ((double)(w12 / fn) + (double)(w21 / fn)), (double)(w22 / fn), v);
        nx = (float)v[0]; ny = (float)v[1]; nz = (float)v[2];
        const float z2 = sum3(nx * nx, ny * ny, nz * nz);
        if (z2 > 0) { const float l = sqrtf(z2); nx /= l; ny /= l; nz /= l; }
    }
    const int orig = __float_as_int(me.w);
    normals[3 * (size_t)orig] = nx; normals[3 * (size_t)orig + 1] = ny; normals[3 * (size_t)orig + 2] = nz;
}

// ---- LoadFromDepth with order-preserving compaction --------------------------------------------
__global__ __launch_bounds__(256) void k_depth_count(const void* __restrict__ depth, int is_u16, float depth_scale, size_t npix,
                                                     unsigned* __restrict__ count) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float z = is_u16 ? (float)((const unsigned short*)depth)[i] / depth_scale : ((const float*)depth)[i];
    count[i] = z > 0 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_depth_scatter(const void* __restrict__ depth, int is_u16, op_camera cam, size_t npix,
                                                       const unsigned* __restrict__ start, float* __restrict__ xyz,
                                                       const unsigned char* __restrict__ rgb, float* __restrict__ colors) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float z = is_u16 ? (float)((const unsigned short*)depth)[i] / cam.depth_scale : ((const float*)depth)[i];
    if (!(z > 0)) return;
    const int r = (int)(i / cam.width), c = (int)(i - (size_t)r * cam.width);
    const unsigned p = start[i];
    xyz[3 * p] = ((float)c - cam.cx) * z / cam.fx; // PointCloud.cpp:90-93
    xyz[3 * p + 1] = ((float)r - cam.cy) * z / cam.fy;
    xyz[3 * p + 2] = z;
    if (colors) { // LoadFromRGBD (PointCloud.cpp:40-42): Point3(b0,b1,b2) / 255.0f in stored channel order
        colors[3 * p] = (float)rgb[3 * i] / 255.0f;
        colors[3 * p + 1] = (float)rgb[3 * i + 1] / 255.0f;
        colors[3 * p + 2] = (float)rgb[3 * i + 2] / 255.0f;
    }
}

int device_exclusive_scan(const unsigned* d_count, size_t n, unsigned* d_start, hipStream_t stream, unsigned* total_out) {
    const size_t nwg = (n + kScanWg - 1) / kScanWg;
    unsigned* d_tot = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_tot, (nwg + 1) * sizeof(unsigned)));
    hipLaunchKernelGGL(k_scan_totals, dim3((unsigned)nwg), dim3(256), 0, stream, d_count, n, d_tot);
    hipLaunchKernelGGL(k_scan_of_totals, dim3(1), dim3(1024), 0, stream, d_tot, nwg);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nwg), dim3(256), 0, stream, d_count, n, (const unsigned*)d_tot, d_start);
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess && total_out) {
        unsigned last_start = 0, last_count = 0;
        e = hipMemcpy(&last_start, d_start + (n - 1), 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&last_count, d_count + (n - 1), 4, hipMemcpyDeviceToHost);
        *total_out = last_start + last_count;
    }
    op::cached_free(d_tot);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "scan failed: %s", hipGetErrorString(e));
    return OP_OK;
}

// nn[source] = target for the queries the host re-decided
__global__ __launch_bounds__(256) void k_patch_nn(const int2* __restrict__ patch, unsigned n, int* __restrict__ nn) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k < n) nn[patch[k].x] = patch[k].y;
}

} // namespace

namespace { template <int, int, int, int> struct SeqRendezvous; } // seq_sums.hpp: where reference-order contexts that run at the same time take their sequential sums
struct op_icp {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t m = 0, n = 0;
    double threshold = 0;
    bool has_normals = false;
    Grid grid{};
    size_t ncell = 0;
    float* tgt_orig = nullptr; // m x 3 (original order)
    float4* tgt = nullptr;     // sorted by cell
    unsigned* cell_start = nullptr; // exclusive scan of the per-cell counts, ncell + 4 entries
    unsigned* sync = nullptr;       // arrival counters of k_icp_iter's cross-workgroup finish
    float* src = nullptr;
    size_t src_cap = 0;
    int *nn = nullptr, *inl = nullptr;
    double *partials = nullptr, *result = nullptr, *stage = nullptr;
    double* result_host = nullptr;      // pinned + mapped: k_reduce_update publishes the sums here (host-solve path)
    double* result_host_dev = nullptr;  // its device-side address
    double seq = 0.0;                   // publication sequence number
    hipEvent_t chunk_ev[8] = {};        // arrival of the chunks of inlier rows at the host (reference-order finish)
    float* T_dev = nullptr;        // start_T (16 floats)
    int n_wg = 0, partials_cap = 0;
    // reference-order finish / strict sums (OP_ICP_OPT_*): ordered inlier rows
    int finish = OP_ICP_FINISH_REFERENCE, sums = OP_ICP_SUMS_FP64;
    float* nrm_orig = nullptr;          // m x 3 target normals in original order (rows of KIND 1)
    unsigned *flag = nullptr, *start = nullptr, *scan_tot = nullptr;
    float *rows_dev = nullptr, *rows_host = nullptr; // src_cap x 9 floats each; rows_host is pinned
    size_t rows_cap = 0;
    // reference-order point-to-plane sums on the device (k_seq_sums, seq_sums.hpp): the 42 results + the row count, and whether the kernel may have its LDS
    float* seq_out = nullptr;
    float* seq_host = nullptr;       // pinned
    unsigned* seq_total = nullptr;
    int seq_ok = -1;                 // -1: not asked yet
    // op_icp_run_enqueue / op_icp_wait: the loop needs the host after every iteration (the 6x6 solve), so an enqueued run proceeds on a host
    // thread of the context's own -- K contexts (each with its stream) register K frame pairs side by side: ICP's only parallel axis (replicas)
    // OP_ICP_TIES_REFERENCE (default): queries whose nearest candidates are exactly equidistant are re-decided on the host in the tree the reference would build
    int ties = OP_ICP_TIES_REFERENCE;
    unsigned* tie_count = nullptr;      // device: grows by one per reported query, never reset between launches
    unsigned tie_total = 0;             // its value once the launches issued so far have run (the host adds sums[29] of every pass)
    unsigned tie_stamp = 0;             // stamp of the last search launch (its records carry it)
    TieRec* tie_rec = nullptr;          // pinned + mapped, src_cap records
    TieRec* tie_rec_dev = nullptr;      // its device-side address
    int2* tie_patch = nullptr;          // pinned + mapped, src_cap entries: (source index, target index) for k_patch_nn
    int2* tie_patch_dev = nullptr;
    size_t tie_cap = 0;
    float* tgt_host = nullptr;          // pinned: the target in original order, downloaded when the first tie shows up (the tie tree searches it)
    float* nrm_host = nullptr;          // pinned: the target's normals, downloaded when a point-to-plane pass first changes a partner
    op_host::NanoTree tie_tree;
    uint64_t tie_queries = 0, tie_changed = 0; // since the context was created
    FinalAux* fin_aux = nullptr;        // device: what the final pass of op_icp_run reports about correspondences it cannot trust (FinalAux)
    unsigned* fin_list = nullptr;       // device, src_cap entries
    size_t fin_cap = 0;
    uint64_t fin_redecided = 0;         // since the context was created
    SeqRendezvous<42, 7, 1, 5>* seq_batch = nullptr; // set for the duration of an op_icp_run_many call: this context's sequential sums are taken in one launch with the other contexts'
    hipEvent_t seq_ev = nullptr;        // "my ordered rows are in place" (recorded on the context's stream for the batch's stream to wait on)
    std::thread worker;
    bool worker_active = false;
    int worker_rc = OP_OK;
    char worker_err[512] = "";
};

namespace {

// one fused pass (transform + NN + inliers + sums + reduction); start_T is read from c->T_dev unless host_T is given.
template <int MODE, bool DETECT = false>
void launch_pass(op_icp* c, bool write_inl, const float* host_T = nullptr, double seq = 0.0, FinalAux* final_aux = nullptr) {
    Mat4 Tv;
    if (host_T) std::memcpy(Tv.m, host_T, sizeof(Tv.m)); else std::memset(Tv.m, 0, sizeof(Tv.m));
    const unsigned per_group = (unsigned)((c->n_wg + kGroups - 1) / kGroups);
    if (DETECT) ++c->tie_stamp;
    hipLaunchKernelGGL((k_icp_iter<MODE, DETECT>), dim3(c->n_wg), dim3(kIterThreads), 0, c->stream, host_T ? (const float*)nullptr : (const float*)c->T_dev, Tv,
                       (const float*)c->src, (unsigned)c->n, c->grid, (const unsigned*)c->cell_start, (const float4*)c->tgt, (unsigned)c->m,
                       (const float*)c->tgt_orig, (const float*)c->nrm_orig, c->threshold * c->threshold, c->nn, write_inl ? c->inl : nullptr, c->partials,
                       c->stage, c->sync, per_group, c->result, host_T ? c->result_host_dev : nullptr, seq, c->tie_count, c->tie_total,
                       MODE == 2 ? reinterpret_cast<TieRec*>(final_aux) : c->tie_rec_dev, c->tie_stamp);
}

// Waits for the rows of sums the launch with sequence number c->seq publishes (one per group of workgroups, in
// host-mapped pinned memory) and adds them in group order.
int wait_rows(op_icp* c, double r[kNSums]) {
    volatile double* pub = c->result_host;
    const int per_group = (c->n_wg + kGroups - 1) / kGroups, n_groups = (c->n_wg + per_group - 1) / per_group;
    for (int k = 0; k < kNSums; ++k) r[k] = 0.0;
    for (int g = 0; g < n_groups; ++g) {
        volatile double* row = pub + (size_t)g * kNSums;
        for (unsigned spin = 0; row[kNSums - 1] != c->seq; ++spin) {
            if ((spin & 0xfff) == 0xfff && hipStreamQuery(c->stream) != hipErrorNotReady) { // finished or failed
                OP_HIP(hipStreamSynchronize(c->stream));
                if (row[kNSums - 1] != c->seq) return fail(OP_ERR_HIP, "icp: the iteration kernel did not publish its sums");
                break;
            }
            __builtin_ia32_pause();
        }
        for (int k = 0; k < kNSums - 1; ++k) r[k] += row[k];
    }
    return OP_OK;
}

int enqueue_pass(op_icp* c, int mode, bool write_inl) {
    const bool detect = c->ties == OP_ICP_TIES_REFERENCE;
    if (mode == 1) { if (detect) launch_pass<1, true>(c, write_inl); else launch_pass<1>(c, write_inl); }
    else if (mode == 0) { if (detect) launch_pass<0, true>(c, write_inl); else launch_pass<0>(c, write_inl); }
    else launch_pass<2>(c, write_inl);
    OP_HIP(hipGetLastError());
    return OP_OK;
}

// host-synchronous single pass with an explicit T (op_icp_iterate)
int ensure_tie_buffers(op_icp* c);
int resolve_ties(op_icp* c, int mode, const float T[16], bool write_inl, double out[kNSums], bool launch_retired, bool nn_is_read = true);
int run_pass(op_icp* c, int mode, const float T[16], bool write_inl, double out[kNSums]) {
    const bool detect = c->ties == OP_ICP_TIES_REFERENCE && mode < 2;
    if (detect) OP_TRY(ensure_tie_buffers(c));
    OP_HIP(hipMemcpyAsync(c->T_dev, T, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    OP_TRY(enqueue_pass(c, mode, write_inl));
    OP_HIP(hipMemcpyAsync(out, c->result, kNSums * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OP_HIP(hipStreamSynchronize(c->stream));
    if (detect) OP_TRY(resolve_ties(c, mode, T, write_inl, out, true));
    return OP_OK;
}

// OP_ICP_TIES_REFERENCE.  The search kernels (DETECT) report the queries whose nearest distance more than one target has -- a handful per
// pass on depth-derived clouds (two float32 squared distances that agree in every bit), every query on a lattice.  Their number comes
// back with the sums (sums[29]), the records through host-mapped memory, so a pass without them costs the marking in the scan and nothing
// else.  For the reported queries resolve_ties repeats the search on the host in the tree nanoflann would build (nn_tree.hpp: the first
// candidate its traversal meets wins).  Where the partner changes, the pair's contribution to the sums is exchanged on the host -- the
// same float expressions as the kernel's, accumulated in fp64 like its sums -- and nn[] is patched by a small kernel behind the pass
// (only the final pass and the pair list read it).  The reference-order modes (write_inl) and floods of ties take the sums again on the
// device instead (MODE 3 / 4 over the stored correspondences).  T = the pose of the pass; `out` = its sums, corrected on return.
constexpr size_t kTieStampChecked = 4096; // records whose arrival the host checks by their stamp (initialised when the buffer is taken)
int ensure_tie_buffers(op_icp* c) {
    if (c->tie_cap >= c->src_cap && c->tie_count) return OP_OK;
    if (c->tie_rec) op::cached_free(c->tie_rec);
    if (c->tie_patch) op::cached_free(c->tie_patch);
    c->tie_rec = nullptr; c->tie_patch = nullptr; c->tie_cap = 0;
    if (!c->tie_count) {
        OP_HIP(op::cached_malloc((void**)&c->tie_count, sizeof(unsigned)));
        OP_HIP(hipMemsetAsync(c->tie_count, 0, sizeof(unsigned), c->stream));
        c->tie_total = 0;
    }
    const size_t cap = std::max<size_t>(c->src_cap, 1);
    OP_HIP(op::cached_host_malloc((void**)&c->tie_rec, cap * sizeof(TieRec)));
    OP_HIP(op::cached_host_malloc((void**)&c->tie_patch, cap * sizeof(int2)));
    OP_HIP(hipHostGetDevicePointer((void**)&c->tie_rec_dev, c->tie_rec, 0));
    OP_HIP(hipHostGetDevicePointer((void**)&c->tie_patch_dev, c->tie_patch, 0));
    for (size_t k = 0; k < std::min(cap, kTieStampChecked); ++k) c->tie_rec[k].stamp = 0xffffffffu; // (a recycled buffer may hold any stamp; beyond these the host synchronises instead)
    c->tie_cap = c->src_cap;
    return OP_OK;
}

// what the pair (source point s with transformed position a, target t with normal n) adds to the sums of k_icp_iter<0 / 1>: the kernel's expressions
inline float h_sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
void pair_contribution(int mode, const float M[16], const float s[3], const float a[3], const float t[3], const float* n, double thr2, double acc[kNSums]) {
    for (int k = 0; k < kNSums; ++k) acc[k] = 0.0;
    const float d0 = (h_sum3(M[0] * s[0], M[1] * s[1], M[2] * s[2]) + M[3]) - t[0];
    const float d1 = (h_sum3(M[4] * s[0], M[5] * s[1], M[6] * s[2]) + M[7]) - t[1];
    const float d2 = (h_sum3(M[8] * s[0], M[9] * s[1], M[10] * s[2]) + M[11]) - t[2];
    const double e = (double)h_sum3(d0 * d0, d1 * d1, d2 * d2);
    if (!(e < thr2)) return;
    acc[27] = e; acc[28] = 1.0;
    if (mode == 1) {
        const float r = h_sum3(n[0] * a[0], n[1] * a[1], n[2] * a[2]) - h_sum3(n[0] * t[0], n[1] * t[1], n[2] * t[2]);
        const float row[6] = {n[0], n[1], n[2], a[1] * n[2] - a[2] * n[1], a[2] * n[0] - a[0] * n[2], a[0] * n[1] - a[1] * n[0]};
        int k = 0;
        for (int p = 0; p < 6; ++p)
            for (int q = p; q < 6; ++q) acc[k++] = (double)(row[p] * row[q]);
        for (int p = 0; p < 6; ++p) acc[21 + p] = (double)(r * row[p]);
    } else {
        for (int p = 0; p < 3; ++p) { acc[p] = a[p]; acc[3 + p] = t[p]; }
        for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) acc[6 + 3 * p + q] = (double)a[p] * t[q];
    }
}

// the target on the host and the (lazily split) tree the reference's nanoflann would build over it
int ensure_tie_tree(op_icp* c) {
    if (c->tie_tree.built()) return OP_OK;
    OP_HIP(op::cached_host_malloc((void**)&c->tgt_host, std::max<size_t>(c->m, 1) * 3 * sizeof(float))); // (pinned: the 3.7 MB come down at the link's rate)
    OP_HIP(hipMemcpy(c->tgt_host, c->tgt_orig, c->m * 3 * sizeof(float), hipMemcpyDeviceToHost));
    c->tie_tree.build(c->tgt_host, c->m, 10, false); // nodes are split as searches reach them: a few tied queries cost ~2 passes over the target, not the whole construction
    return OP_OK;
}

// searches [lo, hi) of `queries` (3 floats each) in the tie tree, a few host threads sharing a large batch (over the finished tree, which is read-only)
void tree_nearest(op_icp* c, const float* queries, size_t n, int* partner) {
    auto decide = [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) partner[k] = c->tie_tree.nearest(queries + 3 * k); };
    const unsigned n_threads = n >= 8192 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (n_threads > 1) {
        c->tie_tree.finish();
        std::vector<std::thread> pool;
        const size_t per = (n + n_threads - 1) / n_threads;
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(decide, std::min<size_t>(t * per, n), std::min<size_t>((t + 1) * per, n));
        for (std::thread& th : pool) th.join();
    } else {
        decide(0, n);
    }
}

// The final pass reported `n_unsure` source points whose stored partner may not be their nearest target under the pose of the last search
// (FinalAux): each is searched again, with that pose, in the tree the reference would search; nn[] is patched behind the pass.
int redecide_final(op_icp* c, const float T_old[16], size_t n_unsure) {
    OP_HIP(hipStreamSynchronize(c->stream));
    unsigned count = 0;
    OP_HIP(hipMemcpy(&count, reinterpret_cast<const char*>(c->fin_aux) + offsetof(FinalAux, count), sizeof(unsigned), hipMemcpyDeviceToHost));
    if ((size_t)count != n_unsure || count > c->n) return fail(OP_ERR_HIP, "icp: the final pass listed %u points to re-decide and counted %zu", count, n_unsure);
    OP_TRY(ensure_tie_buffers(c)); // (tie_patch)
    OP_TRY(ensure_tie_tree(c));
    std::vector<unsigned> idx(count);
    OP_HIP(hipMemcpy(idx.data(), c->fin_list, count * sizeof(unsigned), hipMemcpyDeviceToHost));
    std::sort(idx.begin(), idx.end()); // (the order the kernel appended them in is arbitrary; tree splits happen in a fixed order this way)
    std::vector<float> src(3 * c->n), q(3 * (size_t)count);
    OP_HIP(hipMemcpy(src.data(), c->src, src.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < count; ++k) { // TransformPoints (Geometry.cpp:19-27), the search kernel's expression
        const float* s3 = &src[3 * (size_t)idx[k]];
        const float* M = T_old;
        const float q0 = ((M[0] * s3[0] + M[1] * s3[1]) + M[2] * s3[2]) + M[3] * 1.0f, q1 = ((M[4] * s3[0] + M[5] * s3[1]) + M[6] * s3[2]) + M[7] * 1.0f;
        const float q2 = ((M[8] * s3[0] + M[9] * s3[1]) + M[10] * s3[2]) + M[11] * 1.0f, q3 = ((M[12] * s3[0] + M[13] * s3[1]) + M[14] * s3[2]) + M[15] * 1.0f;
        q[3 * k] = q0 / q3; q[3 * k + 1] = q1 / q3; q[3 * k + 2] = q2 / q3;
    }
    std::vector<int> partner(count);
    tree_nearest(c, q.data(), count, partner.data());
    for (size_t k = 0; k < count; ++k) c->tie_patch[k] = make_int2((int)idx[k], partner[k]);
    hipLaunchKernelGGL(k_patch_nn, dim3((count + 255u) / 256u), dim3(256), 0, c->stream, (const int2*)c->tie_patch_dev, count, c->nn);
    OP_HIP(hipGetLastError());
    c->fin_redecided += count;
    return OP_OK;
}

int resolve_ties(op_icp* c, int mode, const float T[16], bool write_inl, double out[kNSums], bool launch_retired, bool nn_is_read) {
    const unsigned n_tied = (unsigned)(out[29] + 0.5);
    c->tie_total += n_tied; // what the device counter now reads
    if (!n_tied) return OP_OK;
    if (n_tied > c->tie_cap) return fail(OP_ERR_HIP, "icp: the search reported %u tied queries for %zu source points", n_tied, c->n);
    if (!launch_retired && n_tied > kTieStampChecked) { OP_HIP(hipStreamSynchronize(c->stream)); launch_retired = true; } // a flood: let the launch retire
    if (!launch_retired) { // the sums were read from published rows: every record carries the launch's stamp once it has arrived
        volatile TieRec* rec = c->tie_rec;
        bool synced = false;
        for (unsigned k = 0; k < n_tied && !synced; ++k)
            for (unsigned spin = 0; rec[k].stamp != c->tie_stamp; ++spin) {
                if ((spin & 0xfff) == 0xfff && hipStreamQuery(c->stream) != hipErrorNotReady) { OP_HIP(hipStreamSynchronize(c->stream)); synced = true; break; }
                __builtin_ia32_pause();
            }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    OP_TRY(ensure_tie_tree(c));
    const TieRec* rec = c->tie_rec;
    std::vector<int> partner(n_tied);
    {   // (a lattice ties every query)
        std::vector<float> q(3 * (size_t)n_tied);
        for (unsigned k = 0; k < n_tied; ++k) { q[3 * k] = rec[k].tp[0]; q[3 * k + 1] = rec[k].tp[1]; q[3 * k + 2] = rec[k].tp[2]; }
        tree_nearest(c, q.data(), n_tied, partner.data());
    }
    size_t changed = 0;
    for (unsigned k = 0; k < n_tied; ++k)
        if (partner[k] != rec[k].best) c->tie_patch[changed++] = make_int2(rec[k].src, partner[k]);
    c->tie_queries += n_tied; c->tie_changed += changed;
    if (!changed) return OP_OK; // the smallest index happened to be the first the tree meets: the sums stand
    const bool on_host = !write_inl && changed <= 4096;
    if (nn_is_read || !on_host) { // nn[] follows in stream order (tie_patch is not written again before the next pass's sums have come back, i.e. after this kernel ran);
        // every search pass rewrites all of nn[], so inside a loop only the last iteration's partners are ever read (final pass, pair list)
        hipLaunchKernelGGL(k_patch_nn, dim3(((unsigned)changed + 255u) / 256u), dim3(256), 0, c->stream, (const int2*)c->tie_patch_dev, (unsigned)changed, c->nn);
        OP_HIP(hipGetLastError());
    }
    if (on_host) {
        const double thr2 = c->threshold * c->threshold;
        double was[kNSums], is[kNSums];
        if (mode == 1 && !c->nrm_host) {
            OP_HIP(op::cached_host_malloc((void**)&c->nrm_host, std::max<size_t>(c->m, 1) * 3 * sizeof(float)));
            OP_HIP(hipMemcpy(c->nrm_host, c->nrm_orig, c->m * 3 * sizeof(float), hipMemcpyDeviceToHost));
        }
        for (unsigned k = 0; k < n_tied; ++k) {
            if (partner[k] == rec[k].best) continue;
            const float* n_old = mode == 1 ? &c->nrm_host[3 * (size_t)rec[k].best] : nullptr;
            const float* n_new = mode == 1 ? &c->nrm_host[3 * (size_t)partner[k]] : nullptr;
            pair_contribution(mode, T, rec[k].s, rec[k].tp, &c->tgt_host[3 * (size_t)rec[k].best], n_old, thr2, was);
            pair_contribution(mode, T, rec[k].s, rec[k].tp, &c->tgt_host[3 * (size_t)partner[k]], n_new, thr2, is);
            for (int q = 0; q < 29; ++q) out[q] += is[q] - was[q];
        }
        return OP_OK;
    }
    OP_HIP(hipMemcpyAsync(c->T_dev, T, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (mode == 1) launch_pass<4>(c, write_inl); else launch_pass<3>(c, write_inl);
    OP_HIP(hipGetLastError());
    OP_HIP(hipMemcpyAsync(out, c->result, kNSums * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OP_HIP(hipStreamSynchronize(c->stream));
    return OP_OK;
}

void expand_plane_sums(const double in[kNSums], double JTJ[36], double JTr[6]) {
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) { JTJ[a * 6 + b] = in[k]; JTJ[b * 6 + a] = in[k]; ++k; }
    for (int a = 0; a < 6; ++a) JTr[a] = in[21 + a];
}

// Compacts the rows of the current inlier set (c->inl, written by a pass with write_inl) in ascending source
// index, copies the first n_rows of them to pinned host memory and waits.  The transform is read from c->T_dev.
// k_seq_sums needs ~150 KB of dynamic LDS (opt-in attribute) and three small buffers; false = sum on the host as before
bool seq_device_ok(op_icp* c) {
    if (c->seq_ok < 0) {
        int lds_max = 0;
        bool ok = hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) == hipSuccess && (size_t)lds_max >= seq_lds_bytes(42, 7, 1) &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seq_sums<42, 7, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(42, 7, 1)) == hipSuccess;
        if (ok) ok = op::cached_malloc((void**)&c->seq_out, 64 * sizeof(float)) == hipSuccess && op::cached_malloc((void**)&c->seq_total, sizeof(unsigned)) == hipSuccess &&
                     op::cached_host_malloc((void**)&c->seq_host, 64 * sizeof(float)) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        c->seq_ok = ok ? 1 : 0;
    }
    return c->seq_ok == 1;
}

// The reference-order contexts of one op_icp_run_many call take their sequential sums TOGETHER when there are five or more of them (seq_sums.hpp: SeqRendezvous): every such context has a submitter
// thread (its iterations synchronise the stream anyway) and each iteration ends in k_seq_sums -- ONE workgroup, ~0.8 ms for 3e5 rows; K independent runs scale to
// 4 x and no further (round 5: 2.3 k iterations/s at K = 4 and at K = 8), K workgroups of one launch do not have that limit.  One batcher per device.
using IcpSeqBatch = SeqRendezvous<42, 7, 1, 5>;
IcpSeqBatch* icp_seq_batch(int device) {
    static IcpSeqBatch pool[16];
    return device >= 0 && device < 16 ? &pool[device] : nullptr;
}

int emit_rows(op_icp* c, int kind, size_t n_rows, const float** rows) {
    if (rows) *rows = nullptr;
    if (!c->n || !n_rows) return OP_OK;
    if (c->rows_cap < c->n) {
        void* old[] = {c->flag, c->start, c->scan_tot, c->rows_dev};
        for (void* p : old)
            if (p) op::cached_free(p);
        if (c->rows_host) op::cached_free(c->rows_host);
        c->flag = c->start = c->scan_tot = nullptr; c->rows_dev = c->rows_host = nullptr; c->rows_cap = 0;
        const size_t cap = c->src_cap;
        OP_HIP(op::cached_malloc((void**)&c->flag, cap * sizeof(unsigned)));
        OP_HIP(op::cached_malloc((void**)&c->start, cap * sizeof(unsigned)));
        OP_HIP(op::cached_malloc((void**)&c->scan_tot, ((cap + kScanWg - 1) / kScanWg + 1) * sizeof(unsigned)));
        OP_HIP(op::cached_malloc((void**)&c->rows_dev, cap * 9 * sizeof(float)));
        OP_HIP(op::cached_host_malloc((void**)&c->rows_host, cap * 9 * sizeof(float)));
        c->rows_cap = cap;
    }
    const size_t n = c->n, nwg = (n + kScanWg - 1) / kScanWg;
    const unsigned g256 = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_inl_flag, dim3(g256), dim3(256), 0, c->stream, (const int*)c->inl, n, c->flag);
    hipLaunchKernelGGL(k_scan_totals, dim3((unsigned)nwg), dim3(256), 0, c->stream, (const unsigned*)c->flag, n, c->scan_tot);
    hipLaunchKernelGGL(k_scan_of_totals, dim3(1), dim3(1024), 0, c->stream, c->scan_tot, nwg);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nwg), dim3(256), 0, c->stream, (const unsigned*)c->flag, n, (const unsigned*)c->scan_tot, c->start);
#define OP_EMIT(K) hipLaunchKernelGGL(k_emit_rows<K>, dim3(g256), dim3(256), 0, c->stream, (const float*)c->T_dev, (const float*)c->src, \
                                      (const float*)c->tgt_orig, (const float*)c->nrm_orig, (const int*)c->inl, (const unsigned*)c->start, n, c->rows_dev)
    if (kind == 1) OP_EMIT(1); else if (kind == 2) OP_EMIT(2); else if (kind == 3) OP_EMIT(3); else OP_EMIT(0);
#undef OP_EMIT
    OP_HIP(hipGetLastError());
    if (!rows) return OP_OK; // enqueue only: the caller copies c->rows_dev itself
    const size_t w = kind == 1 ? 9 : 6;
    OP_HIP(hipMemcpyAsync(c->rows_host, c->rows_dev, n_rows * w * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    OP_HIP(hipStreamSynchronize(c->stream));
    *rows = c->rows_host;
    return OP_OK;
}

} // namespace

// While a run enqueued with op_icp_run_enqueue is in flight its worker thread owns the context (nn, tie buffers, fin_aux, seq, the stream): every other
// entry point refuses instead of racing with it.
#define OP_ICP_NOT_BUSY(c, what) do { if ((c)->worker_active) return fail(OP_ERR_INVALID, what ": a run enqueued with op_icp_run_enqueue has not been waited for (op_icp_wait)"); } while (0)

extern "C" {

// extent_divisor > 0: the cell is the largest extent of the bounding box / extent_divisor instead of the threshold
// (EstimateNormals' k-NN grid; the box comes from the device either way)
static int icp_create(const float* tgt_xyz, const float* tgt_normals, size_t m, double threshold, double extent_divisor, int mem, int device,
                      op_icp** out) {
    if (!out) return fail(OP_ERR_INVALID, "null out");
    *out = nullptr;
    if (!tgt_xyz && m) return fail(OP_ERR_INVALID, "null target");
    if (!(threshold > 0) && !(extent_divisor > 0)) return fail(OP_ERR_INVALID, "threshold must be > 0");
    if (m >= kMaxPoints) return fail(OP_ERR_INVALID, "target too large (at most %zu points)", kMaxPoints - 1);
    OP_TRY(op::use_device(device));
    op_icp* c = new op_icp();
    c->device = device; c->m = m; c->threshold = threshold; c->has_normals = tgt_normals != nullptr;
    auto bail = [&](int rc) { op_icp_destroy(c); return rc; };
#define OP_HIP_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
    OP_HIP_C(op::cached_stream(&c->stream));
    const size_t m1 = m ? m : 1;
    OP_HIP_C(op::cached_malloc((void**)&c->tgt_orig, m1 * 3 * sizeof(float)));
    OP_HIP_C(op::cached_malloc((void**)&c->tgt, (m + 1) * sizeof(float4))); // + the dummy record of the neighbour scan
    {
        const float inf = std::numeric_limits<float>::infinity();
        const float dummy[4] = {inf, inf, inf, 0.0f};
        OP_HIP_C(hipMemcpy(c->tgt + m, dummy, sizeof(dummy), hipMemcpyHostToDevice));
    }
    float* d_nrm = nullptr;
    const hipMemcpyKind kind = mem == OP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (m) { // device sources: ordered on the context's stream (a device-to-device hipMemcpy does not block the host)
        if (mem == OP_MEM_DEVICE) OP_HIP_C(hipMemcpyAsync(c->tgt_orig, tgt_xyz, m * 3 * sizeof(float), kind, c->stream));
        else OP_HIP_C(hipMemcpy(c->tgt_orig, tgt_xyz, m * 3 * sizeof(float), kind));
    }
    if (c->has_normals) {
        OP_HIP_C(op::cached_malloc((void**)&d_nrm, m1 * 3 * sizeof(float)));
        c->nrm_orig = d_nrm; // owned by the context from here on (freed by op_icp_destroy)
        if (m) {
            if (mem == OP_MEM_DEVICE) OP_HIP_C(hipMemcpyAsync(d_nrm, tgt_normals, m * 3 * sizeof(float), kind, c->stream));
            else OP_HIP_C(hipMemcpy(d_nrm, tgt_normals, m * 3 * sizeof(float), kind));
        }
    }
    // bounding box -> grid
    unsigned* d_box = nullptr;
    OP_HIP_C(op::cached_malloc((void**)&d_box, 6 * sizeof(unsigned)));
    unsigned init[6] = {0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    OP_HIP_C(hipMemcpy(d_box, init, sizeof(init), hipMemcpyHostToDevice));
    if (m) hipLaunchKernelGGL(k_bbox, dim3(128), dim3(256), 0, c->stream, (const float*)c->tgt_orig, m, d_box);
    OP_HIP_C(hipStreamSynchronize(c->stream));
    unsigned box[6];
    OP_HIP_C(hipMemcpy(box, d_box, sizeof(box), hipMemcpyDeviceToHost));
    op::cached_free(d_box);
    float mx[3], mn[3];
    for (int k = 0; k < 3; ++k) { mx[k] = dec_f(box[k]); mn[k] = dec_f(box[3 + k]); }
    if (!m || !(mx[0] >= mn[0])) { for (int k = 0; k < 3; ++k) { mx[k] = 0; mn[k] = 0; } }
    // cell >= threshold (slightly larger so that float rounding of the cell index cannot hide a
    // neighbour closer than threshold); grow it if the grid would exceed kMaxCells
    double cell = threshold * 1.001;
    if (extent_divisor > 0) {
        const float ext = std::max(mx[0] - mn[0], std::max(mx[1] - mn[1], mx[2] - mn[2]));
        cell = ext > 0 ? (double)ext / extent_divisor : 1.0;
        c->threshold = cell;
    }
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(mx[k]) || !std::isfinite(mn[k]) || !std::isfinite((double)mx[k] - (double)mn[k]))
            return bail(fail(OP_ERR_INVALID, "target bounding box is not finite"));
    for (int grow = 0;; ++grow) { // bounded: the extent is finite, so cell *= 1.26 reaches it within ~400 steps of doubles
        double tot = 1;
        for (int k = 0; k < 3; ++k) tot *= std::floor(((double)mx[k] - (double)mn[k]) / cell) + 2.0;
        if (tot <= (double)kMaxCells) break;
        if (grow > 4096 || !std::isfinite(cell)) return bail(fail(OP_ERR_INVALID, "cannot size the search grid (threshold %g)", threshold));
        cell *= 1.26;
    }
    c->grid.ox = mn[0]; c->grid.oy = mn[1]; c->grid.oz = mn[2];
    c->grid.inv_cell = (float)(1.0 / cell);
    c->grid.gx = (int)std::floor((mx[0] - mn[0]) / cell) + 2;
    c->grid.gy = (int)std::floor((mx[1] - mn[1]) / cell) + 2;
    c->grid.gz = (int)std::floor((mx[2] - mn[2]) / cell) + 2;
    c->ncell = (size_t)c->grid.gx * c->grid.gy * c->grid.gz;
    // cell_start = exclusive scan of the per-cell counts over ncell + 4 entries (the padding holds the total), so a
    // run of x-adjacent cells is [cell_start[first], cell_start[last + 1]) and one 16-byte load sees both ends
    const size_t n_tab = c->ncell + 4;
    OP_HIP_C(op::cached_malloc((void**)&c->cell_start, n_tab * sizeof(unsigned)));
    unsigned* d_count = nullptr;
    OP_HIP_C(op::cached_malloc((void**)&d_count, n_tab * sizeof(unsigned)));
    auto drop = [&]() { op::cached_free(d_count); };
    hipError_t e = hipMemsetAsync(d_count, 0, n_tab * sizeof(unsigned), c->stream);
    if (e != hipSuccess) { drop(); return bail(fail(OP_ERR_HIP, "grid build failed: %s", hipGetErrorString(e))); }
    if (m) hipLaunchKernelGGL(k_cell_count, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->tgt_orig, m, c->grid, d_count);
    int rc = device_exclusive_scan(d_count, n_tab, c->cell_start, c->stream, nullptr);
    if (rc != OP_OK) { drop(); return bail(rc); }
    if (m) hipLaunchKernelGGL(k_cell_scatter, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->tgt_orig, m,
                              c->grid, (const unsigned*)c->cell_start, d_count, c->tgt);
    e = hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) return bail(fail(OP_ERR_HIP, "grid build failed: %s", hipGetErrorString(e)));
    OP_HIP_C(op::cached_malloc((void**)&c->sync, (kGroups + 1) * sizeof(unsigned)));
    OP_HIP_C(hipMemset(c->sync, 0, (kGroups + 1) * sizeof(unsigned)));
    OP_HIP_C(op::cached_malloc((void**)&c->result, kNSums * sizeof(double)));
    OP_HIP_C(op::cached_malloc((void**)&c->T_dev, 16 * sizeof(float)));
    OP_HIP_C(op::cached_malloc((void**)&c->stage, (size_t)kGroups * kNSums * sizeof(double)));
    OP_HIP_C(op::cached_host_malloc((void**)&c->result_host, (size_t)kGroups * kNSums * sizeof(double)));
    OP_HIP_C(hipHostGetDevicePointer((void**)&c->result_host_dev, c->result_host, 0));
    std::memset(c->result_host, 0, (size_t)kGroups * kNSums * sizeof(double));
    for (hipEvent_t& ev : c->chunk_ev) OP_HIP_C(op::cached_event(&ev));
#undef OP_HIP_C
    *out = c;
    return OP_OK;
}

int op_icp_create(const float* tgt_xyz, const float* tgt_normals, size_t m, double threshold, int mem, int device, op_icp** out) {
    if (!(threshold > 0)) return fail(OP_ERR_INVALID, "threshold must be > 0");
    OP_TRY(icp_create(tgt_xyz, tgt_normals, m, threshold, 0.0, mem, device, out));
    // OP_RUNTIME_OPT_ICP_DEFAULT_SUMS: the reference's own sequential float32 sums unless the process opted into the fp64 reduction (the mode that is
    // within north_star's 1e-4 of the CPU path on every pair is the default of the drop-in surface)
    (*out)->sums = op::runtime_options().icp_default_sums.load();
    return OP_OK;
}

int op_icp_destroy(op_icp* c) {
    if (c && c->worker_active) { c->worker.join(); c->worker_active = false; }
    if (!c) return OP_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
#ifdef ICP_TRACE
    {
        std::vector<unsigned long long> t(8 * 8192);
        (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_icp_trace), t.size() * 8);
        const int nw = c->n_wg * (kIterThreads / 64);
        unsigned long long t0 = ~0ull, t_end = 0;
        for (int w = 0; w < nw; ++w) { t0 = std::min(t0, t[w * 8]); t_end = std::max(t_end, t[w * 8 + 6]); }
        double sum[7] = {0}, mx[7] = {0};
        for (int w = 0; w < nw; ++w)
            for (int k = 0; k < 7; ++k) { const double d = (double)(t[w * 8 + k] - (k ? t[w * 8 + k - 1] : t0)); sum[k] += d; mx[k] = std::max(mx[k], d); }
        fprintf(stderr, "icp trace (10 ns ticks, %d waves): span %llu; mean/max start %.0f/%.0f cells %.0f/%.0f centre %.0f/%.0f rest %.0f/%.0f gather %.0f/%.0f reduce %.0f/%.0f arrive %.0f/%.0f\n",
                nw, t_end - t0, sum[0] / nw, mx[0], sum[1] / nw, mx[1], sum[2] / nw, mx[2], sum[3] / nw, mx[3], sum[4] / nw, mx[4], sum[5] / nw, mx[5], sum[6] / nw, mx[6]);
    }
#endif
    void* ptrs[] = {c->tgt_orig, c->tgt, c->sync, c->cell_start, c->src, c->nn, c->inl, c->partials, c->result,
                    c->T_dev, c->stage, c->nrm_orig, c->flag, c->start, c->scan_tot, c->rows_dev};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (c->result_host) op::cached_free(c->result_host);
    for (hipEvent_t ev : c->chunk_ev)
        op::release_event(ev, c->device);
    if (c->rows_host) op::cached_free(c->rows_host);
    if (c->seq_ev) op::release_event(c->seq_ev, c->device);
    if (c->seq_out) op::cached_free(c->seq_out);
    if (c->seq_total) op::cached_free(c->seq_total);
    if (c->seq_host) op::cached_free(c->seq_host);
    if (c->fin_aux) op::cached_free(c->fin_aux);
    if (c->fin_list) op::cached_free(c->fin_list);
    if (c->tgt_host) op::cached_free(c->tgt_host);
    if (c->nrm_host) op::cached_free(c->nrm_host);
    if (c->tie_count) op::cached_free(c->tie_count);
    if (c->tie_rec) op::cached_free(c->tie_rec);
    if (c->tie_patch) op::cached_free(c->tie_patch);
    op::release_stream(c->stream, c->device);
    delete c;
    return OP_OK;
}

int op_release_cached_memory(void) {
    op::release_cached_memory();
    return OP_OK;
}

int op_icp_set_option(op_icp* c, int option, int value) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_set_option");
    if (option == OP_ICP_OPT_FINISH && (value == OP_ICP_FINISH_REFERENCE || value == OP_ICP_FINISH_FP64)) { c->finish = value; return OP_OK; }
    if (option == OP_ICP_OPT_SUMS && (value == OP_ICP_SUMS_FP64 || value == OP_ICP_SUMS_REFERENCE_F32)) { c->sums = value; return OP_OK; }
    if (option == OP_ICP_OPT_TIES && (value == OP_ICP_TIES_LOWEST_INDEX || value == OP_ICP_TIES_REFERENCE)) { c->ties = value; return OP_OK; }
    return fail(OP_ERR_INVALID, "op_icp_set_option: unknown option %d / value %d", option, value);
}

int op_icp_tie_stats(op_icp* c, uint64_t* tied_queries, uint64_t* changed) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_tie_stats");
    if (tied_queries) *tied_queries = c->tie_queries;
    if (changed) *changed = c->tie_changed;
    return OP_OK;
}

int op_icp_final_stats(op_icp* c, uint64_t* redecided) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_final_stats");
    if (redecided) *redecided = c->fin_redecided;
    return OP_OK;
}

int op_icp_set_source(op_icp* c, const float* src_xyz, size_t n, int mem) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_set_source");
    OP_HIP(hipSetDevice(c->device));
    if (!src_xyz && n) return fail(OP_ERR_INVALID, "null source");
    if (n >= kMaxPoints) return fail(OP_ERR_INVALID, "source too large (at most %zu points)", kMaxPoints - 1);
    if (n > c->src_cap) {
        void* old[] = {c->src, c->nn, c->inl, c->partials};
        for (void* p : old)
            if (p) op::cached_free(p);
        c->src = nullptr; c->nn = nullptr; c->inl = nullptr; c->partials = nullptr; c->partials_cap = 0;
        OP_HIP(op::cached_malloc((void**)&c->src, n * 3 * sizeof(float)));
        OP_HIP(op::cached_malloc((void**)&c->nn, n * sizeof(int)));
        OP_HIP(op::cached_malloc((void**)&c->inl, n * sizeof(int)));
        c->src_cap = n;
    }
    c->n = n;
    if (n) { // a device-to-device copy does not block the host: ordered on the context's stream, ahead of the kernels that read it
        if (mem == OP_MEM_DEVICE) OP_HIP(hipMemcpyAsync(c->src, src_xyz, n * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        else OP_HIP(hipMemcpy(c->src, src_xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    }
    int wg = (int)((n + kIterThreads - 1) / kIterThreads); // one source point per thread
    if (wg < 1) wg = 1;
    if (!c->partials || wg > c->partials_cap) {
        op::cached_free(c->partials);
        c->partials = nullptr;
        OP_HIP(op::cached_malloc((void**)&c->partials, (size_t)wg * kNSums * sizeof(double)));
        c->partials_cap = wg;
    }
    c->n_wg = wg;
    return OP_OK;
}

int op_icp_iterate(op_icp* c, const float T[16], int mode, double sums[42], uint64_t* n_inliers, double* sum_sq_err) {
    if (!c || !T || !sums) return fail(OP_ERR_INVALID, "null argument");
    OP_ICP_NOT_BUSY(c, "op_icp_iterate");
    OP_HIP(hipSetDevice(c->device));
    if (!c->src && c->n) return fail(OP_ERR_INVALID, "op_icp_set_source has not been called");
    if (mode == OP_ICP_POINT_TO_PLANE && !c->has_normals)
        return fail(OP_ERR_NO_NORMALS, "[ERROR]::[ICPPointToPlane]::target point cloud need to have normals.");
    double r[kNSums];
    OP_TRY(run_pass(c, mode == OP_ICP_POINT_TO_PLANE ? 1 : 0, T, false, r));
    std::memset(sums, 0, 42 * sizeof(double));
    if (mode == OP_ICP_POINT_TO_PLANE) expand_plane_sums(r, sums, sums + 36);
    else std::memcpy(sums, r, 15 * sizeof(double));
    if (n_inliers) *n_inliers = (uint64_t)(r[28] + 0.5);
    if (sum_sq_err) *sum_sq_err = r[27];
    return OP_OK;
}

static int icp_run_impl(op_icp* c, int mode, const float init_T[16], int max_iteration, op_icp_result* result, int32_t* pairs, size_t pairs_cap,
                        int32_t* per_iter_inliers, float* per_iter_T);

int op_icp_run(op_icp* c, int mode, const float init_T[16], int max_iteration, op_icp_result* result, int32_t* pairs, size_t pairs_cap,
               int32_t* per_iter_inliers, float* per_iter_T) {
    if (!c || !init_T || !result) return fail(OP_ERR_INVALID, "null argument");
    OP_ICP_NOT_BUSY(c, "op_icp_run");
    return icp_run_impl(c, mode, init_T, max_iteration, result, pairs, pairs_cap, per_iter_inliers, per_iter_T);
}

// ---- one registration = head (checks, resets) -> the iteration loop -> finish (final CountInliers, RegistrationResult).  The fp64-mode loop is a small state
// machine (IcpLoop: launch an iteration / complete it) so that ONE host thread can keep several contexts' iterations in flight (op_icp_run_many).
static int icp_run_head(op_icp* c, int mode, int max_iteration) {
    OP_HIP(hipSetDevice(c->device));
    if (mode == OP_ICP_POINT_TO_PLANE && !c->has_normals) // ICP.cpp:159-163: error line + default result
        return fail(OP_ERR_NO_NORMALS, "[ERROR]::[ICPPointToPlane]::target point cloud need to have normals.");
    if (!c->src && c->n) return fail(OP_ERR_INVALID, "op_icp_set_source has not been called");
    if (max_iteration <= 0 && c->n) OP_HIP(hipMemsetAsync(c->nn, 0xff, c->n * sizeof(int), c->stream)); // corresponding_index stays -1
    OP_HIP(hipMemsetAsync(c->sync, 0, (kGroups + 1) * sizeof(unsigned), c->stream)); // the counters reset themselves; this covers an aborted launch
    if (c->tie_count) { OP_HIP(hipMemsetAsync(c->tie_count, 0, sizeof(unsigned), c->stream)); c->tie_total = 0; } // likewise
    return OP_OK;
}

// ICP.cpp:177-199, fp64-reduction mode.  The reduced sums of every iteration come back to the host, which does the 6x6 solve (JacobiSVD
// semantics incl. its rank threshold -- the synthetic room's JTJ is rank-deficient, so the threshold matters) and the SE3
// exp, or the Kabsch step of PointToPoint, as north_star prescribes.  The round trip is kept short: the pose goes
// down as a by-value kernel argument and the sums come up through host-mapped pinned memory that the iteration
// kernel publishes with a sequence number the host spins on (no memcpy, no stream sync).  (PointToPoint's step used to
// run in a one-thread kernel after every iteration: 6 us of single-lane fp64 against 1 us of the same code on the host.)
struct IcpLoop {
    op_icp* c = nullptr;
    int pass_mode = 1, max_iteration = 0, it = 0;
    bool detect = false;
    float cur[16], last_search_T[16];
    int32_t* per_iter_inliers = nullptr;
    float* per_iter_T = nullptr;
};
static int icp_loop_begin(IcpLoop& L, op_icp* c, int mode, const float init_T[16], int max_iteration, int32_t* per_iter_inliers, float* per_iter_T) {
    L.c = c; L.pass_mode = mode == OP_ICP_POINT_TO_PLANE ? 1 : 0; L.max_iteration = max_iteration; L.it = 0;
    L.per_iter_inliers = per_iter_inliers; L.per_iter_T = per_iter_T;
    std::memcpy(L.cur, init_T, sizeof(L.cur));
    std::memcpy(L.last_search_T, init_T, sizeof(L.last_search_T));
    L.detect = c->ties == OP_ICP_TIES_REFERENCE;
    if (L.detect) OP_TRY(ensure_tie_buffers(c));
    return OP_OK;
}
static int icp_loop_launch(IcpLoop& L) { // enqueue iteration L.it (does not wait)
    op_icp* c = L.c;
    std::memcpy(L.last_search_T, L.cur, sizeof(L.cur));
    c->seq += 1.0;
    if (L.detect) { if (L.pass_mode == 1) launch_pass<1, true>(c, false, L.cur, c->seq); else launch_pass<0, true>(c, false, L.cur, c->seq); }
    else if (L.pass_mode == 1) launch_pass<1>(c, false, L.cur, c->seq);
    else launch_pass<0>(c, false, L.cur, c->seq);
    OP_HIP(hipGetLastError());
    return OP_OK;
}
static int icp_loop_complete(IcpLoop& L) { // wait for the sums of iteration L.it, solve, chain the pose
    op_icp* c = L.c;
    double r[kNSums];
    float tmp_T[16];
    OP_TRY(wait_rows(c, r));
    if (L.detect) OP_TRY(resolve_ties(c, L.pass_mode, L.cur, false, r, false, L.it == L.max_iteration - 1)); // nothing to do unless the pass reported tied queries (r[29])
    if (L.pass_mode == 1) {
        double JTJ[36], JTr[6];
        float x[6];
        expand_plane_sums(r, JTJ, JTr);
        op_host::solve6_psd(JTJ, JTr, x);  // ICP.cpp:137-138
        op_host::se3_exp(x, tmp_T);        // ICP.cpp:143
    } else {
        op_host::kabsch_from_sums(r[28], r, r + 3, r + 6, tmp_T); // ICP.cpp:79
    }
    op_host::mat4_mul(tmp_T, L.cur, L.cur); // ICP.cpp:198
    if (L.per_iter_inliers) L.per_iter_inliers[L.it] = (int32_t)(r[28] + 0.5);
    if (L.per_iter_T) std::memcpy(L.per_iter_T + 16 * L.it, L.cur, sizeof(L.cur));
    ++L.it;
    return OP_OK;
}

static int icp_run_finish(op_icp* c, const float start_T[16], const float last_search_T[16], int max_iteration, op_icp_result* result, int32_t* pairs, size_t pairs_cap);

static int icp_run_impl(op_icp* c, int mode, const float init_T[16], int max_iteration, op_icp_result* result, int32_t* pairs, size_t pairs_cap,
                        int32_t* per_iter_inliers, float* per_iter_T) {
    OP_TRY(icp_run_head(c, mode, max_iteration));
    float start_T[16], last_search_T[16];
    std::memcpy(last_search_T, init_T, sizeof(last_search_T));
    double r[kNSums];
    const int pass_mode = mode == OP_ICP_POINT_TO_PLANE ? 1 : 0;
    const bool strict = c->sums == OP_ICP_SUMS_REFERENCE_F32;
    if (strict) {
        // Validation mode: every iteration's inlier rows come to the host in inlier order and are summed there
        // sequentially in float32, as the reference's loops do (ICP.cpp:121-136 / Geometry.cpp:117-133 via :76-79);
        // the search, the inlier test and the rows themselves still come from the kernels.
        float cur[16], tmp_T[16];
        std::memcpy(cur, init_T, sizeof(cur));
        for (int it = 0; it < max_iteration; ++it) {
            std::memcpy(last_search_T, cur, sizeof(cur));
            OP_TRY(run_pass(c, pass_mode, cur, true, r)); // leaves `cur` in c->T_dev; with OP_ICP_TIES_REFERENCE, tied queries are re-decided inside
            const size_t n_it = (size_t)(r[28] + 0.5);
            const float* rows = nullptr;
            if (pass_mode == 1 && seq_device_ok(c) && n_it) {
                // the 36 + 6 sequential float32 sums by one wave on the device (k_seq_sums: the tracker's kernel, same row layout {J[6], r}): the ordered rows never
                // leave HBM, 42 numbers come back -- ~0.8 ms for 3e5 inliers instead of an 11 MB transfer and a pass on one host core
                OP_TRY(emit_rows(c, 3, n_it, nullptr));
                const unsigned n_rows_u = (unsigned)n_it;
                OP_HIP(hipMemcpyAsync(c->seq_total, &n_rows_u, sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
                hipError_t eb = hipErrorNotReady;
                if (c->seq_batch) // with the other contexts of the op_icp_run_many call: one launch, a workgroup each (hipErrorNotReady: too few of them -- alone, below)
                    eb = c->seq_batch->submit(c->rows_dev, c->seq_total, c->seq_out, c->seq_host, c->seq_ev, c->stream);
                if (eb != hipSuccess && eb != hipErrorNotReady) return fail(OP_ERR_HIP, "icp: the batched sequential sums failed: %s", hipGetErrorString(eb));
                if (eb == hipErrorNotReady) {
                    hipLaunchKernelGGL((k_seq_sums<42, 7, 1>), dim3(1), dim3(kSeqThreads), seq_lds_bytes(42, 7, 1), c->stream, (const float*)c->rows_dev, (const unsigned*)c->seq_total, c->seq_out);
                    OP_HIP(hipMemcpyAsync(c->seq_host, c->seq_out, 43 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
                    OP_HIP(hipStreamSynchronize(c->stream));
                }
                double JTJ[36], JTr[6];
                float x[6];
                for (int k = 0; k < 36; ++k) JTJ[k] = c->seq_host[k];
                for (int k = 0; k < 6; ++k) JTr[k] = c->seq_host[36 + k];
                op_host::solve6_psd<true>(JTJ, JTr, x);
                op_host::se3_exp(x, tmp_T);
                op_host::mat4_mul(tmp_T, cur, cur);
                if (per_iter_inliers) per_iter_inliers[it] = (int32_t)n_it;
                if (per_iter_T) std::memcpy(per_iter_T + 16 * it, cur, sizeof(cur));
                continue;
            }
            if (c->seq_batch) c->seq_batch->pass(); // nothing for the batched launch from this context in this iteration
            OP_TRY(emit_rows(c, pass_mode == 1 ? 1 : 2, n_it, &rows));
            if (pass_mode == 1) {
                double JTJ[36], JTr[6];
                float x[6];
                op_host::plane_sums_reference_order(rows, n_it, JTJ, JTr);
                op_host::solve6_psd<true>(JTJ, JTr, x);
                op_host::se3_exp(x, tmp_T);
            } else {
                op_host::kabsch_reference_order<true>(rows, n_it, tmp_T);
            }
            op_host::mat4_mul(tmp_T, cur, cur);
            if (per_iter_inliers) per_iter_inliers[it] = (int32_t)n_it;
            if (per_iter_T) std::memcpy(per_iter_T + 16 * it, cur, sizeof(cur));
        }
        std::memcpy(start_T, cur, sizeof(cur));
    } else {
        IcpLoop L;
        OP_TRY(icp_loop_begin(L, c, mode, init_T, max_iteration, per_iter_inliers, per_iter_T));
        while (L.it < max_iteration) { OP_TRY(icp_loop_launch(L)); OP_TRY(icp_loop_complete(L)); }
        std::memcpy(start_T, L.cur, sizeof(L.cur));
        std::memcpy(last_search_T, L.last_search_T, sizeof(last_search_T));
    }
    return icp_run_finish(c, start_T, last_search_T, max_iteration, result, pairs, pairs_cap);
}

static int icp_run_finish(op_icp* c, const float start_T[16], const float last_search_T[16], int max_iteration, op_icp_result* result, int32_t* pairs, size_t pairs_cap) {
    const bool strict = c->sums == OP_ICP_SUMS_REFERENCE_F32;
    double r[kNSums];
    // ICP.cpp:206-221: CountInliers with the final start_T over the last NN set, then Kabsch over
    // (original source, target) pairs
    // The sums of the final pass come back like the loop's (rows in host-mapped memory, no copy, no stream sync).
    // (c->T_dev is not brought up to date: the row emission of the finish works on the ORIGINAL source points.)
    FinalAux* aux = nullptr;
    FinalAux h; // (lives until the pass that reads its device copy has been waited for)
    if (max_iteration > 0 && c->n) { // (without an iteration there was no search: every correspondence is "none", as in the reference)
        if (c->fin_cap < c->n) {
            if (c->fin_aux) op::cached_free(c->fin_aux);
            if (c->fin_list) op::cached_free(c->fin_list);
            c->fin_aux = nullptr; c->fin_list = nullptr; c->fin_cap = 0;
            OP_HIP(op::cached_malloc((void**)&c->fin_aux, sizeof(FinalAux)));
            OP_HIP(op::cached_malloc((void**)&c->fin_list, c->src_cap * sizeof(unsigned)));
            c->fin_cap = c->src_cap;
        }
        std::memcpy(h.T_old, last_search_T, sizeof(h.T_old));
        h.reach = 0.9995f / c->grid.inv_cell; // every target within one cell edge of a query lies in the 27 cells the search scans (0.05 % for the rounding of the cell assignment)
        h.reach2 = h.reach * h.reach;
        h.thr = (float)c->threshold;
        h.count = 0u;
        h.list = c->fin_list;
        OP_HIP(hipMemcpyAsync(c->fin_aux, &h, sizeof(h), hipMemcpyHostToDevice, c->stream)); // (pageable source: the runtime stages it before the call returns)
        aux = c->fin_aux;
    }
    c->seq += 1.0;
    launch_pass<2>(c, true, start_T, c->seq, aux);
    OP_HIP(hipGetLastError());
    OP_TRY(wait_rows(c, r));
    if (r[30] > 0.5) { // correspondences the 27-cell search cannot vouch for under the pose they are now measured with (FinalAux): re-decided on the host, pass repeated
        OP_TRY(redecide_final(c, last_search_T, (size_t)(r[30] + 0.5)));
        c->seq += 1.0;
        launch_pass<2>(c, true, start_T, c->seq);
        OP_HIP(hipGetLastError());
        OP_TRY(wait_rows(c, r));
    }
    const double n_inl = r[28];
    result->n_inliers = (uint64_t)(n_inl + 0.5);
    result->rmse = std::sqrt(r[27] / n_inl);
    result->iterations = max_iteration;
    std::memcpy(result->last_T, start_T, sizeof(result->last_T));
    if (c->finish == OP_ICP_FINISH_REFERENCE) {
        // RegistrationResult::T as the reference forms it (ICP.cpp:215-221 -> Geometry.cpp:117-133): sequential
        // float32 sums over the correspondence_set in ascending source index
        // the rows come up in chunks and the first of the two sequential passes runs on each chunk as it lands
        const size_t n_rows = (size_t)result->n_inliers;
#ifdef ICP_TRACE
        auto now2 = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tf1 = now2();
#endif
        OP_TRY(emit_rows(c, 0, n_rows, nullptr));
        op_host::KabschReferenceOrder fit;
#ifndef ICP_ROW_CHUNKS
#define ICP_ROW_CHUNKS 3 // copies run at 33 GB/s for 1 MB pieces and at 53 GB/s from 4 MB on (tests/tools/pcie_probe.py): few, large pieces
#endif
        constexpr size_t kChunks = ICP_ROW_CHUNKS;
        static_assert(kChunks >= 1 && kChunks <= sizeof(c->chunk_ev) / sizeof(c->chunk_ev[0]), "one event per chunk");
        const size_t per = (n_rows + kChunks - 1) / kChunks;
        size_t n_ev = 0;
        for (size_t lo = 0; lo < n_rows; lo += per, ++n_ev) {
            const size_t cnt = std::min(per, n_rows - lo);
            OP_HIP(hipMemcpyAsync(c->rows_host + 6 * lo, c->rows_dev + 6 * lo, cnt * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            OP_HIP(hipEventRecord(c->chunk_ev[n_ev], c->stream));
        }
        for (size_t k = 0, lo = 0; k < n_ev; ++k, lo += per) {
            // a blocking wait: polling hipEventQuery in a loop slowed the copies themselves down (290 -> 265 us for the 7.3 MB)
            const hipError_t q = hipEventSynchronize(c->chunk_ev[k]);
            if (q != hipSuccess) return fail(OP_ERR_HIP, "icp: copying the inlier rows failed: %s", hipGetErrorString(q));
            fit.add_rows(c->rows_host + 6 * lo, std::min(per, n_rows - lo));
        }
#ifdef ICP_TRACE
        const double tf2 = now2();
#endif
        if (strict) fit.finish<true>(c->rows_host, n_rows, result->T);
        else fit.finish<false>(c->rows_host, n_rows, result->T);
#ifdef ICP_TRACE
        fprintf(stderr, "icp finish trace: rows to host + first pass %.1f us, second pass + fit %.1f us\n", (tf2 - tf1) * 1e6, (now2() - tf2) * 1e6);
#endif
    } else {
        op_host::kabsch_from_sums(n_inl, r, r + 3, r + 6, result->T); // order-free fp64 reduction
    }
    OP_HIP(hipStreamSynchronize(c->stream)); // the sums were read from published rows: the stream itself may still be draining
    if (pairs && c->n) {
        std::vector<int> inl(c->n);
        OP_HIP(hipMemcpy(inl.data(), c->inl, c->n * sizeof(int), hipMemcpyDeviceToHost));
        size_t k = 0;
        for (size_t i = 0; i < c->n && k < pairs_cap; ++i)
            if (inl[i] >= 0) { pairs[2 * k] = (int32_t)i; pairs[2 * k + 1] = inl[i]; ++k; }
    }
    return OP_OK;
}

// K registrations on K contexts, driven by ONE host thread (round-5 review: four submitter threads contend in the runtime's launch path -- k_icp_iter 21 -> 33 us,
// host side 19 -> 33 us at K = 4).  fp64-mode contexts: all K iterations are enqueued, then the thread goes round: wait for context k's sums (they arrive in
// host-mapped memory), solve, enqueue its next iteration, move on -- while it looks at one context the other K - 1 iterations run on the chip.  The finishes
// (final CountInliers + the reference-order Kabsch over ~3e5 rows on a host core, ~0.8 ms each) run side by side on helper threads.  Contexts in the
// reference-order mode synchronise their stream every iteration anyway: they get a submitter thread each, exactly as op_icp_run_enqueue gives them.
// init_T: K x 16 floats (NULL = identity for all); results: K entries.  Returns the first error.
int op_icp_run_many(op_icp* const* ctxs, int k, int mode, const float* init_T, int max_iteration, op_icp_result* results) {
    if (!ctxs || k < 1 || !results) return fail(OP_ERR_INVALID, "op_icp_run_many: null argument");
    static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < k; ++i) {
        if (!ctxs[i]) return fail(OP_ERR_INVALID, "op_icp_run_many: null context %d", i);
        OP_ICP_NOT_BUSY(ctxs[i], "op_icp_run_many");
        for (int j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return fail(OP_ERR_INVALID, "op_icp_run_many: context %d given twice", i);
    }
    int rc = OP_OK;
    char first_err[sizeof(op::g_last_error)] = {0};
    auto note = [&](int r) { if (r != OP_OK && rc == OP_OK) { rc = r; std::snprintf(first_err, sizeof(first_err), "%s", op::g_last_error); } };
    std::vector<IcpLoop> loops((size_t)k);
    std::vector<int> live; // indices of fp64-mode contexts whose loop is running here
    std::vector<int> threaded;
    std::vector<int> strict; // reference-order contexts
    for (int i = 0; i < k; ++i) {
        op_icp* c = ctxs[i];
        const float* T0 = init_T ? init_T + 16 * (size_t)i : kIdentity;
        if (c->sums == OP_ICP_SUMS_REFERENCE_F32) { // own submitter thread; the sequential sums of all of them in one launch per round (SeqBatch)
            if (mode == OP_ICP_POINT_TO_PLANE && hipSetDevice(c->device) == hipSuccess && seq_device_ok(c)) {
                if (!c->seq_ev && op::cached_event(&c->seq_ev) != hipSuccess) { c->seq_ev = nullptr; (void)hipGetLastError(); }
                IcpSeqBatch* b = icp_seq_batch(c->device);
                if (c->seq_ev && b && b->usable(c->device)) { c->seq_batch = b; b->join(); }
            }
            strict.push_back(i); // (its submitter thread starts below, once every participant of the batch is counted)
            continue;
        }
        int r = icp_run_head(c, mode, max_iteration);
        if (r == OP_OK) r = icp_loop_begin(loops[(size_t)i], c, mode, T0, max_iteration, nullptr, nullptr);
        if (r == OP_OK) live.push_back(i); else note(r);
    }
    for (int i : strict) {
        op_icp* c = ctxs[i];
        const float* T0 = init_T ? init_T + 16 * (size_t)i : kIdentity;
        std::array<float, 16> T0a;
        std::memcpy(T0a.data(), T0, sizeof(float) * 16);
        op_icp_result* res_i = &results[i];
        c->worker_active = true; c->worker_rc = OP_OK; c->worker_err[0] = 0;
        try {
            c->worker = std::thread([=] {
                c->worker_rc = icp_run_impl(c, mode, T0a.data(), max_iteration, res_i, nullptr, 0, nullptr, nullptr);
                if (c->worker_rc != OP_OK) std::snprintf(c->worker_err, sizeof(c->worker_err), "%s", op::g_last_error);
                if (c->seq_batch) c->seq_batch->leave(); // (on every exit: nobody may go on waiting for this context)
            });
            threaded.push_back(i);
        } catch (const std::exception& e) {
            c->worker_active = false;
            if (c->seq_batch) { c->seq_batch->leave(); c->seq_batch = nullptr; }
            note(fail(OP_ERR_INVALID, "op_icp_run_many: could not start a submitter thread: %s", e.what()));
        }
    }
    std::vector<int> finishing = live; // (a context whose loop fails drops out below)
    if (max_iteration > 0) {
        std::vector<int> active;
        for (int i : live) { // first iteration of every context
            if (ctxs[i]->device != ctxs[live[0]]->device) (void)hipSetDevice(ctxs[i]->device);
            const int r = icp_loop_launch(loops[(size_t)i]);
            if (r == OP_OK) active.push_back(i); else { note(r); finishing.erase(std::find(finishing.begin(), finishing.end(), i)); }
        }
        while (!active.empty()) {
            for (size_t a = 0; a < active.size();) {
                const int i = active[a];
                IcpLoop& L = loops[(size_t)i];
                (void)hipSetDevice(L.c->device);
                int r = icp_loop_complete(L);
                if (r == OP_OK && L.it < max_iteration) r = icp_loop_launch(L);
                if (r != OP_OK) { note(r); finishing.erase(std::find(finishing.begin(), finishing.end(), i)); }
                if (r != OP_OK || L.it >= max_iteration) active.erase(active.begin() + (long)a); else ++a;
            }
        }
    }
    // the finishes side by side (each is mostly one host core summing rows): helper threads for all but the first
    {
        std::vector<std::thread> helpers;
        std::vector<int> frc(finishing.size(), OP_OK);
        std::vector<std::string> ferr(finishing.size());
        auto fin = [&](size_t q) {
            const int i = finishing[q];
            IcpLoop& L = loops[(size_t)i];
            frc[q] = icp_run_finish(L.c, L.cur, L.last_search_T, max_iteration, &results[i], nullptr, 0);
            if (frc[q] != OP_OK) ferr[q] = op::g_last_error; // (thread-local: hand it over)
        };
        for (size_t q = 1; q < finishing.size(); ++q) {
            try { helpers.emplace_back(fin, q); } catch (const std::exception&) { fin(q); } // no thread to be had: do it here
        }
        if (!finishing.empty()) fin(0);
        for (auto& t : helpers) t.join();
        for (size_t q = 0; q < finishing.size(); ++q)
            if (frc[q] != OP_OK && rc == OP_OK) { rc = frc[q]; std::snprintf(first_err, sizeof(first_err), "%s", ferr[q].c_str()); }
    }
    for (int i : threaded) { note(op_icp_wait(ctxs[i])); ctxs[i]->seq_batch = nullptr; }
    if (rc != OP_OK) return fail(rc, "%s", first_err);
    return OP_OK;
}

int op_icp_run_enqueue(op_icp* c, int mode, const float init_T[16], int max_iteration, op_icp_result* result, int32_t* pairs, size_t pairs_cap) {
    if (!c || !init_T || !result) return fail(OP_ERR_INVALID, "null argument");
    if (c->worker_active) return fail(OP_ERR_INVALID, "op_icp_run_enqueue: an enqueued run has not been waited for");
    std::array<float, 16> T0;
    std::memcpy(T0.data(), init_T, sizeof(float) * 16);
    c->worker_active = true; c->worker_rc = OP_OK; c->worker_err[0] = 0;
    try {
        c->worker = std::thread([=] {
            c->worker_rc = icp_run_impl(c, mode, T0.data(), max_iteration, result, pairs, pairs_cap, nullptr, nullptr);
            if (c->worker_rc != OP_OK) std::snprintf(c->worker_err, sizeof(c->worker_err), "%s", op::g_last_error); // (the error text is thread-local: hand it over)
        });
    } catch (const std::exception& e) { // std::system_error (no thread to be had) must not cross the extern "C" boundary
        c->worker_active = false;
        return fail(OP_ERR_INVALID, "op_icp_run_enqueue: could not start the submitter thread: %s", e.what());
    }
    return OP_OK;
}

int op_icp_wait(op_icp* c) {
    if (!c) return fail(OP_ERR_INVALID, "null argument");
    if (!c->worker_active) return fail(OP_ERR_INVALID, "op_icp_wait: nothing has been enqueued");
    c->worker.join();
    c->worker_active = false;
    if (c->worker_rc != OP_OK) return fail(c->worker_rc, "%s", c->worker_err);
    return OP_OK;
}

int op_icp_register(int mode, const float* src_xyz, size_t n, const float* tgt_xyz, const float* tgt_normals, size_t m, const float init_T[16],
                    int max_iteration, double threshold, int device, op_icp_result* result, int32_t* pairs, size_t pairs_cap) {
    if (mode == OP_ICP_POINT_TO_PLANE && !tgt_normals)
        return fail(OP_ERR_NO_NORMALS, "[ERROR]::[ICPPointToPlane]::target point cloud need to have normals.");
    op_icp* c = nullptr;
    OP_TRY(op_icp_create(tgt_xyz, tgt_normals, m, threshold, OP_MEM_HOST, device, &c));
    int rc = op_icp_set_source(c, src_xyz, n, OP_MEM_HOST);
    if (rc == OP_OK) rc = op_icp_run(c, mode, init_T, max_iteration, result, pairs, pairs_cap, nullptr, nullptr);
    op_icp_destroy(c);
    return rc;
}


// The two estimators of the registration module as stand-alone calls over caller-supplied correspondences.
static int pair_sums_run(int mode, const float* a, size_t na_floats, const float* b, size_t nb_floats, const float* nrm, const int32_t* inliers,
                         size_t n, int mem, int device, double out[kNSums]) {
    OP_TRY(op::use_device(device));
    const int n_wg = 256;
    float *d_a = nullptr, *d_b = nullptr, *d_n = nullptr;
    int* d_i = nullptr;
    double *d_part = nullptr, *d_out = nullptr;
    hipError_t e = hipSuccess;
    auto up = [&](const void* src, size_t bytes, void** dst) {
        if (e != hipSuccess || !src || !bytes) return;
        if (mem == OP_MEM_DEVICE) { *dst = const_cast<void*>(src); return; }
        e = op::cached_malloc(dst, bytes);
        if (e == hipSuccess) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    up(a, na_floats * 4, (void**)&d_a); up(b, nb_floats * 4, (void**)&d_b); up(nrm, nb_floats * 4, (void**)&d_n); up(inliers, n * 8, (void**)&d_i);
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_part, (size_t)n_wg * kNSums * sizeof(double));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_out, kNSums * sizeof(double));
    if (e == hipSuccess) {
        if (mode == 1) hipLaunchKernelGGL(k_pair_sums<1>, dim3(n_wg), dim3(kIterThreads), 0, nullptr, (const float*)d_a, (const float*)d_b, (const float*)d_n, (const int*)d_i, n, d_part);
        else hipLaunchKernelGGL(k_pair_sums<0>, dim3(n_wg), dim3(kIterThreads), 0, nullptr, (const float*)d_a, (const float*)nullptr, (const float*)nullptr, (const int*)nullptr, n, d_part);
        hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(1024), 0, nullptr, (const double*)d_part, n_wg, d_out);
        e = hipMemcpy(out, d_out, kNSums * sizeof(double), hipMemcpyDeviceToHost);
    }
    if (mem != OP_MEM_DEVICE) { op::cached_free(d_a); op::cached_free(d_b); op::cached_free(d_n); op::cached_free(d_i); }
    op::cached_free(d_part); op::cached_free(d_out);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "pair sums failed: %s", hipGetErrorString(e));
    return OP_OK;
}

// host copy of a caller array (the reference-order sums run on one host thread)
static int host_view(const void* p, size_t bytes, int mem, std::vector<unsigned char>& keep, const void** out) {
    *out = p;
    if (mem != OP_MEM_DEVICE || !bytes) return OP_OK;
    keep.resize(bytes);
    OP_HIP(hipMemcpy(keep.data(), p, bytes, hipMemcpyDeviceToHost));
    *out = keep.data();
    return OP_OK;
}

int op_estimate_rigid_point_to_plane_ex(const float* source_xyz, size_t n_source, const float* target_xyz, const float* target_normals,
                                        size_t n_target, const int32_t* inliers, size_t n_inliers, int mem, int device, int sums, float T[16]) {
    if (!T || (n_inliers && (!source_xyz || !target_xyz || !target_normals || !inliers))) return fail(OP_ERR_INVALID, "null argument");
    double JTJ[36], JTr[6];
    float x[6];
    if (sums == OP_ICP_SUMS_REFERENCE_F32) { // ICP.cpp:121-136 as written: one thread, float32, inlier order
        OP_TRY(op::use_device(device));
        std::vector<unsigned char> k0, k1, k2, k3;
        const void *hs, *ht, *hn, *hi;
        OP_TRY(host_view(source_xyz, n_source * 12, mem, k0, &hs)); OP_TRY(host_view(target_xyz, n_target * 12, mem, k1, &ht));
        OP_TRY(host_view(target_normals, n_target * 12, mem, k2, &hn)); OP_TRY(host_view(inliers, n_inliers * 8, mem, k3, &hi));
        std::vector<float> rows(n_inliers * 9);
        const float *S = (const float*)hs, *Tg = (const float*)ht, *N = (const float*)hn;
        const int32_t* I = (const int32_t*)hi;
        for (size_t i = 0; i < n_inliers; ++i) {
            const size_t a = (size_t)I[2 * i], b = (size_t)I[2 * i + 1];
            if (a >= n_source || b >= n_target) return fail(OP_ERR_INVALID, "inlier %zu out of range", i);
            for (int k = 0; k < 3; ++k) { rows[9 * i + k] = S[3 * a + k]; rows[9 * i + 3 + k] = Tg[3 * b + k]; rows[9 * i + 6 + k] = N[3 * b + k]; }
        }
        op_host::plane_sums_reference_order(rows.data(), n_inliers, JTJ, JTr);
        op_host::solve6_psd<true>(JTJ, JTr, x);
    } else {
        double r[kNSums] = {0};
        if (n_inliers) OP_TRY(pair_sums_run(1, source_xyz, n_source * 3, target_xyz, n_target * 3, target_normals, inliers, n_inliers, mem, device, r));
        expand_plane_sums(r, JTJ, JTr);
        op_host::solve6_psd(JTJ, JTr, x);  // ICP.cpp:137-138
    }
    op_host::se3_exp(x, T);            // ICP.cpp:143
    return OP_OK;
}

int op_estimate_rigid_point_to_plane(const float* source_xyz, size_t n_source, const float* target_xyz, const float* target_normals, size_t n_target,
                                     const int32_t* inliers, size_t n_inliers, int mem, int device, float T[16]) {
    return op_estimate_rigid_point_to_plane_ex(source_xyz, n_source, target_xyz, target_normals, n_target, inliers, n_inliers, mem, device,
                                               OP_ICP_SUMS_FP64, T);
}

int op_estimate_rigid_transformation_ex(const float* pairs_xyz6, size_t n_pairs, int mem, int device, int finish, float T[16]) {
    if (!T || (n_pairs && !pairs_xyz6)) return fail(OP_ERR_INVALID, "null argument");
    if (finish == OP_ICP_FINISH_REFERENCE) { // Geometry.cpp:117-133 as written: one thread, float32, pair order
        OP_TRY(op::use_device(device));
        std::vector<unsigned char> keep;
        const void* hp;
        OP_TRY(host_view(pairs_xyz6, n_pairs * 24, mem, keep, &hp));
        op_host::kabsch_reference_order((const float*)hp, n_pairs, T);
        return OP_OK;
    }
    double r[kNSums] = {0};
    if (n_pairs) OP_TRY(pair_sums_run(0, pairs_xyz6, n_pairs * 6, nullptr, 0, nullptr, nullptr, n_pairs, mem, device, r));
    op_host::kabsch_from_sums(r[28], r, r + 3, r + 6, T); // Geometry.cpp:107-151
    return OP_OK;
}

int op_estimate_rigid_transformation(const float* pairs_xyz6, size_t n_pairs, int mem, int device, float T[16]) {
    return op_estimate_rigid_transformation_ex(pairs_xyz6, n_pairs, mem, device, OP_ICP_FINISH_REFERENCE, T);
}

static int points_from_images(const op_camera* cam, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, int device, float* xyz_out,
                              float* colors_out, size_t* n) {
    if (!cam || !depth || !xyz_out || !n || ((rgb == nullptr) != (colors_out == nullptr))) return fail(OP_ERR_INVALID, "null argument");
    if (cam->width <= 0 || cam->height <= 0) return fail(OP_ERR_INVALID, "invalid camera");
    OP_TRY(op::use_device(device));
    const size_t npix = (size_t)cam->width * cam->height;
    const size_t dbytes = npix * (depth_fmt == OP_DEPTH_U16 ? 2 : 4);
    void* d_depth = nullptr;
    unsigned *d_count = nullptr, *d_start = nullptr;
    float *d_xyz = nullptr, *d_col = nullptr;
    unsigned char* d_rgb = nullptr;
    const unsigned char* rsrc = rgb;
    int rc = OP_OK;
    hipError_t e = hipSuccess;
    const void* dsrc = depth;
    if (mem == OP_MEM_HOST) {
        e = op::cached_malloc(&d_depth, dbytes);
        if (e == hipSuccess) e = hipMemcpy(d_depth, depth, dbytes, hipMemcpyHostToDevice);
        dsrc = d_depth;
    }
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_count, npix * 4);
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_start, npix * 4);
    if (e == hipSuccess && mem == OP_MEM_HOST) e = op::cached_malloc((void**)&d_xyz, npix * 12);
    if (e == hipSuccess && mem == OP_MEM_HOST && rgb) {
        e = op::cached_malloc((void**)&d_col, npix * 12);
        if (e == hipSuccess) e = op::cached_malloc((void**)&d_rgb, npix * 3);
        if (e == hipSuccess) e = hipMemcpy(d_rgb, rgb, npix * 3, hipMemcpyHostToDevice);
        rsrc = d_rgb;
    }
    unsigned total = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_depth_count, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, nullptr, dsrc, depth_fmt == OP_DEPTH_U16,
                           cam->depth_scale, npix, d_count);
        rc = device_exclusive_scan(d_count, npix, d_start, nullptr, &total);
        if (rc == OP_OK) {
            float* dst = mem == OP_MEM_HOST ? d_xyz : xyz_out;
            float* cdst = rgb ? (mem == OP_MEM_HOST ? d_col : colors_out) : nullptr;
            hipLaunchKernelGGL(k_depth_scatter, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, nullptr, dsrc, depth_fmt == OP_DEPTH_U16, *cam,
                               npix, (const unsigned*)d_start, dst, rsrc, cdst);
            e = hipDeviceSynchronize();
            if (e == hipSuccess && mem == OP_MEM_HOST && total) e = hipMemcpy(xyz_out, d_xyz, (size_t)total * 12, hipMemcpyDeviceToHost);
            if (e == hipSuccess && mem == OP_MEM_HOST && total && rgb) e = hipMemcpy(colors_out, d_col, (size_t)total * 12, hipMemcpyDeviceToHost);
        }
    }
    if (d_depth) op::cached_free(d_depth);
    if (d_count) op::cached_free(d_count);
    if (d_start) op::cached_free(d_start);
    if (d_xyz) op::cached_free(d_xyz);
    if (d_col) op::cached_free(d_col);
    if (d_rgb) op::cached_free(d_rgb);
    if (rc != OP_OK) return rc;
    if (e != hipSuccess) return fail(OP_ERR_HIP, "points_from_depth failed: %s", hipGetErrorString(e));
    *n = total;
    return OP_OK;
}

int op_points_from_depth(const op_camera* cam, const void* depth, int depth_fmt, int mem, int device, float* xyz_out, size_t* n) {
    return points_from_images(cam, depth, depth_fmt, nullptr, mem, device, xyz_out, nullptr, n);
}

int op_points_from_rgbd(const op_camera* cam, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, int device, float* xyz_out,
                        float* colors_out, size_t* n) {
    if (!rgb || !colors_out) return fail(OP_ERR_INVALID, "null argument");
    return points_from_images(cam, depth, depth_fmt, rgb, mem, device, xyz_out, colors_out, n);
}

int op_estimate_normals(const float* xyz, size_t n, float radius, int knn, int mem, int device, float* normals_out) {
    if (!xyz || !normals_out) return fail(OP_ERR_INVALID, "null argument");
    if (knn < 1 || knn > kNrmMaxK) return fail(OP_ERR_INVALID, "knn must be in [1, %d]", kNrmMaxK);
    if (n == 0) return OP_OK;
    // grid cell = extent / 300: a 640x480 depth cloud of a 6 m room (4-10 mm spacing) gets 2 cm cells, and most points
    // find their 30 neighbours within the first ring (27 cells); measured 1.35 / 1.03 / 1.12 / 1.12 / 1.46 ms for
    // divisors 400 / 300 / 250 / 200 / 150 (tools/ab_normals_cell.sh)
    op_icp* c = nullptr;
    OP_TRY(icp_create(xyz, nullptr, n, 0.0, 300.0, mem, device, &c));
    float* d_nrm = nullptr;
    hipError_t e = op::cached_malloc((void**)&d_nrm, n * 12);
    if (e == hipSuccess) e = hipMemsetAsync(d_nrm, 0, n * 12, c->stream);
    if (e == hipSuccess) {
        const float cell = 1.0f / c->grid.inv_cell;
        hipLaunchKernelGGL(k_estimate_normals, dim3((unsigned)((n + kNrmThreads - 1) / kNrmThreads)), dim3(kNrmThreads), 0, c->stream, c->grid,
                           (const unsigned*)c->cell_start, (const float4*)c->tgt, n, knn, radius, cell, d_nrm);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess && mem == OP_MEM_DEVICE) { // ordered on the stream and finished before d_nrm goes back to the buffer cache
        e = hipMemcpyAsync(normals_out, d_nrm, n * 12, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    } else if (e == hipSuccess) {
        e = hipMemcpy(normals_out, d_nrm, n * 12, hipMemcpyDeviceToHost);
    }
    if (d_nrm) op::cached_free(d_nrm);
    op_icp_destroy(c);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "estimate_normals failed: %s", hipGetErrorString(e));
    return OP_OK;
}

} // extern "C"
