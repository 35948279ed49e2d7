// icp_iter.hip -- ONE pass of a registration (icp_core.hpp lists the translation units): k_icp_iter<MODE, DETECT> fuses TransformPoints + the exact 1-NN over
// the cell grid + CountInliers + the normal-equation / Kabsch sums and finishes their reduction itself; its launch and the host's wait for the published rows;
// the re-decision of exactly equidistant candidates (OP_ICP_TIES_REFERENCE) and of the final count's doubtful correspondences in the tree nanoflann would build.
#include "icp_core.hpp"

namespace {

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

// nn[source] = target for the queries the host re-decided
__global__ __launch_bounds__(256) void k_patch_nn(const int2* __restrict__ patch, unsigned n, int* __restrict__ nn) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k < n) nn[patch[k].x] = patch[k].y;
}

} // namespace

namespace opi {

// one fused pass (transform + NN + inliers + sums + reduction); start_T is read from c->T_dev unless host_T is given.
template <int MODE, bool DETECT = false>
static void launch_pass_t(op_icp* c, bool write_inl, const float* host_T = nullptr, double seq = 0.0, FinalAux* final_aux = nullptr) {
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
    if (mode == 1) { if (detect) launch_pass_t<1, true>(c, write_inl); else launch_pass_t<1>(c, write_inl); }
    else if (mode == 0) { if (detect) launch_pass_t<0, true>(c, write_inl); else launch_pass_t<0>(c, write_inl); }
    else launch_pass_t<2>(c, write_inl);
    OP_HIP(hipGetLastError());
    return OP_OK;
}

// host-synchronous single pass with an explicit T (op_icp_iterate)
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
    if (mode == 1) launch_pass_t<4>(c, write_inl); else launch_pass_t<3>(c, write_inl);
    OP_HIP(hipGetLastError());
    OP_HIP(hipMemcpyAsync(out, c->result, kNSums * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OP_HIP(hipStreamSynchronize(c->stream));
    return OP_OK;
}

// run-time dispatch onto the instantiations of k_icp_iter (the loop and the finish in icp.hip)
void launch_pass(op_icp* c, int kmode, bool detect, bool write_inl, const float* host_T, double seq, FinalAux* final_aux) {
    switch (kmode) {
        case 0: if (detect) launch_pass_t<0, true>(c, write_inl, host_T, seq, final_aux); else launch_pass_t<0>(c, write_inl, host_T, seq, final_aux); break;
        case 1: if (detect) launch_pass_t<1, true>(c, write_inl, host_T, seq, final_aux); else launch_pass_t<1>(c, write_inl, host_T, seq, final_aux); break;
        case 2: launch_pass_t<2>(c, write_inl, host_T, seq, final_aux); break;
        case 3: launch_pass_t<3>(c, write_inl, host_T, seq, final_aux); break;
        default: launch_pass_t<4>(c, write_inl, host_T, seq, final_aux); break;
    }
}

void icp_trace_dump(op_icp* c) {
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
#else
    (void)c;
#endif
}

} // namespace opi
