// icp.hip -- a registration (icp_core.hpp lists the translation units): the iteration loop in both summation modes, the ordered inlier rows and the sequential
// float32 sums behind the reference-order modes, the finish (RegistrationResult), op_icp_run / _many / _enqueue / _register and the stand-alone estimators.
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
#include "icp_core.hpp"
#include "seq_sums.hpp"

namespace {

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

// The reference-order contexts of one op_icp_run_many call take their sequential sums TOGETHER when there are nine or more of them (seq_sums.hpp: SeqRendezvous): every such context has a submitter
// thread (its iterations synchronise the stream anyway) and each iteration ends in k_seq_sums -- ONE workgroup, ~1.4 ms for 3e5 rows; K independent runs scale to
// the number of hardware queues of the process (GPU_MAX_HW_QUEUES) and no further, K workgroups of one launch do not have that limit.  One rendezvous per device.
using IcpSeqBatch = SeqRendezvous<42, 7, 1, 9>;
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
    const size_t n = c->n;
    const unsigned g256 = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_inl_flag, dim3(g256), dim3(256), 0, c->stream, (const int*)c->inl, n, c->flag);
    scan_launch((const unsigned*)c->flag, n, c->scan_tot, c->start, c->stream);
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

extern "C" {

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
    if (L.detect) { if (L.pass_mode == 1) launch_pass(c, 1, true, false, L.cur, c->seq); else launch_pass(c, 0, true, false, L.cur, c->seq); }
    else if (L.pass_mode == 1) launch_pass(c, 1, false, false, L.cur, c->seq);
    else launch_pass(c, 0, false, false, L.cur, c->seq);
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
                // leave HBM, 42 numbers come back -- ~1.4 ms for 3e5 inliers (10 shader cycles per row) instead of an 11 MB transfer and a pass on one host core
                OP_TRY(emit_rows(c, 3, n_it, nullptr));
                const unsigned n_rows_u = (unsigned)n_it;
                OP_HIP(hipMemcpyAsync(c->seq_total, &n_rows_u, sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
                hipError_t eb = hipErrorNotReady;
                if (c->seq_batch) // with the other contexts of the op_icp_run_many call: one launch, a workgroup each (hipErrorNotReady: too few of them -- alone, below)
                    eb = static_cast<IcpSeqBatch*>(c->seq_batch)->submit(c->rows_dev, c->seq_total, c->seq_out, c->seq_host, c->seq_ev, c->stream);
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
            if (c->seq_batch) static_cast<IcpSeqBatch*>(c->seq_batch)->pass(); // nothing for the batched launch from this context in this iteration
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
    OP_HIP(hipSetDevice(c->device)); // (op_icp_run_many finishes on helper threads, whose current device is the process default)
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
    launch_pass(c, 2, false, true, start_T, c->seq, aux);
    OP_HIP(hipGetLastError());
    OP_TRY(wait_rows(c, r));
    if (r[30] > 0.5) { // correspondences the 27-cell search cannot vouch for under the pose they are now measured with (FinalAux): re-decided on the host, pass repeated
        OP_TRY(redecide_final(c, last_search_T, (size_t)(r[30] + 0.5)));
        c->seq += 1.0;
        launch_pass(c, 2, false, true, start_T, c->seq);
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
                if (c->seq_batch) static_cast<IcpSeqBatch*>(c->seq_batch)->leave(); // (on every exit: nobody may go on waiting for this context)
            });
            threaded.push_back(i);
        } catch (const std::exception& e) {
            c->worker_active = false;
            if (c->seq_batch) { static_cast<IcpSeqBatch*>(c->seq_batch)->leave(); c->seq_batch = nullptr; }
            note(fail(OP_ERR_INVALID, "op_icp_run_many: could not start a submitter thread: %s", e.what()));
        }
    }
    // fp64-mode contexts: at most kInFlight of them are ACTIVE at a time, each with one iteration enqueued; the submitter waits for the oldest launch's sums, solves,
    // enqueues that context's next iteration.  One k_icp_iter launch already fills more than half of the chip (4 800 of 8 192 wave slots): two overlap, so
    // more iterations in flight only slow each other down.  A context whose loop is over hands its finish (final CountInliers + the reference-order Kabsch over
    // ~3e5 rows on a host core, ~1 ms) to a helper thread and the next waiting context takes its place: finishes overlap the other contexts' iterations.
    {
        const size_t kInFlight = (size_t)std::max(1, op::runtime_options().icp_many_in_flight.load()); // OP_RUNTIME_OPT_ICP_MANY_IN_FLIGHT (default 4)
        std::vector<std::thread> helpers;
        std::vector<int> frc((size_t)k, OP_OK);
        std::vector<std::string> ferr((size_t)k);
        auto fin = [&](int i) {
            IcpLoop& L = loops[(size_t)i];
            frc[(size_t)i] = icp_run_finish(L.c, L.cur, L.last_search_T, max_iteration, &results[i], nullptr, 0);
            if (frc[(size_t)i] != OP_OK) ferr[(size_t)i] = op::g_last_error; // (thread-local: hand it over)
        };
        auto start_finish = [&](int i) {
            try { helpers.emplace_back(fin, i); } catch (const std::exception&) { fin(i); } // no thread to be had: do it here
        };
        std::deque<int> waiting(live.begin(), live.end()), flying;
        while (!waiting.empty() || !flying.empty()) {
            while (flying.size() < kInFlight && !waiting.empty()) {
                const int i = waiting.front(); waiting.pop_front();
                if (max_iteration <= 0) { start_finish(i); continue; }
                (void)hipSetDevice(ctxs[i]->device);
                const int r = icp_loop_launch(loops[(size_t)i]);
                if (r == OP_OK) flying.push_back(i); else note(r);
            }
            if (flying.empty()) continue;
            const int i = flying.front(); flying.pop_front(); // the oldest launch: its sums arrive first
            IcpLoop& L = loops[(size_t)i];
            (void)hipSetDevice(L.c->device);
            int r = icp_loop_complete(L);
            if (r == OP_OK && L.it < max_iteration) { r = icp_loop_launch(L); if (r == OP_OK) flying.push_back(i); }
            else if (r == OP_OK) start_finish(i);
            if (r != OP_OK) note(r);
        }
        for (auto& t : helpers) t.join();
        for (int i = 0; i < k; ++i)
            if (frc[(size_t)i] != OP_OK && rc == OP_OK) { rc = frc[(size_t)i]; std::snprintf(first_err, sizeof(first_err), "%s", ferr[(size_t)i].c_str()); }
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

} // extern "C"
