// icp_core.hpp -- point-to-plane / point-to-point ICP for gfx950 (MI355X): what the translation units of the registration path share -- constants, the search
// grid, the records that cross between kernels and host (TieRec, FinalAux), small device helpers, the host object op_icp, the host functions that cross files.
//   icp_grid.hip   the context and its search structure: grid build (bbox -> cell counts -> scan -> scatter), op_icp_create / destroy / set_source / options;
//                  the other users of that grid and of the scan: EstimateNormals (k_estimate_normals), LoadFromDepth / LoadFromRGBD (k_depth_*)
//   icp_iter.hip   one pass: k_icp_iter<MODE, DETECT> (transform + 1-NN + CountInliers + sums + their reduction), its launch / wait, the re-decision of exactly
//                  equidistant candidates and of the final count's doubtful correspondences in the tree nanoflann would build
//   icp.hip        a registration: the iteration loop in both summation modes, ordered inlier rows + sequential float32 sums (k_emit_rows, k_seq_sums), the
//                  finish (RegistrationResult), op_icp_run / _many / _enqueue / _register, the stand-alone estimators (k_pair_sums)
// C-ABI entry points op_icp_* / op_points_from_* / op_estimate_* are declared in include/onepiece_hip.h.
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
#pragma once
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
#include <deque>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "host_math.hpp"
#include "nn_tree.hpp"

namespace opi {

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
constexpr int kScanWg = 1024; // the exclusive scan (icp_grid.hip): elements per workgroup (256 threads x 4)
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


} // namespace opi

using namespace opi;

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
    void* seq_batch = nullptr;          // icp.hip: the SeqRendezvous<42, 7, 1, 9> of the device (seq_sums.hpp), set for the duration of an op_icp_run_many call: this context's sequential sums are taken in one launch with the other contexts'
    hipEvent_t seq_ev = nullptr;        // "my ordered rows are in place" (recorded on the context's stream for the batch's stream to wait on)
    std::thread worker;
    bool worker_active = false;
    int worker_rc = OP_OK;
    char worker_err[512] = "";
};

// While a run enqueued with op_icp_run_enqueue is in flight its worker thread owns the context (nn, tie buffers, fin_aux, seq, the stream): every other
// entry point refuses instead of racing with it.
#define OP_ICP_NOT_BUSY(c, what) do { if ((c)->worker_active) return fail(OP_ERR_INVALID, what ": a run enqueued with op_icp_run_enqueue has not been waited for (op_icp_wait)"); } while (0)


namespace opi {
// icp_grid.hip
int device_exclusive_scan(const unsigned* d_count, size_t n, unsigned* d_start, hipStream_t stream, unsigned* total_out);
void scan_launch(const unsigned* d_count, size_t n, unsigned* d_tot, unsigned* d_start, hipStream_t stream); // the three scan kernels, totals in the caller's buffer ((n + 1023) / 1024 + 1 words)
// icp_iter.hip
// one fused pass (transform + NN + inliers + sums + reduction) of kernel mode 0 .. 4 (k_icp_iter's MODE); start_T is read from c->T_dev unless host_T is given
void launch_pass(op_icp* c, int kmode, bool detect, bool write_inl, const float* host_T = nullptr, double seq = 0.0, FinalAux* final_aux = nullptr);
int wait_rows(op_icp* c, double r[kNSums]);
int run_pass(op_icp* c, int mode, const float T[16], bool write_inl, double out[kNSums]);
int ensure_tie_buffers(op_icp* c);
int resolve_ties(op_icp* c, int mode, const float T[16], bool write_inl, double out[kNSums], bool launch_retired, bool nn_is_read = true);
int redecide_final(op_icp* c, const float T_old[16], size_t n_unsure);
void icp_trace_dump(op_icp* c); // (-DICP_TRACE development builds)
} // namespace opi
