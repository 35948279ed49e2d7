// odometry.hip -- dense RGB-D frame-to-frame tracker on gfx950.
//
// Replaces odometry::Odometry::MultiScaleComputing (Odometry/Odometry.cpp:621-687) and what it calls
// (Odometry/DenseOdometryFunction.cpp): per iteration
//     ComputeCorrespondencePixelWise (:72-128)   -> k_track_assoc + the chain walk in k_track_iter
//     ComputeJTJandJTr{Hybrid,Photo,Depth}Term   -> k_track_iter (fp64 wave reduce-scatter + LDS)
//     JTJ.ldlt().solve(-JTr), Se3ToSE3, T update -> k_track_solve (one workgroup, thread 0 solves)
// and afterwards the correspondence_set / rmse of DenseTracking (Odometry.cpp:676-683, :606) in the
// k_emit_* kernels.  The whole coarse-to-fine loop is enqueued without a host round trip: the pose,
// the per-level image descriptors and the early-out flag (ratio > MAX_INLIER_RATIO_DENSE) live in a
// device-side TrackState.
//
// The reference's "z-buffer" (AddElementToCorrespondenceMap, :9-27) reads wraping_depth at the TARGET
// pixel but writes it at the SOURCE pixel, in raster order.  With p(s) the target cell of source
// pixel s, td(s) its transformed depth and ok(s) the geometric gate, that sequential loop is
//     acc(s) = ok(s) && ( p(s) >= s  ||  !acc(p(s))  ||  td(p(s)) > td(s) )
// which only ever refers to a SMALLER raster index, so every thread resolves it independently by
// walking p(.) until a link is decided without recursion; acc(s) is the parity of the walk length.
//
// op_tracker_dense_tracking additionally runs Odometry::DenseTracking's image preparation (k_prep_*,
// NormalizeIntensity) on the device, and splits into _enqueue / op_tracker_wait so that several trackers
// (one HIP stream each) keep independent frame pairs in flight; one call's ~100 launches are captured
// into a hipGraph on first use and replayed afterwards.
#include "odometry_core.hpp"
#include "seq_sums.hpp"

using namespace op;
using namespace opt;

namespace {

// ---- association (DenseOdometryFunction.cpp:89-114) + the acceptance link of every pixel ---------
// ok(s), p(s), td(s) as the reference computes them; then, because acc(s) only needs acc(p(s)) when
// p(s) < s, ok(p(s)) and td(p(s)) <= td(s), the thread ALSO evaluates the association of pixel p(s)
// itself (no inter-thread dependency) and stores a 16-bit link code:
//   kInvalid   not ok(s)                     (never the target of a link)
//   kTerminal  accepted without recursion
//   kFar       depends on p(s) but s - p(s) does not fit 16 bits (consult pair_p)
//   d + 3      depends on pixel s - d        (acc(s) = !acc(s - d))
constexpr unsigned short kInvalid = 0, kTerminal = 1, kFar = 2;
constexpr int kLinkBias = 3;

struct Proj { float krk[9], kt[3]; };

__device__ __forceinline__ int2 associate(const LevelDev& L, const Proj& P, int s) {
    const int i = s / L.w, j = s - i * L.w;
    const float d_s = L.sd[s];
    int p = -1;
    float td = 0.0f;
    if (!isnan(d_s)) {
        const float fj = (float)j, fi = (float)i;
        // d_s * KRK_inv * Point3(j,i,1.0) + Kt: Eigen evaluates (d_s*KRK_inv) first, rows as a0+(a1+a2)
        const float uv0 = sum3((d_s * P.krk[0]) * fj, (d_s * P.krk[1]) * fi, (d_s * P.krk[2]) * 1.0f) + P.kt[0];
        const float uv1 = sum3((d_s * P.krk[3]) * fj, (d_s * P.krk[4]) * fi, (d_s * P.krk[5]) * 1.0f) + P.kt[1];
        td = sum3((d_s * P.krk[6]) * fj, (d_s * P.krk[7]) * fi, (d_s * P.krk[8]) * 1.0f) + P.kt[2];
        // (int)(x / z + 0.5): float quotient, double sum, truncation toward zero (so (-1,0) -> 0)
        const double ax = (double)(uv0 / td) + 0.5, ay = (double)(uv1 / td) + 0.5;
        if (ax > -1.0 && ax < (double)L.w && ay > -1.0 && ay < (double)L.h) { // NaN fails
            const int u_t = (int)ax, v_t = (int)ay;
            const float d_t = L.td[v_t * L.w + u_t];
            // |d_t - td| < MAX_DIFF_DEPTH (double 0.05): for floats that is < 0.05f exactly.  A stored
            // depth of exactly -1 is indistinguishable from "unset" in the reference -> never a pair.
            if (!isnan(d_t) && fabsf(d_t - td) < 0.05f && td != -1.0f) p = v_t * L.w + u_t;
        }
    }
    return make_int2(p, __float_as_int(td));
}

__global__ __launch_bounds__(kThreads) void k_track_assoc(const TrackState* __restrict__ st, int l, int* __restrict__ pair_p,
                                                          unsigned short* __restrict__ code) {
    if (st->stop_level == l) return;
    __shared__ Proj s_P;
    const LevelDev L = st->lv[l];
    if (threadIdx.x == 0) op_host::track_projection(L.fx, L.fy, L.cx, L.cy, st->T, s_P.krk, s_P.kt);
    __syncthreads();
    const Proj P = s_P;
    // XCD-aware order (common.hpp): an XCD works on one band of the image, so its L2 holds one band of the target images
    const int npix = L.w * L.h, s = (int)op::xcd_slab_index(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;
    if (s >= npix) return;
    const int2 c = associate(L, P, s);
    unsigned short cd = kInvalid;
    if (c.x >= 0) {
        cd = kTerminal;
        if (c.x < s) {
            const int2 c2 = associate(L, P, c.x);
            // wraping_depth(p) is set (!= -1) iff p was accepted; if it is, accept s only when it is nearer
            if (c2.x >= 0 && !(__int_as_float(c2.y) > __int_as_float(c.y)))
                cd = (s - c.x) + kLinkBias <= 0xffff ? (unsigned short)((s - c.x) + kLinkBias) : kFar;
        }
    }
    pair_p[s] = c.x;
    code[s] = cd;
}

// ---- acceptance (parity of the link chain) + Jacobian rows + normal-equation partials ----------
// Chains are 40-150 links long on real motion and run backwards in raster order, so each workgroup
// first stages the link codes of a window ending at its last own pixel in LDS (the whole level when
// it fits: 2 bytes per pixel, up to ~150 KB of the 160 KB LDS) and walks there (~50 ns per link
// instead of ~1 us per dependent L2/HBM gather).  Links that leave the window fall back to global.
constexpr int kIterThreads = 1024;

// Jacobian rows + residuals of one correspondence (source pixel s, target pixel t) at pose T (rows 0..2 of the 4x4):
// DenseOdometryFunction.cpp:146-296, float, in the reference's operation order.  TERM 0 hybrid (photometric row, then
// geometric row, each scaled by sqrt(0.5)), 1 photometric, 2 geometric.  Returns the row count.
template <int TERM>
__device__ __forceinline__ int track_rows_dev(const LevelDev& L, const float* T12, int s, int t, float J[2][6], float r[2]) {
    const int i = s / L.w, j = s - i * L.w;
    // source_XYZ[v_s][u_s] (Geometry.cpp:84-100)
    const float z = L.sd[s];
    float p0 = -1.0f, p1 = -1.0f, p2 = -1.0f;
    if (z > 0) { p0 = ((float)j - L.cx) * z / L.fx; p1 = ((float)i - L.cy) * z / L.fy; p2 = z; }
    const float q0 = sum3(T12[0] * p0, T12[1] * p1, T12[2] * p2) + T12[3];
    const float q1 = sum3(T12[4] * p0, T12[5] * p1, T12[6] * p2) + T12[7];
    const float q2 = sum3(T12[8] * p0, T12[9] * p1, T12[10] * p2) + T12[11];
    const float invz = (float)(1.0 / (double)q2);
    const float sq_img = (float)0.70710678118654757, sq_dep = (float)0.70710678118654757; // sqrt(1-0.5), sqrt(0.5)
    int rows = 0;
    if (TERM == 0 || TERM == 1) { // photometric row (:146-193 / :262-283)
        const float diff = L.tc[t] - L.sc[s];
        const float dIdx = 0.125f * L.tcdx[t], dIdy = 0.125f * L.tcdy[t]; // SOBEL_SCALE
        const float c0 = dIdx * L.fx * invz, c1 = dIdy * L.fy * invz;
        const float c2 = -(c0 * q0 + c1 * q1) * invz;
        const float jr[6] = {c0, c1, c2, -q2 * c1 + q1 * c2, q2 * c0 - q0 * c2, -q1 * c0 + q0 * c1};
#pragma unroll
        for (int k = 0; k < 6; ++k) J[rows][k] = TERM == 0 ? sq_img * jr[k] : jr[k];
        r[rows] = TERM == 0 ? sq_img * diff : diff;
        ++rows;
    }
    if (TERM == 0 || TERM == 2) { // geometric row (:194-241 / :284-294)
        float dDdx = 0.125f * L.tddx[t], dDdy = 0.125f * L.tddy[t];
        if (isnan(dDdx)) dDdx = 0.0f;
        if (isnan(dDdy)) dDdy = 0.0f;
        const float diff = L.td[t] - q2;
        const float d0 = dDdx * L.fx * invz, d1 = dDdy * L.fy * invz;
        const float d2 = -(d0 * q0 + d1 * q1) * invz;
        const float jr[6] = {d0, d1, d2 - 1.0f, (-q2 * d1 + q1 * d2) - q1, (q2 * d0 - q0 * d2) + q0, -q1 * d0 + q0 * d1};
#pragma unroll
        for (int k = 0; k < 6; ++k) J[rows][k] = TERM == 0 ? sq_dep * jr[k] : jr[k];
        r[rows] = TERM == 0 ? sq_dep * diff : diff;
        ++rows;
    }
    return rows;
}

// Validation mode (OP_TRACK_SUMS_REFERENCE_F32): the rows of every accepted pixel, written per source pixel
// ({J0[6], r0, J1[6], r1} = 14 floats) so that the host can sum them in raster order in float32, as the reference's
// sequential loop does (DenseOdometryFunction.cpp:297-381).
template <int TERM>
__global__ __launch_bounds__(kThreads) void k_track_rows(const TrackState* __restrict__ st, int l, const int* __restrict__ pair_t,
                                                         float* __restrict__ rows) {
    __shared__ float s_T[12];
    const LevelDev L = st->lv[l];
    if (threadIdx.x < 12) s_T[threadIdx.x] = st->T[threadIdx.x];
    __syncthreads();
    const int s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= L.w * L.h) return;
    const int t = pair_t[s];
    if (t < 0) return;
    float J[2][6], r[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        r[m] = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) J[m][k] = 0.0f;
    }
    track_rows_dev<TERM>(L, s_T, s, t, J, r);
    float* o = rows + (size_t)s * 14;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int k = 0; k < 6; ++k) o[7 * m + k] = J[m][k];
        o[7 * m + 6] = r[m];
    }
}

// ---- reference-order sums on the device (OP_TRACK_SUMS_REFERENCE_F32) ---------------------------------------------
// The reference adds every accepted pixel's J J^T and J r to float32 accumulators one after the other, in raster order
// (DenseOdometryFunction.cpp:297-381), and NormalizeIntensity does the same with two intensity sums (:131-141).  The rounding of
// those 10^5..10^6 sequential additions is part of the reference's result (it moves the pose by up to 2e-4), and a sequential float
// sum cannot be re-associated.  What CAN run side by side are the 36 + 6 accumulators: one wave owns them, a lane each, and walks the
// accepted rows in order doing nothing but `acc += product` -- one dependent v_add_f32 per row.  Everything else is taken off that
// chain: k_rows_count / k_emit_scan / k_track_rows_compact put the accepted pixels' rows into raster order without gaps, and inside
// k_seq_sums the other 15 waves of the workgroup turn the rows of the NEXT tile into the 42 products per row (transposed in LDS so
// that the summing wave reads four consecutive rows of its accumulator with one ds_read_b128) while the summing wave consumes the
// current tile.  Same operands in the same order as the host loop of op_host::track_sums_reference_order: bit-identical sums.
// A full-resolution iteration (~300 k rows) takes ~1 ms instead of a 17 MB transfer + 10 ms on one host core.
// the accepted pixels of a level per workgroup of kThreads pixels
__global__ __launch_bounds__(kThreads) void k_rows_count(const TrackState* __restrict__ st, int l, const int* __restrict__ pair_t, unsigned* __restrict__ wg_count) {
    __shared__ unsigned s_c[kThreads / 64];
    const int npix = st->lv[l].w * st->lv[l].h;
    const int s = blockIdx.x * kThreads + threadIdx.x;
    const bool a = s < npix && pair_t[s] >= 0;
    const unsigned long long m = __ballot(a);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) wg_count[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

// position of an accepted pixel among the accepted pixels of the level, in raster order (wg_off: exclusive scan of k_rows_count's counts)
__device__ __forceinline__ unsigned raster_rank(bool a, const unsigned* __restrict__ wg_off, unsigned* s_c) {
    const unsigned long long m = __ballot(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_c[wave] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned idx = wg_off[blockIdx.x] + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) idx += s_c[w];
    return idx;
}

// the Jacobian rows {J[6], r} of every accepted pixel -- two for the hybrid term, one otherwise -- compacted in raster order
// (k_track_rows writes them per source pixel, with gaps)
template <int TERM>
__global__ __launch_bounds__(kThreads) void k_track_rows_compact(const TrackState* __restrict__ st, int l, const int* __restrict__ pair_t,
                                                                 const unsigned* __restrict__ wg_off, float* __restrict__ rows) {
    __shared__ float s_T[12];
    __shared__ unsigned s_c[kThreads / 64];
    const LevelDev L = st->lv[l];
    if (threadIdx.x < 12) s_T[threadIdx.x] = st->T[threadIdx.x];
    const int s = blockIdx.x * kThreads + threadIdx.x;
    const int t = s < L.w * L.h ? pair_t[s] : -1;
    const unsigned idx = raster_rank(t >= 0, wg_off, s_c); // (its barrier also publishes s_T)
    if (t < 0) return;
    float J[2][6], r[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        r[m] = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) J[m][k] = 0.0f;
    }
    track_rows_dev<TERM>(L, s_T, s, t, J, r);
    constexpr int kRows = TERM == 0 ? 2 : 1;
    float* o = rows + (size_t)idx * (7 * kRows);
#pragma unroll
    for (int m = 0; m < kRows; ++m) {
#pragma unroll
        for (int k = 0; k < 6; ++k) o[7 * m + k] = J[m][k];
        o[7 * m + 6] = r[m];
    }
}

// NormalizeIntensity's operands (DenseOdometryFunction.cpp:131-141): {source intensity at s, target intensity at p(s)} of every accepted pixel
__global__ __launch_bounds__(kThreads) void k_norm_pairs_compact(const float* __restrict__ gs, const float* __restrict__ gt, int npix, const int* __restrict__ pair_t,
                                                                 const unsigned* __restrict__ wg_off, float* __restrict__ out2) {
    __shared__ unsigned s_c[kThreads / 64];
    const int s = blockIdx.x * kThreads + threadIdx.x;
    const int t = s < npix ? pair_t[s] : -1;
    const unsigned idx = raster_rank(t >= 0, wg_off, s_c);
    if (t < 0) return;
    out2[2 * (size_t)idx] = gs[s];
    out2[2 * (size_t)idx + 1] = gt[t];
}

// The bookkeeping of k_track_solve for a pose computed on the host (validation mode).
struct PoseArg { float m[16]; };
__global__ void k_track_apply(TrackState* __restrict__ st, int l, int it, PoseArg T, unsigned long long n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < 16; ++i) st->T[i] = T.m[i];
    st->n_last = n;
    st->last_level = l;
    st->iters_done = st->iters_done + 1;
    st->per_iter_count[it] = (int)n;
    for (int i = 0; i < 16; ++i) st->per_iter_T[16 * it + i] = T.m[i];
    if ((double)((float)n / (float)(st->full_h * st->full_w)) > 0.9) st->stop_level = l;
}

template <int TERM>
__global__ __launch_bounds__(kIterThreads) void k_track_iter(const TrackState* __restrict__ st, int l, int win_cap,
                                                             const int* __restrict__ pair_p, const unsigned short* __restrict__ code,
                                                             int* __restrict__ pair_t, double* __restrict__ partials) {
    if (st->stop_level == l) return;
    extern __shared__ unsigned short s_code[];
    __shared__ double s_red[kIterThreads / 64][kNSums];
    __shared__ float s_T[12];
    const LevelDev L = st->lv[l];
    if (threadIdx.x < 12) s_T[threadIdx.x] = st->T[threadIdx.x];
    const int npix = L.w * L.h;
    const int wg = (int)op::xcd_slab_index(blockIdx.x, gridDim.x);  // XCD-aware order, as in k_track_assoc
    const int own0 = wg * kIterThreads, own1 = min(own0 + kIterThreads, npix); // one own pixel per thread
    const int win0 = max(0, own1 - win_cap) & ~7;              // 16-byte aligned window start
    {   // stage the window, 8 codes (16 B) per load; links that point before the window become kFar
        const int n8 = (own1 - win0 + 7) >> 3;
        const uint4* src = reinterpret_cast<const uint4*>(code + win0);
        uint4* dst = reinterpret_cast<uint4*>(s_code);
        for (int k = threadIdx.x; k < n8; k += kIterThreads) {
            uint4 v = src[k];
            if (win0 > 0 && k < 8192) {                        // only the first 65535 entries can point outside
                unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned lo = w[q] & 0xffffu, hi = w[q] >> 16;
                    const int e = k * 8 + q * 2;
                    if ((int)lo - kLinkBias > e) lo = kFar;
                    if ((int)hi - kLinkBias > e + 1) hi = kFar;
                    w[q] = lo | (hi << 16);
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            dst[k] = v;
        }
    }
    __syncthreads();

    const int s = own0 + threadIdx.x;
    bool accepted = false;
    if (s < own1) {
        int cur = s - win0, parity = 0;
        unsigned cd = s_code[cur];
        if (cd != kInvalid) {
            while (cd > kFar) {                                // acc(cur) = !acc(cur - d): LDS-only loop
                cur -= (int)cd - kLinkBias;
                parity ^= 1;
                cd = s_code[cur];
            }
            if (cd == kFar) {                                  // the chain leaves the window: finish in global memory
                int g = cur + win0;
                unsigned c = kFar;
                while (c != kTerminal) {
                    g = c == kFar ? pair_p[g] : g - ((int)c - kLinkBias);
                    parity ^= 1;
                    c = code[g];
                }
            }
            accepted = parity == 0;
        }
        pair_t[s] = accepted ? pair_p[s] : -1;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double total;
    if (TERM == 3) { // NormalizeIntensity sums (DenseOdometryFunction.cpp:129-139)
        double acc[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = 0.0;
        if (accepted) {
            acc[0] = (double)L.sc[s];
            acc[1] = (double)L.tc[pair_p[s]];
            acc[28] = 1.0;
        }
        wave_reduce_scatter32(acc);
        total = acc[0];
    } else {
        // the pixel's Jacobian rows (zero when it is not accepted); the 29 sums are formed from them inside the wave
        // reduction, so that 16 instead of 32 fp64 values are live: <= 64 VGPRs, i.e. two 1024-thread workgroups per CU,
        // which is what lets the 300 workgroups of a 640x480 level run in ONE round on 256 CUs
        float J[2][6], r[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            r[m] = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) J[m][k] = 0.0f;
        }
        if (accepted) track_rows_dev<TERM>(L, s_T, s, pair_p[s], J, r);
        // sum k of the pixel: float products as the reference forms them, summed in double ([0..20] upper triangle of
        // J^T J row by row, [21..26] J^T r, [27] r^2, [28] count).  A row that does not exist is all zero.
        auto val = [&](auto kc) -> double {
            constexpr int k = decltype(kc)::value;
            constexpr int NR = TERM == 0 ? 2 : 1;
            double v = 0.0;
            if constexpr (k < 21) {
                constexpr int a = k < 6 ? 0 : k < 11 ? 1 : k < 15 ? 2 : k < 18 ? 3 : k < 20 ? 4 : 5;
                constexpr int first = a == 0 ? 0 : a == 1 ? 6 : a == 2 ? 11 : a == 3 ? 15 : a == 4 ? 18 : 20;
                constexpr int b = a + (k - first);
#pragma unroll
                for (int m = 0; m < NR; ++m) v += (double)(J[m][a] * J[m][b]);
            } else if constexpr (k < 27) {
#pragma unroll
                for (int m = 0; m < NR; ++m) v += (double)(J[m][k - 21] * r[m]);
            } else if constexpr (k == 27) {
#pragma unroll
                for (int m = 0; m < NR; ++m) v += (double)(r[m] * r[m]);
            } else if constexpr (k == 28) {
                v = accepted ? 1.0 : 0.0;
            }
            return v;
        };
        total = wave_reduce_scatter32_lazy(val);
    }
    if ((lane & 1) == 0) s_red[wave][lane >> 1] = total;
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double v = 0;
        for (int w = 0; w < kIterThreads / 64; ++w) v += s_red[w][threadIdx.x];
        partials[(size_t)wg * kNSums + threadIdx.x] = v;
    }
}

// ---- second reduction pass + solve + pose update (DenseOdometryFunction.cpp:404-413) ------------
__global__ __launch_bounds__(1024) void k_track_solve(TrackState* __restrict__ st, int l, int it, const double* __restrict__ partials,
                                                      int n_partials) {
    if (st->stop_level == l) return;
    __shared__ double s[32][kNSums];
    __shared__ double tot[kNSums];
    const int k = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
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
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double JTJ[36], JTr[6];
        float x[6], D[16], cur[16];
        int q = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) { JTJ[a * 6 + b] = tot[q]; JTJ[b * 6 + a] = tot[q]; ++q; }
        for (int a = 0; a < 6; ++a) JTr[a] = tot[21 + a];
        op_host::ldlt_solve6(JTJ, JTr, x);
        op_host::se3_exp(x, D);
        for (int i = 0; i < 16; ++i) cur[i] = st->T[i];
        op_host::mat4_mul(D, cur, cur);               // relative_pose = delta_matrix * relative_pose
        for (int i = 0; i < 16; ++i) st->T[i] = cur[i];
        const unsigned long long n = (unsigned long long)(tot[28] + 0.5);
        st->n_last = n;
        st->last_level = l;
        st->iters_done = st->iters_done + 1;
        st->per_iter_count[it] = (int)n;
        for (int i = 0; i < 16; ++i) st->per_iter_T[16 * it + i] = cur[i];
        // Odometry.cpp:669: (float)size / (height*width) > MAX_INLIER_RATIO_DENSE, FULL-resolution h*w
        if ((double)((float)n / (float)(st->full_h * st->full_w)) > 0.9) st->stop_level = l;
    }
}

} // namespace

struct op_tracker {
    int device = 0;
    hipStream_t stream = nullptr;
    TrackState* st = nullptr;        // device
    TrackState* st_host = nullptr;   // pinned: header uploaded for the loop
    TrackState* st_host_norm = nullptr; // pinned: header uploaded for the NormalizeIntensity pass
    TrackState* st_back = nullptr;   // pinned: state downloaded after the run
    bool pending = false;            // a run has been enqueued and not yet waited for
    bool pending_logs = false, pending_points = false;
    // The reference-order modes (OP_TRACK_SUMS_REFERENCE_F32*) need the host after every iteration, so an "enqueued" run of theirs executes on a host
    // thread of the tracker's own: op_tracker_dense_tracking_enqueue still returns at once and several trackers still run side by side (their
    // sequential one-wave sums on different CUs).  op_tracker_wait joins it.
    std::thread worker;
    bool worker_active = false;
    int worker_rc = OP_OK;
    char worker_err[512] = "";
    size_t pix_cap = 0;              // workspace capacity in pixels
    int* pair_p = nullptr;           // per source pixel: candidate target pixel index p(s) or -1
    int* pair_t = nullptr;           // per source pixel: accepted target pixel index or -1
    unsigned short* code = nullptr;  // per source pixel: acceptance link code
    int lds_cap = 0;                 // dynamic LDS bytes available to one k_track_iter workgroup
    int sums = OP_TRACK_SUMS_FP64;   // OP_TRACK_OPT_SUMS
    bool seq_ok = false;             // k_seq_sums may have its LDS (else OP_TRACK_SUMS_REFERENCE_F32 sums on the host like _F32_HOST)
    float* rows_dev = nullptr;       // validation mode: 14 floats per source pixel
    float* rows_host = nullptr;      // pinned
    int* pair_host = nullptr;        // pinned
    size_t rows_cap = 0;             // pixels
    size_t rows_host_cap = 0;        // pixels (rows_host / pair_host)
    float* seq_out = nullptr;        // device: the 42 (+ count) results of k_seq_sums
    float* seq_host = nullptr;       // pinned copy
    unsigned* seq_total = nullptr;   // device: number of accepted pixels (k_emit_scan)
    hipEvent_t seq_ev = nullptr;     // "my ordered rows are in place" (for the rendezvous of trackers that sum together, seq_sums.hpp)
    int lds_total = 0, lds_static = 0, n_cu = 256;
    double* partials = nullptr;
    unsigned* wg_count = nullptr;
    int4* pix_out = nullptr;
    float* pts_out = nullptr;
    float* images = nullptr;         // device copies of OP_MEM_HOST pyramids
    size_t images_cap = 0;           // floats
    // op_tracker_dense_tracking: raw frames and the pyramids it builds
    unsigned char* raw_rgb = nullptr;
    unsigned char* raw_depth = nullptr;
    size_t raw_cap = 0;              // pixels
    float* pyr = nullptr;            // [frame 2][kind 6][levels] images, see pyr_image()
    size_t pyr_cap = 0;              // floats
    float* norm_scales = nullptr;    // 2 floats
    int pyr_w = 0, pyr_h = 0, pyr_levels = 0;
    PrepFrames* prep_host = nullptr; // pinned: this call's frame pointers
    PrepFrames* prep_dev = nullptr;
    // hipGraph of one whole dense-tracking call (device frames, no point pairs): the ~100 launches of a call
    // cost ~0.3 ms of host time when issued one by one, which caps several trackers working concurrently
    hipGraphExec_t graph_exec = nullptr;
    unsigned long long graph_key[4] = {0, 0, 0, 0};
    int graph_ok = 1;                // cleared when capture/instantiate fails (then launches are issued directly)
};

namespace {

int tracker_reserve(op_tracker* t, size_t npix, size_t image_floats) {
    if (npix > t->pix_cap) {
        (void)hipFree(t->pair_t); (void)hipFree(t->pair_p); (void)hipFree(t->code); (void)hipFree(t->partials); (void)hipFree(t->wg_count); (void)hipFree(t->pix_out); (void)hipFree(t->pts_out);
        t->pair_t = nullptr; t->pair_p = nullptr; t->code = nullptr; t->partials = nullptr; t->wg_count = nullptr; t->pix_out = nullptr; t->pts_out = nullptr;
        t->pix_cap = 0;
        const size_t n_wg = (npix + kThreads - 1) / kThreads;
        OP_HIP(hipMalloc(&t->pair_t, npix * sizeof(int)));
        OP_HIP(hipMalloc(&t->pair_p, npix * sizeof(int)));
        OP_HIP(hipMalloc(&t->code, (npix + 16) * sizeof(unsigned short))); // window staging reads whole 16 B groups
        OP_HIP(hipMalloc(&t->partials, n_wg * kNSums * sizeof(double)));
        OP_HIP(hipMalloc(&t->wg_count, n_wg * sizeof(unsigned)));
        OP_HIP(hipMalloc(&t->pix_out, npix * sizeof(int4)));
        OP_HIP(hipMalloc(&t->pts_out, npix * 6 * sizeof(float)));
        t->pix_cap = npix;
    }
    if (image_floats > t->images_cap) {
        (void)hipFree(t->images);
        t->images = nullptr; t->images_cap = 0;
        OP_HIP(hipMalloc(&t->images, image_floats * sizeof(float)));
        t->images_cap = image_floats;
    }
    return OP_OK;
}

// Per-level launch geometry of k_track_iter: one own pixel per thread; the LDS window is the whole
// level when that fits, and at most half of the LDS when there are more workgroups than CUs (so two
// workgroups share a CU and every workgroup of the level is resident at once).
struct IterGeom { int n_wg, win_cap; size_t lds_bytes; };
IterGeom iter_geom(const op_tracker* t, size_t npix) {
    IterGeom g;
    g.n_wg = (int)((npix + kIterThreads - 1) / kIterThreads);
    const size_t want = (npix + 8) * sizeof(unsigned short);  // the whole level (+ alignment slack)
    size_t cap = (size_t)t->lds_cap;
    if (g.n_wg > t->n_cu) cap = ((size_t)t->lds_total / 2 - (size_t)t->lds_static) & ~(size_t)255;
    g.lds_bytes = (((want < cap ? want : cap)) + 15) & ~(size_t)15;
    if (g.lds_bytes > cap) g.lds_bytes = cap & ~(size_t)15;
    g.win_cap = (int)(g.lds_bytes / sizeof(unsigned short)) - 8; // the window start is rounded down to 8 entries
    return g;
}

template <int TERM>
void launch_iter(op_tracker* t, int l, const IterGeom& g) {
    hipLaunchKernelGGL(k_track_iter<TERM>, dim3(g.n_wg), dim3(kIterThreads), g.lds_bytes, t->stream, t->st, l, g.win_cap,
                       t->pair_p, t->code, t->pair_t, t->partials);
}

} // namespace

extern "C" {

int op_tracker_create(int device, op_tracker** out) {
    if (!out) return fail(OP_ERR_INVALID, "op_tracker_create: out is NULL");
    OP_TRY(use_device(device));
    op_tracker* t = new op_tracker();
    t->device = device;
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&t->st, sizeof(TrackState)) != hipSuccess ||
        hipHostMalloc(&t->st_host, sizeof(TrackState)) != hipSuccess ||
        hipHostMalloc(&t->st_host_norm, sizeof(TrackState)) != hipSuccess ||
        hipHostMalloc(&t->st_back, sizeof(TrackState)) != hipSuccess ||
        hipHostMalloc(&t->prep_host, sizeof(PrepFrames)) != hipSuccess || hipMalloc(&t->prep_dev, sizeof(PrepFrames)) != hipSuccess) {
        op_tracker_destroy(t);
        return fail(OP_ERR_HIP, "op_tracker_create: allocating tracker state failed");
    }
    // k_track_iter stages link codes in as much LDS as a workgroup may have (160 KB on gfx950)
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || lds_max < 32768) lds_max = 65536;
    const int lds_static = (int)(sizeof(double) * (kIterThreads / 64) * kNSums + 64) + 256;
    t->lds_total = lds_max; t->lds_static = lds_static;
    t->lds_cap = lds_max - lds_static;
    if (hipDeviceGetAttribute(&t->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || t->n_cu <= 0) t->n_cu = 256;
    bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_track_iter<0>), hipFuncAttributeMaxDynamicSharedMemorySize, t->lds_cap) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&k_track_iter<1>), hipFuncAttributeMaxDynamicSharedMemorySize, t->lds_cap) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&k_track_iter<2>), hipFuncAttributeMaxDynamicSharedMemorySize, t->lds_cap) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&k_track_iter<3>), hipFuncAttributeMaxDynamicSharedMemorySize, t->lds_cap) == hipSuccess;
    if (!attr_ok) { (void)hipGetLastError(); t->lds_total = 65536; t->lds_cap = 65536 - lds_static; }
    // k_seq_sums (reference-order sums) double-buffers its product tiles in ~150 KB of LDS
    t->seq_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seq_sums<42, 14, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(42, 14, 2)) == hipSuccess &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seq_sums<42, 7, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(42, 7, 1)) == hipSuccess &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seq_sums<2, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(2, 2, 1)) == hipSuccess &&
                (size_t)lds_max >= seq_lds_bytes(42, 14, 2);
    if (!t->seq_ok) (void)hipGetLastError();
    if (!op::runtime_options().tracker_graph.load()) t->graph_ok = 0; // OP_RUNTIME_OPT_TRACKER_GRAPH
    t->sums = op::runtime_options().tracker_default_sums.load();       // OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS: the reference's own sums unless the process opted out
    *out = t;
    return OP_OK;
}

int op_tracker_set_option(op_tracker* t, int option, int value) {
    if (!t) return fail(OP_ERR_INVALID, "null tracker");
    if (t->worker_active) return fail(OP_ERR_INVALID, "op_tracker_set_option: an enqueued run has not been waited for");
    if (option == OP_TRACK_OPT_SUMS && (value == OP_TRACK_SUMS_FP64 || value == OP_TRACK_SUMS_REFERENCE_F32 || value == OP_TRACK_SUMS_REFERENCE_F32_HOST)) { t->sums = value; return OP_OK; }
    return fail(OP_ERR_INVALID, "op_tracker_set_option: unknown option %d / value %d", option, value);
}

int op_tracker_destroy(op_tracker* t) {
    if (!t) return OP_OK;
    if (t->worker_active) { t->worker.join(); t->worker_active = false; }
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    (void)hipFree(t->rows_dev);
    (void)hipFree(t->seq_out); (void)hipFree(t->seq_total);
    if (t->seq_ev) (void)hipEventDestroy(t->seq_ev);
    if (t->seq_host) (void)hipHostFree(t->seq_host);
    if (t->rows_host) (void)hipHostFree(t->rows_host);
    if (t->pair_host) (void)hipHostFree(t->pair_host);
    (void)hipFree(t->pair_t); (void)hipFree(t->pair_p); (void)hipFree(t->code); (void)hipFree(t->partials); (void)hipFree(t->wg_count); (void)hipFree(t->pix_out); (void)hipFree(t->pts_out);
    (void)hipFree(t->images); (void)hipFree(t->st); (void)hipFree(t->raw_rgb); (void)hipFree(t->raw_depth); (void)hipFree(t->pyr); (void)hipFree(t->norm_scales); (void)hipFree(t->prep_dev);
    if (t->prep_host) (void)hipHostFree(t->prep_host);
    if (t->graph_exec) (void)hipGraphExecDestroy(t->graph_exec);
    if (t->st_host) (void)hipHostFree(t->st_host);
    if (t->st_host_norm) (void)hipHostFree(t->st_host_norm);
    if (t->st_back) (void)hipHostFree(t->st_back);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return OP_OK;
}

// Enqueues the coarse-to-fine loop + result assembly over the level descriptors already stored in
// t->st_host->lv (device pointers) on t->stream and returns; track_finish() synchronises and reads the
// result.  The pinned buffers are only touched between a finish and the next enqueue.
static void fill_loop_header(op_tracker* t, int full_width, int full_height, int term_type, const float init_T[16]) {
    TrackState* h = t->st_host;
    std::memcpy(h->T, init_T, sizeof(h->T));
    h->full_w = full_width; h->full_h = full_height; h->term = term_type;
    h->stop_level = -1; h->iters_done = 0; h->last_level = -1; h->n_last = 0; h->n_emit = 0; h->rmse = 0; h->success = 0;
}

// Trackers in the reference-order mode that run at the same time (the pairs in flight of a tracking + fusion pipeline, each on its own host thread and stream) can take
// their per-iteration sequential sums TOGETHER: one k_seq_sums_many launch with a workgroup per tracker instead of a one-workgroup launch per stream (seq_sums.hpp:
// SeqRendezvous; with twelve or more trackers running, below that each launches its own).  OFF by default (OP_RUNTIME_OPT_TRACKER_BATCH_SUMS): what the pipeline
// needed was a hardware queue per tracker stream (GPU_MAX_HW_QUEUES = 16: 349 -> 489 frames/s with 16 pairs in flight, profiles/r06_track_hw_queues.txt); meeting in
// lock step costs every round the largest pyramid level's sum (profiles/r06_track_depth_probe.txt).  One rendezvous per device; hybrid term only.
using TrackSeqBatch = SeqRendezvous<42, 14, 2, 12>;
static TrackSeqBatch* track_seq_batch(int device) {
    static TrackSeqBatch pool[16];
    return device >= 0 && device < 16 ? &pool[device] : nullptr;
}
struct TrackBatchMembership { // leaves on every exit of the run
    TrackSeqBatch* b = nullptr;
    ~TrackBatchMembership() { if (b) b->leave(); }
};

static int track_enqueue(op_tracker* t, int n_levels, const int32_t* iters_per_level, int full_width, int full_height, int term_type,
                         const float init_T[16], bool want_points, bool want_logs) {
    TrackState* h = t->st_host;
    fill_loop_header(t, full_width, full_height, term_type, init_T);
    // header of the state only (the per-iteration logs are outputs)
    OP_HIP(hipMemcpyAsync(t->st, h, offsetof(TrackState, per_iter_count), hipMemcpyHostToDevice, t->stream));
    size_t max_pix = 0;
    int it = 0;
    const bool strict = t->sums != OP_TRACK_SUMS_FP64;
    const bool on_device = t->sums == OP_TRACK_SUMS_REFERENCE_F32 && t->seq_ok; // the sequential float32 sums in k_seq_sums; else on one host thread
    if (strict && !t->seq_out) {
        OP_HIP(hipMalloc(&t->seq_out, 64 * sizeof(float)));
        OP_HIP(hipHostMalloc(&t->seq_host, 64 * sizeof(float), hipHostMallocDefault));
        OP_HIP(hipMalloc(&t->seq_total, sizeof(unsigned)));
    }
    TrackSeqBatch* batch = nullptr;
    TrackBatchMembership member;
    if (strict && on_device && term_type == 0 && op::runtime_options().tracker_batch_sums.load()) {
        if (!t->seq_ev && hipEventCreateWithFlags(&t->seq_ev, hipEventDisableTiming) != hipSuccess) { t->seq_ev = nullptr; (void)hipGetLastError(); }
        TrackSeqBatch* b = track_seq_batch(t->device);
        if (t->seq_ev && b && b->usable(t->device)) { batch = b; b->join(); member.b = b; }
    }
    float cur[16];
    std::memcpy(cur, init_T, sizeof(cur));
    for (int l = n_levels - 1; l >= 0; --l) {
        const size_t np = (size_t)h->lv[l].w * h->lv[l].h;
        max_pix = np > max_pix ? np : max_pix;
        const int n_wg_a = (int)((np + kThreads - 1) / kThreads);
        const IterGeom g = iter_geom(t, np);
        if (strict && np > t->rows_cap) {
            if (t->rows_dev) OP_HIP(hipFree(t->rows_dev));
            t->rows_dev = nullptr; t->rows_cap = 0;
            OP_HIP(hipMalloc(&t->rows_dev, np * 14 * sizeof(float)));
            t->rows_cap = np;
        }
        if (strict && !on_device && np > t->rows_host_cap) { // the host-sum variant's pinned copies (not needed when k_seq_sums does the sums)
            if (t->rows_host) OP_HIP(hipHostFree(t->rows_host));
            if (t->pair_host) OP_HIP(hipHostFree(t->pair_host));
            t->rows_host = nullptr; t->pair_host = nullptr; t->rows_host_cap = 0;
            OP_HIP(hipHostMalloc(&t->rows_host, np * 14 * sizeof(float), hipHostMallocDefault));
            OP_HIP(hipHostMalloc(&t->pair_host, np * sizeof(int), hipHostMallocDefault));
            t->rows_host_cap = np;
        }
        for (int j = 0; j < iters_per_level[l]; ++j, ++it) {
            hipLaunchKernelGGL(k_track_assoc, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, t->pair_p, t->code);
            if (term_type == 0) launch_iter<0>(t, l, g);
            else if (term_type == 1) launch_iter<1>(t, l, g);
            else launch_iter<2>(t, l, g);
            if (!strict) {
                hipLaunchKernelGGL(k_track_solve, dim3(1), dim3(1024), 0, t->stream, t->st, l, it, t->partials, g.n_wg);
                continue;
            }
            // Reference-order sums: association, acceptance and the Jacobian rows come from the kernels; the sums are taken in raster
            // order in float32 like the reference's loop -- by k_seq_sums on the device (42 numbers come back), or, in the _HOST variant,
            // on ONE host thread after a transfer of all rows -- then the solve / exp / pose update on the host, and the new pose goes
            // back into the device state for the next iteration.
            float JTJ[36], JTr[6], x[6], D[16];
            size_t n_pairs = 0;
            bool batched_now = false;
            if (on_device) {
                hipLaunchKernelGGL(k_rows_count, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, (const int*)t->pair_t, t->wg_count);
                launch_emit_scan(dim3(1), dim3(1024), t->stream, t->st, t->wg_count, n_wg_a, t->seq_total);
                if (term_type == 0) {
                    hipLaunchKernelGGL(k_track_rows_compact<0>, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, (const int*)t->pair_t, (const unsigned*)t->wg_count, t->rows_dev);
                    hipError_t eb = hipErrorNotReady;
                    if (batch) { // with the other trackers running now: one launch, a workgroup each; the sums land in t->seq_host (hipErrorNotReady: too few of them)
                        OP_HIP(hipGetLastError());
                        eb = batch->submit(t->rows_dev, t->seq_total, t->seq_out, t->seq_host, t->seq_ev, t->stream);
                        if (eb != hipSuccess && eb != hipErrorNotReady) return fail(OP_ERR_HIP, "tracker: the batched sequential sums failed: %s", hipGetErrorString(eb));
                    }
                    batched_now = eb == hipSuccess;
                    if (!batched_now)
                        hipLaunchKernelGGL((k_seq_sums<42, 14, 2>), dim3(1), dim3(kSeqThreads), seq_lds_bytes(42, 14, 2), t->stream, (const float*)t->rows_dev, (const unsigned*)t->seq_total, t->seq_out);
                } else {
                    if (term_type == 1) hipLaunchKernelGGL(k_track_rows_compact<1>, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, (const int*)t->pair_t, (const unsigned*)t->wg_count, t->rows_dev);
                    else hipLaunchKernelGGL(k_track_rows_compact<2>, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, (const int*)t->pair_t, (const unsigned*)t->wg_count, t->rows_dev);
                    hipLaunchKernelGGL((k_seq_sums<42, 7, 1>), dim3(1), dim3(kSeqThreads), seq_lds_bytes(42, 7, 1), t->stream, (const float*)t->rows_dev, (const unsigned*)t->seq_total, t->seq_out);
                }
                OP_HIP(hipGetLastError());
                if (!batched_now) {
                    OP_HIP(hipMemcpyAsync(t->seq_host, t->seq_out, 43 * sizeof(float), hipMemcpyDeviceToHost, t->stream));
                    OP_HIP(hipStreamSynchronize(t->stream));
                }
                std::memcpy(JTJ, t->seq_host, sizeof(JTJ));
                std::memcpy(JTr, t->seq_host + 36, sizeof(JTr));
                unsigned n32;
                std::memcpy(&n32, t->seq_host + 42, sizeof(n32));
                n_pairs = n32;
            } else {
                if (term_type == 0) hipLaunchKernelGGL(k_track_rows<0>, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, (const int*)t->pair_t, t->rows_dev);
                else if (term_type == 1) hipLaunchKernelGGL(k_track_rows<1>, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, (const int*)t->pair_t, t->rows_dev);
                else hipLaunchKernelGGL(k_track_rows<2>, dim3(n_wg_a), dim3(kThreads), 0, t->stream, t->st, l, (const int*)t->pair_t, t->rows_dev);
                OP_HIP(hipGetLastError());
                OP_HIP(hipMemcpyAsync(t->rows_host, t->rows_dev, np * 14 * sizeof(float), hipMemcpyDeviceToHost, t->stream));
                OP_HIP(hipMemcpyAsync(t->pair_host, t->pair_t, np * sizeof(int), hipMemcpyDeviceToHost, t->stream));
                OP_HIP(hipStreamSynchronize(t->stream));
                op_host::track_sums_reference_order(t->rows_host, t->pair_host, np, term_type == 0 ? 2 : 1, JTJ, JTr, &n_pairs);
            }
            op_host::ldlt_solve6_float_sums(JTJ, JTr, x);          // DenseOdometryFunction.cpp:404
            op_host::se3_exp(x, D);
            op_host::mat4_mul(D, cur, cur);                          // relative_pose = delta_matrix * relative_pose
            PoseArg pa;
            std::memcpy(pa.m, cur, sizeof(cur));
            hipLaunchKernelGGL(k_track_apply, dim3(1), dim3(1), 0, t->stream, t->st, l, it, pa, (unsigned long long)n_pairs);
            // Odometry.cpp:669: early-out of the level at ratio > MAX_INLIER_RATIO_DENSE (the device sets stop_level likewise)
            if ((double)((float)n_pairs / (float)(full_height * full_width)) > 0.9) { it += 1; break; }
        }
    }
    const int n_wg_max = (int)((max_pix + kThreads - 1) / kThreads);
    launch_emit_count(dim3(n_wg_max), dim3(kThreads), t->stream, t->st, t->pair_t, t->wg_count);
    launch_emit_scan(dim3(1), dim3(1024), t->stream, t->st, t->wg_count, n_wg_max);
    launch_emit_scatter(dim3(n_wg_max), dim3(kThreads), t->stream, t->st, t->pair_t, t->wg_count, t->pix_out,
                       want_points ? t->pts_out : nullptr, t->partials);
    launch_emit_finish(dim3(1), dim3(1024), t->stream, t->st, t->partials, n_wg_max);
    OP_HIP(hipGetLastError());
    OP_HIP(hipMemcpyAsync(t->st_back, t->st, want_logs ? sizeof(TrackState) : offsetof(TrackState, per_iter_count), hipMemcpyDeviceToHost,
                          t->stream));
    t->pending = true; t->pending_logs = want_logs; t->pending_points = want_points;
    return OP_OK;
}

static int track_finish(op_tracker* t, op_track_result* result, int32_t* pixel_corr, float* point_corr, size_t corr_cap,
                        int32_t* per_iter_count, float* per_iter_T) {
    if (!t->pending) return fail(OP_ERR_INVALID, "tracker: nothing has been enqueued");
    t->pending = false;
    OP_HIP(hipStreamSynchronize(t->stream));
    const TrackState* h = t->st_back;
    std::memcpy(result->T, h->T, sizeof(result->T));
    result->rmse = h->rmse;
    result->n_correspondences = h->last_level < 0 ? 0 : h->n_last;
    result->tracking_success = h->success;
    result->iterations = h->iters_done;
    if (h->last_level >= 0 && h->n_emit != h->n_last)
        return fail(OP_ERR_HIP, "tracker: emitted %llu correspondences, counted %llu", h->n_emit, h->n_last);
    if ((per_iter_count || per_iter_T) && !t->pending_logs) return fail(OP_ERR_INVALID, "tracker: per-iteration logs were not requested at enqueue");
    if (per_iter_count) std::memcpy(per_iter_count, h->per_iter_count, sizeof(int) * h->iters_done);
    if (per_iter_T) std::memcpy(per_iter_T, h->per_iter_T, sizeof(float) * 16 * h->iters_done);
    const size_t n = (size_t)result->n_correspondences;
    if ((pixel_corr || point_corr) && n) {
        if (n > corr_cap) return fail(OP_ERR_CAPACITY, "tracker: %zu correspondences exceed corr_cap %zu", n, corr_cap);
        if (point_corr && !t->pending_points) return fail(OP_ERR_INVALID, "tracker: point correspondences were not requested at enqueue");
        if (pixel_corr) OP_HIP(hipMemcpyAsync(pixel_corr, t->pix_out, n * sizeof(int4), hipMemcpyDeviceToHost, t->stream));
        if (point_corr) OP_HIP(hipMemcpyAsync(point_corr, t->pts_out, n * 6 * sizeof(float), hipMemcpyDeviceToHost, t->stream));
        OP_HIP(hipStreamSynchronize(t->stream));
    }
    return OP_OK;
}

static int track_run(op_tracker* t, int n_levels, const int32_t* iters_per_level, int full_width, int full_height, int term_type,
                     const float init_T[16], op_track_result* result, int32_t* pixel_corr, float* point_corr, size_t corr_cap,
                     int32_t* per_iter_count, float* per_iter_T) {
    OP_TRY(track_enqueue(t, n_levels, iters_per_level, full_width, full_height, term_type, init_T, point_corr != nullptr,
                         per_iter_count || per_iter_T));
    return track_finish(t, result, pixel_corr, point_corr, corr_cap, per_iter_count, per_iter_T);
}

static int check_iters(const char* who, int n_levels, const int32_t* iters_per_level, int term_type) {
    if (n_levels < 1 || n_levels > kMaxLevels) return fail(OP_ERR_INVALID, "%s: n_levels %d not in [1,%d]", who, n_levels, kMaxLevels);
    if (term_type < 0 || term_type > 2) return fail(OP_ERR_INVALID, "%s: term_type %d not in {0,1,2}", who, term_type);
    int total = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (iters_per_level[l] < 0) return fail(OP_ERR_INVALID, "%s: negative iteration count at level %d", who, l);
        total += iters_per_level[l];
    }
    if (total > kMaxIters) return fail(OP_ERR_INVALID, "%s: %d iterations exceed the limit %d", who, total, kMaxIters);
    return OP_OK;
}

int op_tracker_track(op_tracker* t, const op_track_level* levels, int n_levels, const int32_t* iters_per_level, int full_width,
                     int full_height, int term_type, const float init_T[16], int mem, op_track_result* result, int32_t* pixel_corr,
                     float* point_corr, size_t corr_cap, int32_t* per_iter_count, float* per_iter_T) {
    if (!t || !levels || !iters_per_level || !init_T || !result) return fail(OP_ERR_INVALID, "op_tracker_track: NULL argument");
    OP_TRY(check_iters("op_tracker_track", n_levels, iters_per_level, term_type));
    if (full_width <= 0 || full_height <= 0) return fail(OP_ERR_INVALID, "op_tracker_track: full resolution %dx%d", full_width, full_height);
    if (mem != OP_MEM_HOST && mem != OP_MEM_DEVICE) return fail(OP_ERR_INVALID, "op_tracker_track: bad mem %d", mem);
    size_t max_pix = 0, image_floats = 0;
    for (int l = 0; l < n_levels; ++l) {
        const op_track_level& L = levels[l];
        if (L.width <= 0 || L.height <= 0 || (size_t)L.width * L.height > (1u << 28))
            return fail(OP_ERR_INVALID, "op_tracker_track: level %d size %dx%d", l, L.width, L.height);
        if (!L.source_color || !L.source_depth || !L.target_color || !L.target_depth || !L.target_color_dx || !L.target_color_dy ||
            !L.target_depth_dx || !L.target_depth_dy)
            return fail(OP_ERR_INVALID, "op_tracker_track: level %d has a NULL image", l);
        const size_t np = (size_t)L.width * L.height;
        max_pix = np > max_pix ? np : max_pix;
        image_floats += 8 * np;
    }
    if (t->pending || t->worker_active) return fail(OP_ERR_INVALID, "op_tracker_track: an enqueued run has not been waited for");
    OP_TRY(use_device(t->device));
    OP_TRY(tracker_reserve(t, max_pix, mem == OP_MEM_HOST ? image_floats : 0));
    TrackState* h = t->st_host;
    float* dst = t->images;
    for (int l = 0; l < n_levels; ++l) {
        const op_track_level& L = levels[l];
        LevelDev& D = h->lv[l];
        D.w = L.width; D.h = L.height; D.fx = L.fx; D.fy = L.fy; D.cx = L.cx; D.cy = L.cy;
        const float* src[8] = {L.source_color, L.source_depth, L.target_color, L.target_depth,
                               L.target_color_dx, L.target_color_dy, L.target_depth_dx, L.target_depth_dy};
        const float** out[8] = {&D.sc, &D.sd, &D.tc, &D.td, &D.tcdx, &D.tcdy, &D.tddx, &D.tddy};
        const size_t np = (size_t)L.width * L.height;
        for (int k = 0; k < 8; ++k) {
            if (mem == OP_MEM_HOST) {
                OP_HIP(hipMemcpyAsync(dst, src[k], np * sizeof(float), hipMemcpyHostToDevice, t->stream));
                *out[k] = dst;
                dst += np;
            } else {
                *out[k] = src[k];
            }
        }
    }
    return track_run(t, n_levels, iters_per_level, full_width, full_height, term_type, init_T, result, pixel_corr, point_corr, corr_cap,
                     per_iter_count, per_iter_T);
}

// image (frame f: 0 source / 1 target, kind k: 0 colour 1 depth 2 colour_dx 3 colour_dy 4 depth_dx 5 depth_dy, level l)
static float* pyr_image(const op_tracker* t, int f, int k, int l) {
    size_t off = 0;
    for (int q = 0; q < l; ++q) off += (size_t)(t->pyr_w >> q) * (t->pyr_h >> q);
    size_t per_set = 0;
    for (int q = 0; q < t->pyr_levels; ++q) per_set += (size_t)(t->pyr_w >> q) * (t->pyr_h >> q);
    return t->pyr + (size_t)(f * 6 + k) * per_set + off;
}

static int dense_tracking_enqueue_now(op_tracker* t, const op_camera* cam, int n_levels, const int32_t* iters_per_level,
                                      const uint8_t* source_rgb, const uint8_t* target_rgb, const void* source_depth, const void* target_depth,
                                      int depth_fmt, const float init_T[16], int term_type, int mem, int want_point_corr);

int op_tracker_dense_tracking_enqueue(op_tracker* t, const op_camera* cam, int n_levels, const int32_t* iters_per_level,
                                      const uint8_t* source_rgb, const uint8_t* target_rgb, const void* source_depth, const void* target_depth,
                                      int depth_fmt, const float init_T[16], int term_type, int mem, int want_point_corr) {
    if (!t || !cam || !iters_per_level || !source_rgb || !target_rgb || !source_depth || !target_depth || !init_T)
        return fail(OP_ERR_INVALID, "op_tracker_dense_tracking: NULL argument");
    if (t->pending || t->worker_active) return fail(OP_ERR_INVALID, "op_tracker_dense_tracking: an enqueued run has not been waited for");
    if (t->sums == OP_TRACK_SUMS_FP64)
        return dense_tracking_enqueue_now(t, cam, n_levels, iters_per_level, source_rgb, target_rgb, source_depth, target_depth, depth_fmt, init_T, term_type, mem, want_point_corr);
    // reference-order sums: the run synchronises with the host every iteration -- on the tracker's own host thread (the images must stay valid until op_tracker_wait, as for any enqueue)
    OP_TRY(check_iters("op_tracker_dense_tracking", n_levels, iters_per_level, term_type));
    const op_camera cam_copy = *cam;
    const std::vector<int32_t> iters(iters_per_level, iters_per_level + n_levels);
    std::array<float, 16> T0;
    std::memcpy(T0.data(), init_T, sizeof(float) * 16);
    t->worker_active = true; t->worker_rc = OP_OK; t->worker_err[0] = 0;
    t->worker = std::thread([=] {
        t->worker_rc = dense_tracking_enqueue_now(t, &cam_copy, n_levels, iters.data(), source_rgb, target_rgb, source_depth, target_depth, depth_fmt, T0.data(), term_type, mem, want_point_corr);
        if (t->worker_rc != OP_OK) std::snprintf(t->worker_err, sizeof(t->worker_err), "%s", op::g_last_error); // (the error text is thread-local: hand it over)
    });
    return OP_OK;
}

static int dense_tracking_enqueue_now(op_tracker* t, const op_camera* cam, int n_levels, const int32_t* iters_per_level,
                                      const uint8_t* source_rgb, const uint8_t* target_rgb, const void* source_depth, const void* target_depth,
                                      int depth_fmt, const float init_T[16], int term_type, int mem, int want_point_corr) {
    if (t->pending) return fail(OP_ERR_INVALID, "op_tracker_dense_tracking: an enqueued run has not been waited for");
    OP_TRY(check_iters("op_tracker_dense_tracking", n_levels, iters_per_level, term_type));
    if (depth_fmt != OP_DEPTH_F32 && depth_fmt != OP_DEPTH_U16) return fail(OP_ERR_INVALID, "op_tracker_dense_tracking: bad depth_fmt %d", depth_fmt);
    if (mem != OP_MEM_HOST && mem != OP_MEM_DEVICE) return fail(OP_ERR_INVALID, "op_tracker_dense_tracking: bad mem %d", mem);
    const int W = cam->width, H = cam->height;
    if (W < 4 || H < 4 || (size_t)W * H > (1u << 28) || (W >> (n_levels - 1)) < 3 || (H >> (n_levels - 1)) < 3)
        return fail(OP_ERR_INVALID, "op_tracker_dense_tracking: %dx%d with %d levels", W, H, n_levels);
    OP_TRY(use_device(t->device));
    const size_t np = (size_t)W * H;
    OP_TRY(tracker_reserve(t, np, 0));
    size_t per_set = 0;
    for (int q = 0; q < n_levels; ++q) per_set += (size_t)(W >> q) * (H >> q);
    if (12 * per_set > t->pyr_cap) {
        (void)hipFree(t->pyr); t->pyr = nullptr; t->pyr_cap = 0;
        OP_HIP(hipMalloc(&t->pyr, 12 * per_set * sizeof(float)));
        t->pyr_cap = 12 * per_set;
    }
    if (!t->norm_scales) OP_HIP(hipMalloc(&t->norm_scales, 2 * sizeof(float)));
    t->pyr_w = W; t->pyr_h = H; t->pyr_levels = n_levels;
    const size_t dbytes = depth_fmt == OP_DEPTH_U16 ? 2 : 4;
    PrepFrames& P = *t->prep_host;
    if (mem == OP_MEM_HOST) {
        if (np > t->raw_cap) {
            (void)hipFree(t->raw_rgb); (void)hipFree(t->raw_depth); t->raw_rgb = nullptr; t->raw_depth = nullptr; t->raw_cap = 0;
            OP_HIP(hipMalloc(&t->raw_rgb, 2 * np * 3));
            OP_HIP(hipMalloc(&t->raw_depth, 2 * np * 4));
            t->raw_cap = np;
        }
        OP_HIP(hipMemcpyAsync(t->raw_rgb, source_rgb, np * 3, hipMemcpyHostToDevice, t->stream));
        OP_HIP(hipMemcpyAsync(t->raw_rgb + np * 3, target_rgb, np * 3, hipMemcpyHostToDevice, t->stream));
        OP_HIP(hipMemcpyAsync(t->raw_depth, source_depth, np * dbytes, hipMemcpyHostToDevice, t->stream));
        OP_HIP(hipMemcpyAsync(t->raw_depth + np * 4, target_depth, np * dbytes, hipMemcpyHostToDevice, t->stream));
        P.rgb[0] = t->raw_rgb; P.rgb[1] = t->raw_rgb + np * 3; P.depth[0] = t->raw_depth; P.depth[1] = t->raw_depth + np * 4;
    } else {
        P.rgb[0] = source_rgb; P.rgb[1] = target_rgb; P.depth[0] = source_depth; P.depth[1] = target_depth;
    }
    P.is_u16 = depth_fmt == OP_DEPTH_U16; P.depth_scale = cam->depth_scale; P.w = W; P.h = H;
    P.out[0] = pyr_image(t, 0, 0, 0); P.out[1] = pyr_image(t, 1, 0, 0); P.out[2] = pyr_image(t, 0, 1, 0); P.out[3] = pyr_image(t, 1, 1, 0);

    // level descriptors (Camera.h:38-42: intrinsics halved per level) + the NormalizeIntensity pass header
    TrackState* h = t->st_host_norm;
    float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
    for (int l = 0; l < n_levels; ++l) {
        LevelDev& D = h->lv[l];
        D.w = W >> l; D.h = H >> l; D.fx = fx; D.fy = fy; D.cx = cx; D.cy = cy;
        D.sc = pyr_image(t, 0, 0, l); D.sd = pyr_image(t, 0, 1, l); D.tc = pyr_image(t, 1, 0, l); D.td = pyr_image(t, 1, 1, l);
        D.tcdx = pyr_image(t, 1, 2, l); D.tcdy = pyr_image(t, 1, 3, l); D.tddx = pyr_image(t, 1, 4, l); D.tddy = pyr_image(t, 1, 5, l);
        fx /= 2; fy /= 2; cx /= 2; cy /= 2;
    }
    {
        const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        std::memcpy(h->T, I4, sizeof(I4));
        h->full_w = W; h->full_h = H; h->term = 3; h->stop_level = -1;
        h->iters_done = 0; h->last_level = -1; h->n_last = 0; h->n_emit = 0; h->rmse = 0; h->success = 0;
        std::memcpy(t->st_host->lv, h->lv, sizeof(h->lv));
    }
    fill_loop_header(t, W, H, term_type, init_T);

    // One call = ~100 small launches.  With device-resident frames the sequence depends on the call only through
    // the three pinned buffers filled above, so it is captured once into a hipGraph and replayed afterwards.
    unsigned long long key[4] = {((unsigned long long)(unsigned)W << 32) | (unsigned)H,
                                 ((unsigned long long)(unsigned)n_levels << 40) | ((unsigned long long)(unsigned)term_type << 32) |
                                     ((unsigned long long)(unsigned)depth_fmt << 16) | (unsigned)t->lds_cap,
                                 0ull, (unsigned long long)(uintptr_t)t->pyr ^ ((unsigned long long)(uintptr_t)t->pair_p << 1)};
    for (int l = 0; l < n_levels; ++l) key[2] = key[2] * 1000003ull + (unsigned long long)(unsigned)iters_per_level[l] + 1ull;
    const bool graph_path = t->graph_ok && mem == OP_MEM_DEVICE && !want_point_corr && t->sums == OP_TRACK_SUMS_FP64; // the validation mode synchronises every iteration: not capturable
    if (graph_path && t->graph_exec && std::memcmp(key, t->graph_key, sizeof(key)) == 0) {
        OP_HIP(hipGraphLaunch(t->graph_exec, t->stream));
        t->pending = true; t->pending_logs = false; t->pending_points = false;
        return OP_OK;
    }
    bool capturing = false;
    if (graph_path) {
        if (t->graph_exec) { (void)hipGraphExecDestroy(t->graph_exec); t->graph_exec = nullptr; }
        if (hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
        else { (void)hipGetLastError(); t->graph_ok = 0; }
    }
    auto enqueue_all = [&]() -> int {
        const int n_wg0 = (int)((np + kThreads - 1) / kThreads);
        OP_HIP(hipMemcpyAsync(t->prep_dev, t->prep_host, sizeof(PrepFrames), hipMemcpyHostToDevice, t->stream));
        launch_prep_convert_blur(dim3((W + kBlurTx - 1) / kBlurTx, (H + kBlurTy - 1) / kBlurTy, 4), dim3(kBlurTx * kBlurTy), t->stream,
                           (const PrepFrames*)t->prep_dev);
        // NormalizeIntensity over the identity-pose correspondences of level 0 (Odometry.cpp:543-544).  Its state header
        // is uploaded from its own pinned buffer (st_host_norm), the loop's from st_host: no host-side wait in between.
        OP_HIP(hipMemcpyAsync(t->st, t->st_host_norm, offsetof(TrackState, per_iter_count), hipMemcpyHostToDevice, t->stream));
        const IterGeom g = iter_geom(t, np);
        hipLaunchKernelGGL(k_track_assoc, dim3(n_wg0), dim3(kThreads), 0, t->stream, t->st, 0, t->pair_p, t->code);
        launch_iter<3>(t, 0, g);
        if (t->sums != OP_TRACK_SUMS_FP64) {
            // reference-order mode: NormalizeIntensity's two means summed like the reference does (DenseOdometryFunction.cpp:131-141):
            // sequentially in float32 over the identity-pose pairs in raster order -- by k_seq_sums, or (_HOST variant) on one host thread
            float mean_s = 0.0f, mean_t = 0.0f;
            size_t cnt = 0;
            if (t->sums == OP_TRACK_SUMS_REFERENCE_F32 && t->seq_ok) {
                if (!t->seq_out) {
                    OP_HIP(hipMalloc(&t->seq_out, 64 * sizeof(float)));
                    OP_HIP(hipHostMalloc(&t->seq_host, 64 * sizeof(float), hipHostMallocDefault));
                    OP_HIP(hipMalloc(&t->seq_total, sizeof(unsigned)));
                }
                float* pairs2 = reinterpret_cast<float*>(t->pix_out); // 2 floats per accepted pixel; pix_out (16 B per pixel) is idle until the run's final emit
                hipLaunchKernelGGL(k_rows_count, dim3(n_wg0), dim3(kThreads), 0, t->stream, t->st, 0, (const int*)t->pair_t, t->wg_count);
                launch_emit_scan(dim3(1), dim3(1024), t->stream, t->st, t->wg_count, n_wg0, t->seq_total);
                hipLaunchKernelGGL(k_norm_pairs_compact, dim3(n_wg0), dim3(kThreads), 0, t->stream, (const float*)pyr_image(t, 0, 0, 0), (const float*)pyr_image(t, 1, 0, 0), (int)np,
                                   (const int*)t->pair_t, (const unsigned*)t->wg_count, pairs2);
                hipLaunchKernelGGL((k_seq_sums<2, 2, 1>), dim3(1), dim3(kSeqThreads), seq_lds_bytes(2, 2, 1), t->stream, (const float*)pairs2, (const unsigned*)t->seq_total, t->seq_out);
                OP_HIP(hipGetLastError());
                OP_HIP(hipMemcpyAsync(t->seq_host, t->seq_out, 3 * sizeof(float), hipMemcpyDeviceToHost, t->stream));
                OP_HIP(hipStreamSynchronize(t->stream));
                unsigned n32;
                std::memcpy(&n32, t->seq_host + 2, sizeof(n32));
                mean_s = t->seq_host[0]; mean_t = t->seq_host[1]; cnt = n32;
            } else {
                std::vector<int> pt(np);
                std::vector<float> gs(np), gt(np);
                OP_HIP(hipMemcpyAsync(pt.data(), t->pair_t, np * sizeof(int), hipMemcpyDeviceToHost, t->stream));
                OP_HIP(hipMemcpyAsync(gs.data(), pyr_image(t, 0, 0, 0), np * sizeof(float), hipMemcpyDeviceToHost, t->stream));
                OP_HIP(hipMemcpyAsync(gt.data(), pyr_image(t, 1, 0, 0), np * sizeof(float), hipMemcpyDeviceToHost, t->stream));
                OP_HIP(hipStreamSynchronize(t->stream));
                for (size_t k = 0; k < np; ++k)
                    if (pt[k] >= 0) { mean_s += gs[k]; mean_t += gt[(size_t)pt[k]]; ++cnt; }
            }
            mean_s /= (float)cnt; mean_t /= (float)cnt;
            const float sc[2] = {(float)(0.5 / (double)mean_s), (float)(0.5 / (double)mean_t)};
            OP_HIP(hipMemcpyAsync(t->norm_scales, sc, sizeof(sc), hipMemcpyHostToDevice, t->stream));
            OP_HIP(hipStreamSynchronize(t->stream)); // `sc` is a stack buffer
        } else
        launch_norm_scales(dim3(1), dim3(1024), t->stream, t->partials, g.n_wg, t->norm_scales);
        launch_norm_apply(dim3(n_wg0, 2), dim3(kThreads), t->stream, pyr_image(t, 0, 0, 0), pyr_image(t, 1, 0, 0), (int)np,
                           t->norm_scales);
        for (int l = 0; l < n_levels; ++l) {
            const int w = W >> l, hh = H >> l;
            if (l > 0) {
                PrepImages D;
                for (int f = 0; f < 2; ++f)
                    for (int k = 0; k < 2; ++k) { D.in[f * 2 + k] = pyr_image(t, f, k, l - 1); D.out[f * 2 + k] = pyr_image(t, f, k, l); }
                D.w = W >> (l - 1); D.h = H >> (l - 1);
                launch_prep_pyrdown(dim3((unsigned)(((size_t)w * hh + kThreads - 1) / kThreads), 4), dim3(kThreads), t->stream, D);
            }
            PrepImages S;
            S.in[0] = pyr_image(t, 1, 0, l); S.in[1] = pyr_image(t, 1, 1, l); S.in[2] = S.in[3] = nullptr;
            S.out[0] = pyr_image(t, 1, 2, l); S.out[1] = pyr_image(t, 1, 3, l); S.out[2] = pyr_image(t, 1, 4, l); S.out[3] = pyr_image(t, 1, 5, l);
            S.w = w; S.h = hh;
            launch_prep_sobel(dim3((unsigned)(((size_t)w * hh + kThreads - 1) / kThreads), 4), dim3(kThreads), t->stream, S);
        }
        return track_enqueue(t, n_levels, iters_per_level, W, H, term_type, init_T, want_point_corr != 0, false);
    };
    const int rc = enqueue_all();
    if (capturing) {
        hipGraph_t graph = nullptr;
        const hipError_t ec = hipStreamEndCapture(t->stream, &graph);
        t->pending = false;
        bool ok = rc == OP_OK && ec == hipSuccess && graph != nullptr;
        if (ok) ok = hipGraphInstantiate(&t->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (ok) {
            std::memcpy(t->graph_key, key, sizeof(key));
            OP_HIP(hipGraphLaunch(t->graph_exec, t->stream));
            t->pending = true; t->pending_logs = false; t->pending_points = false;
            return OP_OK;
        }
        // capture is not usable here: issue the launches directly from now on
        (void)hipGetLastError();
        t->graph_exec = nullptr; t->graph_ok = 0;
        return enqueue_all();
    }
    return rc;
}

int op_tracker_wait(op_tracker* t, op_track_result* result, int32_t* pixel_corr, float* point_corr, size_t corr_cap) {
    if (!t || !result) return fail(OP_ERR_INVALID, "op_tracker_wait: NULL argument");
    OP_TRY(use_device(t->device));
    if (t->worker_active) { // a reference-order run on the tracker's own host thread
        t->worker.join();
        t->worker_active = false;
        if (t->worker_rc != OP_OK) { t->pending = false; return fail(t->worker_rc, "%s", t->worker_err); }
    }
    return track_finish(t, result, pixel_corr, point_corr, corr_cap, nullptr, nullptr);
}

int op_tracker_dense_tracking(op_tracker* t, const op_camera* cam, int n_levels, const int32_t* iters_per_level, const uint8_t* source_rgb,
                              const uint8_t* target_rgb, const void* source_depth, const void* target_depth, int depth_fmt,
                              const float init_T[16], int term_type, int mem, op_track_result* result, int32_t* pixel_corr,
                              float* point_corr, size_t corr_cap) {
    if (!result) return fail(OP_ERR_INVALID, "op_tracker_dense_tracking: NULL argument");
    OP_TRY(op_tracker_dense_tracking_enqueue(t, cam, n_levels, iters_per_level, source_rgb, target_rgb, source_depth, target_depth, depth_fmt,
                                             init_T, term_type, mem, point_corr != nullptr));
    return op_tracker_wait(t, result, pixel_corr, point_corr, corr_cap);
}

int op_tracker_read_pyramid(op_tracker* t, int frame, int kind, int level, float* out, size_t cap) {
    if (!t || !out) return fail(OP_ERR_INVALID, "op_tracker_read_pyramid: NULL argument");
    if (t->pending || t->worker_active) return fail(OP_ERR_INVALID, "op_tracker_read_pyramid: an enqueued run has not been waited for");
    if (!t->pyr || frame < 0 || frame > 1 || kind < 0 || kind > 5 || level < 0 || level >= t->pyr_levels || (frame == 0 && kind > 1))
        return fail(OP_ERR_INVALID, "op_tracker_read_pyramid: no such image (frame %d kind %d level %d)", frame, kind, level);
    const size_t n = (size_t)(t->pyr_w >> level) * (t->pyr_h >> level);
    if (cap < n) return fail(OP_ERR_CAPACITY, "op_tracker_read_pyramid: cap %zu < %zu", cap, n);
    OP_TRY(use_device(t->device));
    OP_HIP(hipMemcpyAsync(out, pyr_image(t, frame, kind, level), n * sizeof(float), hipMemcpyDeviceToHost, t->stream));
    OP_HIP(hipStreamSynchronize(t->stream));
    return OP_OK;
}

int op_tracker_correspondences(op_tracker* t, const op_track_level* level, const float T[16], int mem, int32_t* pixel_corr,
                               size_t corr_cap, size_t* n) {
    if (!t || !level || !T || !n) return fail(OP_ERR_INVALID, "op_tracker_correspondences: NULL argument");
    // The acceptance pass lives in k_track_iter, so run exactly one iteration of one level (its pose
    // update is discarded) and emit that iteration's pairs.
    op_track_result res;
    const int32_t iters[1] = {1};
    OP_TRY(op_tracker_track(t, level, 1, iters, level->width, level->height, OP_TRACK_DEPTH, T, mem, &res, pixel_corr, nullptr,
                            corr_cap, nullptr, nullptr));
    *n = (size_t)res.n_correspondences;
    return OP_OK;
}

int op_track_projection(const float cam4[4], const float T[16], float KRK_inv[9], float Kt[3]) {
    if (!cam4 || !T || !KRK_inv || !Kt) return fail(OP_ERR_INVALID, "op_track_projection: NULL argument");
    op_host::track_projection(cam4[0], cam4[1], cam4[2], cam4[3], T, KRK_inv, Kt);
    return OP_OK;
}

int op_ldlt_solve6(const double JTJ[36], const double JTr[6], float x[6]) {
    if (!JTJ || !JTr || !x) return fail(OP_ERR_INVALID, "op_ldlt_solve6: NULL argument");
    op_host::ldlt_solve6(JTJ, JTr, x);
    return OP_OK;
}

int op_dense_track(const op_track_level* levels, int n_levels, const int32_t* iters_per_level, int full_width, int full_height,
                   int term_type, const float init_T[16], int mem, int device, op_track_result* result, int32_t* pixel_corr,
                   float* point_corr, size_t corr_cap) {
    op_tracker* t = nullptr;
    OP_TRY(op_tracker_create(device, &t));
    const int rc = op_tracker_track(t, levels, n_levels, iters_per_level, full_width, full_height, term_type, init_T, mem, result,
                                    pixel_corr, point_corr, corr_cap, nullptr, nullptr);
    op_tracker_destroy(t);
    return rc;
}

} // extern "C"
