// odometry_emit.hip -- the result sets of a track (odometry_core.hpp lists the translation units).
#include "odometry_core.hpp"

using namespace op;
using namespace opt;

namespace {

// ---- correspondence_set / pixel_correspondence_set / rmse (Odometry.cpp:676-687, :606) ---------
// Ordered (raster) compaction of the last executed iteration's accepted pixels.
__global__ __launch_bounds__(kThreads) void k_emit_count(const TrackState* __restrict__ st, const int* __restrict__ pair_t,
                                                         unsigned* __restrict__ wg_count) {
    __shared__ unsigned s_c[kThreads / 64];
    const int ll = st->last_level;
    const int npix = ll < 0 ? 0 : st->lv[ll].w * st->lv[ll].h;
    const int s = blockIdx.x * kThreads + threadIdx.x;
    const bool a = s < npix && pair_t[s] >= 0;
    const unsigned long long m = __ballot(a);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) wg_count[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ __launch_bounds__(1024) void k_emit_scan(TrackState* __restrict__ st, unsigned* __restrict__ wg_count, int n_wg, unsigned* __restrict__ total_out = nullptr) {
    // exclusive scan of n_wg counts by one workgroup (n_wg <= a few thousand), in place
    __shared__ unsigned s_part[1024];
    const int per = (n_wg + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, n_wg);
    unsigned sum = 0;
    for (int i = lo; i < hi; ++i) sum += wg_count[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = s_part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) { const unsigned c = wg_count[i]; wg_count[i] = run; run += c; }
    if (threadIdx.x == 1023) { st->n_emit = s_part[1023]; if (total_out) *total_out = s_part[1023]; }
}

__global__ __launch_bounds__(kThreads) void k_emit_scatter(const TrackState* __restrict__ st, const int* __restrict__ pair_t,
                                                           const unsigned* __restrict__ wg_off,
                                                           int4* __restrict__ pix_out, float* __restrict__ pts_out,
                                                           double* __restrict__ partials) {
    __shared__ unsigned s_c[kThreads / 64];
    __shared__ double s_e[kThreads / 64];
    const int ll = st->last_level;
    const int npix = ll < 0 ? 0 : st->lv[ll].w * st->lv[ll].h;
    const int s = blockIdx.x * kThreads + threadIdx.x;
    const bool a = s < npix && pair_t[s] >= 0;
    const unsigned long long m = __ballot(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_c[wave] = (unsigned)__popcll(m);
    __syncthreads();
    double e = 0.0;
    if (a) {
        unsigned idx = wg_off[blockIdx.x] + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) idx += s_c[w];
        const int W = st->lv[ll].w;
        const int v_s = s / W, u_s = s - v_s * W, t = pair_t[s];
        pix_out[idx] = make_int4(v_s, u_s, t / W, t - (t / W) * W);
        // source / target image_xyz of LEVEL 0, both at the SOURCE pixel (Odometry.cpp:676-683)
        const LevelDev& L0 = st->lv[0];
        const size_t o = (size_t)v_s * L0.w + u_s;
        const float zs = L0.sd[o], zt = L0.td[o];
        float p[3] = {-1.0f, -1.0f, -1.0f}, q[3] = {-1.0f, -1.0f, -1.0f};
        if (zs > 0) { p[0] = ((float)u_s - L0.cx) * zs / L0.fx; p[1] = ((float)v_s - L0.cy) * zs / L0.fy; p[2] = zs; }
        if (zt > 0) { q[0] = ((float)u_s - L0.cx) * zt / L0.fx; q[1] = ((float)v_s - L0.cy) * zt / L0.fy; q[2] = zt; }
        if (pts_out) {
            float* o6 = pts_out + (size_t)idx * 6;
            o6[0] = p[0]; o6[1] = p[1]; o6[2] = p[2]; o6[3] = q[0]; o6[4] = q[1]; o6[5] = q[2];
        }
        // ComputeReprojectionError3D (Geometry.cpp:48-61): (T*(p,1)).head<3>()/w - q, squaredNorm
        const float* T = st->T;
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = ((T[r * 4] * p[0] + T[r * 4 + 1] * p[1]) + T[r * 4 + 2] * p[2]) + T[r * 4 + 3] * 1.0f;
        const float e0 = h[0] / h[3] - q[0], e1 = h[1] / h[3] - q[1], e2 = h[2] / h[3] - q[2];
        e = (double)sum3(e0 * e0, e1 * e1, e2 * e2);
    }
    e = wave_sum_d(e);
    if (lane == 0) s_e[wave] = e;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_e[0] + s_e[1]) + (s_e[2] + s_e[3]);
}

__global__ __launch_bounds__(1024) void k_emit_finish(TrackState* __restrict__ st, const double* __restrict__ partials, int n_wg) {
    __shared__ double s_w[16];
    double v = 0;
    for (int i = threadIdx.x; i < n_wg; i += 1024) v += partials[i];
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < 16; ++w) t += s_w[w];
        const unsigned long long n = st->last_level < 0 ? 0ull : st->n_last;
        st->rmse = sqrt(t / (double)n);                      // n == 0 -> NaN, as the reference's 0/0
        st->success = (double)((float)n / (float)(st->full_h * st->full_w)) >= 0.3 ? 1 : 0; // MIN_INLIER_RATIO_DENSE
    }
}

} // namespace

namespace opt {

void launch_emit_count(dim3 grid, dim3 block, hipStream_t stream, const TrackState* st, const int* pair_t, unsigned* wg_count) {
    hipLaunchKernelGGL(k_emit_count, grid, block, 0, stream, st, pair_t, wg_count);
}

void launch_emit_scan(dim3 grid, dim3 block, hipStream_t stream, TrackState* st, unsigned* wg_count, int n_wg, unsigned* total_out) {
    hipLaunchKernelGGL(k_emit_scan, grid, block, 0, stream, st, wg_count, n_wg, total_out);
}

void launch_emit_scatter(dim3 grid, dim3 block, hipStream_t stream, const TrackState* st, const int* pair_t, const unsigned* wg_off, int4* pix_out, float* pts_out, double* partials) {
    hipLaunchKernelGGL(k_emit_scatter, grid, block, 0, stream, st, pair_t, wg_off, pix_out, pts_out, partials);
}

void launch_emit_finish(dim3 grid, dim3 block, hipStream_t stream, TrackState* st, const double* partials, int n_wg) {
    hipLaunchKernelGGL(k_emit_finish, grid, block, 0, stream, st, partials, n_wg);
}

} // namespace opt
