// odometry_core.hpp -- what the translation units of the dense RGB-D tracker share: constants, the device-side state of a track (TrackState), the argument
// structs of the image preparation, and the launchers of the kernels that live outside odometry.hip.
//   odometry.hip        the tracker: association + acceptance chain + Jacobian rows + sums + solve (k_track_*: the iteration family), the reference-order row
//                       compaction, the host object and the C-ABI (op_tracker_*, op_dense_track)
//   odometry_prep.hip   Odometry::DenseTracking's image preparation (k_prep_convert_blur / pyrdown / sobel) and NormalizeIntensity's scale (k_norm_*)
//   odometry_emit.hip   correspondence_set / pixel_correspondence_set / rmse of the last executed iteration (k_emit_*)
#pragma once
#include <array>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"
#include "host_math.hpp"

namespace opt {

constexpr int kMaxLevels = 8;
constexpr int kMaxIters = 256;      // total iterations over all levels
constexpr int kThreads = 256;
constexpr int kNSums = 32;          // [0..20] JTJ upper triangle, [21..26] JTr, [27] sum r^2, [28] count

struct LevelDev {
    int w, h;
    float fx, fy, cx, cy;
    const float *sc, *sd, *tc, *td, *tcdx, *tcdy, *tddx, *tddy;
};

struct TrackState {
    float T[16];
    LevelDev lv[kMaxLevels];
    int full_w, full_h, term;
    int stop_level;                  // level whose remaining iterations are skipped (-1: none)
    int iters_done;
    int last_level;                  // level of the last executed iteration (-1: none)
    unsigned long long n_last;       // its correspondence count
    unsigned long long n_emit;
    double rmse;
    int success;
    int per_iter_count[kMaxIters];
    float per_iter_T[kMaxIters * 16];
};

__device__ __forceinline__ float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct PrepFrames {
    const unsigned char* rgb[2];
    const void* depth[2];
    int is_u16;
    float depth_scale;
    int w, h;
    float* out[4];      // src gray, tgt gray, src depth, tgt depth (level 0)
};

struct PrepImages { const float* in[4]; float* out[4]; int w, h; }; // w, h of the INPUT images

constexpr int kBlurTx = 32, kBlurTy = 8;

// launchers (one per kernel of odometry_prep.hip / odometry_emit.hip; same arguments as the kernel)
void launch_prep_convert_blur(dim3 grid, dim3 block, hipStream_t stream, const PrepFrames* Pp);
void launch_prep_pyrdown(dim3 grid, dim3 block, hipStream_t stream, PrepImages P);
void launch_prep_sobel(dim3 grid, dim3 block, hipStream_t stream, PrepImages P);
void launch_norm_scales(dim3 grid, dim3 block, hipStream_t stream, const double* partials, int n_partials, float* scales);
void launch_norm_apply(dim3 grid, dim3 block, hipStream_t stream, float* gs, float* gt, int npix, const float* scales);
void launch_emit_count(dim3 grid, dim3 block, hipStream_t stream, const TrackState* st, const int* pair_t, unsigned* wg_count);
void launch_emit_scan(dim3 grid, dim3 block, hipStream_t stream, TrackState* st, unsigned* wg_count, int n_wg, unsigned* total_out = nullptr);
void launch_emit_scatter(dim3 grid, dim3 block, hipStream_t stream, const TrackState* st, const int* pair_t, const unsigned* wg_off, int4* pix_out, float* pts_out, double* partials);
void launch_emit_finish(dim3 grid, dim3 block, hipStream_t stream, TrackState* st, const double* partials, int n_wg);

} // namespace opt
