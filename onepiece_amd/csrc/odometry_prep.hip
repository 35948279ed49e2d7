// odometry_prep.hip -- image preparation of Odometry::DenseTracking on the device (odometry_core.hpp lists the translation units).
#include "odometry_core.hpp"

using namespace op;
using namespace opt;

namespace {

// ---- image preparation of Odometry::DenseTracking (Odometry.cpp:436-449,609-620) -----------------
// The reference delegates this stage to OpenCV (cvtColor, GaussianBlur 3x3, pyrDown, Sobel 3x3), which it
// does not vendor: the kernels below implement OpenCV's published kernels / BORDER_REFLECT_101 in float
// (horizontal pass, then vertical, taps accumulated in order) and are checked against the restatement
// of the same definitions in oracle/ -- not against OpenCV.  ConvertDepthTo32FNaN, the /255 intensity
// scale and NormalizeIntensity are reference code (DenseOdometryFunction.cpp:28-71,129-145).
__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return i < 0 ? 0 : i;
}

// z: 0/1 = grey of frame 0/1, 2/3 = depth of frame 0/1.  The frame pointers are picked with selects (indexing the
// by-value struct with a run-time z would move it to scratch memory).
__device__ __forceinline__ float prep_raw(const PrepFrames& P, int z, int y, int x) {
    const size_t k = (size_t)y * P.w + x;
    if (z < 2) {
        const unsigned char* c = (z == 0 ? P.rgb[0] : P.rgb[1]) + 3 * k;
        const int g = (c[0] * 4899 + c[1] * 9617 + c[2] * 1868 + 8192) >> 14; // cvtColor RGB2GRAY, 8-bit
        return (float)(g & 255) / 255.0f;
    }
    const void* dp = z == 2 ? P.depth[0] : P.depth[1];
    if (P.is_u16) {
        const unsigned short d = static_cast<const unsigned short*>(dp)[k];
        return ((double)d > 0.5 * (double)P.depth_scale && (float)d < 4.0f * P.depth_scale) ? (float)d / P.depth_scale : __builtin_nanf("");
    }
    const float d = static_cast<const float*>(dp)[k];
    return ((double)d > 0.5 && d < 4.0f) ? d : __builtin_nanf("");
}

// conversion + GaussianBlur(3x3, sigma 0) = [1 2 1]/4 separable, for the four level-0 images at once.
// 32 x 8 output tile per workgroup: the (32+2) x (8+2) converted input values are staged in LDS once (the grey
// conversion alone is three byte loads + integer math per tap otherwise), the nine taps then come from LDS.
__global__ __launch_bounds__(kBlurTx * kBlurTy) void k_prep_convert_blur(const PrepFrames* __restrict__ Pp) {
    const PrepFrames P = *Pp;   // per-call frame pointers live in device memory so that a captured graph can be replayed
    __shared__ float s_in[kBlurTy + 2][kBlurTx + 2];
    const int z = blockIdx.z;
    const int x0 = blockIdx.x * kBlurTx, y0 = blockIdx.y * kBlurTy;
    for (int k = threadIdx.x; k < (kBlurTy + 2) * (kBlurTx + 2); k += kBlurTx * kBlurTy) {
        const int ly = k / (kBlurTx + 2), lx = k - ly * (kBlurTx + 2);
        // BORDER_REFLECT_101 on the image, clamped for the part of the tile that hangs over the image
        const int yy = reflect101(min(y0 + ly - 1, P.h), P.h), xx = reflect101(min(x0 + lx - 1, P.w), P.w);
        s_in[ly][lx] = prep_raw(P, z, yy, xx);
    }
    __syncthreads();
    const int lx = threadIdx.x % kBlurTx, ly = threadIdx.x / kBlurTx;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= P.w || y >= P.h) return;
    float hrow[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) hrow[r] = (0.25f * s_in[ly + r][lx] + 0.5f * s_in[ly + r][lx + 1]) + 0.25f * s_in[ly + r][lx + 2];
    float* out = z == 0 ? P.out[0] : (z == 1 ? P.out[1] : (z == 2 ? P.out[2] : P.out[3]));
    out[(size_t)y * P.w + x] = (0.25f * hrow[0] + 0.5f * hrow[1]) + 0.25f * hrow[2];
}

// pyrDown to (w/2, h/2): [1 4 6 4 1]/16 separable at the even samples, four images at once
__global__ __launch_bounds__(kThreads) void k_prep_pyrdown(PrepImages P) {
    const int z = blockIdx.y, w2 = P.w / 2, h2 = P.h / 2, s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= w2 * h2) return;
    const int y = s / w2, x = s - y * w2;
    const float k[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    const float* in = P.in[z];
    int xs[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) xs[c] = reflect101(2 * x - 2 + c, P.w);
    float v = 0.0f;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const float* row = in + (size_t)reflect101(2 * y - 2 + r, P.h) * P.w;
        float hsum = k[0] * row[xs[0]];
#pragma unroll
        for (int c = 1; c < 5; ++c) hsum = hsum + k[c] * row[xs[c]];
        v = r == 0 ? k[0] * hsum : v + k[r] * hsum;
    }
    P.out[z][s] = v;
}

// Sobel 3x3: z = 0/1 -> d/dx, d/dy of in[0] into out[0], out[1]; z = 2/3 -> of in[1] into out[2], out[3]
__global__ __launch_bounds__(kThreads) void k_prep_sobel(PrepImages P) {
    const int z = blockIdx.y, s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= P.w * P.h) return;
    const int y = s / P.w, x = s - y * P.w;
    const float* in = P.in[z >> 1];
    const float dk[3] = {-1.0f, 0.0f, 1.0f}, sk[3] = {1.0f, 2.0f, 1.0f};
    const bool dx = (z & 1) == 0;
    const int xs[3] = {reflect101(x - 1, P.w), x, reflect101(x + 1, P.w)};
    float v = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* row = in + (size_t)reflect101(y - 1 + r, P.h) * P.w;
        const float kx0 = dx ? dk[0] : sk[0], kx1 = dx ? dk[1] : sk[1], kx2 = dx ? dk[2] : sk[2];
        const float hsum = (kx0 * row[xs[0]] + kx1 * row[xs[1]]) + kx2 * row[xs[2]];
        const float ky = dx ? sk[r] : dk[r];
        v = r == 0 ? ky * hsum : v + ky * hsum;
    }
    P.out[z][s] = v;
}

// NormalizeIntensity (DenseOdometryFunction.cpp:129-145): means over the identity-pose pairs (summed in
// double here, sequentially in float there), scale = float(0.5 / mean), img = img * scale + 0.
__global__ __launch_bounds__(1024) void k_norm_scales(const double* __restrict__ partials, int n_partials, float* __restrict__ scales) {
    __shared__ double s_w[3][16];
    double a = 0, b = 0, n = 0;
    for (int i = threadIdx.x; i < n_partials; i += 1024) {
        a += partials[(size_t)i * kNSums + 0]; b += partials[(size_t)i * kNSums + 1]; n += partials[(size_t)i * kNSums + 28];
    }
    a = wave_sum_d(a); b = wave_sum_d(b); n = wave_sum_d(n);
    if ((threadIdx.x & 63) == 0) { s_w[0][threadIdx.x >> 6] = a; s_w[1][threadIdx.x >> 6] = b; s_w[2][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0, tb = 0, tn = 0;
        for (int w = 0; w < 16; ++w) { ta += s_w[0][w]; tb += s_w[1][w]; tn += s_w[2][w]; }
        const float ms = (float)ta / (float)tn, mt = (float)tb / (float)tn;
        scales[0] = (float)(0.5 / (double)ms);
        scales[1] = (float)(0.5 / (double)mt);
    }
}
__global__ __launch_bounds__(kThreads) void k_norm_apply(float* __restrict__ gs, float* __restrict__ gt, int npix, const float* __restrict__ scales) {
    const int s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= npix) return;
    float* img = blockIdx.y ? gt : gs;
    img[s] = img[s] * scales[blockIdx.y] + 0.0f;
}

} // namespace

namespace opt {

void launch_prep_convert_blur(dim3 grid, dim3 block, hipStream_t stream, const PrepFrames* Pp) {
    hipLaunchKernelGGL(k_prep_convert_blur, grid, block, 0, stream, Pp);
}

void launch_prep_pyrdown(dim3 grid, dim3 block, hipStream_t stream, PrepImages P) {
    hipLaunchKernelGGL(k_prep_pyrdown, grid, block, 0, stream, P);
}

void launch_prep_sobel(dim3 grid, dim3 block, hipStream_t stream, PrepImages P) {
    hipLaunchKernelGGL(k_prep_sobel, grid, block, 0, stream, P);
}

void launch_norm_scales(dim3 grid, dim3 block, hipStream_t stream, const double* partials, int n_partials, float* scales) {
    hipLaunchKernelGGL(k_norm_scales, grid, block, 0, stream, partials, n_partials, scales);
}

void launch_norm_apply(dim3 grid, dim3 block, hipStream_t stream, float* gs, float* gt, int npix, const float* scales) {
    hipLaunchKernelGGL(k_norm_apply, grid, block, 0, stream, gs, gt, npix, scales);
}

} // namespace opt
