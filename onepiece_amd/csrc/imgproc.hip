// imgproc.hip -- the depth preprocessing every fusion driver of the reference runs right before
// CubeHandler::IntegrateImage (example/ImageSequenceIntegration.cpp:36-38, DenseFusion.cpp:92-94, ...):
//
//     tool::ConvertDepthTo32F(depth, refined_depth, camera.GetDepthScale());   Tool/ImageProcessing.cpp:68-91
//     tool::BilateralFilter(refined_depth, filtered_depth);                    Tool/ImageProcessing.cpp:64-67
//                                                         = cv::bilateralFilter(src, dst, 7, 0.03, 4.5)
//
// OpenCV is not vendored by the reference, so the filter is implemented from cv::bilateralFilter's documented
// definition for CV_32FC1 (see include/onepiece_hip.h, op_bilateral_filter_depth) and is outside the pinned
// parity claim; the CPU restatement of the same definition is oracle/onepiece_oracle.c:orc_bilateral_filter.
//
// One kernel: a workgroup stages a (16 + 2R) x (64 + 2R) window (BORDER_REFLECT_101, uint16 -> metres folded in)
// in LDS; each thread produces 4 pixels of a 64 x 16 tile.  The kernel is VALU-bound (29 taps x one exponential per
// pixel at d = 7), so the two Gaussian factors are folded into ONE base-2 exponential per tap,
//     w = 2^(log2(e) * (-(i^2 + j^2) / (2 sigma_space^2)) + log2(e) * (-(dv)^2 / (2 sigma_color^2))),
// evaluated by the hardware v_exp_f32 (1 ulp); the spatial exponents of the default radius are compile-time
// multiples of one scalar.  A weight error of 1e-6 relative moves a weighted mean of values that lie within a few
// sigma_color of each other by < 1e-7 m, far inside the 2e-6 bar of tests/test_imgproc_gpu.py.
#include "common.hpp"

namespace {
using op::fail;

constexpr int kTileW = 64, kTileH = 16, kRowsPerThread = 4; // 256 threads
constexpr int kMaxRadius = 15;

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

struct BilateralArgs {
    const void* src;
    float* dst;
    int width, height;
    int u16;
    float depth_scale;
    int radius;
    float gauss_color, gauss_space; // log2(e) * (-0.5 / sigma^2)
};

template <int R> // R > 0: compile-time radius (unrolled); R == 0: run-time radius
__global__ __launch_bounds__(256) void k_bilateral(BilateralArgs A) {
    extern __shared__ float s_mem[];
    const int rad = R > 0 ? R : A.radius;
    const int win_w = kTileW + 2 * rad, win_h = kTileH + 2 * rad, ksz = 2 * rad + 1;
    float* s_win = s_mem;
    float* s_exp = s_mem + win_w * win_h; // run-time radius only: spatial exponent per tap, -inf outside the disc
    const int tid = threadIdx.y * kTileW + threadIdx.x;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const size_t img = (size_t)blockIdx.z * A.width * A.height;
    if (R == 0)
        for (int k = tid; k < ksz * ksz; k += 256) {
            const int i = k / ksz - rad, j = k % ksz - rad;
            const int r2 = i * i + j * j;
            s_exp[k] = r2 > rad * rad ? -INFINITY : (float)r2 * A.gauss_space;
        }
    for (int k = tid; k < win_w * win_h; k += 256) {
        const int wy = k / win_w, wx = k - wy * win_w;
        const int sy = reflect101(y0 + wy - rad, A.height), sx = reflect101(x0 + wx - rad, A.width);
        const size_t p = img + (size_t)sy * A.width + sx;
        // ConvertDepthTo32F: (float)u16 / depth_scale, float input passes through
        s_win[k] = A.u16 ? (float)((const unsigned short*)A.src)[p] / A.depth_scale : ((const float*)A.src)[p];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= A.width) return;
#pragma unroll
    for (int q = 0; q < kRowsPerThread; ++q) {
        const int ty = threadIdx.y + q * (kTileH / kRowsPerThread), y = y0 + ty;
        if (y >= A.height) continue;
        const float* c = s_win + (ty + rad) * win_w + threadIdx.x + rad;
        const float v0 = *c;
        float sum = 0.0f, wsum = 0.0f;
        if (R > 0) {
#pragma unroll
            for (int i = -R; i <= R; ++i)
#pragma unroll
                for (int j = -R; j <= R; ++j) {
                    if (i * i + j * j > R * R) continue;
                    const float v = c[i * win_w + j];
                    const float w = __builtin_amdgcn_exp2f((float)(i * i + j * j) * A.gauss_space + (v - v0) * (v - v0) * A.gauss_color);
                    sum += v * w;
                    wsum += w;
                }
        } else {
            for (int i = -rad; i <= rad; ++i)
                for (int j = -rad; j <= rad; ++j) {
                    const float v = c[i * win_w + j];
                    const float w = __builtin_amdgcn_exp2f(s_exp[(i + rad) * ksz + (j + rad)] + (v - v0) * (v - v0) * A.gauss_color);
                    sum += v * w;
                    wsum += w;
                }
        }
        A.dst[img + (size_t)y * A.width + x] = sum / wsum;
    }
}

struct DevStream { // one lazily created stream per device for calls that pass no stream
    hipStream_t s[16] = {};
};
DevStream g_streams;

} // namespace

extern "C" int op_bilateral_filter_depth(const void* depth, int depth_format, float depth_scale, int width, int height, int n_images, int d,
                                         float sigma_color, float sigma_space, int mem, int device, void* stream, float* out) {
    if (!depth || !out) return fail(OP_ERR_INVALID, "null argument");
    if (width <= 0 || height <= 0 || n_images < 0 || (long long)width * height > (1LL << 30)) return fail(OP_ERR_INVALID, "invalid image size");
    if (depth_format != OP_DEPTH_F32 && depth_format != OP_DEPTH_U16) return fail(OP_ERR_INVALID, "unknown depth format %d", depth_format);
    if (depth_format == OP_DEPTH_U16 && !(depth_scale > 0)) return fail(OP_ERR_INVALID, "depth_scale must be positive");
    if (mem != OP_MEM_HOST && mem != OP_MEM_DEVICE) return fail(OP_ERR_INVALID, "unknown memory kind %d", mem);
    if (mem == OP_MEM_HOST && stream) return fail(OP_ERR_INVALID, "a stream can only be given with device buffers");
    // cv::bilateralFilter: sigma <= 0 -> 1; d <= 0 -> radius = round(sigma_space * 1.5); radius >= 1
    if (!(sigma_color > 0)) sigma_color = 1.0f;
    if (!(sigma_space > 0)) sigma_space = 1.0f;
    int radius = d <= 0 ? (int)lrint((double)sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    if (radius > kMaxRadius) return fail(OP_ERR_INVALID, "filter diameter %d exceeds the supported %d", d, 2 * kMaxRadius + 1);
    OP_TRY(op::use_device(device));
    if (n_images == 0) return OP_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!st) {
        if (device >= 16) return fail(OP_ERR_INVALID, "device %d out of range", device);
        if (!g_streams.s[device]) OP_HIP(hipStreamCreateWithFlags(&g_streams.s[device], hipStreamNonBlocking));
        st = g_streams.s[device];
    }
    const size_t npix = (size_t)width * height * n_images;
    const size_t in_bytes = npix * (depth_format == OP_DEPTH_U16 ? 2 : 4);
    void* d_in = nullptr;
    float* d_out = nullptr;
    BilateralArgs A{};
    if (mem == OP_MEM_HOST) {
        OP_HIP(op::cached_malloc(&d_in, in_bytes));
        hipError_t e = op::cached_malloc((void**)&d_out, npix * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(d_in, depth, in_bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            op::cached_free(d_in);
            if (d_out) op::cached_free(d_out);
            return fail(OP_ERR_HIP, "bilateral filter staging failed: %s", hipGetErrorString(e));
        }
        A.src = d_in;
        A.dst = d_out;
    } else {
        A.src = depth;
        A.dst = out;
    }
    A.width = width;
    A.height = height;
    A.u16 = depth_format == OP_DEPTH_U16;
    A.depth_scale = depth_scale;
    A.radius = radius;
    const double log2e = 1.4426950408889634;
    A.gauss_color = (float)(log2e * -0.5 / ((double)sigma_color * sigma_color));
    A.gauss_space = (float)(log2e * -0.5 / ((double)sigma_space * sigma_space));
    const dim3 grid((width + kTileW - 1) / kTileW, (height + kTileH - 1) / kTileH, n_images), block(kTileW, kTileH / kRowsPerThread);
    const size_t lds = ((size_t)(kTileW + 2 * radius) * (kTileH + 2 * radius) + (size_t)(2 * radius + 1) * (2 * radius + 1)) * sizeof(float);
    if (radius == 3)
        hipLaunchKernelGGL(k_bilateral<3>, grid, block, lds, st, A);
    else
        hipLaunchKernelGGL(k_bilateral<0>, grid, block, lds, st, A);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && mem == OP_MEM_HOST) e = hipMemcpyAsync(out, d_out, npix * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && !stream) e = hipStreamSynchronize(st);
    if (mem == OP_MEM_HOST) {
        op::cached_free(d_in);
        op::cached_free(d_out);
    }
    if (e != hipSuccess) return fail(OP_ERR_HIP, "bilateral filter failed: %s", hipGetErrorString(e));
    return OP_OK;
}
