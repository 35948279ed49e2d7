// hbm_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE (and the TCC_EA0 request counters behind them) on
// kernels whose HBM traffic is known exactly, in the access shapes the fusion kernels use.  MI355X_MICROARCH.md calibrates
// FETCH_SIZE only for 16 B/lane streaming reads (x2 on gfx950) and calls every other width uncalibrated; k_integrate
// reads 4 B per lane (one 256-byte plane row per wave instruction).
//
// Every kernel moves BYTES = 4 GiB (16x the 256 MiB Infinity Cache, so nothing is served on-die) and has its own
// name so the per-kernel rows of the counter csv can be matched:
//   k_read_b4 / b8 / b16      streaming read, 4 / 8 / 16 bytes per lane, wave-contiguous (256 / 512 / 1024 B per instruction)
//   k_read_planes_b4          k_integrate's shape: a 512-thread workgroup per 10 240-byte block, 5 plane rows of 4 B/lane
//   k_rmw_planes_b4           the same with the five rows written back (read + write = 2 x BYTES)
//   k_write_b4 / k_write_b16  streaming write
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/hbm_calib.bin      (one counter group per pass)
// Build: hipcc --offload-arch=gfx950 -O2 tools/hbm_calib.hip -o tools/hbm_calib.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read_b4(const float* in, size_t n, float* sink) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_b8(const float2* in, size_t n, float* sink) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float2 v = in[i]; acc += v.x + v.y; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_b16(const float4* in, size_t n, float* sink) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[i]; acc += (v.x + v.y) + (v.z + v.w); }
    if (acc == 123.456f) sink[0] = acc;
}
// k_integrate's block shape: 5 planes x 512 floats per block, thread = voxel, one row of 64 floats per wave and plane
__global__ __launch_bounds__(512) void k_read_planes_b4(const float* pool, size_t n_blocks, float* sink) {
    float acc = 0;
    for (size_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const float* vox = pool + b * 2560 + threadIdx.x;
        acc += vox[0] + vox[512] + vox[1024] + vox[1536] + vox[2048];
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(512) void k_rmw_planes_b4(float* pool, size_t n_blocks, float one) {
    for (size_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        float* vox = pool + b * 2560 + threadIdx.x;
        const float s = vox[0], w = vox[512], c0 = vox[1024], c1 = vox[1536], c2 = vox[2048];
        vox[0] = s * one; vox[512] = w * one; vox[1024] = c0 * one; vox[1536] = c1 * one; vox[2048] = c2 * one;
    }
}
__global__ __launch_bounds__(256) void k_write_b4(float* out, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
__global__ __launch_bounds__(256) void k_write_b16(float4* out, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(v, v, v, v);
}

int main(int argc, char** argv) {
    const size_t BYTES = (argc > 1 ? (size_t)std::atoll(argv[1]) : 4096) << 20; // MiB
    const int reps = argc > 2 ? std::atoi(argv[2]) : 3;
    const size_t n_blocks = BYTES / 10240;
    const size_t plane_bytes = n_blocks * 10240;
    void* buf; float* sink;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(buf, 0, BYTES));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8 * 4;
    std::printf("buffer %zu bytes (%.3f GiB); k_*_planes move %zu bytes per direction\n", BYTES, BYTES / 1073741824.0, plane_bytes);
    auto timed = [&](const char* name, size_t bytes, auto launch) {
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("%-18s rep %d: %zu bytes in %.3f ms = %.1f GB/s\n", name, r, bytes, ms, bytes / ms / 1e6);
        }
    };
    timed("k_read_b4", BYTES, [&] { hipLaunchKernelGGL(k_read_b4, dim3(grid), dim3(256), 0, 0, (const float*)buf, BYTES / 4, sink); });
    timed("k_read_b8", BYTES, [&] { hipLaunchKernelGGL(k_read_b8, dim3(grid), dim3(256), 0, 0, (const float2*)buf, BYTES / 8, sink); });
    timed("k_read_b16", BYTES, [&] { hipLaunchKernelGGL(k_read_b16, dim3(grid), dim3(256), 0, 0, (const float4*)buf, BYTES / 16, sink); });
    timed("k_read_planes_b4", plane_bytes, [&] { hipLaunchKernelGGL(k_read_planes_b4, dim3(256 * 16), dim3(512), 0, 0, (const float*)buf, n_blocks, sink); });
    timed("k_rmw_planes_b4", 2 * plane_bytes, [&] { hipLaunchKernelGGL(k_rmw_planes_b4, dim3(256 * 16), dim3(512), 0, 0, (float*)buf, n_blocks, 1.0f); });
    timed("k_write_b4", BYTES, [&] { hipLaunchKernelGGL(k_write_b4, dim3(grid), dim3(256), 0, 0, (float*)buf, BYTES / 4, 0.0f); });
    timed("k_write_b16", BYTES, [&] { hipLaunchKernelGGL(k_write_b16, dim3(grid), dim3(256), 0, 0, (float4*)buf, BYTES / 16, 0.0f); });
    CK(hipDeviceSynchronize());
    return 0;
}
