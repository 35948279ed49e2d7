// queue_probe.hip -- how many kernels of DIFFERENT HIP streams does the chip run side by side?  K streams each get one long one-workgroup kernel (a dependent
// fp32 add chain, like k_seq_sums' summing wave); if they ran concurrently the wall time would stay that of one kernel for any K <= the number of CUs.
// Measured on MI355X (profiles/r06_queue_probe.txt): ceil(K / Q) kernel times with Q = GPU_MAX_HW_QUEUES (default 4; 8 and 16 tried): the limit is the number of
// hardware queues the runtime maps its streams onto, not the chip.  ONE launch of K workgroups takes one kernel time for any K <= 32.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/queue_probe.bin && GPU_MAX_HW_QUEUES=8 tools/queue_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_chain(float* out, int n) {
    float a = (float)threadIdx.x;
    for (int i = 0; i < n; ++i) a = a + 1.0f; // (the compiler may not re-associate a float chain)
    if (a == -1.0f) out[threadIdx.x] = a;
}
// the same work as ONE launch with K workgroups (what k_seq_sums_many does)
__global__ void k_chain_many(float* out, int n) {
    float a = (float)threadIdx.x + (float)blockIdx.x;
    for (int i = 0; i < n; ++i) a = a + 1.0f;
    if (a == -1.0f) out[threadIdx.x] = a;
}

int main() {
    const int n = 400000;
    float* d = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess) { printf("no device\n"); return 1; }
    std::vector<hipStream_t> s(32);
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s[0], d, n);
    hipDeviceSynchronize();
    for (int K : {1, 2, 3, 4, 5, 6, 8, 12, 16, 32}) {
        double best = 1e9, best_many = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            double t = now();
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s[k], d, n);
            hipDeviceSynchronize();
            best = std::min(best, now() - t);
            t = now();
            hipLaunchKernelGGL(k_chain_many, dim3(K), dim3(64), 0, s[0], d, n);
            hipDeviceSynchronize();
            best_many = std::min(best_many, now() - t);
        }
        printf("K = %2d one-workgroup kernels on K streams: %7.3f ms   |   one launch of K workgroups: %7.3f ms\n", K, best, best_many);
    }
    return 0;
}
