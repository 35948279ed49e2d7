// graph_probe.hip -- is a chain of ~100 small dependent kernels cheaper to start as ONE hipGraph launch than as 100 stream launches?  (The dense tracker
// enqueues ~100 kernels per frame pair; with four pairs in flight its rate is bound by kernel starts)
// Measures chains/s for K streams (K = 1, 4, 8), each running chains of N kernels of ~3 us: plain launches from one host thread vs one graph launch per chain.
// Build: hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o tools/graph_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_small(float* p, int n, int spin) { // a few microseconds of dependent work in a handful of workgroups
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.0f;
    for (int k = 0; k < spin; ++k) v = v * 1.0000001f + 1e-7f;
    if (i < n) p[i] = v;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 100, chains = argc > 2 ? atoi(argv[2]) : 200, spin = argc > 3 ? atoi(argv[3]) : 300;
    const int n = 64 * 256;
    for (int K : {1, 4, 8}) {
        std::vector<hipStream_t> st(K);
        std::vector<float*> buf(K);
        std::vector<hipGraphExec_t> ge(K);
        for (int s = 0; s < K; ++s) {
            CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
            CK(hipMalloc((void**)&buf[s], n * sizeof(float)));
            CK(hipMemset(buf[s], 0, n * sizeof(float)));
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st[s], buf[s], n, spin);
            CK(hipStreamEndCapture(st[s], &g));
            CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
        }
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                for (int c = 0; c < chains; ++c) {
                    const int s = c % K;
                    if (mode == 0) for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st[s], buf[s], n, spin);
                    else CK(hipGraphLaunch(ge[s], st[s]));
                }
                const double host = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                CK(hipDeviceSynchronize());
                const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rep == 1)
                    printf("%d stream(s), %s: %.0f chains/s (%.2f us per kernel), host enqueue time %.1f us per chain\n", K, mode ? "one graph launch per chain" : "plain launches           ",
                           chains / all, all / chains / N * 1e6 * 1.0, host / chains * 1e6);
            }
        }
        for (int s = 0; s < K; ++s) { (void)hipGraphExecDestroy(ge[s]); (void)hipFree(buf[s]); (void)hipStreamDestroy(st[s]); }
    }
    return 0;
}
