// bar_probe.hip -- can the host write a flag straight into device memory, and what does a host -> kernel -> host ping cost that way against a kernel
// polling host-mapped memory?  (Round 5: priced for a resident ICP loop.)  Build: hipcc --offload-arch=gfx950 -O2 tools/bar_probe.hip -o tools/bar_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_pong(const volatile unsigned* flag, volatile unsigned* ack, unsigned rounds) {
    for (unsigned k = 1; k <= rounds; ++k) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load((const unsigned*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != k) {
            if (wall_clock64() - t0 > 50000000) { __hip_atomic_store((unsigned*)ack, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
            __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store((unsigned*)ack, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static double ping(volatile unsigned* flag_host_view, const unsigned* flag_dev_view, volatile unsigned* ack_host, unsigned* ack_dev, unsigned rounds) {
    *flag_host_view = 0; *ack_host = 0;
    hipLaunchKernelGGL(k_pong, dim3(1), dim3(64), 0, 0, (const volatile unsigned*)flag_dev_view, (volatile unsigned*)ack_dev, rounds);
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned k = 1; k <= rounds; ++k) {
        *flag_host_view = k;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        while (*ack_host != k) { if (*ack_host == 0xdeadu) return -1; }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipDeviceSynchronize();
    return dt / rounds * 1e6;
}

int main() {
    unsigned *ack_host = nullptr, *ack_dev = nullptr, *flag_pinned = nullptr, *flag_pinned_dev = nullptr;
    HK(hipHostMalloc((void**)&ack_host, 64, hipHostMallocMapped));
    HK(hipHostGetDevicePointer((void**)&ack_dev, ack_host, 0));
    HK(hipHostMalloc((void**)&flag_pinned, 64, hipHostMallocMapped));
    HK(hipHostGetDevicePointer((void**)&flag_pinned_dev, flag_pinned, 0));
    printf("flag in host-mapped memory (the kernel polls over PCIe): %.2f us per host -> kernel -> host round\n", ping(flag_pinned, flag_pinned_dev, ack_host, ack_dev, 2000));
    unsigned* fine = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&fine, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        HK(hipMemset(fine, 0, 4096));
        struct sigaction sa, old;
        std::memset(&sa, 0, sizeof(sa));
        sa.sa_handler = on_segv;
        sigaction(SIGSEGV, &sa, &old);
        if (sigsetjmp(g_jmp, 1) == 0) {
            volatile unsigned* p = fine;
            *p = 5; // a host store into device memory through the BAR
            const unsigned back = *p;
            printf("host store + load through the device pointer: wrote 5, read %u\n", back);
            sigaction(SIGSEGV, &old, nullptr);
            printf("flag in fine-grained DEVICE memory (the host writes through the BAR, the kernel polls its own memory): %.2f us per round\n", ping(fine, fine, ack_host, ack_dev, 2000));
        } else {
            sigaction(SIGSEGV, &old, nullptr);
            printf("host access to the device pointer faults: device memory is not host-visible on this stack\n");
        }
    }
    return 0;
}
