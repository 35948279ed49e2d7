// common.hpp -- shared plumbing of the C-ABI library (error reporting, HIP checks, device helpers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/onepiece_hip.h"

namespace op {

extern thread_local char g_last_error[512];

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

#define OP_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ::op::fail(OP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                              __FILE__, __LINE__);                                                \
    } while (0)

#define OP_TRY(expr)                  \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != OP_OK) return rc_; \
    } while (0)

// Selects `device` after checking that a usable GPU exists; there is no CPU fallback anywhere.
inline int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(OP_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(OP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    OP_HIP(hipSetDevice(device));
    return OP_OK;
}

constexpr int kWave = 64; // gfx950 wavefront width

// XCD-aware workgroup order.  Workgroup p of a 1-D grid is dispatched to XCD p % 8 and every XCD has its own 4 MiB
// L2 (MI355X_MICROARCH.md), so with the plain order eight neighbouring workgroups -- which gather neighbouring data --
// land on eight different L2s and each L2 sees the whole working set.  This bijection of [0, n) hands XCD x the
// contiguous slab of logical indices [start_x, start_x + q + (x < r)), q = n / 8, r = n % 8.
__device__ __forceinline__ unsigned xcd_slab_index(unsigned p, unsigned n) {
    const unsigned x = p & 7u, slot = p >> 3, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + slot;
}

// Wave-wide sums of 32 doubles per lane by recursive halving ("reduce-scatter"): at the step with lane
// mask M a lane keeps one half of its values and receives the partner's copy of that half, so the work
// halves every step (16+8+4+2+1+1 = 32 fp64 adds per lane instead of 32 x 6 for a butterfly per value).
// On return lane L holds in v[0] the wave total of element (L >> 1) & 31.
template <int H, int M, int N>
__device__ __forceinline__ void wave_halve(double (&v)[N], int lane) {
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double keep = up ? v[i + H] : v[i];
        const double give = up ? v[i] : v[i + H];
        v[i] = keep + __shfl_xor(give, M, 64);
    }
}
__device__ __forceinline__ void wave_reduce_scatter32(double (&v)[32]) {
    const int lane = threadIdx.x & 63;
    wave_halve<16, 32>(v, lane);
    wave_halve<8, 16>(v, lane);
    wave_halve<4, 8>(v, lane);
    wave_halve<2, 4>(v, lane);
    wave_halve<1, 2>(v, lane);
    v[0] += __shfl_xor(v[0], 1, 64);
}
// The same reduction when the 32 per-lane values are cheap to (re)compute: `val(k)` (k a compile-time constant after
// unrolling) is evaluated inside the first halving step, so only 16 doubles are ever live (32 VGPRs instead of 64).
// Returns what wave_reduce_scatter32 leaves in v[0]; bit-identical to it.
template <int K> struct index_c { static constexpr int value = K; };
template <int I, class F>
__device__ __forceinline__ void lazy_first_step(F& val, bool up, double (&v)[16]) {
    if constexpr (I < 16) {
        const double lo = val(index_c<I>{}), hi = val(index_c<I + 16>{});
        v[I] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, 32, 64);
        lazy_first_step<I + 1>(val, up, v);
    }
}
// val is called as val(index_c<k>{}) so that k is a constant expression inside it
template <class F>
__device__ __forceinline__ double wave_reduce_scatter32_lazy(F val) {
    const int lane = threadIdx.x & 63;
    const bool up = (lane & 32) != 0;
    double v[16];
    lazy_first_step<0>(val, up, v);
    wave_halve<8, 16>(v, lane);
    wave_halve<4, 8>(v, lane);
    wave_halve<2, 4>(v, lane);
    wave_halve<1, 2>(v, lane);
    return v[0] + __shfl_xor(v[0], 1, 64);
}

} // namespace op
