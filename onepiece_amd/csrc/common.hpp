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
//
// How the halves change hands matters more than the 32 adds (tools/valu_ubench.hip, MI355X): `up ? a : b` on a double
// is two VOP2 v_cndmask_b32 reading VCC, and BACK-TO-BACK VOP2 v_cndmask issue at ~11 cycles each instead of ~1.4 --
// the four selects of a step cost 43 cycles, 1300 per reduction.  So:
//   * lane masks 32 and 16 (24 of the 31 steps) use gfx950's v_permlane32_swap / v_permlane16_swap: swapping the odd
//     rows of a (the value the lower half keeps) with the even rows of b (the value the upper half keeps) leaves
//     {own a, partner's a} in the lower rows and {partner's b, own b} in the upper ones -- a' + b' is the step, with no
//     select and no LDS permute;
//   * the other masks select with v_bfi_b32 against an all-ones / all-zeros lane word.
__device__ __forceinline__ double pack_d(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ double select_d(unsigned m, double if_set, double if_clear) { // m: all ones or all zeros
    const unsigned sl = (unsigned)__double2loint(if_set), sh = (unsigned)__double2hiint(if_set);
    const unsigned cl = (unsigned)__double2loint(if_clear), ch = (unsigned)__double2hiint(if_clear);
    return pack_d((m & sl) | (~m & cl), (m & sh) | (~m & ch));
}
template <int M>
__device__ __forceinline__ double halve_step(double a /* kept where (lane & M) == 0 */, double b /* kept elsewhere */, unsigned up_mask) {
    if constexpr (M == 32 || M == 16) {
        (void)up_mask;
        const unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
        const unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
        if constexpr (M == 32) {
            const auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false), h = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
            return pack_d(l[0], h[0]) + pack_d(l[1], h[1]);
        } else {
            const auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false), h = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
            return pack_d(l[0], h[0]) + pack_d(l[1], h[1]);
        }
    } else {
        const double keep = select_d(up_mask, b, a), give = select_d(up_mask, a, b);
        return keep + __shfl_xor(give, M, 64);
    }
}
template <int H, int M, int N>
__device__ __forceinline__ void wave_halve(double (&v)[N], int lane) {
    const unsigned up_mask = (lane & M) != 0 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < H; ++i) v[i] = halve_step<M>(v[i], v[i + H], up_mask);
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
        (void)up;
        v[I] = halve_step<32>(lo, hi, 0u);
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
