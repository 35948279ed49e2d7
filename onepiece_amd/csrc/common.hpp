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

} // namespace op
