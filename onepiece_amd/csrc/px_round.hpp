// px_round.hpp -- the reference's pixel rounding  int u = fx*X/Z + 0.5 + cx  (Integrator.cpp:20-21,
// 61-62) for host and device.
//
// Reference semantics: a = (fx*X)/Z is a float; the literal 0.5 promotes the sum to double;
// t = (double)a + 0.5 + (double)c is exact for any pixel-scale a; the result is t truncated toward
// zero.  Non-finite or out-of-int-range t is UB in C++ (INT_MIN from cvttsd2si on x86) and is
// reported as INT_MIN here, which every caller rejects through its `u < 0` test.
//
// px_round_dp() evaluates exactly that in double.  px_round_sp() produces the SAME integer for every
// a whenever the caller's bounds test (0 <= u < width) can pass, using only fp32 compares and
// integer adds: a = trunc(a) + af and c = floor(c) + cf are exact splits, and the position of
// af + cf + 0.5 relative to 0, 1, 2 is decided by comparing af with the exactly representable
// thresholds -0.5-cf, 0.5-cf, 1.5-cf (exact when |c| >= 1 or c == 0; px_split() checks this and
// also excludes cf == 0.5, the one case where the reference's double sum is itself inexact in a
// way that matters -- callers then fall back to px_round_dp).
// fp64 adds/converts run at a fraction of the fp32 rate on CDNA4; the integrate kernel is
// ALU-bound once frames are batched, so this matters.  Equivalence is tested exhaustively-ish in
// tests/test_abi_cpu.py::test_pixel_rounding_fast_path_equals_double_formula (4e7 cases incl.
// values within a few ulp of every rounding boundary).
#pragma once
#include <climits>
#include <cmath>

#if defined(__HIPCC__)
#define OP_HD __host__ __device__ __forceinline__
#else
#define OP_HD inline
#endif

struct PxSplit {
    int ci;          // floor(c)
    float hm, h0, h1; // -0.5 - cf, 0.5 - cf, 1.5 - cf
    int exact;       // thresholds exactly representable -> fast path allowed
};

OP_HD PxSplit px_split(float c) {
    PxSplit s;
    const float fl = floorf(c);
    const float cf = c - fl;
    s.ci = (int)fl;
    s.hm = -0.5f - cf; s.h0 = 0.5f - cf; s.h1 = 1.5f - cf;
    // cf == 0.5 is excluded: there cf + 0.5 is an integer, and for |a| < ~1e-13 the reference's
    // double additions round a away (t lands exactly on the integer) while this path stays exact.
    s.exact = (fabsf(c) >= 1.0f || c == 0.0f) && fabsf(c) < 1.0e6f && cf != 0.5f;
    return s;
}

OP_HD int px_round_dp(float a, float c) {
    const double t = (double)a + 0.5 + (double)c;
    if (!(t > -2147483649.0 && t < 2147483648.0)) return INT_MIN;
    return (int)t;
}

OP_HD int px_round_sp(float a, const PxSplit& s) {
    if (!(fabsf(a) < 1.0e9f)) return INT_MIN; // NaN, inf and values no image can contain
    const float ai = truncf(a);
    const float af = a - ai; // exact, same sign as a, |af| < 1
    const int fl = (int)ai + s.ci - 1 + (af >= s.hm) + (af >= s.h0) + (af >= s.h1); // floor(t)
    if (fl >= 0) return fl;
    // t < 0: truncation toward zero = ceil(t); t is an integer only when af sits on a threshold
    const bool integral = af == s.hm || af == s.h0 || af == s.h1;
    return integral ? fl : fl + 1;
}

// ---- in-image pixel of one axis (what the kernels use) ----------------------------------------------------------
// The kernels only ever need the pixel when it lies INSIDE the image (Integrator.cpp:63 rejects everything else), and
// inside the image the truncation above is a floor except on t in (-1, 0), which truncates to 0:
//     0 <= trunc(t) <= extent - 1   <=>   -1 < t < extent   <=>   t_lo < a < t_hi,   t_lo = -(c + 1.5), t_hi = extent - c - 0.5
//     trunc(t) = max(floor(t), 0) there,   floor(t) = trunc(a) + floor(K) - 1 + [af >= h0] + [af >= h1],
// with K = c + 0.5, kf = K - floor(K), af = a - trunc(a) (exact), h0 = -kf, h1 = 1 - kf.  Two range compares on `a`, two
// threshold compares on `af`, no integer range tests, no special handling of NaN / inf / huge values (they fail the range
// compares).  Exact whenever t_lo, t_hi, h0, h1 are representable in fp32 and kf != 0 (then a tiny |a| cannot move the
// reference's double sum across an integer); px_axis() checks that and callers fall back to the double formula otherwise.
struct PxAxis {
    float t_lo, t_hi, h0, h1;
    int base;  // floor(c + 0.5) - 1
    int exact;
};

OP_HD PxAxis px_axis(float c, int extent) {
    PxAxis s;
    const double K = (double)c + 0.5, ki = floor(K), kf = K - ki;
    const double t_lo = -((double)c + 1.5), t_hi = (double)extent - (double)c - 0.5, h0 = -kf, h1 = 1.0 - kf;
    s.t_lo = (float)t_lo; s.t_hi = (float)t_hi; s.h0 = (float)h0; s.h1 = (float)h1;
    s.base = (int)ki - 1;
    s.exact = (double)s.t_lo == t_lo && (double)s.t_hi == t_hi && (double)s.h0 == h0 && (double)s.h1 == h1 && kf != 0.0 &&
              fabs((double)c) < 1.0e6 && extent > 0 && extent < (1 << 20);
    return s;
}

// true + the pixel when it is inside [0, extent), false otherwise
OP_HD bool px_pixel_sp(float a, const PxAxis& s, int& u) {
    const float ai = truncf(a);
    const float af = a - ai; // exact
    const int fl = (int)ai + s.base + (af >= s.h0) + (af >= s.h1);
    u = fl < 0 ? 0 : fl;
    return a > s.t_lo && a < s.t_hi;
}
OP_HD bool px_pixel_dp(float a, float c, int extent, int& u) {
    u = px_round_dp(a, c);
    return u >= 0 && u < extent;
}
