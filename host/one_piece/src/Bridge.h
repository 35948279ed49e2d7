// Bridge.h -- small conversions between the public C++ types and the C-ABI's PODs (library-internal).
#pragma once
#include <cstdlib>
#include <iostream>

#include "Geometry/Geometry.h"
#include "onepiece_hip.h"

namespace one_piece {
namespace bridge {

inline void RowMajor(const geometry::Matrix4& m, float out[16]) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = m(r, c);
}
inline geometry::Matrix4 FromRowMajor(const float in[16]) {
    geometry::Matrix4 m;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m(r, c) = in[r * 4 + c];
    return m;
}
// the reference tests `depth.depth() == CV_32FC1` and treats everything else as unsigned short (Integrator.cpp:26-29)
inline int DepthFormat(const cv::Mat& depth) { return depth.depth() == CV_32F ? OP_DEPTH_F32 : OP_DEPTH_U16; }
// Every device object of the class surface is created through this call -- which is therefore where the surface makes its one explicit runtime
// request, before its first object touches HIP: eight hardware queues (op_runtime_configure: the pipelined tracker's streams must not share a
// queue; a value the application has set is never overwritten).  The C-ABI library itself changes nothing in the process on its own.
inline int Device() {
    static const int configured = op_runtime_configure(16);
    (void)configured;
    const char* e = std::getenv("ONEPIECE_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}
// reference behaviour on failure: a coloured line on std::cout, then the caller returns early
inline bool Failed(int rc, const char* where) {
    if (rc == OP_OK) return false;
    std::cout << RED << "[ERROR]::[" << where << "]::" << op_last_error() << RESET << std::endl;
    return true;
}
inline const float* Floats(const geometry::Point3List& v) { return v.empty() ? nullptr : v[0].data(); }
inline float* Floats(geometry::Point3List& v) { return v.empty() ? nullptr : v[0].data(); }
static_assert(sizeof(geometry::Point3) == 3 * sizeof(float), "Point3List must be a contiguous xyz float array");

} // namespace bridge
} // namespace one_piece
