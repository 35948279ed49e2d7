// Integration/Integrator.h -- integration::Integrator's public state (reference: src/Integration/Integrator.h:10-27).
// Callers read and set `truncation` directly; the per-voxel update itself (Integrator.cpp:36-94) is the k_integrate
// kernel behind op_volume_integrate, and GetSDF (:8-35) is evaluated inside k_select.
#pragma once

namespace one_piece {
namespace integration {

class Integrator {
  public:
    Integrator() = default;
    void SetTruncation(float _trunc) { truncation = _trunc; }
    float truncation = 0.1;
    // constant weight kept for source compatibility; the update uses the literal 1.0 (Integrator.cpp:77)
    float Weight = 1.0 / (2 * truncation);
};

} // namespace integration
} // namespace one_piece
