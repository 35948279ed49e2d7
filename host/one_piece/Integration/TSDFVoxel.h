// Integration/TSDFVoxel.h -- the 20-byte voxel record {sdf, weight, color} of the block hash, host side
// (reference semantics: src/Integration/TSDFVoxel.h:8-82).  The device keeps the same five floats as SoA planes; this
// class is what GetCubeMap() hands back and what host code combines.  Defaults mark an unobserved voxel:
// sdf = 999, weight = 0, color = (-1, -1, -1).
#pragma once
#include "Geometry/Geometry.h"

namespace one_piece {
namespace integration {

class TSDFVoxel {
  public:
    float sdf = 999;
    float weight = 0;
    geometry::Point3 color = geometry::Point3(-1, -1, -1);

    TSDFVoxel() = default;
    TSDFVoxel(float _sdf, float _weight, const geometry::Point3& _color) : sdf(_sdf), weight(_weight), color(_color) {}

    // observed and inside the unit band (:75-78)
    bool IsValid() const { return !(sdf >= 1 || weight <= 0); }

    // weighted running mean of two observations (:24-39); an unobserved operand yields the other one
    TSDFVoxel operator+(const TSDFVoxel& other) const {
        if (weight == 0) return other;
        if (other.weight == 0) return *this;
        TSDFVoxel merged; // stays at the defaults if the weights cancel
        merged.weight = weight + other.weight;
        if (merged.weight != 0) {
            merged.sdf = (weight * sdf + other.weight * other.sdf) / merged.weight;
            merged.color = (color * weight + other.color * other.weight) / merged.weight;
        }
        return merged;
    }
    void operator+=(const TSDFVoxel& other) { *this = *this + other; }

    // plain sums, used while interpolating (:40-54): weights add, values add without normalisation
    TSDFVoxel add(const TSDFVoxel& other) const {
        if (weight == 0) return other;
        if (other.weight == 0) return *this;
        return TSDFVoxel(sdf + other.sdf, weight + other.weight, color + other.color);
    }
    // scaling of an interpolation term (:59-74): a zero factor or an unobserved voxel gives the default voxel
    TSDFVoxel operator*(float factor) const {
        if (factor == 0 || weight == 0) return TSDFVoxel();
        return TSDFVoxel(sdf * factor, weight * factor, color * factor);
    }
    TSDFVoxel operator/(float divisor) const { return (*this) * (1 / divisor); }
};

} // namespace integration
} // namespace one_piece
