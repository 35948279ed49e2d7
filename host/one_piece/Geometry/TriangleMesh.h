// Geometry/TriangleMesh.h -- geometry::TriangleMesh as CubeHandler::ExtractTriangleMesh fills it (reference:
// src/Geometry/TriangleMesh.h:10-51: points / normals / colors / triangles, Reset, HasColors, sizes, WriteToPLY).
// Simplification, normals and loaders are outside the hot path and are not re-declared.
#pragma once
#include <memory>
#include <string>

#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"

namespace one_piece {
namespace geometry {

class TriangleMesh {
  public:
    bool HasColors() const { return colors.size() == points.size() && colors.size() > 0; }
    bool HasNormals() const { return normals.size() == points.size() && normals.size() > 0; }
    void Reset() { triangles.clear(); points.clear(); normals.clear(); colors.clear(); }
    void Transform(const geometry::TransformationMatrix& T);
    std::shared_ptr<geometry::PointCloud> GetPointCloud() const;
    size_t GetPointSize() const { return points.size(); }
    size_t GetTriangleSize() const { return triangles.size(); }
    bool WriteToPLY(const std::string& fileName) const;

    geometry::Point3uiList triangles;
    geometry::Point3List points;
    geometry::Point3List normals;
    geometry::Point3List colors;
};

} // namespace geometry
} // namespace one_piece
