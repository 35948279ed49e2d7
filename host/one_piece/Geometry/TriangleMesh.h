// Geometry/TriangleMesh.h -- geometry::TriangleMesh with the reference's public surface (src/Geometry/TriangleMesh.h:10-51)
// as CubeHandler::ExtractTriangleMesh fills it and the fusion drivers post-process it (example/ImageIntegration.cpp:41-46:
// ComputeNormals + WriteToPLY; example/ImageSequenceIntegration.cpp:56-58: ClusteringSimplify + WriteToPLY).  Host C++,
// outside the hot path: documented behaviour, no bit-parity claim.  QuadricSimplify (a full
// quadric-error edge-collapse simplifier, src/Geometry/MeshSimplification.cpp) is NOT provided.
// Inside the reference tree (-DONEPIECE_IN_REFERENCE_TREE) this header steps aside for the reference's own declaration.
#pragma once
#ifdef ONEPIECE_IN_REFERENCE_TREE
#include_next "Geometry/TriangleMesh.h"
#else
#include <memory>
#include <string>
#include <vector>

#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"

namespace one_piece {
namespace geometry {

class TriangleMesh {
  public:
    bool LoadFromPLY(const std::string& filename);
    bool LoadFromOBJ(const std::string& filename);
    bool LoadFromFile(const std::string& filename);
    // per vertex: the normalised sum of the unit normals (p2 - p1) x (p3 - p1) of its triangles (TriangleMesh.cpp:95-127)
    void ComputeNormals();
    void Transform(const geometry::TransformationMatrix& T);
    bool HasColors() const { return colors.size() == points.size() && colors.size() > 0; }
    bool HasNormals() const { return normals.size() == points.size() && normals.size() > 0; }
    void Reset() { triangles.clear(); points.clear(); normals.clear(); colors.clear(); }
    // concatenation with re-based triangle indices (TriangleMesh.cpp:72-94)
    void LoadFromMeshes(const std::vector<TriangleMesh>& meshes);
    // vertex clustering on a grid of edge grid_len: the corners of all triangles that fall into one cell collapse onto
    // the mean of those corner positions, triangles with two corners in one cell disappear (MeshSimplification.cpp:579-657)
    std::shared_ptr<geometry::TriangleMesh> ClusteringSimplify(float grid_len) const;
    // drops every edge-connected component with at most min_points vertices (MeshSimplification.cpp:658-740)
    std::shared_ptr<geometry::TriangleMesh> Prune(size_t min_points) const;
    std::shared_ptr<geometry::PointCloud> GetPointCloud() const;
    size_t GetPointSize() const { return points.size(); }
    size_t GetTriangleSize() const { return triangles.size(); }
    bool WriteToPLY(const std::string& fileName) const;
    bool WriteToOBJ(const std::string& fileName) const;

    geometry::Point3uiList triangles;
    geometry::Point3List points;
    geometry::Point3List normals;
    geometry::Point3List colors;
};

} // namespace geometry
} // namespace one_piece
#endif // ONEPIECE_IN_REFERENCE_TREE
