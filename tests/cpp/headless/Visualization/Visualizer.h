// Headless stand-in for the reference's OpenGL/Pangolin viewer (src/Visualization/Visualizer.h), used ONLY to compile
// the reference's example sources on a machine without a display stack (tests/test_reference_examples.py,
// oracle/tools/build_ref_examples.sh).  The GUI is out of scope; this class draws nothing: it reports
// what it was handed and returns, so an example's main() runs to its end on the GPU box.
#pragma once
#include <iostream>
#include <string>

#include "Geometry/Geometry.h"
#include "Geometry/PointCloud.h"
#include "Geometry/TriangleMesh.h"

namespace one_piece {
namespace visualization {

class Visualizer {
  public:
    void AddPointCloud(const geometry::PointCloud& pcd) { std::cout << "[headless viewer] point cloud with " << pcd.points.size() << " points" << std::endl; }
    void AddTriangleMesh(const geometry::TriangleMesh& mesh) {
        std::cout << "[headless viewer] mesh with " << mesh.points.size() << " vertices, " << mesh.triangles.size() << " triangles" << std::endl;
    }
    void AddCameraSet(const geometry::SE3List&, const geometry::Point3List&) {}
    void Show() { std::cout << "[headless viewer] Show()" << std::endl; }
    void ShowOnce() {}
    void Initialize(const std::string& = "OnePiece") {}
    void SetDrawColor(bool) {}
    void SetDrawNormal(bool) {}
    void DrawPhongRendering() {}
    void SetModelViewMatrix(const geometry::TransformationMatrix&, bool = true) {}
    void Reset() {}
};

} // namespace visualization
} // namespace one_piece
