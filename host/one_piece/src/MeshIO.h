// MeshIO.h -- PLY / OBJ readers and writers behind PointCloud / TriangleMesh::{LoadFromPLY, LoadFromOBJ, WriteToPLY,
// WriteToOBJ} (library-internal; the reference does this with tinyply / tinyobjloader, src/Tool/PLYManager.cpp,
// OBJManager.cpp).  Own implementation of the two published formats:
//   PLY  ascii and binary_little_endian; element `vertex` with x y z (float / double), optional nx ny nz, optional
//        red green blue (uchar -> /255, or float), any further properties skipped; element `face` with a list property
//        vertex_indices / vertex_index (triangles kept, larger polygons fanned); other elements skipped.
//        Written: binary little endian, float x y z [nx ny nz] [uchar red green blue], face list uchar + 3 x uint --
//        the layout the reference's writer produces (PLYManager.cpp:243-254).
//   OBJ  `v x y z [r g b]`, `vn`, `f a b c` with a, a/t, a/t/n or a//n indices (1-based, negative = relative).
#pragma once
#include <string>

#include "Geometry/Geometry.h"

namespace one_piece {
namespace meshio {

bool ReadPly(const std::string& file, geometry::Point3List& points, geometry::Point3List& normals, geometry::Point3List& colors,
             geometry::Point3uiList* triangles);
bool WritePly(const std::string& file, const geometry::Point3List& points, const geometry::Point3List& normals, const geometry::Point3List& colors,
              const geometry::Point3uiList* triangles);
bool ReadObj(const std::string& file, geometry::Point3List& points, geometry::Point3List& normals, geometry::Point3List& colors,
             geometry::Point3uiList* triangles);
bool WriteObj(const std::string& file, const geometry::Point3List& points, const geometry::Point3List& normals, const geometry::Point3List& colors,
              const geometry::Point3uiList* triangles);

} // namespace meshio
} // namespace one_piece
