// Integration/MarchingCube.h -- where ExtractTriangleMesh gets its case tables from.
//
// The reference triangulates with the 256 x 16 table `MCLookTable` and the 12 x 2 `EdgeIndexPairs` of
// src/Integration/MarchingCubePredefined.h -- data of the reference (and not the commonly published table).  The
// device kernel is table-agnostic (op_volume_extract_mesh takes both tables as arguments), so:
//   * inside the reference tree, call SetMarchingCubeTables(&MCLookTable[0][0], &EdgeIndexPairs[0][0]) once (the
//     one-line binding INTEGRATION.md shows) and every mesh is the reference's, vertex for vertex;
//   * otherwise the library falls back to a table it GENERATES at first use (GenerateMarchingCubeTables): for each of
//     the 256 sign cases the crossing edges are joined into closed loops by walking the cube's faces (ambiguous faces
//     keep the negative -- inside -- corners separated) and every loop is fan-triangulated.  The surface is watertight
//     across cells, but its triangulation is this library's, not the reference's.
#pragma once
#include <cstdint>

namespace one_piece {
namespace integration {

// tri_table: 256 rows x 16 edge ids, each row terminated by -1; edge_pairs: 12 x 2 corner ids (corner numbering =
// CubePara::CornerXYZOffset).  The pointers must stay valid (they are read at every extraction).
void SetMarchingCubeTables(const int* tri_table, const int* edge_pairs);
// the tables in use: the ones set above, else the generated default
void GetMarchingCubeTables(const int** tri_table, const int** edge_pairs);
// fills tri_table[256*16] and edge_pairs[12*2] with the generated default (see above)
void GenerateMarchingCubeTables(int* tri_table, int* edge_pairs);

} // namespace integration
} // namespace one_piece
