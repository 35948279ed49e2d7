// MarchingCube.cpp -- case tables for ExtractTriangleMesh: the caller's (SetMarchingCubeTables) or a generated default.
#include "Integration/MarchingCube.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace one_piece {
namespace integration {

namespace {
const int* g_tri = nullptr;
const int* g_edges = nullptr;
int g_default_tri[256 * 16];
int g_default_edges[12 * 2];
std::once_flag g_default_once;

// corner numbering of CubePara::CornerXYZOffset: bottom ring (z = 0) 0..3 counter-clockwise seen from +z, top ring 4..7
const int kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// edges: bottom ring, top ring, verticals
const int kEdge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
// the six faces as corner cycles (either orientation: loop orientation is fixed afterwards from the geometry)
const int kFace[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}};

// both cube edges lie on one face of the cube
bool OnOneFace(int ea, int eb) {
    for (int axis = 0; axis < 3; ++axis)
        for (int side = 0; side < 2; ++side)
            if (kCorner[kEdge[ea][0]][axis] == side && kCorner[kEdge[ea][1]][axis] == side && kCorner[kEdge[eb][0]][axis] == side && kCorner[kEdge[eb][1]][axis] == side) return true;
    return false;
}
// triangles of the sub-polygon loop[i .. j] (its closing edge i-j is given) without an in-face diagonal; appends to out, false = none exists
bool TriangulateAvoidingFaces(const std::vector<int>& loop, int i, int j, std::vector<int>& out) {
    if (j - i < 2) return true;
    for (int k = i + 1; k < j; ++k) {
        if (k - i > 1 && OnOneFace(loop[i], loop[k])) continue;
        if (j - k > 1 && OnOneFace(loop[k], loop[j])) continue;
        const size_t mark = out.size();
        out.push_back(loop[i]); out.push_back(loop[k]); out.push_back(loop[j]);
        if (TriangulateAvoidingFaces(loop, i, k, out) && TriangulateAvoidingFaces(loop, k, j, out)) return true;
        out.resize(mark);
    }
    return false;
}

int EdgeBetween(int a, int b) {
    for (int e = 0; e < 12; ++e)
        if ((kEdge[e][0] == a && kEdge[e][1] == b) || (kEdge[e][0] == b && kEdge[e][1] == a)) return e;
    return -1;
}
} // namespace

void GenerateMarchingCubeTables(int* tri_table, int* edge_pairs) {
    for (int e = 0; e < 12; ++e) { edge_pairs[2 * e] = kEdge[e][0]; edge_pairs[2 * e + 1] = kEdge[e][1]; }
    for (int c = 0; c < 256; ++c) {
        int* row = tri_table + 16 * c;
        for (int k = 0; k < 16; ++k) row[k] = -1;
        // bit i of the case = corner i has sdf > 0 (outside), the convention of DetermineCase (MarchingCube.cpp:17-24);
        // "inside" below = bit clear
        // 1. per face, join the crossing edges pairwise.  A face with four crossings is ambiguous: a segment cuts off each corner whose
        //    case bit is SET (the two such corners are kept apart, the other two joined) -- the resolution of the reference's table
        //    (MarchingCubePredefined.h:17-274) in every one of its rows: tests/golden/mc_table_golden.json holds the polygons of that table,
        //    and the rows generated here cut all 256 cases along the same ones (same loops, same facing, same triangle counts).  What still
        //    differs is which diagonals triangulate a polygon (a fan from the loop's first edge here).
        int link[12][2], nlink[12];
        for (int e = 0; e < 12; ++e) { nlink[e] = 0; link[e][0] = link[e][1] = -1; }
        auto join = [&](int a, int b) { link[a][nlink[a]++] = b; link[b][nlink[b]++] = a; };
        for (int f = 0; f < 6; ++f) {
            int fe[4], crossing[4], ncross = 0;
            for (int k = 0; k < 4; ++k) {
                const int a = kFace[f][k], b = kFace[f][(k + 1) & 3];
                fe[k] = EdgeBetween(a, b);
                crossing[k] = ((c >> a) & 1) != ((c >> b) & 1);
                ncross += crossing[k];
            }
            if (ncross == 2) {
                int first = -1;
                for (int k = 0; k < 4; ++k)
                    if (crossing[k]) { if (first < 0) first = k; else join(fe[first], fe[k]); }
            } else if (ncross == 4) {
                for (int k = 0; k < 4; ++k) // corner kFace[f][k] sits between face edges k-1 and k
                    if ((c >> kFace[f][k]) & 1) join(fe[(k + 3) & 3], fe[k]);
            }
        }
        // 2. walk the closed loops, fan-triangulate each, orient every triangle so that its normal points from the inside
        //    (negative) corners towards the outside
        bool used[12] = {false};
        int out = 0;
        for (int e0 = 0; e0 < 12; ++e0) {
            if (used[e0] || nlink[e0] != 2) continue;
            std::vector<int> loop;
            int prev = -1, cur = e0;
            while (!used[cur]) {
                used[cur] = true;
                loop.push_back(cur);
                const int next = link[cur][0] != prev ? link[cur][0] : link[cur][1];
                prev = cur; cur = next;
            }
            // edge midpoints stand in for the crossing points when fixing the orientation
            auto mid = [&](int e, double m[3]) { for (int k = 0; k < 3; ++k) m[k] = 0.5 * (kCorner[kEdge[e][0]][k] + kCorner[kEdge[e][1]][k]); };
            double centre[3] = {0, 0, 0}, inside[3] = {0, 0, 0}, normal[3] = {0, 0, 0};
            int n_in = 0;
            for (size_t k = 0; k < loop.size(); ++k) { double m[3]; mid(loop[k], m); for (int a = 0; a < 3; ++a) centre[a] += m[a] / loop.size(); }
            for (size_t k = 0; k < loop.size(); ++k) // inside corners touched by this loop
                for (int s = 0; s < 2; ++s) {
                    const int corner = kEdge[loop[k]][s];
                    if (!((c >> corner) & 1)) { for (int a = 0; a < 3; ++a) inside[a] += kCorner[corner][a]; ++n_in; }
                }
            for (int a = 0; a < 3; ++a) inside[a] /= n_in;
            for (size_t k = 0; k < loop.size(); ++k) { // Newell normal of the polygon
                double p[3], q[3];
                mid(loop[k], p); mid(loop[(k + 1) % loop.size()], q);
                normal[0] += (p[1] - q[1]) * (p[2] + q[2]); normal[1] += (p[2] - q[2]) * (p[0] + q[0]); normal[2] += (p[0] - q[0]) * (p[1] + q[1]);
            }
            const double side = normal[0] * (centre[0] - inside[0]) + normal[1] * (centre[1] - inside[1]) + normal[2] * (centre[2] - inside[2]);
            if (side < 0) for (size_t a = 0, b = loop.size() - 1; a < b; ++a, --b) { const int t = loop[a]; loop[a] = loop[b]; loop[b] = t; }
            // 3. triangulate the polygon without a diagonal that lies IN a face of the cube (both of its end points on edges of one face): there the
            //    neighbouring cell runs its own segments, and a triangle edge on top of them leaves the surface open along it.  The reference's
            //    table has no such diagonal in any row (mc_table_golden.json: in_face_diagonals).  Fans first, from each vertex in loop order; then
            //    (never needed for the 256 cases, kept for completeness) any triangulation, by recursion over the apex of the edge (first, last).
            const int n = (int)loop.size();
            std::vector<int> tris;
            bool done = false;
            for (int s = 0; s < n && !done; ++s) {
                bool ok = true;
                for (int k = 2; k + 1 < n && ok; ++k) ok = !OnOneFace(loop[s], loop[(s + k) % n]);
                if (!ok) continue;
                for (int k = 1; k + 1 < n; ++k) { tris.push_back(loop[s]); tris.push_back(loop[(s + k) % n]); tris.push_back(loop[(s + k + 1) % n]); }
                done = true;
            }
            if (!done) done = TriangulateAvoidingFaces(loop, 0, n - 1, tris);
            if (!done) for (int k = 1; k + 1 < n; ++k) { tris.push_back(loop[0]); tris.push_back(loop[k]); tris.push_back(loop[k + 1]); }
            for (size_t k = 0; k + 2 < tris.size() && out + 3 <= 15; k += 3) { row[out++] = tris[k]; row[out++] = tris[k + 1]; row[out++] = tris[k + 2]; }
        }
    }
}

void SetMarchingCubeTables(const int* tri_table, const int* edge_pairs) { g_tri = tri_table; g_edges = edge_pairs; }

void GetMarchingCubeTables(const int** tri_table, const int** edge_pairs) {
    if (!g_tri || !g_edges) {
        std::call_once(g_default_once, [] { GenerateMarchingCubeTables(g_default_tri, g_default_edges); });
        *tri_table = g_default_tri; *edge_pairs = g_default_edges;
        return;
    }
    *tri_table = g_tri; *edge_pairs = g_edges;
}

} // namespace integration
} // namespace one_piece

// C entry for tests / other languages: the generated default tables
extern "C" void op_host_generate_mc_tables(int* tri_table_256x16, int* edge_pairs_12x2) {
    one_piece::integration::GenerateMarchingCubeTables(tri_table_256x16, edge_pairs_12x2);
}
