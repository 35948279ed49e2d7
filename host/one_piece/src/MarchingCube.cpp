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
        // 1. per face, join the crossing edges pairwise: a segment cuts off each INSIDE corner whose two face edges both
        //    cross (ambiguous faces thereby keep inside corners apart); what is left is a single pair
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
                    if (!((c >> kFace[f][k]) & 1)) join(fe[(k + 3) & 3], fe[k]);
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
            for (size_t k = 1; k + 1 < loop.size() && out + 3 <= 15; ++k) { row[out++] = loop[0]; row[out++] = loop[k]; row[out++] = loop[k + 1]; }
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
