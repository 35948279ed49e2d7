// gen_mc_golden.cpp -- generates tests/golden/mc_table_golden.json.
//
// Runs ONLY in the build container: it includes the reference's Integration/MarchingCubePredefined.h where it lies (the one header on the
// mesh-extraction path that compiles without OpenCV) and writes, for each of the 256 sign configurations, a SIGNATURE of the reference's
// triangle row -- not the row: the number of triangles and the oriented boundary loops of the patch they form (cycles of cube-edge
// numbers; a directed triangle edge a->b is interior when b->a occurs too).  Two tables with the same signatures cut every cell along the
// same polygons with the same facing and differ at most in which diagonals triangulate a polygon.  Also written: the twelve corner pairs
// of EdgeIndexPairs (the numbering the signatures are expressed in) and, per case, the order-independent set of triangles as sorted
// vertex triples rotated to their smallest edge first (what decides whether two tables give the SAME triangles, not just the same polygons),
// folded into one 64-bit FNV-1a hash per case.
//
// What the fixture pins: host/one_piece's generated default tables (Integration/MarchingCube.h, used when a caller has not handed in the
// reference's own) against the reference's -- tests/test_oracle_golden.py::test_generated_mc_tables_*.
// Build + run: oracle/tools/gen_golden.sh
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
#include "Integration/MarchingCubePredefined.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out.json\n", argv[0]); return 2; }
    using namespace one_piece::integration;
    FILE* f = std::fopen(argv[1], "w");
    if (!f) return 1;
    std::fprintf(f, "{\n \"source\": \"signatures of MCLookTable / EdgeIndexPairs of the reference's Integration/MarchingCubePredefined.h (oracle/tools/gen_mc_golden.cpp)\",\n \"edge_pairs\": [");
    for (int e = 0; e < 12; ++e) std::fprintf(f, "%s[%d, %d]", e ? ", " : "", EdgeIndexPairs[e][0], EdgeIndexPairs[e][1]);
    std::fprintf(f, "],\n \"cases\": [\n");
    for (int c = 0; c < 256; ++c) {
        std::vector<int> row;
        for (int k = 0; k < 16 && MCLookTable[c][k] >= 0; ++k) row.push_back(MCLookTable[c][k]);
        const int nt = (int)row.size() / 3;
        std::map<std::pair<int, int>, int> dir;
        for (int t = 0; t < nt; ++t)
            for (int k = 0; k < 3; ++k) dir[std::make_pair(row[3 * t + k], row[3 * t + (k + 1) % 3])]++;
        std::map<int, int> next; // boundary: a -> b without b -> a
        bool manifold = true;
        for (const auto& d : dir) {
            if (d.second != 1) manifold = false;
            if (!dir.count(std::make_pair(d.first.second, d.first.first))) { if (next.count(d.first.first)) manifold = false; next[d.first.first] = d.first.second; }
        }
        std::vector<std::vector<int> > loops;
        std::map<int, bool> seen;
        for (const auto& s : next) {
            if (seen[s.first]) continue;
            std::vector<int> loop;
            int cur = s.first;
            while (!seen[cur] && next.count(cur)) { seen[cur] = true; loop.push_back(cur); cur = next[cur]; }
            std::rotate(loop.begin(), std::min_element(loop.begin(), loop.end()), loop.end());
            loops.push_back(loop);
        }
        std::sort(loops.begin(), loops.end());
        std::vector<std::vector<int> > tris;
        for (int t = 0; t < nt; ++t) {
            std::vector<int> v(row.begin() + 3 * t, row.begin() + 3 * t + 3);
            std::rotate(v.begin(), std::min_element(v.begin(), v.end()), v.end());
            tris.push_back(v);
        }
        std::sort(tris.begin(), tris.end());
        uint64_t h = 1469598103934665603ull;
        for (const auto& t : tris) for (int v : t) { h ^= (uint64_t)(v + 1); h *= 1099511628211ull; }
        // interior diagonals (a -> b with b -> a present) whose two cube edges lie on one face of the cube: such a diagonal lies IN that face, where the
        // neighbouring cell's own segments run -- a triangulation with one is not watertight against every neighbour
        int in_face = 0;
        for (const auto& d : dir) {
            const int a = d.first.first, b = d.first.second;
            if (a < b && dir.count(std::make_pair(b, a))) {
                const int* pa = EdgeIndexPairs[a]; const int* pb = EdgeIndexPairs[b];
                const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
                for (int axis = 0; axis < 3; ++axis)
                    for (int side = 0; side < 2; ++side)
                        if (corner[pa[0]][axis] == side && corner[pa[1]][axis] == side && corner[pb[0]][axis] == side && corner[pb[1]][axis] == side) ++in_face;
            }
        }
        std::fprintf(f, "  {\"case\": %d, \"triangles\": %d, \"manifold\": %s, \"in_face_diagonals\": %d, \"triangle_set_fnv1a\": \"%016llx\", \"loops\": [", c, nt, manifold ? "true" : "false", in_face, (unsigned long long)h);
        for (size_t l = 0; l < loops.size(); ++l) {
            std::fprintf(f, "%s[", l ? ", " : "");
            for (size_t k = 0; k < loops[l].size(); ++k) std::fprintf(f, "%s%d", k ? ", " : "", loops[l][k]);
            std::fprintf(f, "]");
        }
        std::fprintf(f, "]}%s\n", c == 255 ? "" : ",");
    }
    std::fprintf(f, " ]\n}\n");
    std::fclose(f);
    return 0;
}
