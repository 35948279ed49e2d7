// nn_tree_check.cpp -- op_host::NanoTree (onepiece_amd/csrc/nn_tree.hpp) against the answers of the real nanoflann.
// Input file (written by tests/test_abi_cpu.py from tests/golden/nanoflann_golden.json): int32 n, int32 nq, n x 3 float32 targets,
// nq x 3 float32 queries, nq int32 expected indices.  Prints the number of mismatches; exit code 0 iff none.
#include <cstdio>
#include <cstdint>
#include <vector>
#include "nn_tree.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t n = 0, nq = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&nq, 4, 1, f) != 1) return 2;
    std::vector<float> t((size_t)n * 3), q((size_t)nq * 3);
    std::vector<int32_t> want(nq);
    if (fread(t.data(), 4, t.size(), f) != t.size() || fread(q.data(), 4, q.size(), f) != q.size() || fread(want.data(), 4, want.size(), f) != want.size()) return 2;
    fclose(f);
    int bad = 0;
    for (int lazy = 0; lazy < 2; ++lazy) { // the finished tree, and the tree whose nodes are split as the searches reach them (what the ICP path uses)
        op_host::NanoTree tree;
        tree.build(t.data(), (size_t)n, 10, lazy == 0);
        for (int i = 0; i < nq; ++i) bad += tree.nearest(&q[3 * (size_t)i]) != want[i];
        if (lazy) { // ... and finished afterwards, it still answers the same
            tree.finish();
            for (int i = 0; i < nq; ++i) bad += tree.nearest(&q[3 * (size_t)i]) != want[i];
        }
    }
    printf("%d queries, %d mismatches\n", nq, bad);
    return bad ? 1 : 0;
}
