// kdtree_check.cpp -- geometry::KDTree<T> of the class surface (host/one_piece/Geometry/KDTree.h) against the answers of the real nanoflann
// behind the reference's wrapper.  Input (written by tests/test_cpp_surface.py from tests/golden/nanoflann_golden.json): int32 dim, n, nq, k,
// kind (0 = KnnSearch, 1 = RadiusSearch with max_result = k), sorted; float32 radius; n x dim float32 targets; nq x dim float32 queries;
// nq int32 found; nq x k int32 indices; nq x k float32 squared distances.  Exit code 0 iff everything agrees bit for bit.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "Geometry/KDTree.h"

using namespace one_piece;

template <int D>
static int run(FILE* f, int n, int nq, int k, int kind, int sorted, float radius) {
    std::vector<float> t((size_t)n * D), q((size_t)nq * D), d2((size_t)nq * k);
    std::vector<int32_t> found(nq), idx((size_t)nq * k);
    if (fread(t.data(), 4, t.size(), f) != t.size() || fread(q.data(), 4, q.size(), f) != q.size() || fread(found.data(), 4, found.size(), f) != found.size() ||
        fread(idx.data(), 4, idx.size(), f) != idx.size() || fread(d2.data(), 4, d2.size(), f) != d2.size()) return 2;
    geometry::KDTree<D> tree;
    geometry::PointXList as_x;       // D != 3: the dynamic-vector overload (what 3DFeature.cpp feeds it); D == 3: the fixed-size list
    geometry::PointList<D> as_fixed;
    for (int i = 0; i < n; ++i) {
        geometry::Vector<D> p;
        for (int d = 0; d < D; ++d) p(d) = t[(size_t)i * D + d];
        as_fixed.push_back(p);
        geometry::VectorX x(D);
        for (int d = 0; d < D; ++d) x(d) = p(d);
        as_x.push_back(x);
    }
    if (D == 3) tree.BuildTree(as_fixed); else tree.BuildTree(as_x);
    int bad = 0;
    for (int i = 0; i < nq; ++i) {
        geometry::Vector<D> p;
        for (int d = 0; d < D; ++d) p(d) = q[(size_t)i * D + d];
        std::vector<int> ids;
        std::vector<float> ds;
        if (kind == 0) tree.KnnSearch(p, ids, ds, k);
        else tree.RadiusSearch(p, ids, ds, radius, (size_t)k, geometry::SearchParameter(128, 1e-8f, sorted != 0));
        if ((int)ids.size() != found[i]) { ++bad; continue; }
        for (size_t j = 0; j < ids.size(); ++j)
            if (ids[j] != idx[(size_t)i * k + j] || std::memcmp(&ds[j], &d2[(size_t)i * k + j], 4) != 0) { ++bad; break; }
    }
    printf("%d queries, %d mismatches\n", nq, bad);
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[6];
    float radius;
    if (fread(h, 4, 6, f) != 6 || fread(&radius, 4, 1, f) != 1) return 2;
    int rc = 2;
    if (h[0] == 3) rc = run<3>(f, h[1], h[2], h[3], h[4], h[5], radius);
    else if (h[0] == 33) rc = run<33>(f, h[1], h[2], h[3], h[4], h[5], radius);
    fclose(f);
    return rc;
}
