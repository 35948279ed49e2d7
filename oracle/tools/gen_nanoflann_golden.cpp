// gen_nanoflann_golden.cpp -- generates tests/golden/nanoflann_golden.json.
//
// Runs ONLY in the build container: it includes the reference's vendored third-party header
// /root/reference/3rdparty/nanoflann/include/nanoflann.hpp (nanoflann 1.3.2) where it lies and drives it the way the reference's
// geometry::KDTree does (KDTree.h:62-98,171-190,230-255: KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, cloud>, cloud, 3>,
// max leaf 10, buildIndex(), knnSearch(query, k, indices, squared distances) with the default SearchParams, i.e. eps = 0, sorted).
// KDTree.h itself cannot be compiled here (it includes <opencv2/core/eigen.hpp>), so the adaptor below is this file's own: a plain
// array of float triples with the three members nanoflann asks a dataset for.  The JSON holds inputs and outputs only (float32
// arrays as base64 of their little-endian bytes, indices as integers); it contains no OnePiece source.
//
// What the fixture pins: the 1-NN of registration::PointToPoint / PointToPlane (ICP.cpp:69,189: k = 1) and the k-NN of
// PointCloud::EstimateNormals (PointCloud.cpp:120: k = knn) -- indices AND squared distances, bit for bit -- on inputs without
// exact-distance ties, and nanoflann's choice among exactly equidistant points (lattice cases), which follows its traversal order.
//
// Build + run: see oracle/tools/gen_golden.sh
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "nanoflann.hpp"

struct Cloud {
    std::vector<float> xyz;
    inline size_t kdtree_get_point_count() const { return xyz.size() / 3; }
    inline float kdtree_get_pt(const size_t idx, const size_t dim) const { return xyz[3 * idx + dim]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};
typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, Cloud>, Cloud, 3> Tree;

// any dimension (Registration/3DFeature.cpp searches 33-bin histograms with KDTree<33>), and the radius search as KDTree.h:131-146 calls it
template <int D>
struct CloudD {
    std::vector<float> v;
    inline size_t kdtree_get_point_count() const { return v.size() / D; }
    inline float kdtree_get_pt(const size_t idx, const size_t dim) const { return v[D * idx + dim]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};

static FILE* out;
static void b64(const char* name, const void* data, size_t bytes, bool comma = true) {
    static const char* A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const unsigned char* p = (const unsigned char*)data;
    fprintf(out, "\"%s\": \"", name);
    for (size_t i = 0; i < bytes; i += 3) {
        const unsigned v = (p[i] << 16) | ((i + 1 < bytes ? p[i + 1] : 0) << 8) | (i + 2 < bytes ? p[i + 2] : 0);
        fputc(A[(v >> 18) & 63], out); fputc(A[(v >> 12) & 63], out);
        fputc(i + 1 < bytes ? A[(v >> 6) & 63] : '=', out); fputc(i + 2 < bytes ? A[v & 63] : '=', out);
    }
    fprintf(out, "\"%s", comma ? ", " : "");
}

// one case: k nearest of every query, in nanoflann's output order
static void run_case(const char* name, const char* what, const Cloud& tgt, const std::vector<float>& q, int k, bool last = false, const char* target_of = nullptr) {
    Tree tree(3, tgt, nanoflann::KDTreeSingleIndexAdaptorParams(10));
    tree.buildIndex();
    const size_t nq = q.size() / 3;
    std::vector<int32_t> idx(nq * k, -1);
    std::vector<float> d2(nq * k, -1.0f);
    std::vector<int32_t> cnt(nq);
    for (size_t i = 0; i < nq; ++i) {
        std::vector<size_t> id(k);
        std::vector<float> dd(k);
        const size_t found = tree.knnSearch(&q[3 * i], k, &id[0], &dd[0]);
        cnt[i] = (int32_t)found;
        for (size_t j = 0; j < found; ++j) { idx[i * k + j] = (int32_t)id[j]; d2[i * k + j] = dd[j]; }
    }
    fprintf(out, "\"%s\": {\"what\": \"%s\", \"k\": %d, \"n_target\": %zu, \"n_query\": %zu, ", name, what, k, tgt.xyz.size() / 3, nq);
    if (target_of) fprintf(out, "\"target_of\": \"%s\", ", target_of);   // the same cloud as that case: not stored twice
    else b64("target_f32", tgt.xyz.data(), tgt.xyz.size() * 4);
    b64("query_f32", q.data(), q.size() * 4);
    b64("found_i32", cnt.data(), cnt.size() * 4);
    b64("index_i32", idx.data(), idx.size() * 4);
    b64("dist2_f32", d2.data(), d2.size() * 4, false);
    fprintf(out, "}%s\n", last ? "" : ",");
}

template <int D>
static void run_case_d(const char* name, const char* what, const CloudD<D>& tgt, const std::vector<float>& q, int k) {
    typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, CloudD<D> >, CloudD<D>, D> TreeD;
    TreeD tree(D, tgt, nanoflann::KDTreeSingleIndexAdaptorParams(10));
    tree.buildIndex();
    const size_t nq = q.size() / D;
    std::vector<int32_t> idx(nq * k, -1), cnt(nq);
    std::vector<float> d2(nq * k, -1.0f);
    for (size_t i = 0; i < nq; ++i) {
        std::vector<size_t> id(k);
        std::vector<float> dd(k);
        const size_t found = tree.knnSearch(&q[D * i], k, &id[0], &dd[0]);
        cnt[i] = (int32_t)found;
        for (size_t j = 0; j < found; ++j) { idx[i * k + j] = (int32_t)id[j]; d2[i * k + j] = dd[j]; }
    }
    fprintf(out, "\"%s\": {\"what\": \"%s\", \"dim\": %d, \"k\": %d, \"n_target\": %zu, \"n_query\": %zu, ", name, what, D, k, tgt.v.size() / D, nq);
    b64("target_f32", tgt.v.data(), tgt.v.size() * 4);
    b64("query_f32", q.data(), q.size() * 4);
    b64("found_i32", cnt.data(), cnt.size() * 4);
    b64("index_i32", idx.data(), idx.size() * 4);
    b64("dist2_f32", d2.data(), d2.size() * 4, false);
    fprintf(out, "},\n");
}

// KDTree::RadiusSearch (KDTree.h:131-146): radiusSearch(point, radius, matches, max_result * 2.5, SearchParams(checks, eps, sorted)), first max_result kept
static void run_radius(const char* name, const char* what, const Cloud& tgt, const std::vector<float>& q, float radius, size_t max_result, bool sorted, const char* target_of) {
    Tree tree(3, tgt, nanoflann::KDTreeSingleIndexAdaptorParams(10));
    tree.buildIndex();
    const size_t nq = q.size() / 3;
    std::vector<int32_t> idx(nq * max_result, -1), cnt(nq);
    std::vector<float> d2(nq * max_result, -1.0f);
    for (size_t i = 0; i < nq; ++i) {
        std::vector<std::pair<size_t, float> > ret;
        size_t n = tree.radiusSearch(&q[3 * i], radius, ret, max_result * 2.5, nanoflann::SearchParams(128, 1e-8f, sorted));
        if (n > max_result) n = max_result;
        cnt[i] = (int32_t)n;
        for (size_t j = 0; j < n; ++j) { idx[i * max_result + j] = (int32_t)ret[j].first; d2[i * max_result + j] = ret[j].second; }
    }
    fprintf(out, "\"%s\": {\"what\": \"%s\", \"dim\": 3, \"k\": %zu, \"radius\": %.9g, \"sorted\": %d, \"n_target\": %zu, \"n_query\": %zu, \"target_of\": \"%s\", ", name, what, max_result, radius, (int)sorted,
            tgt.xyz.size() / 3, nq, target_of);
    b64("query_f32", q.data(), q.size() * 4);
    b64("found_i32", cnt.data(), cnt.size() * 4);
    b64("index_i32", idx.data(), idx.size() * 4);
    b64("dist2_f32", d2.data(), d2.size() * 4, false);
    fprintf(out, "},\n");
}

int main(int argc, char** argv) {
    out = fopen(argc > 1 ? argv[1] : "nanoflann_golden.json", "w");
    if (!out) return 1;
    std::mt19937 g(20260929);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    fprintf(out, "{\n\"generator\": \"oracle/tools/gen_nanoflann_golden.cpp against /root/reference/3rdparty/nanoflann/include/nanoflann.hpp (version 0x%x), g++ -O3 -msse4.2; "
                 "adaptor = KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, cloud>, cloud, 3>, leaf 10, knnSearch with default SearchParams\",\n", NANOFLANN_VERSION);

    { // 1. uniform cloud, independent queries: 1-NN
        Cloud t; std::vector<float> q;
        for (int i = 0; i < 2000 * 3; ++i) t.xyz.push_back(u(g));
        for (int i = 0; i < 700 * 3; ++i) q.push_back(1.1f * u(g));   // (some queries lie outside the target's box)
        run_case("uniform_1nn", "2000 uniform targets in [-1,1]^3, 700 queries in [-1.1,1.1]^3, k = 1 (ICP.cpp:69,189)", t, q, 1);
    }
    { // 2. a depth-image-like surface and the same surface after a small rigid motion: the shape of ICP's own inputs
        Cloud t; std::vector<float> q;
        const float c = std::cos(0.03f), s = std::sin(0.03f);
        for (int v = 0; v < 40; ++v)
            for (int x = 0; x < 52; ++x) {
                const float z = 1.5f + 0.2f * std::sin(0.11f * x) * std::cos(0.07f * v) + 0.002f * u(g);
                const float X = (x * 12 + 6 - 318.771f) / 514.817f * z, Y = (v * 12 + 6 - 238.447f) / 515.375f * z;
                t.xyz.push_back(X); t.xyz.push_back(Y); t.xyz.push_back(z);
                const float z2 = z + 0.001f * u(g);
                q.push_back(c * X + s * z2 + 0.01f); q.push_back(Y - 0.015f); q.push_back(-s * X + c * z2 + 0.02f);
            }
        run_case("surface_1nn", "52 x 40 back-projected wavy sheet; queries = the sheet moved by 0.03 rad about y and (0.01,-0.015,0.02) m, k = 1", t, q, 1);
    }
    { // 3. exact-distance ties: a lattice with exactly representable coordinates plus duplicated points
        Cloud t; std::vector<float> q;
        for (int z = 0; z < 10; ++z) for (int y = 0; y < 10; ++y) for (int x = 0; x < 10; ++x) { t.xyz.push_back(0.125f * x); t.xyz.push_back(0.125f * y); t.xyz.push_back(0.125f * z); }
        std::uniform_int_distribution<int> pick(0, 999), cell(0, 8);
        for (int i = 0; i < 60; ++i) { const int j = pick(g); for (int c = 0; c < 3; ++c) t.xyz.push_back(t.xyz[3 * j + c]); }         // duplicates: distance-0 ties
        for (int i = 0; i < 300; ++i) { q.push_back(0.125f * cell(g) + 0.0625f); q.push_back(0.125f * cell(g) + 0.0625f); q.push_back(0.125f * cell(g) + 0.0625f); } // cell centres: 8-way
        for (int i = 0; i < 300; ++i) { q.push_back(0.125f * cell(g) + 0.0625f); q.push_back(0.125f * cell(g)); q.push_back(0.125f * cell(g)); }                     // edge midpoints: 2-way
        for (int i = 0; i < 200; ++i) { const int j = 1000 + pick(g) % 60; for (int c = 0; c < 3; ++c) q.push_back(t.xyz[3 * j + c]); }                                // on a duplicated lattice point
        run_case("lattice_ties_1nn", "10^3 lattice of spacing 1/8 + 60 duplicated points; queries at cell centres, edge midpoints, lattice points: every query has exactly equidistant candidates, k = 1", t, q, 1);
    }
    { // 4. k = 30 with the query a member of the cloud: PointCloud::EstimateNormals' search
        Cloud t; std::vector<float> q;
        for (int i = 0; i < 3000 * 3; ++i) t.xyz.push_back(0.5f * u(g));
        for (int i = 0; i < 150 * 3; ++i) q.push_back(t.xyz[i]);
        run_case("uniform_knn30", "3000 uniform points in [-0.5,0.5]^3, queries = its first 150 points, k = 30 (PointCloud.cpp:120)", t, q, 30);
    }
    { // 5. fewer points than k
        Cloud t; std::vector<float> q;
        for (int i = 0; i < 7 * 3; ++i) t.xyz.push_back(u(g));
        for (int i = 0; i < 5 * 3; ++i) q.push_back(u(g));
        run_case("tiny_knn30", "7 points, k = 30: knnSearch returns 7", t, q, 30);
    }
    { // 5b. a noisy surface whose coordinates are quantised to 1/128: duplicated points and equidistant pairs all over a deep tree
        Cloud t; std::vector<float> q;
        auto qz = [](float v) { return std::floor(v * 128.0f + 0.5f) / 128.0f; };
        for (int i = 0; i < 4000; ++i) { const float x = u(g), y = u(g); t.xyz.push_back(qz(x)); t.xyz.push_back(qz(y)); t.xyz.push_back(qz(0.3f * std::sin(3.0f * x) * std::cos(2.0f * y) + 0.01f * u(g))); }
        for (int i = 0; i < 1000; ++i) { const float x = u(g), y = u(g); q.push_back(qz(x)); q.push_back(qz(y)); q.push_back(qz(0.3f * std::sin(3.0f * x) * std::cos(2.0f * y) + 0.02f * u(g))); }
        run_case("quantised_1nn", "4000 points of a wavy sheet with coordinates rounded to 1/128 (duplicates, equidistant pairs), 1000 queries rounded alike, k = 1", t, q, 1);
        std::vector<float> q2(q.begin(), q.begin() + 3 * 200);
        run_case("quantised_knn30", "the same cloud, 200 of the queries, k = 30", t, q2, 30, false, "quantised_1nn");
    }
    { // 5c. the capped radius search of KDTree::RadiusSearch on the quantised sheet (its squared-distance radius, 2.5 x cap, then cut): sorted and in traversal order
        Cloud t; std::vector<float> q;
        std::mt19937 g2(20260929 + 7);
        std::uniform_real_distribution<float> u2(-1.f, 1.f);
        auto qz = [](float v) { return std::floor(v * 128.0f + 0.5f) / 128.0f; };
        // (the same generator state as case 5b is not available here: the cloud is regenerated and stored under its own name)
        for (int i = 0; i < 1500; ++i) { const float x = u2(g2), y = u2(g2); t.xyz.push_back(qz(x)); t.xyz.push_back(qz(y)); t.xyz.push_back(qz(0.3f * std::sin(3.0f * x) * std::cos(2.0f * y))); }
        for (int i = 0; i < 150; ++i) { const float x = u2(g2), y = u2(g2); q.push_back(qz(x)); q.push_back(qz(y)); q.push_back(qz(0.3f * std::sin(3.0f * x) * std::cos(2.0f * y))); }
        run_case("radius_cloud_1nn", "1500 points of the wavy sheet rounded to 1/128, 150 queries, k = 1 (holds the cloud of the radius cases)", t, q, 1);
        run_radius("radius_sorted", "RadiusSearch(radius 0.01 (squared), max_result 20): sorted by distance, at most 50 collected before the cut", t, q, 0.01f, 20, true, "radius_cloud_1nn");
        run_radius("radius_unsorted", "the same, SearchParameter::sorted = false: traversal order", t, q, 0.01f, 20, false, "radius_cloud_1nn");
        run_radius("radius_capped", "radius 0.05 (squared), max_result 8: the cap of 20 collected neighbours stops the traversal early", t, q, 0.05f, 8, true, "radius_cloud_1nn");
    }
    { // 5d. 33 dimensions with small integer coordinates (FPFH histograms are sums of integer quotients: equal distances are common)
        CloudD<33> t; std::vector<float> q;
        std::uniform_int_distribution<int> bin(0, 4);
        for (int i = 0; i < 250 * 33; ++i) t.v.push_back((float)bin(g));
        for (int i = 0; i < 60 * 33; ++i) q.push_back((float)bin(g));
        run_case_d<33>("hist33_knn5", "250 points in 33 dimensions with coordinates in {0..4}, 60 queries, k = 5", t, q, 5);
    }
    { // 6. k = 30 on the lattice: ties inside the result list
        Cloud t; std::vector<float> q;
        for (int z = 0; z < 8; ++z) for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) { t.xyz.push_back(0.25f * x); t.xyz.push_back(0.25f * y); t.xyz.push_back(0.25f * z); }
        for (int i = 0; i < 100 * 3; ++i) q.push_back(t.xyz[3 * 73 + i]);
        run_case("lattice_knn30", "8^3 lattice of spacing 1/4, queries = 100 of its points, k = 30: the sorted distance list is unique, the indices within a distance shell are not", t, q, 30, true);
    }
    fprintf(out, "}\n");
    fclose(out);
    return 0;
}
