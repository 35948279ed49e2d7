// Geometry/KDTree.h -- geometry::KDTree<T> and geometry::SearchParameter (reference: src/Geometry/KDTree.h:12-259, the NANO_IMPLAMENTATION
// branch: nanoflann 1.3.2 behind a thin wrapper).  A host-side structure in the reference and here.  The searches run over the tree
// nanoflann would build from the same points (../../include/onepiece_nanotree.hpp), so results agree with the reference's index for
// index: order of a k-NN list, the pick among exactly equidistant points, and which neighbours a capped radius search stops at.
//
// As in the reference: distances in and out are SQUARED L2 (`dists`, and `radius` is compared with them as given -- KDTree.h:133,248-252);
// SearchParameter::checks is accepted and ignored (nanoflann ignores it too), KnnSearch / KnnRadiusSearch use nanoflann's default search
// parameters (eps = 0) whatever `sp` says, RadiusSearch passes sp.eps and sp.sorted on and collects at most 2.5 x max_result neighbours
// before keeping the first max_result of them.
#pragma once
#include <algorithm>
#include <iostream>
#include <utility>
#include <vector>
#include "Geometry/Geometry.h"
#include "Tool/ConsoleColor.h"
#include "onepiece_nanotree.hpp"

namespace one_piece {
namespace geometry {

class SearchParameter {
public:
    SearchParameter(int _checks = 256, float _eps = 1e-8, bool _sorted = true) {
        checks = _checks;
        eps = _eps;
        sorted = _sorted;
    }
    bool sorted = true;
    float eps = 1e-8;
    int checks = 32;
};

template <int T = 3>
class KDTree {
public:
    KDTree(int _max_leaf = 10) : max_leaf(_max_leaf) {}

    void BuildTree(const geometry::PointXList& points) {
        flat.resize(points.size() * (size_t)T);
        for (size_t i = 0; i != points.size(); ++i) {
            if (points[i].rows() != T) {
                std::cout << RED << "[ERROR]::[BuildKDTree]::The dimension of point is not equal to the dimension of kdtree." << RESET << std::endl;
                flat.clear();
                tree.build(nullptr, 0, (size_t)max_leaf);
                return;
            }
            for (int d = 0; d < T; ++d) flat[i * (size_t)T + d] = points[i](d);
        }
        tree.build(flat.data(), points.size(), (size_t)max_leaf);
    }
    void BuildTree(const geometry::PointList<T>& points) {
        flat.resize(points.size() * (size_t)T);
        for (size_t i = 0; i != points.size(); ++i)
            for (int d = 0; d < T; ++d) flat[i * (size_t)T + d] = points[i](d);
        tree.build(flat.data(), points.size(), (size_t)max_leaf);
    }

    // The reference spells every search out four times (dynamic / fixed-size query x int / size_t indices); here the twelve public overloads
    // forward to three private templates.
    // ---- RadiusSearch (KDTree.h:99-146)
    void RadiusSearch(const geometry::VectorX& point, std::vector<int>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter& sp = SearchParameter()) { WithinRadius(point, indices, dists, radius, max_result, sp, "[ERROR]::[RadiusSearch]::Wrong dimension!"); }
    void RadiusSearch(const geometry::VectorX& point, std::vector<size_t>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter& sp = SearchParameter()) { WithinRadius(point, indices, dists, radius, max_result, sp, "[ERROR]::[RadiusSearch]::Wrong dimension!"); }
    void RadiusSearch(const geometry::Vector<T>& point, std::vector<int>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter sp = SearchParameter()) { WithinRadius(point, indices, dists, radius, max_result, sp, nullptr); }
    void RadiusSearch(const geometry::Vector<T>& point, std::vector<size_t>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter sp = SearchParameter()) { WithinRadius(point, indices, dists, radius, max_result, sp, nullptr); }

    // ---- KnnSearch (KDTree.h:147-196); the reference calls knnSearch with nanoflann's default parameters whatever `sp` says
    void KnnSearch(const geometry::VectorX& point, std::vector<int>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        (void)sp; Nearest(point, indices, dists, k, "[ERROR]::[KnnSearch]::Wrong dimension!");
    }
    void KnnSearch(const geometry::VectorX& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        (void)sp; Nearest(point, indices, dists, k, "[ERROR]::[KnnSearch]::Wrong dimension!");
    }
    void KnnSearch(const geometry::Vector<T>& point, std::vector<int>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        (void)sp; Nearest(point, indices, dists, k, nullptr);
    }
    void KnnSearch(const geometry::Vector<T>& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        (void)sp; Nearest(point, indices, dists, k, nullptr);
    }

    // ---- KnnRadiusSearch (KDTree.h:197-255): the k nearest, then the prefix whose squared distance does not exceed `radius`
    void KnnRadiusSearch(const geometry::VectorX& point, std::vector<int>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) { (void)sp; if (Nearest(point, indices, dists, k, "[ERROR]::[KnnSearch]::Wrong dimension!")) CutAt(radius, indices, dists); }
    void KnnRadiusSearch(const geometry::VectorX& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) { (void)sp; if (Nearest(point, indices, dists, k, "[ERROR]::[KnnSearch]::Wrong dimension!")) CutAt(radius, indices, dists); }
    void KnnRadiusSearch(const geometry::Vector<T>& point, std::vector<int>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) { (void)sp; if (Nearest(point, indices, dists, k, nullptr)) CutAt(radius, indices, dists); }
    void KnnRadiusSearch(const geometry::Vector<T>& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) { (void)sp; if (Nearest(point, indices, dists, k, nullptr)) CutAt(radius, indices, dists); }

protected:
    // the query as T floats; a dynamic vector of another length is refused with the reference's message (and the outputs are left alone, as there)
    template <class Query>
    bool Coordinates(const Query& point, float* q, const char* complaint) const {
        if (complaint && (int)point.rows() != T) {
            std::cout << RED << complaint << RESET << std::endl;
            return false;
        }
        for (int d = 0; d < T; ++d) q[d] = point(d);
        return true;
    }
    template <class Query, class Index>
    bool Nearest(const Query& point, std::vector<Index>& indices, std::vector<float>& dists, int k, const char* complaint) const {
        float q[T];
        if (!Coordinates(point, q, complaint)) return false;
        const size_t want = k > 0 ? (size_t)k : 0;
        std::vector<size_t> found(want);
        std::vector<float> d2(want);
        const size_t n = want ? tree.knn(q, want, found.data(), d2.data()) : 0;
        indices.resize(n);
        dists.resize(n);
        for (size_t i = 0; i < n; ++i) { indices[i] = static_cast<Index>(found[i]); dists[i] = d2[i]; }
        return true;
    }
    // at most 2.5 x max_result neighbours are collected (in traversal order, then sorted if asked), the first max_result of them kept
    template <class Query, class Index>
    void WithinRadius(const Query& point, std::vector<Index>& indices, std::vector<float>& dists, double radius, size_t max_result, const SearchParameter& sp,
                      const char* complaint) const {
        float q[T];
        if (!Coordinates(point, q, complaint)) return;
        std::vector<std::pair<size_t, float> > matches;
        const size_t n = std::min(tree.radius(q, static_cast<float>(radius), matches, static_cast<size_t>(max_result * 2.5), sp.eps, sp.sorted), max_result);
        indices.resize(n);
        dists.resize(n);
        for (size_t i = 0; i < n; ++i) { indices[i] = static_cast<Index>(matches[i].first); dists[i] = matches[i].second; }
    }
    template <class Index>
    static void CutAt(float radius, std::vector<Index>& indices, std::vector<float>& dists) {
        size_t keep = 0;
        while (keep != dists.size() && !(dists[keep] > radius)) ++keep;
        indices.resize(keep);
        dists.resize(keep);
    }

    op_host::NanoTreeT<T> tree;
    std::vector<float> flat; // the points, T floats each (nanoflann's dataset adaptor reads the caller's list; here they are copied once)
    int max_leaf = 10;
};

} // namespace geometry
} // namespace one_piece
