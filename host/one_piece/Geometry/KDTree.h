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

    // ---- RadiusSearch (KDTree.h:99-146)
    void RadiusSearch(const geometry::VectorX& point, std::vector<int>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter& sp = SearchParameter()) {
        std::vector<size_t> _indices;
        RadiusSearch(point, _indices, dists, radius, max_result, sp);
        Narrow(_indices, indices);
    }
    void RadiusSearch(const geometry::VectorX& point, std::vector<size_t>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter& sp = SearchParameter()) {
        if (point.rows() != T) {
            std::cout << RED << "[ERROR]::[RadiusSearch]::Wrong dimension!" << RESET << std::endl;
            return;
        }
        geometry::Vector<T> _point;
        for (int d = 0; d < T; ++d) _point(d) = point(d);
        RadiusSearch(_point, indices, dists, radius, max_result, sp);
    }
    void RadiusSearch(const geometry::Vector<T>& point, std::vector<int>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter sp = SearchParameter()) {
        std::vector<size_t> _indices;
        RadiusSearch(point, _indices, dists, radius, max_result, sp);
        Narrow(_indices, indices);
    }
    void RadiusSearch(const geometry::Vector<T>& point, std::vector<size_t>& indices, std::vector<float>& dists, double radius, size_t max_result,
                      const SearchParameter sp = SearchParameter()) {
        float q[T];
        for (int d = 0; d < T; ++d) q[d] = point(d);
        std::vector<std::pair<size_t, float> > ret_matches;
        size_t search_num = tree.radius(q, static_cast<float>(radius), ret_matches, static_cast<size_t>(max_result * 2.5), sp.eps, sp.sorted);
        if (search_num > max_result) search_num = max_result;
        indices.resize(search_num);
        dists.resize(search_num);
        for (size_t i = 0; i < search_num; ++i) {
            indices[i] = ret_matches[i].first;
            dists[i] = ret_matches[i].second;
        }
    }

    // ---- KnnSearch (KDTree.h:147-196)
    void KnnSearch(const geometry::VectorX& point, std::vector<int>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        std::vector<size_t> _indices;
        KnnSearch(point, _indices, dists, k, sp);
        Narrow(_indices, indices);
    }
    void KnnSearch(const geometry::VectorX& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        if (point.rows() != T) {
            std::cout << RED << "[ERROR]::[KnnSearch]::Wrong dimension!" << RESET << std::endl;
            return;
        }
        geometry::Vector<T> _point;
        for (int d = 0; d < T; ++d) _point(d) = point(d);
        KnnSearch(_point, indices, dists, k, sp);
    }
    void KnnSearch(const geometry::Vector<T>& point, std::vector<int>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        std::vector<size_t> _indices;
        KnnSearch(point, _indices, dists, k, sp);
        Narrow(_indices, indices);
    }
    void KnnSearch(const geometry::Vector<T>& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, const SearchParameter& sp = SearchParameter()) {
        (void)sp; // the reference calls knnSearch with nanoflann's default parameters whatever sp says
        float q[T];
        for (int d = 0; d < T; ++d) q[d] = point(d);
        const size_t kk = k > 0 ? (size_t)k : 0;
        indices.resize(kk);
        std::vector<float> out_dist_sqr(kk);
        const size_t search_num = kk ? tree.knn(q, kk, &indices[0], &out_dist_sqr[0]) : 0;
        indices.resize(search_num);
        dists.resize(search_num);
        for (size_t i = 0; i < search_num; ++i) dists[i] = out_dist_sqr[i];
    }

    // ---- KnnRadiusSearch (KDTree.h:197-255): the k nearest, then the prefix whose squared distance does not exceed `radius`
    void KnnRadiusSearch(const geometry::VectorX& point, std::vector<int>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) {
        std::vector<size_t> _indices;
        KnnRadiusSearch(point, _indices, dists, k, radius, sp);
        Narrow(_indices, indices);
    }
    void KnnRadiusSearch(const geometry::VectorX& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) {
        if (point.rows() != T) {
            std::cout << RED << "[ERROR]::[KnnSearch]::Wrong dimension!" << RESET << std::endl;
            return;
        }
        geometry::Vector<T> _point;
        for (int d = 0; d < T; ++d) _point(d) = point(d);
        KnnRadiusSearch(_point, indices, dists, k, radius, sp);
    }
    void KnnRadiusSearch(const geometry::Vector<T>& point, std::vector<int>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) {
        std::vector<size_t> _indices;
        KnnRadiusSearch(point, _indices, dists, k, radius, sp);
        Narrow(_indices, indices);
    }
    void KnnRadiusSearch(const geometry::Vector<T>& point, std::vector<size_t>& indices, std::vector<float>& dists, int k, float radius,
                         const SearchParameter& sp = SearchParameter()) {
        KnnSearch(point, indices, dists, k, sp);
        size_t in_radius = 0;
        for (; in_radius != indices.size(); ++in_radius)
            if (dists[in_radius] > radius) break;
        indices.resize(in_radius);
        dists.resize(in_radius);
    }

protected:
    static void Narrow(const std::vector<size_t>& from, std::vector<int>& to) {
        to.resize(from.size());
        for (size_t i = 0; i != from.size(); ++i) to[i] = static_cast<int>(from[i]);
    }

    op_host::NanoTreeT<T> tree;
    std::vector<float> flat; // the points, T floats each (nanoflann's dataset adaptor reads the caller's list; here they are copied once)
    int max_leaf = 10;
};

} // namespace geometry
} // namespace one_piece
