// onepiece_nanotree.hpp -- host-side (CPU) nearest-neighbour searches that answer the way the reference's kd-tree does.
//
// The reference searches with nanoflann 1.3.2 (3rdparty/nanoflann/include/nanoflann.hpp -- with one local addition, a neighbour cap in
// the radius result set) through Geometry/KDTree.h:62-98: KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, points>, points, D>, leaf
// size 10.  Which of several EXACTLY equidistant points it reports, in which order a k-NN list comes back, and which neighbours a capped
// radius search stops at are all decided by its depth-first traversal, i.e. by the tree: which dimension each node cuts, where, and how
// the three-way partition permutes the index array on the way.  This header builds the same tree from the published algorithm (bounding
// box, middle split with the spread test, plane split, near child first with the incremental per-dimension bound) and runs the same
// three searches over it; float arithmetic throughout, as in the reference's instantiation.
//
// Users: the ICP path (onepiece_amd/csrc/icp_iter.hip re-decides tied queries with nearest()), and the class surface's geometry::KDTree<D>
// (host/one_piece/Geometry/KDTree.h).  Checked against the real library's answers: tests/golden/nanoflann_golden.json.
// Nodes are split on first visit (build() only lays down the root): a handful of queries -- the ICP path's tied ones -- then cost the few
// root-to-leaf paths they walk, ~2 passes over the points, instead of the whole O(n log n) construction; finish() completes the tree (what
// geometry::KDTree does, and what concurrent queries need: splitting is not thread-safe, searching a finished tree is).
// Header-only, C++11, no dependencies.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace op_host {

template <int D>
class NanoTreeT {
public:
    bool built() const { return ready_; }
    size_t size() const { return n_; }

    // points: n x D floats, kept by reference (the caller owns them for the life of the tree).  complete = false leaves every node to be
    // split when a search first reaches it.
    void build(const float* points, size_t n, size_t leaf_size = 10, bool complete = true) {
        pts_ = points; n_ = n; leaf_ = leaf_size; ready_ = true; finished_ = false;
        order_.resize(n);
        for (size_t i = 0; i < n; ++i) order_[i] = i;
        xyz_.assign(points, points + n * (size_t)D); // the coordinates in the order of `order_`: splits and leaves scan them front to back
        nodes_.clear();
        if (!n) { finished_ = true; return; }
        nodes_.reserve(complete ? n / 4 + 16 : 256);
        Node root;
        for (int d = 0; d < D; ++d) root.cell.lo[d] = root.cell.hi[d] = points[d];
        for (size_t k = 1; k < n; ++k)
            for (int d = 0; d < D; ++d) {
                const float v = points[(size_t)D * k + d];
                if (v < root.cell.lo[d]) root.cell.lo[d] = v;
                if (v > root.cell.hi[d]) root.cell.hi[d] = v;
            }
        root.begin = 0; root.end = n;
        root_ = root.cell;
        nodes_.push_back(root);
        if (complete) finish();
    }
    // splits every node that has not been visited yet; afterwards searches do not modify the tree (and may run concurrently)
    void finish() {
        if (finished_) return;
        for (size_t id = 0; id < nodes_.size(); ++id) split((int32_t)id); // (children are appended behind their parent: one sweep reaches them all)
        finished_ = true;
    }

    // knnSearch(query, k, indices, squared distances): the k nearest in ascending distance, equally distant ones in the order the
    // traversal met them; returns how many were found (< k only if the tree holds fewer points, or for NaN / infinite queries)
    size_t knn(const float* q, size_t k, size_t* indices, float* dists) const {
        if (!n_ || !k) return 0;
        KnnSet set{indices, dists, k, 0};
        dists[k - 1] = FLT_MAX;
        search(q, set, 1.0f);
        return set.count;
    }

    // index of the nearest point as knnSearch(query, 1, ...) reports it; -1 when it reports none
    int nearest(const float* q) const {
        size_t idx = 0;
        float d = 0;
        return knn(q, 1, &idx, &d) ? (int)idx : -1;
    }

    // radiusSearch(query, radius, out, max_neighbors, SearchParams(_, eps, sorted)): every point whose squared distance is below `radius`
    // (the metric is squared: so is the bound), in traversal order, stopping once max_neighbors (> 0) have been collected; then sorted
    // by distance if asked.  Returns out.size().
    size_t radius(const float* q, float radius, std::vector<std::pair<size_t, float> >& out, size_t max_neighbors, float eps, bool sorted) const {
        out.clear();
        if (!n_) return 0;
        RadiusSet set{&out, radius, max_neighbors};
        search(q, set, 1 + eps);
        if (sorted) std::sort(out.begin(), out.end(), [](const std::pair<size_t, float>& a, const std::pair<size_t, float>& b) { return a.second < b.second; });
        return out.size();
    }

private:
    struct Box { float lo[D], hi[D]; };
    // a node covers order_[begin, end) and remembers the cell it was handed; `state`: 0 = not looked at yet, 1 = leaf, 2 = split into low_part / high_part
    struct Node { Box cell; size_t begin = 0, end = 0; int32_t low_part = -1, high_part = -1; int axis = 0, state = 0; float below = 0, above = 0; };

    // nanoflann's KNNResultSet: an equally distant newcomer goes BEHIND the entries already there, and is dropped when the set is full
    struct KnnSet {
        size_t* indices; float* dists; size_t capacity, count;
        float worst() const { return dists[capacity - 1]; }
        bool add(float dist, size_t index) {
            size_t i;
            for (i = count; i > 0; --i) {
                if (dists[i - 1] > dist) { if (i < capacity) { dists[i] = dists[i - 1]; indices[i] = indices[i - 1]; } }
                else break;
            }
            if (i < capacity) { dists[i] = dist; indices[i] = index; }
            if (count < capacity) count++;
            return true;
        }
    };
    // the reference's RadiusResultSet (with its neighbour cap: the search stops at the first candidate AFTER the cap was reached)
    struct RadiusSet {
        std::vector<std::pair<size_t, float> >* out; float radius; size_t max_neighbors;
        float worst() const { return radius; }
        bool add(float dist, size_t index) {
            if (max_neighbors > 0 && out->size() >= max_neighbors) return false;
            if (dist < radius) out->push_back(std::make_pair(index, dist));
            return true;
        }
    };

    const float* pts_ = nullptr;
    size_t n_ = 0, leaf_ = 10;
    bool ready_ = false;
    mutable bool finished_ = false;
    mutable std::vector<size_t> order_;   // nanoflann's vind: a leaf covers order_[begin .. end)
    mutable std::vector<float> xyz_;      // xyz_[D * slot ..] = the point order_[slot] (kept in step with every swap of order_)
    mutable std::vector<Node> nodes_;     // (searches split the nodes they are the first to reach)
    Box root_;

    float coord(size_t slot, int d) const { return xyz_[(size_t)D * slot + d]; }

    void range(size_t begin, size_t count, int d, float& mn, float& mx) const {
        mn = mx = coord(begin, d);
        for (size_t i = 1; i < count; ++i) {
            const float v = coord(begin + i, d);
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }

    // three-way partition of order_[begin, begin + count) about `cut` along d: [< cut | == cut | > cut); returns the two boundaries
    std::pair<size_t, size_t> partition(size_t begin, size_t count, int d, float cut) const {
        size_t* ind = order_.data() + begin;
        float* c = xyz_.data() + (size_t)D * begin;
        auto exchange = [&](size_t i, size_t j) {
            std::swap(ind[i], ind[j]);
            for (int k = 0; k < D; ++k) std::swap(c[(size_t)D * i + k], c[(size_t)D * j + k]);
        };
        size_t l = 0, r = count - 1;
        for (;;) {
            while (l <= r && c[(size_t)D * l + d] < cut) ++l;
            while (r && l <= r && c[(size_t)D * r + d] >= cut) --r;
            if (l > r || !r) break;
            exchange(l, r);
            ++l; --r;
        }
        const size_t first = l;
        r = count - 1;
        for (;;) {
            while (l <= r && c[(size_t)D * l + d] <= cut) ++l;
            while (r && l <= r && c[(size_t)D * r + d] > cut) --r;
            if (l > r || !r) break;
            exchange(l, r);
            ++l; --r;
        }
        return std::make_pair(first, l);
    }

    // decides what node `id` is: a leaf, or two children about a cut (nanoflann's divideTree + middleSplit_ for this one node).  The distances a
    // search needs at the node -- the largest coordinate below the cut, the smallest above it, which nanoflann takes from the children's bounding
    // boxes on its way back up -- are read off the two parts directly.
    void split(int32_t id) const {
        if (nodes_[id].state) return;
        const size_t begin = nodes_[id].begin, end = nodes_[id].end, count = end - begin;
        if (count <= leaf_) { nodes_[id].state = 1; return; }
        const Box box = nodes_[id].cell;
        // the dimension: among those whose cell span is within 1e-5 of the widest, the one over which the points spread most
        const float eps = 0.00001f;
        float widest = box.hi[0] - box.lo[0];
        for (int d = 1; d < D; ++d) { const float span = box.hi[d] - box.lo[d]; if (span > widest) widest = span; }
        float best_spread = -1;
        int axis = 0;
        float lo_of[D], hi_of[D]; // the points' range along the candidate dimensions -- all of them in ONE pass when there are few (the big nodes near the root are
        bool have[D];             // what a handful of queries on an unfinished tree pay for)
        for (int d = 0; d < D; ++d) have[d] = false;
        if (D <= 4) {
            for (int d = 0; d < D; ++d) { lo_of[d] = hi_of[d] = coord(begin, d); have[d] = true; }
            const float* p = xyz_.data() + (size_t)D * begin;
            for (size_t i = 1; i < count; ++i) {
                p += D;
                for (int d = 0; d < D; ++d) { if (p[d] < lo_of[d]) lo_of[d] = p[d]; if (p[d] > hi_of[d]) hi_of[d] = p[d]; }
            }
        }
        for (int d = 0; d < D; ++d) {
            const float span = box.hi[d] - box.lo[d];
            if (span > (1 - eps) * widest) {
                if (!have[d]) { range(begin, count, d, lo_of[d], hi_of[d]); have[d] = true; }
                const float spread = hi_of[d] - lo_of[d];
                if (spread > best_spread) { axis = d; best_spread = spread; }
            }
        }
        // the cut: the middle of the cell, pulled into the range of the points
        const float middle = (box.lo[axis] + box.hi[axis]) / 2;
        if (!have[axis]) range(begin, count, axis, lo_of[axis], hi_of[axis]);
        const float mn = lo_of[axis], mx = hi_of[axis];
        const float cut = middle < mn ? mn : (middle > mx ? mx : middle);
        const std::pair<size_t, size_t> lim = partition(begin, count, axis, cut);
        const size_t half = count / 2;
        const size_t take = lim.first > half ? lim.first : (lim.second < half ? lim.second : half);
        Node low, high;
        low.cell = box; low.cell.hi[axis] = cut; low.begin = begin; low.end = begin + take;
        high.cell = box; high.cell.lo[axis] = cut; high.begin = begin + take; high.end = end;
        float unused, below, above;
        range(low.begin, take, axis, unused, below);
        range(high.begin, count - take, axis, above, unused);
        const int32_t a = (int32_t)nodes_.size();
        nodes_.push_back(low);
        nodes_.push_back(high);
        Node& nd = nodes_[id]; // (taken after the push_backs: the vector may have moved)
        nd.low_part = a; nd.high_part = a + 1; nd.axis = axis; nd.below = below; nd.above = above; nd.state = 2;
    }

    template <class Set>
    void search(const float* q, Set& set, float eps_error) const {
        float per_dim[D], bound = 0.0f;
        for (int d = 0; d < D; ++d) per_dim[d] = 0.0f;
        for (int d = 0; d < D; ++d) {
            if (q[d] < root_.lo[d]) { per_dim[d] = (q[d] - root_.lo[d]) * (q[d] - root_.lo[d]); bound += per_dim[d]; }
            if (q[d] > root_.hi[d]) { per_dim[d] = (q[d] - root_.hi[d]) * (q[d] - root_.hi[d]); bound += per_dim[d]; }
        }
        descend(0, q, bound, per_dim, set, eps_error);
    }

    // false = the result set wants no more points
    template <class Set>
    bool descend(int32_t id, const float* q, float bound, float* per_dim, Set& set, float eps_error) const {
        split(id);
        const Node nd = nodes_[id]; // (a copy: splitting further down may move the vector)
        if (nd.state == 1) {
            const float worst = set.worst(); // nanoflann reads the result set's worst distance once per leaf
            for (size_t s = nd.begin; s < nd.end; ++s) {
                const float* p = xyz_.data() + (size_t)D * s;
                float dist = 0;
                for (int d = 0; d < D; ++d) { const float diff = q[d] - p[d]; dist += diff * diff; }
                if (dist < worst && !set.add(dist, order_[s])) return false;
            }
            return true;
        }
        const float v = q[nd.axis], d_below = v - nd.below, d_above = v - nd.above;
        const bool low_first = (d_below + d_above) < 0;
        const float cut = low_first ? (v - nd.above) * (v - nd.above) : (v - nd.below) * (v - nd.below);
        if (!descend(low_first ? nd.low_part : nd.high_part, q, bound, per_dim, set, eps_error)) return false;
        const float kept = per_dim[nd.axis];
        bound = bound + cut - kept;
        per_dim[nd.axis] = cut;
        if (bound * eps_error <= set.worst()) {
            if (!descend(low_first ? nd.high_part : nd.low_part, q, bound, per_dim, set, eps_error)) return false;
        }
        per_dim[nd.axis] = kept;
        return true;
    }
};

} // namespace op_host
