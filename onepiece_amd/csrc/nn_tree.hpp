// nn_tree.hpp -- host-side nearest neighbour of ONE query the way the reference's kd-tree answers it, for the rare queries whose
// nearest candidates are EXACTLY equidistant (OP_ICP_OPT_TIES = OP_ICP_TIES_REFERENCE).
//
// The reference searches with nanoflann 1.3.2 (3rdparty/nanoflann, driven by Geometry/KDTree.h:62-98,171-190: single-index
// adaptor over float triples, L2_Simple metric, leaf size 10, eps = 0).  Among equally distant points it returns the one its
// depth-first traversal reaches first, and that order is a property of the tree: which dimension each node cuts, where, and how
// the three-way partition permutes the index array on the way.  The device search (uniform grid, icp.hip) returns the smallest
// index instead, and reports the queries for which that choice was not forced.  For those -- none on depth-derived clouds, all of
// them on a lattice -- this file builds the same tree nanoflann would (published algorithm: bounding box, middle split with the
// spread test, plane split, near child first with the incremental per-dimension bound, a one-entry result set that keeps the
// first of equals) and repeats the search on the host.  Float arithmetic throughout, as in the reference's instantiation.
// Checked against the real library's answers in tests/golden/nanoflann_golden.json (tests/test_icp_gpu.py).
#pragma once
#include <cfloat>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace op_host {

class NanoTree {
public:
    bool built() const { return ready_; }
    size_t size() const { return n_; }

    // points: n x 3 floats, kept by reference (the caller owns them for the life of the tree)
    void build(const float* points, size_t n) {
        pts_ = points; n_ = n; ready_ = true;
        order_.resize(n);
        for (size_t i = 0; i < n; ++i) order_[i] = i;
        nodes_.clear();
        if (!n) return;
        nodes_.reserve(n / 4 + 16);
        Box root;
        for (int d = 0; d < 3; ++d) root.lo[d] = root.hi[d] = points[d];
        for (size_t k = 1; k < n; ++k)
            for (int d = 0; d < 3; ++d) {
                const float v = points[3 * k + d];
                if (v < root.lo[d]) root.lo[d] = v;
                if (v > root.hi[d]) root.hi[d] = v;
            }
        split(0, n, root);
        root_ = root;
    }

    // index of the nearest point as nanoflann's knnSearch(query, 1, ...) reports it; -1 when it reports none (empty tree, or no
    // point at a distance below FLT_MAX: NaN / infinite queries)
    int nearest(const float q[3]) const {
        if (!n_) return -1;
        Best best{FLT_MAX, -1};
        float per_dim[3] = {0.0f, 0.0f, 0.0f}, bound = 0.0f;
        for (int d = 0; d < 3; ++d) {
            if (q[d] < root_.lo[d]) { per_dim[d] = (q[d] - root_.lo[d]) * (q[d] - root_.lo[d]); bound += per_dim[d]; }
            if (q[d] > root_.hi[d]) { per_dim[d] = (q[d] - root_.hi[d]) * (q[d] - root_.hi[d]); bound += per_dim[d]; }
        }
        descend(0, q, bound, per_dim, best);
        return best.index;
    }

private:
    struct Box { float lo[3], hi[3]; };
    struct Node { int32_t low_part = -1, high_part = -1; size_t begin = 0, end = 0; int axis = 0; float below = 0, above = 0; }; // children: the points below / above the cut (-1 = leaf)
    struct Best { float dist; int index; };

    const float* pts_ = nullptr;
    size_t n_ = 0;
    bool ready_ = false;
    std::vector<size_t> order_;   // nanoflann's vind: leaf i covers order_[begin .. end)
    std::vector<Node> nodes_;
    Box root_{};

    float coord(size_t slot, int d) const { return pts_[3 * order_[slot] + d]; }

    void range(size_t begin, size_t count, int d, float& mn, float& mx) const {
        mn = mx = coord(begin, d);
        for (size_t i = 1; i < count; ++i) {
            const float v = coord(begin + i, d);
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }

    // three-way partition of order_[begin, begin + count) about `cut` along d: [< cut | == cut | > cut); returns the two boundaries
    std::pair<size_t, size_t> partition(size_t begin, size_t count, int d, float cut) {
        size_t* ind = order_.data() + begin;
        auto at = [&](size_t i) { return pts_[3 * ind[i] + d]; };
        size_t l = 0, r = count - 1;
        for (;;) {
            while (l <= r && at(l) < cut) ++l;
            while (r && l <= r && at(r) >= cut) --r;
            if (l > r || !r) break;
            std::swap(ind[l], ind[r]);
            ++l; --r;
        }
        const size_t first = l;
        r = count - 1;
        for (;;) {
            while (l <= r && at(l) <= cut) ++l;
            while (r && l <= r && at(r) > cut) --r;
            if (l > r || !r) break;
            std::swap(ind[l], ind[r]);
            ++l; --r;
        }
        return {first, l};
    }

    // builds the subtree over order_[begin, end); `box` comes in as the cell and goes out as the bounding box of the points
    int32_t split(size_t begin, size_t end, Box& box) {
        const int32_t id = (int32_t)nodes_.size();
        nodes_.emplace_back();
        const size_t count = end - begin;
        if (count <= 10) { // KDTree(int _max_leaf = 10)
            nodes_[id].begin = begin; nodes_[id].end = end;
            for (int d = 0; d < 3; ++d) box.lo[d] = box.hi[d] = coord(begin, d);
            for (size_t k = begin + 1; k < end; ++k)
                for (int d = 0; d < 3; ++d) {
                    const float v = coord(k, d);
                    if (box.lo[d] > v) box.lo[d] = v;
                    if (box.hi[d] < v) box.hi[d] = v;
                }
            return id;
        }
        // the dimension: among those whose cell span is within 1e-5 of the widest, the one over which the points spread most
        const float eps = 0.00001f;
        float widest = box.hi[0] - box.lo[0];
        for (int d = 1; d < 3; ++d) { const float span = box.hi[d] - box.lo[d]; if (span > widest) widest = span; }
        float best_spread = -1;
        int axis = 0;
        for (int d = 0; d < 3; ++d) {
            const float span = box.hi[d] - box.lo[d];
            if (span > (1 - eps) * widest) {
                float mn, mx;
                range(begin, count, d, mn, mx);
                const float spread = mx - mn;
                if (spread > best_spread) { axis = d; best_spread = spread; }
            }
        }
        // the cut: the middle of the cell, pulled into the range of the points
        const float middle = (box.lo[axis] + box.hi[axis]) / 2;
        float mn, mx;
        range(begin, count, axis, mn, mx);
        const float cut = middle < mn ? mn : (middle > mx ? mx : middle);
        const std::pair<size_t, size_t> lim = partition(begin, count, axis, cut);
        const size_t half = count / 2;
        const size_t take = lim.first > half ? lim.first : (lim.second < half ? lim.second : half);
        Box low = box, high = box;
        low.hi[axis] = cut;
        const int32_t a = split(begin, begin + take, low);
        high.lo[axis] = cut;
        const int32_t b = split(begin + take, end, high);
        Node& nd = nodes_[id]; // (taken after the recursion: the vector may have moved)
        nd.low_part = a; nd.high_part = b; nd.axis = axis; nd.begin = begin; nd.end = end;
        nd.below = low.hi[axis]; nd.above = high.lo[axis];
        for (int d = 0; d < 3; ++d) { box.lo[d] = low.lo[d] < high.lo[d] ? low.lo[d] : high.lo[d]; box.hi[d] = low.hi[d] > high.hi[d] ? low.hi[d] : high.hi[d]; }
        return id;
    }

    void descend(int32_t id, const float q[3], float bound, float per_dim[3], Best& best) const {
        const Node& nd = nodes_[id];
        if (nd.low_part < 0) {
            const float worst = best.dist; // nanoflann reads the result set's worst distance once per leaf
            for (size_t s = nd.begin; s < nd.end; ++s) {
                const float* p = pts_ + 3 * order_[s];
                float dist = 0;
                for (int d = 0; d < 3; ++d) { const float diff = q[d] - p[d]; dist += diff * diff; }
                // a one-entry result set: an entry is replaced only by a strictly nearer point, so the first of equals stays
                if (dist < worst && best.dist > dist) { best.dist = dist; best.index = (int)order_[s]; }
            }
            return;
        }
        const float v = q[nd.axis], d_below = v - nd.below, d_above = v - nd.above;
        const bool low_first = (d_below + d_above) < 0;
        const float cut = low_first ? (v - nd.above) * (v - nd.above) : (v - nd.below) * (v - nd.below);
        descend(low_first ? nd.low_part : nd.high_part, q, bound, per_dim, best);
        const float kept = per_dim[nd.axis];
        bound = bound + cut - kept;
        per_dim[nd.axis] = cut;
        if (bound * 1.0f <= best.dist) descend(low_first ? nd.high_part : nd.low_part, q, bound, per_dim, best);
        per_dim[nd.axis] = kept;
    }
};

} // namespace op_host
