// RansacRigid.cpp -- geometry::EstimateRigidTransformationRANSAC (Geometry/Ransac.h).  Host C++; see the header for what is restated and what
// is not pinned.  The sampler is std::mt19937 with a FIXED seed (the reference's is seeded from std::random_device and runs one engine per OpenMP
// thread, GRANSAC.hpp:38-44): runs of this library are reproducible, runs of the reference are not.
#include "Geometry/Ransac.h"

#include <algorithm>
#include <cmath>
#include <random>
#include <thread>

namespace one_piece {
namespace geometry {

namespace {

struct Draw { int idx[MIN_INLIER_SIZE_RANSAC_TRANSFORMATION]; size_t inliers; };

// |R p + t - q| < threshold, as TransformationModel::ComputeDistanceMeasure forms it (TransformationModel.hpp:37-49)
inline bool Inlier(const TransformationMatrix& T, const PointCorrespondence& c, float threshold) {
    float d[3];
    for (int r = 0; r < 3; ++r) d[r] = (T(r, 0) * c.first(0) + T(r, 1) * c.first(1) + T(r, 2) * c.first(2)) + T(r, 3) - c.second(r);
    return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < threshold;
}

} // namespace

TransformationMatrix EstimateRigidTransformationRANSAC(const PointCorrespondenceSet& correspondence_set, PointCorrespondenceSet& inliers,
                                                       std::vector<int>& inlier_ids, int max_iteration, float threshold) {
    const size_t n = correspondence_set.size();
    constexpr int kModel = MIN_INLIER_SIZE_RANSAC_TRANSFORMATION;
    if (n < static_cast<size_t>(kModel)) {
        std::cout << YELLOW << "[Warning]::[FitPlaneRANSAC]::Too few canidate point pair." << RESET << std::endl; // (the reference's wording, Ransac.cpp:12)
        return TransformationMatrix::Zero();
    }
    if (n == static_cast<size_t>(kModel) || max_iteration <= 0) return TransformationMatrix::Zero(); // GRANSAC::Estimate declines (GRANSAC.hpp:72-76)
    // Draws are independent: the iterations are split over a few host threads, each with its own engine (seeded by its index), and the
    // winner is the draw with the most inliers, the earliest iteration on ties -- the same result for any thread count.
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
    if (static_cast<size_t>(max_iteration) * n < (1u << 22)) nt = 1;
    const int per = (max_iteration + static_cast<int>(nt) - 1) / static_cast<int>(nt);
    std::vector<Draw> best(nt);
    std::vector<int> best_iter(nt, -1);
    auto work = [&](unsigned t) {
        Draw b; b.inliers = 0;
        std::vector<int> perm(n);
        PointCorrespondenceSet sample(static_cast<size_t>(kModel));
        for (int it = static_cast<int>(t) * per; it < std::min(max_iteration, (static_cast<int>(t) + 1) * per); ++it) {
            std::mt19937 engine(0x9e3779b9u ^ static_cast<unsigned>(it) * 2654435761u); // one stream per ITERATION: independent of the thread count
            for (size_t i = 0; i < n; ++i) perm[i] = static_cast<int>(i);
            Draw d; d.inliers = 0;
            for (int k = 0; k < kModel; ++k) { // partial Fisher-Yates: 8 distinct indices, uniformly
                std::uniform_int_distribution<int> pick(k, static_cast<int>(n) - 1);
                std::swap(perm[static_cast<size_t>(k)], perm[static_cast<size_t>(pick(engine))]);
                d.idx[k] = perm[static_cast<size_t>(k)];
                sample[static_cast<size_t>(k)] = correspondence_set[static_cast<size_t>(d.idx[k])];
            }
            const TransformationMatrix T = EstimateRigidTransformation(sample);
            for (size_t i = 0; i < n; ++i) d.inliers += Inlier(T, correspondence_set[i], threshold) ? 1u : 0u;
            if (d.inliers > b.inliers) { b = d; best_iter[t] = it; }
        }
        best[t] = b;
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    int w = -1;
    for (unsigned t = 0; t < nt; ++t)
        if (best_iter[t] >= 0 && (w < 0 || best[t].inliers > best[static_cast<size_t>(w)].inliers)) w = static_cast<int>(t); // threads hold ascending iteration ranges: first best wins
    if (w < 0) return TransformationMatrix::Zero(); // no draw had a single inlier: the reference's best model stays null
    PointCorrespondenceSet model(static_cast<size_t>(kModel));
    for (int k = 0; k < kModel; ++k) model[static_cast<size_t>(k)] = correspondence_set[static_cast<size_t>(best[static_cast<size_t>(w)].idx[k])];
    const TransformationMatrix T = EstimateRigidTransformation(model);
    for (size_t i = 0; i < n; ++i)
        if (Inlier(T, correspondence_set[i], threshold)) { inliers.push_back(correspondence_set[i]); inlier_ids.push_back(static_cast<int>(i)); }
    return T;
}

} // namespace geometry
} // namespace one_piece
