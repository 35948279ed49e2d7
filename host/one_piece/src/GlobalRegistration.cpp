// GlobalRegistration.cpp -- FPFH features (Registration/3DFeature.h) and feature-based global registration (Registration/GlobalRegistration.h).
// Host C++ on example/DenseFusion's submap path (DenseSlam.cpp:66-118); a restatement of what the reference computes
// (src/Registration/3DFeature.cpp, GlobalRegistration.cpp), written for this library's types; see the two headers for the arithmetic that is
// kept and the one deviation.  The neighbour searches are exact (a uniform grid for the 3-D radius search, a threaded exhaustive scan for the
// 33-D nearest feature) where the reference asks nanoflann with 1024 checks.
#include "Registration/GlobalRegistration.h"

#include <algorithm>
#include <cmath>
#include <thread>
#include <unordered_map>

namespace one_piece {
namespace registration {

namespace {

constexpr int kBins = 11, kDim = 3 * kBins;

template <class F>
void ParallelFor(size_t n, size_t grain, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
    if (n < grain * 2 || nt == 1) { f(static_cast<size_t>(0), n); return; }
    const size_t per = (n + nt - 1) / nt;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t lo = t * per, hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back([=] { f(lo, hi); });
    }
    for (auto& x : th) x.join();
}

// neighbours of every point: squared distance strictly below r2, ascending (index breaks ties), at most max_n of them, the point itself first
void RadiusNeighbours(const geometry::Point3List& pts, float r2, int max_n, std::vector<std::vector<int> >& out) {
    const size_t n = pts.size();
    out.assign(n, std::vector<int>());
    if (n == 0) return;
    const float cell = std::sqrt(r2);
    std::unordered_map<geometry::Point3i, std::vector<int>, geometry::VoxelGridHasher> grid;
    auto cell_of = [cell](const geometry::Point3& p) {
        return geometry::Point3i(static_cast<int>(std::floor(p(0) / cell)), static_cast<int>(std::floor(p(1) / cell)), static_cast<int>(std::floor(p(2) / cell)));
    };
    for (size_t i = 0; i < n; ++i) grid[cell_of(pts[i])].push_back(static_cast<int>(i));
    ParallelFor(n, 256, [&](size_t lo, size_t hi) {
        std::vector<std::pair<float, int> > cand;
        for (size_t i = lo; i < hi; ++i) {
            cand.clear();
            const geometry::Point3i c = cell_of(pts[i]);
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dz = -1; dz <= 1; ++dz) {
                        auto it = grid.find(geometry::Point3i(c(0) + dx, c(1) + dy, c(2) + dz));
                        if (it == grid.end()) continue;
                        for (int j : it->second) {
                            const float d2 = (pts[static_cast<size_t>(j)] - pts[i]).squaredNorm();
                            if (d2 < r2) cand.push_back(std::make_pair(d2, j));
                        }
                    }
            std::sort(cand.begin(), cand.end());
            const size_t keep = std::min(cand.size(), static_cast<size_t>(max_n > 0 ? max_n : 0));
            out[i].resize(keep);
            for (size_t k = 0; k < keep; ++k) out[i][k] = cand[k].second;
        }
    });
}

inline int Bin(double x) { // floor(11 x) clamped to [0, 10] (3DFeature.cpp:62-71)
    int b = static_cast<int>(std::floor(kBins * x));
    return b > kBins - 1 ? kBins - 1 : (b < 0 ? 0 : b);
}

float ComputeRMSE(const geometry::PointCorrespondenceSet& inliers, const geometry::TransformationMatrix& T) { // GlobalRegistration.cpp:8-16
    float sum_error = 0.0;
    for (size_t i = 0; i != inliers.size(); ++i) {
        const geometry::Point3& p = inliers[i].first;
        float d[3];
        for (int r = 0; r < 3; ++r) d[r] = (T(r, 0) * p(0) + T(r, 1) * p(1) + T(r, 2) * p(2)) + T(r, 3) - inliers[i].second(r);
        sum_error += d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    }
    return std::sqrt(sum_error / inliers.size());
}

// matches -> pruning x 3 -> RANSAC -> result: the common tail of the two RansacRegistration overloads (GlobalRegistration.cpp:163-208, 226-263)
std::shared_ptr<RegistrationResult> RegisterMatched(const geometry::Point3List& source_points, const geometry::Point3List& target_points,
                                                    const FeatureSet& source_features, const FeatureSet& target_features, int max_iteration,
                                                    float threshold, float unscale) {
    geometry::FMatchSet matches;
    FeatureMatching3D(source_features, target_features, matches);
    std::default_random_engine engine; // default-seeded in the reference as well (GlobalRegistration.cpp:166,230): this part IS deterministic there
    RejectMatchesRanSaPC(source_points, target_points, engine, matches);
    RejectMatchesRanSaPC(source_points, target_points, engine, matches);
    RejectMatchesRanSaPC(source_points, target_points, engine, matches);
    geometry::PointCorrespondenceSet correspondence_set;
    for (size_t i = 0; i != matches.size(); ++i)
        correspondence_set.push_back(std::make_pair(source_points[static_cast<size_t>(matches[i].first)] / unscale, target_points[static_cast<size_t>(matches[i].second)] / unscale));
    geometry::PointCorrespondenceSet inliers;
    std::vector<int> inlier_ids;
    RegistrationResult result;
    result.T = geometry::EstimateRigidTransformationRANSAC(correspondence_set, inliers, inlier_ids, max_iteration, threshold);
    result.correspondence_set = inliers;
    result.rmse = ComputeRMSE(inliers, result.T);
    for (size_t i = 0; i < inlier_ids.size(); ++i) result.correspondence_set_index.push_back(matches[static_cast<size_t>(inlier_ids[i])]);
    return std::make_shared<RegistrationResult>(result);
}

} // namespace

PairDescriptor ComputePairDescriptor(const geometry::Point3& ps, const geometry::Point3& ns, const geometry::Point3& pt, const geometry::Point3& nt) {
    const geometry::Point3 delta = pt - ps;
    const float distance = delta.norm();
    const geometry::Point3 dir = delta / distance;
    const geometry::Point3 u = ns, v = u.cross(dir); // (v is not normalised in the reference either)
    if (v.norm() == 0) return PairDescriptor::Zero();
    const geometry::Point3 w = u.cross(v);
    PairDescriptor result;
    result(3) = distance;
    result(1) = v.dot(nt);
    result(2) = u.dot(dir);
    result(0) = static_cast<float>(std::atan2(w.dot(nt), u.dot(nt)));
    return result;
}

void ComputeFPFHFeature(const geometry::PointCloud& pcd, FeatureSet& fpfh_features, int knn, float radius) {
    const size_t n = pcd.points.size();
    Feature zero;
    zero.resize(kDim);
    zero.setZero();
    fpfh_features.assign(n, zero);
    if (n == 0 || pcd.normals.size() != n) return;
    std::vector<std::vector<int> > found;
    RadiusNeighbours(pcd.points, radius, knn, found); // `radius` is compared with SQUARED distances, like the reference's search (3DFeature.h header)
    // simplified histograms of every point over its neighbours (3DFeature.cpp:30-84); neighbours[i] = found[i] without the point itself
    std::vector<std::vector<float> > spfh(n, std::vector<float>(kDim, 0.0f));
    ParallelFor(n, 256, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const int m = static_cast<int>(found[i].size());
            if (m - 1 <= 0) continue;
            const double each = 100 / (m - 1); // integer quotient, as written in the reference
            for (int j = 1; j < m; ++j) {
                const size_t q = static_cast<size_t>(found[i][static_cast<size_t>(j)]);
                const PairDescriptor d = ComputePairDescriptor(pcd.points[i], pcd.normals[i], pcd.points[q], pcd.normals[q]);
                spfh[i][static_cast<size_t>(Bin((d(0) + M_PI) / (2.0 * M_PI)))] += static_cast<float>(each);
                spfh[i][static_cast<size_t>(kBins + Bin((d(1) + 1) / 2.0))] += static_cast<float>(each);
                spfh[i][static_cast<size_t>(2 * kBins + Bin((d(2) + 1) / 2.0))] += static_cast<float>(each);
            }
        }
    });
    ParallelFor(n, 256, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            float acc[kDim] = {0};
            double sum[3] = {0, 0, 0};
            for (size_t j = 1; j < found[i].size(); ++j) {
                const size_t q = static_cast<size_t>(found[i][j]);
                const float dist = (pcd.points[i] - pcd.points[q]).norm();
                if (dist == 0.0f) continue;
                const float w = 1 / dist;
                for (int b = 0; b < kDim; ++b) acc[b] += w * spfh[q][static_cast<size_t>(b)];
                for (int k = 0; k < 3; ++k) {
                    float s = 0;
                    for (int b = 0; b < kBins; ++b) s += spfh[q][static_cast<size_t>(k * kBins + b)];
                    sum[k] += s;
                }
            }
            Feature& f = fpfh_features[i];
            for (int k = 0; k < 3; ++k) {
                const float scale = sum[k] != 0 ? static_cast<float>(100.0 / sum[k]) : 0.0f; // (the reference stores NaN when the sum is 0)
                for (int b = 0; b < kBins; ++b) f(k * kBins + b) = acc[k * kBins + b] * scale + spfh[i][static_cast<size_t>(k * kBins + b)];
            }
        }
    });
}

void FeatureMatching3D(const FeatureSet& source_feature, const FeatureSet& target_feature, geometry::FMatchSet& matching_index) {
    const size_t ns = source_feature.size(), nt = target_feature.size();
    std::vector<int> nearest(ns, -1);
    if (nt) {
        std::vector<float> tgt(nt * kDim); // contiguous copy: the scan is memory-bound
        for (size_t j = 0; j < nt; ++j)
            for (int b = 0; b < kDim; ++b) tgt[j * kDim + b] = b < target_feature[j].rows() ? target_feature[j](b) : 0.0f;
        ParallelFor(ns, 16, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                float q[kDim];
                for (int b = 0; b < kDim; ++b) q[b] = b < source_feature[i].rows() ? source_feature[i](b) : 0.0f;
                float best = 0; int arg = -1;
                for (size_t j = 0; j < nt; ++j) {
                    const float* t = &tgt[j * kDim];
                    float d2 = 0;
                    for (int b = 0; b < kDim; ++b) { const float e = q[b] - t[b]; d2 += e * e; }
                    if (arg < 0 || d2 < best) { best = d2; arg = static_cast<int>(j); }
                }
                nearest[i] = arg;
            }
        });
    }
    matching_index.clear();
    for (size_t i = 0; i != ns; ++i)
        if (nearest[i] >= 0) matching_index.push_back(std::make_pair(static_cast<int>(i), nearest[i]));
}

void RejectMatchesRanSaPC(const geometry::Point3List& source_points, const geometry::Point3List& target_points, std::default_random_engine& engine,
                          geometry::FMatchSet& init_matches, int candidate_num, float difference) {
    const size_t N = init_matches.size();
    geometry::FMatchSet filtered_matches;
    if (N == 0) return;
    filtered_matches.reserve(N);
    std::uniform_int_distribution<int> uniform(0, static_cast<int>(N) - 1);
    for (size_t i = 0; i < N; i++) {
        const geometry::Point3& ref_point = source_points[static_cast<size_t>(init_matches[i].first)];
        const geometry::Point3& new_point = target_points[static_cast<size_t>(init_matches[i].second)];
        bool keeps_a_distance = false;
        for (int j = 0; j < candidate_num && !keeps_a_distance; j++) {
            const geometry::FMatch& other = init_matches[static_cast<size_t>(uniform(engine))];
            const float d1 = (source_points[static_cast<size_t>(other.first)] - ref_point).norm();
            const float d2 = (target_points[static_cast<size_t>(other.second)] - new_point).norm();
            keeps_a_distance = std::fabs(d1 - d2) <= difference * d1;
        }
        if (keeps_a_distance) filtered_matches.push_back(init_matches[i]);
    }
    init_matches = filtered_matches;
}

std::tuple<geometry::PointCloud, FeatureSet> DownSampleAndExtractFeature(const geometry::PointCloud& pcd, const RANSACParameter& r_para) {
    std::shared_ptr<geometry::PointCloud> ptr = pcd.DownSample(r_para.voxel_len);
    if (!ptr->HasNormals()) ptr->EstimateNormals(r_para.search_radius_normal, r_para.max_nn_normal);
    FeatureSet features;
    ComputeFPFHFeature(*ptr, features, r_para.max_nn, r_para.search_radius);
    return std::make_tuple(*ptr, features);
}

std::shared_ptr<RegistrationResult> RansacRegistration(const geometry::PointCloud& source_feature_pcd, const geometry::PointCloud& target_feature_pcd,
                                                       const FeatureSet& source_features, const FeatureSet& target_features,
                                                       const RANSACParameter& r_para) {
    return RegisterMatched(source_feature_pcd.points, target_feature_pcd.points, source_features, target_features, r_para.max_iteration,
                           static_cast<float>(r_para.threshold), 1.0f);
}

std::shared_ptr<RegistrationResult> RansacRegistration(const geometry::PointCloud& _source_pcd, const geometry::PointCloud& _target_pcd,
                                                       const RANSACParameter& r_para) {
    geometry::PointCloud source_pcd = _source_pcd, target_pcd = _target_pcd;
    const float scaling = static_cast<float>(r_para.scaling);
    if (r_para.scaling != 1) { // features are computed on the scaled clouds, the transform on the unscaled pairs (GlobalRegistration.cpp:126-132,186-191)
        for (size_t i = 0; i != source_pcd.points.size(); ++i) source_pcd.points[i] = source_pcd.points[i] * scaling;
        for (size_t i = 0; i != target_pcd.points.size(); ++i) target_pcd.points[i] = target_pcd.points[i] * scaling;
    }
    std::shared_ptr<geometry::PointCloud> source_ptr = source_pcd.DownSample(r_para.voxel_len), target_ptr = target_pcd.DownSample(r_para.voxel_len);
    if (!source_ptr->HasNormals()) source_ptr->EstimateNormals(r_para.search_radius_normal, r_para.max_nn_normal);
    if (!target_ptr->HasNormals()) target_ptr->EstimateNormals(r_para.search_radius_normal, r_para.max_nn_normal);
    FeatureSet source_features, target_features;
    ComputeFPFHFeature(*source_ptr, source_features, r_para.max_nn, r_para.search_radius);
    ComputeFPFHFeature(*target_ptr, target_features, r_para.max_nn, r_para.search_radius);
    return RegisterMatched(source_ptr->points, target_ptr->points, source_features, target_features, r_para.max_iteration,
                           static_cast<float>(r_para.threshold / r_para.scaling), scaling);
}

} // namespace registration
} // namespace one_piece
