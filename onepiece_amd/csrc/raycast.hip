// raycast.hip -- the raycaster north_star names ("integrate/raycast ... ray-marching laid out for coalesced HBM reads of the
// voxel-block hash"); volume_core.hpp lists the translation units.
//
// The reference has NO raycast; its closest relative is the trilinear gather of VoxelCube::ReadVoxelInterpolate
// (/root/reference/src/Integration/VoxelCube.cpp:6-50).  The definition is this library's own, restated on the CPU in
// oracle/onepiece_oracle.c (orc_volume_raycast) and validated against the analytic synthetic scene:
//
//   * the ray of pixel (u, v): origin = the pose's translation, direction d = R ((u - cx) / fx, (v - cy) / fy, 1), so the ray
//     parameter t is the z-depth in the camera frame;
//   * samples lie on a FIXED lattice t_k = near + k * res (k = 0 .. kmax, t_kmax <= far): p_k = origin + t_k d;
//   * s_k = trilinear sdf at p_k over the 8 voxel centres around it (g = p * (1 / res) - 0.5, base voxel floor(g)), VALID when
//     all 8 voxels are observed (weight > 0);
//   * the hit is the smallest k >= 1 with s_(k-1) valid and > 0, s_k valid and <= 0; depth = t_(k-1) + (t_k - t_(k-1)) *
//     s_(k-1) / (s_(k-1) - s_k) (0 = no hit).  Normal = normalised central difference of the trilinear sdf at +-res/2 around
//     the hit point, colour = trilinear colour there.
//
// The crossing test is LOCAL to the lattice pair (k-1, k), so the first hit of a ray is a minimum over independent pieces of
// the ray -- which is what lets the march run BLOCK-MAJOR instead of ray-major (round 5; rounds 1-4 marched one thread per
// ray: every sample was 16 scattered 4-byte loads behind a per-lane hash probe, 1.16 ms per 640 x 480 view of the 164 k-block
// room volume, profiles/r05_raycast_before.*):
//   R0 k_rc_visible  one thread per allocated block: the block's sample domain (the points whose base voxel lies in it) against
//                    the view frustum -> the list of visible blocks with their pixel boxes; also resets the depth image.
//   R0b k_rc_neighbours  one lane per (visible block, neighbour): the pool slots of the 26 neighbours go into the block's list entry.
//   R1 k_rc_march    one workgroup per visible block: the block's sdf + weight planes and the shell its 26 neighbours contribute
//                    (voxels -1 .. 9 of the block's frame on every axis) are read ONCE into an 11 x 11 x 11 LDS tile (unobserved =
//                    NaN, which the trilinear sum propagates by itself); a block without an observed sdf <= 0 in reach of its own
//                    samples cannot hold the second sample of a crossing and is dropped; otherwise every pixel of the block's
//                    box marches its lattice samples INSIDE the block from LDS (8 ds_reads per sample, no hash probe, no global
//                    load).  The sample BEFORE the first in-block one lies one step upstream -- inside the tile unless the step
//                    exceeds a voxel along an axis (then, rarely, through the hash; measured with the 9^3 tile of the first
//                    version: two hash-path samples per ray were 60 % of the kernel).  The first crossing goes to the depth
//                    image with atomicMin on the float's bits.  Pixels whose current depth already lies in front of the block
//                    skip it.
//   R2 k_rc_finish   per pixel: bits -> depth; normals / colours zeroed.
//   R3 k_rc_shade    (when normals or colours are asked for) block-major again, over the blocks that hold hit points: the same LDS tile
//                    serves the six gradient samples of every hit pixel; colours are gathered through the entry's neighbour slots.
// The result does not depend on the order blocks are processed in (tests compare it with the CPU restatement bit for bit).
#include "volume_core.hpp"

namespace {

constexpr unsigned kRcCountWords = 16 + 16 * 64;
constexpr unsigned kNoHit = 0x7f800000u; // +inf: the depth image while blocks are marched (positive floats order like their bits)
constexpr int kTileEdge = 11;            // the block with one voxel in front of it and two behind it on every axis: voxels -1 .. 9
constexpr int kTileVox = kTileEdge * kTileEdge * kTileEdge;
// The sample domain of block k -- the points whose base voxel lies in it -- is [(8k + 0.5) res, (8k + 8.5) res) per axis; the tests that
// bound it (k_rc_visible's pixel box, k_rc_march's slab test) add 0.02 voxels for the rounding of their own arithmetic.  Membership of a
// sample is decided exactly, from its base voxel.
constexpr float kDomLo = 0.48f, kDomExt = 8.04f;

// a visible block: its id, pixel box (u0, v0) + w x h, and the pool slots of its 3 x 3 x 3 neighbourhood (nb[13] = itself; < 0: absent),
// filled in by k_rc_neighbours so that the march starts loading voxels one hop after it has read its entry
struct __attribute__((aligned(16))) RcBlock { int kx, ky, kz, u0, v0, w, h, pad; int nb[28]; };
static_assert(sizeof(RcBlock) == 144, "36 dwords: nine 16-byte loads");

struct RcView {
    float P[12];   // camera -> world, rows 0..2
    float Pi[12];  // world -> camera, rows 0..2
    float fx, fy, cx, cy;
    int width, height;
    float res, inv_res, near_d, far_d;
    int kmax;      // last lattice index (t_kmax <= far); -1: no lattice point
};

// ---- the definition's arithmetic, shared by the LDS march and the global-memory gathers (same operations, same order) ----------
__device__ __forceinline__ float rc_t(const RcView& W, int k) { return W.near_d + (float)k * W.res; }

struct RcCell { int i0, i1, i2; float f0, f1, f2; };
__device__ __forceinline__ RcCell rc_cell(const RcView& W, float x, float y, float z) {
    const float g0 = x * W.inv_res - 0.5f, g1 = y * W.inv_res - 0.5f, g2 = z * W.inv_res - 0.5f;
    const float b0 = floorf(g0), b1 = floorf(g1), b2 = floorf(g2);
    return RcCell{(int)b0, (int)b1, (int)b2, g0 - b0, g1 - b1, g2 - b2};
}
__device__ __forceinline__ float rc_weight(const RcCell& c, int k) {
    const float wx = (k & 1) ? c.f0 : 1.0f - c.f0, wy = (k & 2) ? c.f1 : 1.0f - c.f1, wz = (k & 4) ? c.f2 : 1.0f - c.f2;
    return (wx * wy) * wz;
}

struct BlockCache { int cx, cy, cz, idx; };

// trilinear sample through the hash (the march's rare previous-sample lookups and k_rc_finish): false = not all 8 voxels observed
template <bool COL>
__device__ bool rc_sample_global(const VolView& V, const RcView& W, BlockCache& bc, float x, float y, float z, float* sdf, float* col) {
    const RcCell c = rc_cell(W, x, y, z);
    float acc = 0, a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int px = c.i0 + (k & 1), py = c.i1 + ((k >> 1) & 1), pz = c.i2 + ((k >> 2) & 1);
        const int bx = px >> 3, by = py >> 3, bz = pz >> 3;
        if (!(bx == bc.cx && by == bc.cy && bz == bc.cz)) { bc.cx = bx; bc.cy = by; bc.cz = bz; bc.idx = table_find(V, bx, by, bz); }
        if (bc.idx < 0) return false;
        const float* t = V.pool + (size_t)bc.idx * kBlockFloats + ((px - bx * 8) + (py - by * 8) * 8 + (pz - bz * 8) * 64);
        if (!(t[kVox] > 0)) return false;
        const float w = rc_weight(c, k);
        acc += w * t[0];
        if (COL) { a0 += w * t[2 * kVox]; a1 += w * t[3 * kVox]; a2 += w * t[4 * kVox]; } // colour planes only at the hit
    }
    *sdf = acc;
    if (COL) { col[0] = a0; col[1] = a1; col[2] = a2; }
    return true;
}

// ---- R0 ------------------------------------------------------------------------------------------------------------------------
// The sample domain of block (kx, ky, kz) -- the points whose base voxel lies in it -- is the block shifted by half a voxel:
// [(8k + 0.5) res, (8k + 8.5) res) per axis; one voxel of margin on every side makes the test conservative (R1 decides
// membership per sample, exactly).  Visible = its camera-frame z range meets [near, far] and its projected box meets the image.
__global__ __launch_bounds__(256) void k_rc_visible(VolView V, RcView W, RcBlock* __restrict__ list, unsigned* __restrict__ n_vis,
                                                    unsigned* __restrict__ depth_bits, unsigned npx) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x, gsz = gridDim.x * 256u;
    for (unsigned i = gid; i < npx; i += gsz) depth_bits[i] = kNoHit;
    unsigned nb = *V.n_blocks;
    if (nb > V.max_blocks) nb = V.max_blocks;
    const int lane = threadIdx.x & 63;
    for (unsigned b0 = blockIdx.x * 256u; b0 < nb; b0 += gsz) { // (uniform per wave: the ballot below needs every lane)
        const unsigned b = b0 + threadIdx.x;
        bool vis = false;
        RcBlock e{};
        if (b < nb) {
            const int kx = V.keys[3 * b], ky = V.keys[3 * b + 1], kz = V.keys[3 * b + 2];
            const float lo0 = ((float)(8 * kx) + kDomLo) * W.res, lo1 = ((float)(8 * ky) + kDomLo) * W.res, lo2 = ((float)(8 * kz) + kDomLo) * W.res;
            const float ext = kDomExt * W.res;
            float zmin = FLT_MAX, zmax = -FLT_MAX, umin = FLT_MAX, umax = -FLT_MAX, vmin = FLT_MAX, vmax = -FLT_MAX;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x = lo0 + ((c & 1) ? ext : 0.0f), y = lo1 + ((c & 2) ? ext : 0.0f), z = lo2 + ((c & 4) ? ext : 0.0f);
                const float q0 = W.Pi[0] * x + W.Pi[1] * y + W.Pi[2] * z + W.Pi[3], q1 = W.Pi[4] * x + W.Pi[5] * y + W.Pi[6] * z + W.Pi[7],
                            q2 = W.Pi[8] * x + W.Pi[9] * y + W.Pi[10] * z + W.Pi[11];
                zmin = fminf(zmin, q2); zmax = fmaxf(zmax, q2);
                const float iz = 1.0f / fmaxf(q2, 1e-6f);
                const float u = W.fx * q0 * iz + W.cx, v = W.fy * q1 * iz + W.cy;
                umin = fminf(umin, u); umax = fmaxf(umax, u); vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            }
            if (zmax >= W.near_d && zmin <= W.far_d) {
                int u0 = 0, v0 = 0, u1 = W.width - 1, v1 = W.height - 1;
                if (zmin > 1e-4f) { // every corner in front of the camera: the box of the projections bounds the block's pixels
                    // (a pixel's ray meets the domain only if the pixel lies inside the box of the corner projections; 0.02 pixels cover their rounding)
                    u0 = max(u0, (int)fminf(fmaxf(ceilf(umin - 0.02f), -1.0f), 3.0e7f)); u1 = min(u1, (int)fmaxf(fminf(floorf(umax + 0.02f), 3.0e7f), -2.0f));
                    v0 = max(v0, (int)fminf(fmaxf(ceilf(vmin - 0.02f), -1.0f), 3.0e7f)); v1 = min(v1, (int)fmaxf(fminf(floorf(vmax + 0.02f), 3.0e7f), -2.0f));
                }
                if (u0 <= u1 && v0 <= v1) {
                    vis = true; e.kx = kx; e.ky = ky; e.kz = kz; e.u0 = u0; e.v0 = v0; e.w = u1 - u0 + 1; e.h = v1 - v0 + 1; e.nb[13] = (int)b;
                    // does every ray through the box advance at most one voxel per lattice step along every axis (|d_a| <= 1)?  d is linear in the pixel: corners decide
                    float dmax = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float dcx = ((float)((c & 1) ? u1 : u0) - W.cx) / W.fx, dcy = ((float)((c & 2) ? v1 : v0) - W.cy) / W.fy;
#pragma unroll
                        for (int a = 0; a < 3; ++a) dmax = fmaxf(dmax, fabsf((W.P[4 * a] * dcx + W.P[4 * a + 1] * dcy) + W.P[4 * a + 2]));
                    }
                    e.pad = dmax <= 0.9999f ? 1 : 0;
                    // bits 8 .. 13: which 64th of [near, far] the block's nearest corner lies in (k_rc_order takes the list front to back)
                    e.pad |= (int)fminf(fmaxf((fmaxf(zmin, W.near_d) - W.near_d) / fmaxf(W.far_d - W.near_d, 1e-6f) * 64.0f, 0.0f), 63.0f) << 8;
                }
            }
        }
        const unsigned long long m = __ballot(vis);
        if (m) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(n_vis, (unsigned)__popcll(m));
            base = __shfl(base, 0, 64);
            if (vis) list[base + (unsigned)__popcll(m & ((1ULL << lane) - 1ULL))] = e;
        }
    }
}

// ---- R0b: one lane per (visible block, neighbour): 26 hash lookups per block, all of them in flight at once ------------------------------------
// The 32 lanes of an entry also look at what is KNOWN about the 27 blocks (summary: has an observed sdf <= 0 / > 0 among its own voxels; stamp =
// the volume's content generation; an absent block has neither; written by earlier views' marches from their tiles and by k_integrate for every
// block a batch of the exact update changes, which is why such a batch does not advance the generation) and mark the entry (pad bit 1)
// when the march would only load the tile to find that no crossing can end in the block:
//   * a crossing ends at an in-block sample <= 0, which needs an observed sdf <= 0 among tile voxels 0 .. 8 = the block and its 7 upper neighbours;
//   * its first sample is > 0 and lies in the tile when the rays step <= 1 voxel per axis (pad bit 0), which needs an observed sdf > 0 among the 27.
// The summaries are supersets of what the tile tests see (whole blocks instead of their border layers), so a marked entry is one the march
// would have dropped anyway: results cannot change, only the work.
__global__ __launch_bounds__(256) void k_rc_neighbours(VolView V, RcBlock* __restrict__ list, const unsigned* __restrict__ n_vis, const unsigned* __restrict__ summary,
                                                       unsigned stamp) {
    const unsigned n = *n_vis, total = n * 32u;
    for (unsigned i0 = blockIdx.x * 256u; i0 < total; i0 += gridDim.x * 256u) { // (uniform per wave: ballots below)
        const unsigned i = i0 + threadIdx.x;
        const unsigned e = i >> 5, j = i & 31u;
        bool known = true, neg = false, pos = false; // (lanes 27 .. 31 and lanes beyond the list: neutral)
        RcBlock* B = nullptr;
        if (i < total && j < 27u) {
            B = list + e;
            const int slot = j == 13u ? B->nb[13] : table_find(V, B->kx + (int)(j % 3u) - 1, B->ky + (int)((j / 3u) % 3u) - 1, B->kz + (int)(j / 9u) - 1);
            if (j != 13u) B->nb[j] = slot;
            if (slot >= 0) {
                const unsigned sm = summary[slot];
                known = (sm >> 2) == stamp;
                neg = (sm & 1u) != 0u; pos = (sm & 2u) != 0u;
            }
        }
        const int half = (threadIdx.x & 32) ? 32 : 0;
        const unsigned upper = (1u << 13) | (1u << 14) | (1u << 16) | (1u << 17) | (1u << 22) | (1u << 23) | (1u << 25) | (1u << 26); // offsets (0|1, 0|1, 0|1)
        const unsigned m_unknown = (unsigned)(__ballot(!known) >> half), m_neg = (unsigned)(__ballot(known && neg) >> half), m_pos = (unsigned)(__ballot(known && pos) >> half);
        if (B && j == 13u) {
            const bool no_end = !(m_unknown & upper) && !(m_neg & upper);
            const bool no_start = (B->pad & 1) && !m_unknown && !m_pos;
            if (no_end || no_start) B->pad |= 2;
        }
    }
}

// ---- R0c: front to back ---------------------------------------------------------------------------------------------------------
// A pixel skips a block when a hit in front of it is already recorded (k_rc_march), so the order blocks are taken in decides how much is marched behind the
// surface: every XCD's contiguous eighth of the list (rc_span: the L2 locality of the tile re-reads) is counting-sorted by the depth bucket k_rc_visible left in
// the entry (64 buckets over [near, far]).  One workgroup per eighth; the order inside a bucket is whatever the atomics give -- the result does not depend on it.
#ifndef RC_ORDER
#define RC_ORDER 1
#endif
__global__ __launch_bounds__(1024) void k_rc_order(const RcBlock* __restrict__ list, const unsigned* __restrict__ n_vis, unsigned* __restrict__ order) {
    __shared__ unsigned s_hist[64], s_cur[64];
    const unsigned n = *n_vis, per = (n + 7u) / 8u, lo = blockIdx.x * per, end = lo + per < n ? lo + per : n;
    if (threadIdx.x < 64) s_hist[threadIdx.x] = 0u;
    __syncthreads();
    for (unsigned e = lo + threadIdx.x; e < end; e += 1024u) atomicAdd(&s_hist[((unsigned)list[e].pad >> 8) & 63u], 1u);
    __syncthreads();
    if (threadIdx.x < 64) { // exclusive scan of 64 counters by one wave
        unsigned v = s_hist[threadIdx.x], x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned y = __shfl_up(x, d, 64); if ((int)threadIdx.x >= d) x += y; }
        s_cur[threadIdx.x] = lo + x - v;
    }
    __syncthreads();
    for (unsigned e = lo + threadIdx.x; e < end; e += 1024u) order[atomicAdd(&s_cur[((unsigned)list[e].pad >> 8) & 63u], 1u)] = e;
}

// ---- R1 ------------------------------------------------------------------------------------------------------------------------
#ifndef RC_MIN_WAVES
#define RC_MIN_WAVES 6 // waves per SIMD the march and shading kernels are compiled for: at 8 (64 VGPRs) the shading pass spills and takes 58 instead of 38 us
#endif
#ifndef RC_WG
#define RC_WG 128      // threads per workgroup = per visible block in flight (profiles/r05_ab_raycast.txt: 64 / 128 / 256)
#endif
constexpr int kRcWg = RC_WG;
#ifndef RC_STAGE_GROUP
#define RC_STAGE_GROUP 11 // tile cells per thread whose loads are in flight together (all of them: one round trip)
#endif
#ifdef RC_STATS // development aid (make EXTRA=-DRC_STATS): what the last views marched; printed by op_volume_raycast
__device__ unsigned long long g_rc_stats[8]; // visible blocks, blocks marched, pixels in boxes, pixels past the slab + depth tests, samples in block, hash lookups of a previous sample, crossings
#define RC_COUNT(K, N) atomicAdd(&g_rc_stats[K], (unsigned long long)(N))
#else
#define RC_COUNT(K, N) do { } while (0)
#endif
// trilinear sdf from the LDS tile at a cell whose local base voxel (l0, l1, l2) lies in [-1, 8]^3 (corners in [-1, 9]); NaN = not valid
__device__ __forceinline__ float rc_tile_sample(const float* __restrict__ tile, const RcCell& c, int l0, int l1, int l2) {
    const float* tl = tile + ((l0 + 1) + kTileEdge * (l1 + 1) + kTileEdge * kTileEdge * (l2 + 1));
    float acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += rc_weight(c, j) * tl[(j & 1) + kTileEdge * ((j >> 1) & 1) + kTileEdge * kTileEdge * (j >> 2)];
    return acc;
}

// ---- the LDS tile of a block, shared by the march and the shading pass --------------------------------------------------------------------------
// Which voxel of which neighbour a thread's tile cells come from does not depend on the block: a table, copied into LDS once per workgroup
// (bits 0-4 neighbour slot, 5-13 voxel id, 14 "an in-block sample can touch it" = tile coordinates 0 .. 8, 15 the cell exists).
constexpr int kStageIter = (kTileVox + kRcWg - 1) / kRcWg;
struct RcCellLut {
    unsigned short v[kStageIter * kRcWg];
    constexpr RcCellLut() : v{} {
        for (int a = 0; a < kStageIter * kRcWg; ++a) {
            const int x = a % kTileEdge - 1, y = (a / kTileEdge) % kTileEdge - 1, z = a / (kTileEdge * kTileEdge) - 1;
            const unsigned slot = (unsigned)(((x + 8) >> 3) + 3 * ((y + 8) >> 3) + 9 * ((z + 8) >> 3)), vid = (unsigned)((x & 7) + (y & 7) * 8 + (z & 7) * 64);
            const unsigned reach = (x >= 0 && y >= 0 && z >= 0 && x <= 8 && y <= 8 && z <= 8) ? 1u : 0u;
            v[a] = (unsigned short)(a < kTileVox ? (slot | (vid << 5) | (reach << 14) | (1u << 15)) : 0u);
        }
    }
};
__device__ const RcCellLut g_rc_cells{}; // (evaluated by the compiler)
__device__ __forceinline__ void rc_tile_map(unsigned short* s_cell, int tid) {
#pragma unroll
    for (int i = 0; i < kStageIter; ++i) s_cell[tid + kRcWg * i] = g_rc_cells.v[tid + kRcWg * i];
}
// Voxels -1 .. 9 of the block's frame on every axis (own 512 + the shell of its 26 neighbours) -> s_sdf: the observed sdf, or NaN; RC_STAGE_GROUP
// cells per thread in flight at a time.  Returns bit 0: an observed sdf <= 0 among the voxels an IN-BLOCK sample can touch, bit 1: an observed
// sdf > 0 anywhere in the tile, bits 2 / 3: the same two among the block's OWN 512 voxels (this thread's cells).
// PLAIN: every voxel of the volume was written by the fusion kernel alone since create / clear (op_volume::plain): a voxel is then either the default
// {sdf 999, weight 0} or a running mean of in-band observations (|sdf| < truncation < 999, weight >= 1), so "observed" is `sdf != 999` and the weight plane
// need not be read at all -- half the loads of the tile.  Any other writer (upload, merge, resampling, file) clears the flag and both planes are read.
template <bool PLAIN>
__device__ __forceinline__ unsigned rc_tile_load(const VolView& V, const int* __restrict__ s_nb, const unsigned short* __restrict__ s_cell, float* __restrict__ s_sdf, int tid) {
    bool neg = false, pos = false, own_neg = false, own_pos = false;
#pragma unroll
    for (int g = 0; g < kStageIter; g += RC_STAGE_GROUP) {
        float sd[RC_STAGE_GROUP], wt[RC_STAGE_GROUP];
        unsigned cl[RC_STAGE_GROUP];
#pragma unroll
        for (int i = 0; i < RC_STAGE_GROUP; ++i) {
            sd[i] = PLAIN ? 999.0f : 0.0f; wt[i] = 0.0f;
            cl[i] = g + i < kStageIter ? (unsigned)s_cell[tid + (g + i) * kRcWg] : 0u;
            if (cl[i] >> 15) {
                const int nbk = s_nb[cl[i] & 31u];
                if (nbk >= 0) {
                    const float* t = V.pool + (size_t)nbk * kBlockFloats + ((cl[i] >> 5) & 511u);
                    sd[i] = t[0];
                    if (!PLAIN) wt[i] = t[kVox];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RC_STAGE_GROUP; ++i)
            if (cl[i] >> 15) {
                const bool ok = PLAIN ? sd[i] != 999.0f : wt[i] > 0;
                const bool own = (cl[i] & 31u) == 13u;
                neg |= ok && sd[i] <= 0 && ((cl[i] >> 14) & 1u);
                pos |= ok && sd[i] > 0;
                own_neg |= own && ok && sd[i] <= 0;
                own_pos |= own && ok && sd[i] > 0;
                s_sdf[tid + kRcWg * (g + i)] = ok ? sd[i] : __builtin_nanf("");
            }
    }
    return (neg ? 1u : 0u) | (pos ? 2u : 0u) | (own_neg ? 4u : 0u) | (own_pos ? 8u : 0u);
}
// The visible list is in pool order -- runs of spatially adjacent blocks -- and a block's voxels are read by up to 27 workgroups (its own and
// its neighbours' shells): XCD x (workgroup b runs on XCD b % 8) takes the x-th contiguous eighth of the list, so that those re-reads
// meet in ONE L2 instead of eight (measured with the list dealt round-robin: L2 hit rate 23 %, 2.8 x the tile bytes fetched over the fabric).
struct RcSpan { unsigned first, end, step; };
__device__ __forceinline__ RcSpan rc_span(unsigned n) {
    const unsigned xcd = blockIdx.x & 7u, per = (n + 7u) / 8u, lo = xcd * per;
    return RcSpan{lo + (blockIdx.x >> 3), lo + per < n ? lo + per : n, gridDim.x >> 3};
}

// hit_blocks: one byte per pool slot, set for the block whose sample domain holds the hit point of a crossing recorded here (k_rc_shade works
// through exactly those blocks and clears the bytes again); nullptr when neither normals nor colours are asked for.
template <bool PLAIN>
__global__ __launch_bounds__(RC_WG, RC_MIN_WAVES) void k_rc_march(VolView V, RcView W, const RcBlock* __restrict__ list, const unsigned* __restrict__ n_vis,
                                                                  unsigned* __restrict__ depth_bits, unsigned char* __restrict__ hit_blocks, unsigned* __restrict__ summary,
                                                                  unsigned stamp, unsigned* __restrict__ counters, const unsigned* __restrict__ order) {
    __shared__ float s_sdf[kTileVox + 5];
    __shared__ int s_ent[36];
    __shared__ unsigned s_flags;
    __shared__ unsigned short s_cell[kStageIter * kRcWg];
    const int tid = threadIdx.x;
    const float o0 = W.P[3], o1 = W.P[7], o2 = W.P[11];
    rc_tile_map(s_cell, tid);
    const RcSpan span = rc_span(*n_vis);
    unsigned n_dropped = 0, n_loaded = 0, n_marched = 0; // (uniform over the workgroup; op_volume_raycast_stats)
    for (unsigned i = span.first; i < span.end; i += span.step) {
        const unsigned e = RC_ORDER ? order[i] : i; // front to back within the XCD's eighth (k_rc_order)
        if (list[e].pad & 2) { ++n_dropped; continue; } // k_rc_neighbours could tell from the summaries of earlier views that no crossing ends here (uniform: no LDS touched yet)
        __syncthreads(); // the previous block's readers are done with the tile and the entry
        if (tid < 36) s_ent[tid] = reinterpret_cast<const int*>(list + e)[tid];
        if (tid == 64) s_flags = 0u;
        __syncthreads();
        const int kx = s_ent[0], ky = s_ent[1], kz = s_ent[2];
        const int* s_nb = s_ent + 8;
        {
            const unsigned mine = rc_tile_load<PLAIN>(V, s_nb, s_cell, s_sdf, tid);
            const unsigned f = (__ballot(mine & 1u) ? 1u : 0u) | (__ballot(mine & 2u) ? 2u : 0u) | (__ballot(mine & 4u) ? 4u : 0u) | (__ballot(mine & 8u) ? 8u : 0u);
            if ((tid & 63) == 0 && f) atomicOr(&s_flags, f);
        }
        if (tid == 0) RC_COUNT(0, 1);
        ++n_loaded;
        __syncthreads(); // the tile and the flags are complete
        if (tid == 0) summary[s_nb[13]] = (stamp << 2) | ((s_flags >> 2) & 3u); // what this view learnt about the block's own voxels, for the next views of the unchanged volume
        // No observed sdf <= 0 in reach of this block's samples: no crossing can END here.  No observed sdf > 0 in the whole tile: no sample BEFORE
        // one of this block's can be positive either, provided it lies in the tile (s_ent[7]: every ray through the box steps <= 1 voxel per axis)
        const unsigned flags = s_flags;
        if (!(flags & 1u) || (!(flags & 2u) && (s_ent[7] & 1))) continue;
        ++n_marched;
#ifdef RC_STAGE_ONLY // (timing experiment only)
        const int npix = s_sdf[tid] == 12345.0f ? s_ent[5] * s_ent[6] : 0;
#else
        const int npix = s_ent[5] * s_ent[6];
#endif
        if (tid == 0) { RC_COUNT(1, 1); RC_COUNT(2, npix); }
        const int bx8 = 8 * kx, by8 = 8 * ky, bz8 = 8 * kz, bu0 = s_ent[3], bv0 = s_ent[4], bw = s_ent[5];
        const float lo0 = ((float)bx8 + kDomLo) * W.res, lo1 = ((float)by8 + kDomLo) * W.res, lo2 = ((float)bz8 + kDomLo) * W.res, ext = kDomExt * W.res;
        for (int q = tid; q < npix; q += kRcWg) {
            const int py = bv0 + q / bw, px = bu0 + q % bw;
            const size_t pix = (size_t)py * W.width + px;
            const float dcx = ((float)px - W.cx) / W.fx, dcy = ((float)py - W.cy) / W.fy;
            const float d0 = (W.P[0] * dcx + W.P[1] * dcy) + W.P[2], d1 = (W.P[4] * dcx + W.P[5] * dcy) + W.P[6], d2 = (W.P[8] * dcx + W.P[9] * dcy) + W.P[10];
            // slab test (conservative; NaN from 0 * inf drops out of fminf / fmaxf)
            float tmin = W.near_d - W.res, tmax = W.far_d + W.res;
            {
                const float i0 = 1.0f / d0, i1 = 1.0f / d1, i2 = 1.0f / d2;
                const float a0 = (lo0 - o0) * i0, b0 = (lo0 + ext - o0) * i0, a1 = (lo1 - o1) * i1, b1 = (lo1 + ext - o1) * i1, a2 = (lo2 - o2) * i2, b2 = (lo2 + ext - o2) * i2;
                tmin = fmaxf(tmin, fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fminf(a2, b2)));
                tmax = fminf(tmax, fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fmaxf(a2, b2)));
            }
            if (!(tmin <= tmax)) continue;
            int k0 = (int)floorf((tmin - W.near_d) * W.inv_res) - 1, k1 = (int)ceilf((tmax - W.near_d) * W.inv_res) + 1;
            if (k0 < 0) k0 = 0;
            if (k1 > W.kmax) k1 = W.kmax;
            if (k0 > k1) continue;
            // a crossing that ends at k >= max(k0, 1) reports a depth >= t_(k-1): nothing to do behind an earlier hit
            if (__uint_as_float(depth_bits[pix]) < rc_t(W, (k0 > 1 ? k0 : 1) - 1)) continue;
            RC_COUNT(3, 1);
            bool prev_in = false;
            float s_prev = 0, t_prev = 0;
            for (int k = k0; k <= k1; ++k) {
                const float t = rc_t(W, k);
                const RcCell c = rc_cell(W, o0 + t * d0, o1 + t * d1, o2 + t * d2);
                const int l0 = c.i0 - bx8, l1 = c.i1 - by8, l2 = c.i2 - bz8;
                if ((unsigned)l0 >= 8u || (unsigned)l1 >= 8u || (unsigned)l2 >= 8u) { prev_in = false; continue; } // another block's sample
                RC_COUNT(4, 1);
                const float acc = rc_tile_sample(s_sdf, c, l0, l1, l2);
                if (acc <= 0.0f) { // (false for NaN = not valid)
                    float sp = s_prev, tp = t_prev;
                    if (!prev_in) { // the previous lattice point belongs to another block (or there is none): one step upstream, normally still inside the tile
                        sp = __builtin_nanf("");
#ifndef RC_NO_PREV // (timing experiment only: wrong results)
                        if (k > 0) {
                            tp = rc_t(W, k - 1);
                            const float x = o0 + tp * d0, y = o1 + tp * d1, z = o2 + tp * d2;
                            const RcCell cp = rc_cell(W, x, y, z);
                            const int m0 = cp.i0 - bx8, m1 = cp.i1 - by8, m2 = cp.i2 - bz8;
                            if ((unsigned)(m0 + 1) < 10u && (unsigned)(m1 + 1) < 10u && (unsigned)(m2 + 1) < 10u) sp = rc_tile_sample(s_sdf, cp, m0, m1, m2);
                            else { // a step of more than one voxel along an axis (|d_a| > 1): through the hash
                                RC_COUNT(5, 1);
                                BlockCache bc{INT_MIN, INT_MIN, INT_MIN, -1};
                                float sv;
                                if (rc_sample_global<false>(V, W, bc, x, y, z, &sv, nullptr)) sp = sv;
                            }
                        }
#endif
                    }
                    if (sp > 0.0f) {
                        const float depth = tp + (t - tp) * (sp / (sp - acc));
                        if (depth > 0.0f) {
                            RC_COUNT(6, 1);
                            atomicMin(depth_bits + pix, __float_as_uint(depth));
                            if (hit_blocks) { // the hit point lies between the two samples: in this block, or in the neighbour the ray came from
                                const RcCell ch = rc_cell(W, o0 + depth * d0, o1 + depth * d1, o2 + depth * d2);
                                const int hb = s_nb[(((ch.i0 - bx8) + 8) >> 3) + 3 * (((ch.i1 - by8) + 8) >> 3) + 9 * (((ch.i2 - bz8) + 8) >> 3)];
                                if (hb >= 0) hit_blocks[hb] = 1; // (absent: the hit point has no observed voxels around it -- no normal, no colour)
                            }
                            break;
                        }
                    }
                }
                prev_in = true; s_prev = acc; t_prev = t;
            }
        }
    }
    if (tid == 0) { // one shard of the call's counters per 64th of the workgroups (atomics on one address serialise)
        unsigned* c = counters + 16u * (blockIdx.x & 63u);
        if (n_dropped) atomicAdd(c, n_dropped);
        if (n_loaded) atomicAdd(c + 1, n_loaded);
        if (n_marched) atomicAdd(c + 2, n_marched);
    }
}

// ---- R2 ------------------------------------------------------------------------------------------------------------------------
// bits -> depth; normals and colours start as zero (k_rc_shade fills in the pixels that were hit)
__global__ __launch_bounds__(256) void k_rc_finish(RcView W, float* __restrict__ depth_io, float* __restrict__ normals_out, float* __restrict__ colors_out) {
    const size_t npx = (size_t)W.width * W.height;
    for (size_t pix = blockIdx.x * 256u + threadIdx.x; pix < npx; pix += (size_t)gridDim.x * 256u) {
        const unsigned bits = reinterpret_cast<const unsigned*>(depth_io)[pix];
        depth_io[pix] = bits == kNoHit ? 0.0f : __uint_as_float(bits);
        if (normals_out) { normals_out[3 * pix] = 0; normals_out[3 * pix + 1] = 0; normals_out[3 * pix + 2] = 0; }
        if (colors_out) { colors_out[3 * pix] = 0; colors_out[3 * pix + 1] = 0; colors_out[3 * pix + 2] = 0; }
    }
}

// ---- R3 ------------------------------------------------------------------------------------------------------------------------
// Normals and colours, block-major like the march: a workgroup takes a visible block that holds hit points (k_rc_march's byte), loads the same
// 11^3 tile, and every pixel of the block's box whose hit point has its base voxel in the block gets
//   normal = normalised (s(p + h e_a) - s(p - h e_a))_a, h = res / 2 -- six trilinear samples, all inside the tile (base voxels -1 .. 8),
//   colour = trilinear colour at p -- 24 gathers from the colour planes through the entry's neighbour slots (no hash probe),
// both zero unless every sample is valid (the definition's rule).  Every hit point lies in the sample domain of exactly one block.
#ifndef RC_SHADE_MIN_WAVES
#define RC_SHADE_MIN_WAVES RC_MIN_WAVES
#endif
template <bool PLAIN>
__global__ __launch_bounds__(RC_WG, RC_SHADE_MIN_WAVES) void k_rc_shade(VolView V, RcView W, const RcBlock* __restrict__ list, const unsigned* __restrict__ n_vis,
                                                                  unsigned char* __restrict__ hit_blocks, const float* __restrict__ depth, float* __restrict__ normals_out,
                                                                  float* __restrict__ colors_out) {
    __shared__ float s_sdf[kTileVox + 5];
    __shared__ int s_ent[36];
    __shared__ unsigned short s_cell[kStageIter * kRcWg];
    const int tid = threadIdx.x;
    const float o0 = W.P[3], o1 = W.P[7], o2 = W.P[11];
    rc_tile_map(s_cell, tid);
    const RcSpan span = rc_span(*n_vis);
    for (unsigned e = span.first; e < span.end; e += span.step) {
        const int self = list[e].nb[13];
        if (!hit_blocks[self]) continue; // (uniform: one byte per block)
        __syncthreads();
        if (tid < 36) s_ent[tid] = reinterpret_cast<const int*>(list + e)[tid];
        if (tid == 64) hit_blocks[self] = 0; // consumed (every block is visited by exactly one workgroup)
        __syncthreads();
        const int* s_nb = s_ent + 8;
        (void)rc_tile_load<PLAIN>(V, s_nb, s_cell, s_sdf, tid);
        __syncthreads();
        const int bx8 = 8 * s_ent[0], by8 = 8 * s_ent[1], bz8 = 8 * s_ent[2], bu0 = s_ent[3], bv0 = s_ent[4], bw = s_ent[5], npix = s_ent[5] * s_ent[6];
        const float h = 0.5f * W.res;
        for (int q = tid; q < npix; q += kRcWg) {
            const int py = bv0 + q / bw, px = bu0 + q % bw;
            const size_t pix = (size_t)py * W.width + px;
            const float hit = depth[pix];
            if (!(hit > 0)) continue;
            const float dcx = ((float)px - W.cx) / W.fx, dcy = ((float)py - W.cy) / W.fy;
            const float d0 = (W.P[0] * dcx + W.P[1] * dcy) + W.P[2], d1 = (W.P[4] * dcx + W.P[5] * dcy) + W.P[6], d2 = (W.P[8] * dcx + W.P[9] * dcy) + W.P[10];
            const float x = o0 + hit * d0, y = o1 + hit * d1, z = o2 + hit * d2;
            const RcCell c = rc_cell(W, x, y, z);
            const int l0 = c.i0 - bx8, l1 = c.i1 - by8, l2 = c.i2 - bz8;
            if ((unsigned)l0 >= 8u || (unsigned)l1 >= 8u || (unsigned)l2 >= 8u) continue; // another block's hit point
            if (colors_out) {
                float col[3] = {0, 0, 0};
                if (rc_tile_sample(s_sdf, c, l0, l1, l2) == rc_tile_sample(s_sdf, c, l0, l1, l2)) { // all 8 voxels observed (not NaN)
                    float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int vx = l0 + (j & 1), vy = l1 + ((j >> 1) & 1), vz = l2 + (j >> 2);
                        const float* t = V.pool + (size_t)s_nb[(vx >> 3) + 3 * (vy >> 3) + 9 * (vz >> 3) + 13] * kBlockFloats + ((vx & 7) + (vy & 7) * 8 + (vz & 7) * 64);
                        const float w = rc_weight(c, j);
                        a0 += w * t[2 * kVox]; a1 += w * t[3 * kVox]; a2 += w * t[4 * kVox];
                    }
                    col[0] = a0; col[1] = a1; col[2] = a2;
                }
                colors_out[3 * pix] = col[0]; colors_out[3 * pix + 1] = col[1]; colors_out[3 * pix + 2] = col[2];
            }
            if (normals_out) {
                float n[3];
                bool ok = true;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const RcCell cp = rc_cell(W, x + (a == 0 ? h : 0.0f), y + (a == 1 ? h : 0.0f), z + (a == 2 ? h : 0.0f));
                    const RcCell cm = rc_cell(W, x - (a == 0 ? h : 0.0f), y - (a == 1 ? h : 0.0f), z - (a == 2 ? h : 0.0f));
                    const float sp = rc_tile_sample(s_sdf, cp, cp.i0 - bx8, cp.i1 - by8, cp.i2 - bz8), sm = rc_tile_sample(s_sdf, cm, cm.i0 - bx8, cm.i1 - by8, cm.i2 - bz8);
                    ok = ok && sp == sp && sm == sm;
                    n[a] = sp - sm;
                }
                const float l2n = sum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
                if (ok && l2n > 0) { const float l = sqrtf(l2n); n[0] /= l; n[1] /= l; n[2] /= l; } else { n[0] = n[1] = n[2] = 0; }
                normals_out[3 * pix] = n[0]; normals_out[3 * pix + 1] = n[1]; normals_out[3 * pix + 2] = n[2];
            }
        }
    }
}

#ifndef RC_NB_GRID
#define RC_NB_GRID 4096
#endif
#ifndef RC_MARCH_GRID
#define RC_MARCH_GRID 16384 // workgroups of k_rc_march, each taking every RC_MARCH_GRID-th visible block (4 x what is resident: evens out blocks of unequal cost)
#endif

} // namespace

extern "C" {

int op_volume_raycast(op_volume* v, const op_camera* cam, const float pose[16], float* depth_out, float* normals_out, float* colors_out, int mem) {
    OP_VOL(v);
    if (!pose || !depth_out) return fail(OP_ERR_INVALID, "null argument");
    const op_camera c = cam ? *cam : v->cam;
    OP_TRY(check_cam(&c));
    OP_TRY(vol_check(v));
    const size_t npx = (size_t)c.width * c.height;
    // the visible-block list (one entry per pool block at most) and the hit-point bytes (one per pool slot; zero between calls)
    if (v->rc_cap < v->max_blocks || !v->rc_list || !v->rc_hit || !v->rc_sum || !v->rc_order) {
        if (v->rc_list) op::cached_free(v->rc_list);
        if (v->rc_hit) op::cached_free(v->rc_hit);
        if (v->rc_sum) op::cached_free(v->rc_sum);
        if (v->rc_order) op::cached_free(v->rc_order);
        v->rc_list = nullptr; v->rc_hit = nullptr; v->rc_sum = nullptr; v->rc_order = nullptr; v->rc_cap = 0;
        OP_HIP(op::cached_malloc(&v->rc_list, sizeof(RcBlock) * (size_t)v->max_blocks));
        OP_HIP(op::cached_malloc((void**)&v->rc_order, sizeof(unsigned) * (size_t)v->max_blocks));
        OP_HIP(op::cached_malloc((void**)&v->rc_hit, (size_t)v->max_blocks));
        OP_HIP(hipMemsetAsync(v->rc_hit, 0, (size_t)v->max_blocks, v->stream));
        OP_HIP(op::cached_malloc((void**)&v->rc_sum, sizeof(unsigned) * (size_t)v->max_blocks));
        OP_HIP(hipMemsetAsync(v->rc_sum, 0, sizeof(unsigned) * (size_t)v->max_blocks, v->stream)); // stamp 0 = nothing known (content_gen starts at 1)
        v->rc_cap = v->max_blocks;
    }
    if (!v->rc_count) OP_HIP(op::cached_malloc((void**)&v->rc_count, sizeof(unsigned) * kRcCountWords)); // [0] the list's length, [16 + 16 s ...] shard s of the call's counters
    OP_HIP(hipMemsetAsync(v->rc_count, 0, sizeof(unsigned) * kRcCountWords, v->stream));
    float *d_depth = depth_out, *d_nrm = normals_out, *d_col = colors_out;
    // OP_MEM_HOST: device temporaries from the buffer cache; released on EVERY exit (a failed allocation or launch must not leave them in the cache's live set)
    auto release_tmp = [&] {
        if (mem != OP_MEM_HOST) return;
        if (d_depth) op::cached_free(d_depth);
        if (d_nrm) op::cached_free(d_nrm);
        if (d_col) op::cached_free(d_col);
        d_depth = d_nrm = d_col = nullptr;
    };
    if (mem == OP_MEM_HOST) {
        d_depth = d_nrm = d_col = nullptr;
        hipError_t ea = op::cached_malloc((void**)&d_depth, npx * 4);
        if (ea == hipSuccess && normals_out) ea = op::cached_malloc((void**)&d_nrm, npx * 12);
        if (ea == hipSuccess && colors_out) ea = op::cached_malloc((void**)&d_col, npx * 12);
        if (ea != hipSuccess) { release_tmp(); return fail(OP_ERR_HIP, "raycast: no device memory for the output images: %s", hipGetErrorString(ea)); }
    }
    RcView W;
    float inv[16];
    op_host::mat4_inverse(pose, inv);
    std::memcpy(W.P, pose, sizeof(W.P));
    std::memcpy(W.Pi, inv, sizeof(W.Pi));
    W.fx = c.fx; W.fy = c.fy; W.cx = c.cx; W.cy = c.cy; W.width = c.width; W.height = c.height;
    W.res = v->res; W.inv_res = 1.0f / v->res; W.near_d = v->near_d; W.far_d = v->far_d;
    // last lattice index: the largest k with near + (float)k * res <= far, evaluated like the kernels evaluate it
    W.kmax = -1;
    if (v->near_d <= v->far_d) {
        long long k = (long long)((v->far_d - v->near_d) / v->res) + 2;
        if (k > (1 << 24)) k = 1 << 24;
        while (k > 0 && !(v->near_d + (float)k * v->res <= v->far_d)) --k;
        W.kmax = (int)k;
    }
    const VolView V = v->view();
    const unsigned work = (unsigned)std::max<size_t>(npx, v->max_blocks);
    hipLaunchKernelGGL(k_rc_visible, dim3(std::min(2048u, (work + 255u) / 256u)), dim3(256), 0, v->stream, V, W, (RcBlock*)v->rc_list, v->rc_count,
                       reinterpret_cast<unsigned*>(d_depth), (unsigned)npx);
    const bool shade = d_nrm || d_col;
    // summaries are stamped with the low 30 bits of the content generation; when those wrap, nothing older may survive
    if ((v->content_gen >> 30) != v->rc_sum_epoch) {
        const hipError_t ew = hipMemsetAsync(v->rc_sum, 0, sizeof(unsigned) * (size_t)v->rc_cap, v->stream);
        if (ew != hipSuccess) { release_tmp(); return fail(OP_ERR_HIP, "raycast: wiping the block summaries failed: %s", hipGetErrorString(ew)); }
        v->rc_sum_epoch = v->content_gen >> 30;
    }
    const unsigned stamp = (unsigned)(v->content_gen & 0x3fffffffull);
    // Stamp 0 is also what the wipe leaves behind ("nothing known"): in the one generation per 2^30 whose low bits are 0 no stored summary may match -- that view
    // marches every visible block (summaries written with stamp 0, by this view or by k_integrate, match no later generation either).
    // OP_VOLUME_OPT_RAYCAST_PRUNE = 0: likewise, no stored summary ever matches (a stored stamp has 30 bits).
    const unsigned stamp_read = (v->rc_prune && stamp != 0u) ? stamp : 0x7fffffffu;
    hipLaunchKernelGGL(k_rc_neighbours, dim3(RC_NB_GRID), dim3(256), 0, v->stream, V, (RcBlock*)v->rc_list, (const unsigned*)v->rc_count, (const unsigned*)v->rc_sum, stamp_read);
    if (RC_ORDER) hipLaunchKernelGGL(k_rc_order, dim3(8), dim3(1024), 0, v->stream, (const RcBlock*)v->rc_list, (const unsigned*)v->rc_count, v->rc_order);
    const bool plain = v->plain && v->trunc < 900.0f; // (see rc_tile_load: the weight plane is not needed to tell observed voxels)
#define OP_RC_MARCH(P) hipLaunchKernelGGL(k_rc_march<P>, dim3(RC_MARCH_GRID), dim3(kRcWg), 0, v->stream, V, W, (const RcBlock*)v->rc_list, (const unsigned*)v->rc_count, \
                                          reinterpret_cast<unsigned*>(d_depth), shade ? v->rc_hit : nullptr, v->rc_sum, stamp, v->rc_count + 16, (const unsigned*)v->rc_order)
    if (plain) OP_RC_MARCH(true); else OP_RC_MARCH(false);
#undef OP_RC_MARCH
    hipLaunchKernelGGL(k_rc_finish, dim3((unsigned)std::min<size_t>(2048, (npx + 255) / 256)), dim3(256), 0, v->stream, W, d_depth, d_nrm, d_col);
#define OP_RC_SHADE(P) hipLaunchKernelGGL(k_rc_shade<P>, dim3(RC_MARCH_GRID), dim3(kRcWg), 0, v->stream, V, W, (const RcBlock*)v->rc_list, (const unsigned*)v->rc_count, v->rc_hit, \
                                          (const float*)d_depth, d_nrm, d_col)
    if (shade) { if (plain) OP_RC_SHADE(true); else OP_RC_SHADE(false); }
#undef OP_RC_SHADE
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
#ifdef RC_STATS
    {
        unsigned long long st[8] = {0}, zero[8] = {0};
        if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_rc_stats), sizeof(st)) == hipSuccess && hipMemcpyToSymbol(HIP_SYMBOL(g_rc_stats), zero, sizeof(zero)) == hipSuccess)
            fprintf(stderr, "rc stats: visible blocks %llu, marched %llu, box pixels %llu, pixels marched %llu, samples %llu, previous-sample lookups %llu, crossings %llu\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6]);
    }
#endif
    if (mem == OP_MEM_HOST) {
        if (e == hipSuccess) e = hipMemcpy(depth_out, d_depth, npx * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && normals_out) e = hipMemcpy(normals_out, d_nrm, npx * 12, hipMemcpyDeviceToHost);
        if (e == hipSuccess && colors_out) e = hipMemcpy(colors_out, d_col, npx * 12, hipMemcpyDeviceToHost);
    }
    release_tmp();
    if (e != hipSuccess) {
        // k_rc_shade clears the hit bytes it consumes; a launch that failed may have left some set, and the next shading call assumes they are zero between calls:
        // drop the raycast buffers, the next call re-creates (and zeroes) them
        (void)hipStreamSynchronize(v->stream);
        if (v->rc_list) op::cached_free(v->rc_list);
        if (v->rc_hit) op::cached_free(v->rc_hit);
        if (v->rc_sum) op::cached_free(v->rc_sum);
        if (v->rc_order) op::cached_free(v->rc_order);
        v->rc_list = nullptr; v->rc_hit = nullptr; v->rc_sum = nullptr; v->rc_order = nullptr; v->rc_cap = 0;
        return fail(OP_ERR_HIP, "raycast failed: %s", hipGetErrorString(e));
    }
    return OP_OK;
}

int op_volume_raycast_stats(op_volume* v, uint64_t* visible_blocks, uint64_t* dropped_unloaded, uint64_t* loaded_blocks, uint64_t* marched_blocks) {
    OP_VOL(v);
    uint64_t out[4] = {0, 0, 0, 0};
    if (v->rc_count) {
        std::vector<unsigned> h(kRcCountWords);
        OP_HIP(hipStreamSynchronize(v->stream));
        OP_HIP(hipMemcpy(h.data(), v->rc_count, sizeof(unsigned) * kRcCountWords, hipMemcpyDeviceToHost));
        out[0] = h[0];
        for (unsigned s = 0; s < 64; ++s) { out[1] += h[16 + 16 * s]; out[2] += h[16 + 16 * s + 1]; out[3] += h[16 + 16 * s + 2]; }
    }
    if (visible_blocks) *visible_blocks = out[0];
    if (dropped_unloaded) *dropped_unloaded = out[1];
    if (loaded_blocks) *loaded_blocks = out[2];
    if (marched_blocks) *marched_blocks = out[3];
    return OP_OK;
}

} // extern "C"
