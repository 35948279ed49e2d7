// select.hip -- the per-frame kernels in front of the voxel update (volume_core.hpp lists the translation units):
//   KA k_prepare_frames: CubeHandler::ComputeBounding (Integration/CubeHandler.cpp:116-145) + packing of the frame
//   KB k_select / k_select_vote + k_select_merge: CubeHandler::PrepareCubes + Integrator::GetSDF (CubeHandler.cpp:147-196, Integrator.cpp:8-35)
//   k_mark_cubes (op_volume_integrate_cubes), k_finish_select (PrepareCubes record mode)
// file:line citations are relative to /root/reference/src.
#include "volume_core.hpp"

namespace {

__global__ void k_finish_select(VolView V, State* st) {
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < (unsigned)(kMaxBatch * kAccSlots * 8); k += gridDim.x * blockDim.x) (&st->acc[0][0][0])[k] = 0u; // as KC does
    const unsigned n = st->n_list[0] < V.max_blocks ? st->n_list[0] : V.max_blocks; // a one-frame batch: list 0 only
    const unsigned nr = st->n_rec < V.max_blocks ? st->n_rec : V.max_blocks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) V.bmask[V.blist[i]] = (bmask_t)0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x) V.sel_list[i] = V.tvals[V.sel_list[i]];
}

// Integrator::IntegrateImage for a caller-chosen cube list (op_volume_integrate_cubes): takes KB's place in a one-frame
// batch -- every listed cube is found or allocated and put on the batch list with the frame's bit, no selection test.
__global__ void k_mark_cubes(VolView V, State* st, const int* __restrict__ keys, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (st->overflow & 3u)) return;
    const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
    if (!key_in_range(x, y, z)) { atomicOr(&st->overflow, 8u); return; }
    bool created;
    const int slot = table_claim(V, st, x, y, z, &created);
    if (slot < 0) return;
    if (atomicOr(&V.bmask[slot], (bmask_t)1) == (bmask_t)0) { // a key listed twice is fused once
        const unsigned pos = atomicAdd(&st->n_list[0], 1u);
        if (pos < V.max_blocks) V.blist[pos] = slot;
    }
}


// ---------------------------------------------------------------------------------------------
// KA: per-frame preparation = ComputeBounding (CubeHandler.cpp:116-145: back-project, transform,
// frustum test, min/max) + packing of the frame into one {depth, rgba} record per pixel so that
// the later gathers are single 8-byte loads + the smallest and largest valid depth of every 16 x 16
// pixel tile (what k_select's coarse test looks at).  grid = (ka_grid(W, H), n_frames); a workgroup
// owns a 64 x 16 pixel rectangle, a thread 4 consecutive pixels of one row.
// One bounding partial per workgroup (no atomics): [max x,y,z, min x,y,z, inside, pad].
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_prepare_frames(KaFwd B, int f0, CamParams C, BatchPtrs Q, uint2* __restrict__ pimg, float2* __restrict__ ptile,
                                                        float* __restrict__ partial, State* st, unsigned seq,
                                                        const unsigned* __restrict__ n_blocks, unsigned* __restrict__ hstat) {
    __shared__ float s_red[4][6];
    __shared__ unsigned s_cnt[4];
    __shared__ float s_tile[4][4][2];
    const int tid = threadIdx.x, f = f0 + (int)blockIdx.y; // frame of the batch
    if (blockIdx.x == 0 && f == 0 && tid == 0) {
        st->n_rec = 0; // new batch: empty lists
        for (int b = 0; b < kBands; ++b) st->n_list[b] = 0;
        st->cur_seq = seq;
        // Progress report for the host (host-mapped pinned memory, read without any synchronisation): this kernel starting
        // means every earlier batch has finished; unless the stream is poisoned by an overflow they all completed.  The host
        // uses it to retire its replay log / staging slots and to grow the pool BEFORE it runs full.
        if (hstat && (st->overflow & 3u) == 0u) {
            hstat[1] = *n_blocks;
            __threadfence_system();
            __hip_atomic_store(&hstat[0], seq - 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (blockIdx.x == 0 && f == 0 && tid < kKcShares) st->kc_next[tid * 16] = 0u;
    if (blockIdx.x == 0 && f == 0) { // the previous launch's k_integrate duration in shader cycles: its longest workgroup
        static_assert(kKcTSlots == 256, "one slot per thread");
        unsigned long long kc = st->kc_t[tid];
        st->kc_t[tid] = 0ull;
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(kc, o, 64); kc = x > kc ? x : kc; }
        __shared__ unsigned long long s_kc[4];
        if ((tid & 63) == 0) s_kc[tid >> 6] = kc;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) kc = s_kc[w] > kc ? s_kc[w] : kc;
            st->stat_kc_ticks += kc;
        }
    }
    const PoseFwd& P = B.f[blockIdx.y];
    const int npix = C.width * C.height;
    const void* dptr = Q.depth[f];
    const unsigned char* cptr = Q.rgb[f];
    uint2* out = pimg + (size_t)f * npix;
    const int wgx = (C.width + kKaW - 1) / kKaW;
    const int gy = (int)blockIdx.x / wgx, gx = (int)blockIdx.x - gy * wgx;
    const int row = gy * kKaH + (tid >> 4), col0 = gx * kKaW + (tid & 15) * 4;
    float mx0 = -FLT_MAX, mx1 = -FLT_MAX, mx2 = -FLT_MAX, mn0 = FLT_MAX, mn1 = FLT_MAX, mn2 = FLT_MAX;
    float tmin = __builtin_inff(), tmax = -__builtin_inff(); // valid depths of the thread's pixels
    unsigned inside = 0;
    if (row < C.height && col0 < C.width) {
        const int pix0 = row * C.width + col0;
        float zz[4];
        unsigned cc[4] = {0u, 0u, 0u, 0u};
        const int nv = C.width - col0 < 4 ? C.width - col0 : 4; // pixels of the row this thread has
        // Integrator.cpp:26-29 / PointCloud.cpp:83-86: float depth, or uint16 / depth_scale.  Aligned rows take one wide load per thread.
        const bool wide = nv == 4 && (C.width & 3) == 0 && ((size_t)dptr & 15u) == 0 && ((size_t)cptr & 3u) == 0;
        if (wide) {
            if (C.depth_u16) {
                const ushort4 d = *reinterpret_cast<const ushort4*>((const unsigned short*)dptr + pix0);
                zz[0] = (float)d.x / C.depth_scale; zz[1] = (float)d.y / C.depth_scale; zz[2] = (float)d.z / C.depth_scale; zz[3] = (float)d.w / C.depth_scale;
            } else {
                const float4 d = *reinterpret_cast<const float4*>((const float*)dptr + pix0);
                zz[0] = d.x; zz[1] = d.y; zz[2] = d.z; zz[3] = d.w;
            }
            if (cptr) {
                const unsigned* c3 = reinterpret_cast<const unsigned*>(cptr + 3 * (size_t)pix0); // 12 bytes = 4 pixels, 4-byte aligned
                const unsigned d0 = c3[0], d1 = c3[1], d2 = c3[2];
                cc[0] = d0 & 0xffffffu; cc[1] = (d0 >> 24) | ((d1 & 0xffffu) << 8); cc[2] = (d1 >> 16) | ((d2 & 0xffu) << 16); cc[3] = d2 >> 8;
            }
            uint4* o4 = reinterpret_cast<uint4*>(out + pix0); // byte 3 of the colour word = 1: the observation count k_integrate's sum form adds up
            o4[0] = make_uint4(__float_as_uint(zz[0]), cc[0] | 0x01000000u, __float_as_uint(zz[1]), cc[1] | 0x01000000u);
            o4[1] = make_uint4(__float_as_uint(zz[2]), cc[2] | 0x01000000u, __float_as_uint(zz[3]), cc[3] | 0x01000000u);
        } else {
            for (int e = 0; e < nv; ++e) {
                const int pix = pix0 + e;
                zz[e] = C.depth_u16 ? (float)((const unsigned short*)dptr)[pix] / C.depth_scale : ((const float*)dptr)[pix];
                if (cptr) cc[e] = (unsigned)cptr[3 * (size_t)pix] | ((unsigned)cptr[3 * (size_t)pix + 1] << 8) | ((unsigned)cptr[3 * (size_t)pix + 2] << 16);
                out[pix] = make_uint2(__float_as_uint(zz[e]), cc[e] | 0x01000000u);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= nv) break;
            const float z = zz[e];
            if (!(z > 0)) continue;
            tmin = fminf(tmin, z); tmax = fmaxf(tmax, z);
            const int i = row, j = col0 + e;
            const float x = ((float)j - C.cx) * z / C.fx; // PointCloud.cpp:90-93
            const float y = ((float)i - C.cy) * z / C.fy;
            const float* M = P.pose;                       // Geometry.cpp:19-27
            const float q0 = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * 1.0f;
            const float q1 = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * 1.0f;
            const float q2 = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * 1.0f;
            const float q3 = ((M[12] * x + M[13] * y) + M[14] * z) + M[15] * 1.0f;
            // TransformPoints divides by w (Geometry.cpp:24-26).  A rigid pose has the bottom row (0, 0, 0, 1), so w is exactly 1 and x / 1 = x:
            // the three IEEE divisions (33 of the ~220 instructions per pixel) only run when some lane's w is not 1 (a projective "pose", NaN).
            float p0 = q0, p1 = q1, p2 = q2;
            if (__builtin_amdgcn_ballot_w64(q3 != 1.0f) != 0ull) { p0 = q0 / q3; p1 = q1 / q3; p2 = q2 / q3; }
            bool in = true; // Frustum::ContainPoint incl. its early "== 0 -> true" (Frustum.h:74-103)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float dist = sum3(P.planes[4 * k] * p0, P.planes[4 * k + 1] * p1, P.planes[4 * k + 2] * p2) + P.planes[4 * k + 3];
                if (dist < 0) { in = false; break; }
                if (dist == 0) break;
            }
            if (in) {
                ++inside;
                mx0 = p0 > mx0 ? p0 : mx0; mx1 = p1 > mx1 ? p1 : mx1; mx2 = p2 > mx2 ? p2 : mx2;
                mn0 = p0 < mn0 ? p0 : mn0; mn1 = p1 < mn1 ? p1 : mn1; mn2 = p2 < mn2 ? p2 : mn2;
            }
        }
    }
    // tiles: 4 lanes share a 16-pixel row segment, lane bits 4 and 5 are the wave's 4 rows, the 4 waves are the tile's 16 rows
    tmin = fminf(tmin, __shfl_xor(tmin, 1, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 1, 64));
    tmin = fminf(tmin, __shfl_xor(tmin, 2, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 2, 64));
    tmin = fminf(tmin, __shfl_xor(tmin, 16, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmin = fminf(tmin, __shfl_xor(tmin, 32, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    mx0 = wave_max(mx0); mx1 = wave_max(mx1); mx2 = wave_max(mx2);
    mn0 = wave_min(mn0); mn1 = wave_min(mn1); mn2 = wave_min(mn2);
    inside = wave_sum(inside);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
        s_red[wave][0] = mx0; s_red[wave][1] = mx1; s_red[wave][2] = mx2;
        s_red[wave][3] = mn0; s_red[wave][4] = mn1; s_red[wave][5] = mn2;
        s_cnt[wave] = inside;
    }
    if ((lane & 0x33) == 0) { s_tile[wave][lane >> 2][0] = tmin; s_tile[wave][lane >> 2][1] = tmax; }
    __syncthreads();
    if (tid < 4) {
        const int tx = gx * (kKaW / kTile) + tid, tw = tiles_w(C.width);
        if (tx < tw && gy < tiles_h(C.height)) {
            float lo = s_tile[0][tid][0], hi = s_tile[0][tid][1];
            for (int w = 1; w < 4; ++w) { lo = fminf(lo, s_tile[w][tid][0]); hi = fmaxf(hi, s_tile[w][tid][1]); }
            ptile[((size_t)f * tiles_h(C.height) + gy) * tw + tx] = make_float2(lo, hi);
        }
    }
    float* pout = partial + ((size_t)f * gridDim.x + blockIdx.x) * 8;
    if (tid < 6) {
        float v = s_red[0][tid];
        for (int w = 1; w < 4; ++w) v = tid < 3 ? fmaxf(v, s_red[w][tid]) : fminf(v, s_red[w][tid]);
        pout[tid] = v;
    } else if (tid == 6) {
        ((unsigned*)pout)[6] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    // ... and into the frame's accumulators, from which every KB workgroup takes the candidate range (they used to fold
    // the frame's 300 partial rows each: a third of that kernel's time).  The rows stay for op_volume_compute_bounding.
    if (tid < 7) {
        const unsigned cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (cnt) { // a workgroup without an in-frustum point contributes nothing (its row is the identity)
            unsigned* a = st->acc[f][blockIdx.x % kAccSlots];
            if (tid < 3) atomicMax(&a[tid], ord_enc(pout[tid]));
            else if (tid < 6) atomicMax(&a[tid], ~ord_enc(pout[tid]));
            else atomicAdd(&a[6], cnt);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// KB: PrepareCubes (CubeHandler.cpp:147-196) for every frame of the batch (blockIdx.y = frame).
// The reference probes EVERY block of the bbox +-1 range (8 corner-voxel GetSDF probes each); ~13 % are selected.  Here the range is cut
// into super-blocks of kSB^3 blocks, and a super-block is first tested as a whole against the tiles' depth range (k_prepare_frames):
//   * its 8 extreme voxel centres are transformed into the camera; all of them farther than 5 cm in front of it => every voxel centre of
//     the super-block projects inside the pixel bounding box of those 8 projections (a projective map keeps convex hulls while z > 0);
//   * the box widened by 2 px + 0.1 % lies outside the image => every probe is off-image (GetSDF = 999), nothing is selected;
//   * else, with [dmin, dmax] the valid depths of the tiles the box touches: dmin - zmax >= truncation + 1 mm or zmin - dmax >= truncation
//     + 1 mm (or no valid depth at all) => every probe has |sdf| >= truncation, nothing is selected.
// The margins are orders of magnitude above the float rounding of either side, so a super-block is only ever dropped when the exact
// per-block test below would reject every one of its blocks; the selected set is the reference's, bit for bit (parity suite, fuzz).
// Surviving super-blocks go through the exact test: one wave per super-block, one lane per block, 8 probes per lane.  A selected block
// is looked up / inserted in the hash table, its batch mask gets the frame's bit, and the first selection in the batch appends it to
// the batch list (collected in LDS, one global append per workgroup).  record != 0 (single-frame PrepareCubes API): also emits
// (table slot, candidate rank) pairs.
// ---------------------------------------------------------------------------------------------
#ifndef KB_SB
#define KB_SB 4
#endif
constexpr int kSB = KB_SB;                      // super-block edge in blocks
constexpr int kSBVol = kSB * kSB * kSB;         // 64 blocks = one wave
#ifndef KB_SBPERWG
#define KB_SBPERWG 8
#endif
constexpr int kSBPerWg = KB_SBPERWG;                    // super-blocks a workgroup tests at a time (8 lanes each)
constexpr int kSelTiles = 64;                   // a super-block whose pixel box touches more tiles skips the depth test (it is close to the camera)
static_assert(kSBVol == 64, "one lane per block of a super-block");

// -- the three steps of the selection, shared by k_select and k_select_vote ---------------------------------------------------------
// Finish ComputeBounding from the frame's accumulators (k_prepare_frames) and turn it into the candidate range (CubeHandler.cpp:147-163): wave 0 of a
// workgroup calls this, lane 0 leaves {i0, j0, k0, ni, nj, nk} in range[] (shared memory; all 0: no candidates) and, if `publish`, the frame's statistics in State.
__device__ __forceinline__ void frame_candidate_range(State* st, const CamParams& C, int f, int lane, bool publish, int* range) {
    unsigned tot = 0, e[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    const unsigned poisoned = lane == 0 ? (st->overflow & 3u) : 0u; // issued together with the accumulator loads: one round trip, not two
    if (lane < kAccSlots) { // lane k < kAccSlots reads set k (one round trip), then a 16-lane fold
        const unsigned* a = st->acc[f][lane];
        tot = a[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) e[c] = a[c];
    }
#pragma unroll
    for (int o = kAccSlots / 2; o > 0; o >>= 1) {
        tot += __shfl_xor(tot, o, 64);
#pragma unroll
        for (int c = 0; c < 6; ++c) { const unsigned x = __shfl_xor(e[c], o, 64); e[c] = x > e[c] ? x : e[c]; }
    }
    if (lane == 0) {
        float b[6];
        for (int c = 0; c < 6; ++c) // nothing in the frustum: the reference's lowest() / max() start values (CubeHandler.cpp:129-130)
            b[c] = tot ? ord_dec(c < 3 ? e[c] : ~e[c]) : (c < 3 ? -FLT_MAX : FLT_MAX);
        // A batch that ran out of pool / table space poisons the stream: its KC and every later batch do nothing (an empty
        // candidate range here), so that the host can grow the volume and REPLAY from the failing batch on -- no frame is
        // ever partially fused (vol_recover).  Read by one thread per workgroup: a per-thread load of this hot line next to
        // the candidate loop doubled the kernel's time.
        if (tot == 0 || poisoned) {
            for (int c = 0; c < 6; ++c) range[c] = 0;
        } else {
            for (int c = 0; c < 3; ++c) {
                // GetCubeID (VoxelCube.h:63-74): floor(p/res) in float -> int, then
                // floor((pb + 0.0)/8) in double == arithmetic shift by 3.
                const int hi = ((int)floorf(b[c] / C.res)) >> 3;
                const int lo = ((int)floorf(b[3 + c] / C.res)) >> 3;
                range[c] = lo - 1;
                range[3 + c] = hi - lo + 3;
            }
        }
        if (publish) {
            for (int c = 0; c < 6; ++c) st->bbox[f][c] = b[c];
            st->n_inside[f] = tot;
        }
    }
}

// Coarse test of the blocks [bi0..bi1] x [bj0..bj1] x [bk0..bk1] (part of a super-block) against frame M / tiles: false when the exact test below would
// reject every one of them (see the comment above k_select).  Called by 8 consecutive lanes, one per corner of the box, with the same arguments otherwise.
__device__ __forceinline__ bool superblock_survives(const CamParams& C, const float* __restrict__ M, const float2* __restrict__ tiles, int tw, int corner,
                                                    int bi0, int bi1, int bj0, int bj1, int bk0, int bk1, float cube_res, float o_lo, float o_hi) {
    const float px = (corner & 1) ? (float)bi1 * cube_res + o_hi : (float)bi0 * cube_res + o_lo;
    const float py = (corner & 2) ? (float)bj1 * cube_res + o_hi : (float)bj0 * cube_res + o_lo;
    const float pz = (corner & 4) ? (float)bk1 * cube_res + o_hi : (float)bk0 * cube_res + o_lo;
    const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    const float qy = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    const float qz = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    float zmin = qz, zmax = qz;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { zmin = fminf(zmin, __shfl_xor(zmin, o, 64)); zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64)); }
    if (!(zmin > 0.05f)) return true; // not all 8 extreme centres well in front of the camera: no shortcut
    // (qz > 0.05 on all 8 lanes.  v_rcp_f32 instead of the exact quotient: 1 ulp against margins of 2 px + 0.1 %, and two divisions were a sixth of this function)
    const float rz = __builtin_amdgcn_rcpf(qz);
    const float uf = (C.fx * q0) * rz + C.cx, vf = (C.fy * qy) * rz + C.cy;
    float umin = uf, umax = uf, vmin = vf, vmax = vf;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        umin = fminf(umin, __shfl_xor(umin, o, 64)); umax = fmaxf(umax, __shfl_xor(umax, o, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, o, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    }
    const float mu = 2.0f + 1e-3f * fmaxf(fabsf(umin), fabsf(umax)), mv = 2.0f + 1e-3f * fmaxf(fabsf(vmin), fabsf(vmax));
    const float u_lo = umin - mu, u_hi = umax + mu, v_lo = vmin - mv, v_hi = vmax + mv;
    const float wmax = (float)(C.width - 1), hmax = (float)(C.height - 1);
    if (!(u_hi >= 0.0f && u_lo <= wmax && v_hi >= 0.0f && v_lo <= hmax))
        return !(u_hi < 0.0f || u_lo > wmax || v_hi < 0.0f || v_lo > hmax); // NaN somewhere: no shortcut
    const int x0 = (int)fmaxf(u_lo, 0.0f), x1 = (int)fminf(u_hi, wmax), y0 = (int)fmaxf(v_lo, 0.0f), y1 = (int)fminf(v_hi, hmax);
    const int tx0 = x0 / kTile, tx1 = x1 / kTile, ty0 = y0 / kTile, ty1 = y1 / kTile;
    const int ntx = tx1 - tx0 + 1, nt = ntx * (ty1 - ty0 + 1);
    if (nt > kSelTiles) return true;
    float dmin = __builtin_inff(), dmax = -__builtin_inff();
    static_assert(kSelTiles == 64, "8 tiles per lane at most");
    float2 d[8]; // the lane's tiles t = corner, corner + 8, ...: independent loads, one round trip (a loop with one dependent load per trip was most of this function's time)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int t = corner + 8 * k;
        const int ty = (int)((float)t * (1.0f / (float)ntx) + 1e-4f); // t / ntx for 0 <= t < 64, 1 <= ntx <= 64 (the quotient's fractional part is 0 or >= 1/64)
        const int tx = t - ty * ntx;
        d[k] = t < nt ? tiles[(ty0 + ty) * tw + tx0 + tx] : make_float2(__builtin_inff(), -__builtin_inff());
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { dmin = fminf(dmin, d[k].x); dmax = fmaxf(dmax, d[k].y); }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { dmin = fminf(dmin, __shfl_xor(dmin, o, 64)); dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64)); }
    const float guard = C.trunc + 1e-3f;
    // (no valid depth in the tiles: dmin = +inf, dmax = -inf, both differences are +inf)
    return !(dmin - zmax >= guard || zmin - dmax >= guard);
}

// Integrator::GetSDF (Integrator.cpp:8-35) probes of the 8 corner voxels {0,7,56,63,448,455,504,511} of the block at (bx, by, bz) for the frame
// with inverse pose rows M and packed image img: all 8 projections first, then all 8 gathers in flight together, then the min.  True when the
// block is selected (CubeHandler.cpp:176-190: min |sdf| < truncation).  pmax = the largest pixel index among the corners (-1: none on the image).
template <bool FAST>
__device__ __forceinline__ bool block_selected(const CamParams& C, const float* __restrict__ M, const uint2* __restrict__ img, float bx, float by, float bz,
                                               float o_lo, float o_hi, int& pmax) {
    int pix[8];
    float zc[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float px = bx + ((corner & 1) ? o_hi : o_lo);
        const float py = by + ((corner & 2) ? o_hi : o_lo);
        const float pz = bz + ((corner & 4) ? o_hi : o_lo);
        const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
        const float q1c = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
        const float q2c = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
        zc[corner] = q2c;
        pix[corner] = project_pixel<FAST>(C, q0, q1c, q2c);
    }
    float dd[8];
    pmax = pix[0];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) { dd[corner] = pix[corner] >= 0 ? __uint_as_float(img[(unsigned)pix[corner]].x) : 0.0f; pmax = max(pmax, pix[corner]); }
    float min_sdf = FLT_MAX;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float sdf = dd[corner] <= 0 ? 999.0f : dd[corner] - zc[corner]; // off-image or d <= 0 -> 999
        const float a = fabsf(sdf);
        if (min_sdf > a) min_sdf = a;
    }
    return min_sdf < C.trunc;
}

#ifndef KB_MINWAVES
#define KB_MINWAVES 7
#endif
template <bool FAST>
__global__ __launch_bounds__(256, KB_MINWAVES) void k_select(BatchInv B, CamParams C, VolView V, const uint2* __restrict__ pimg, const float2* __restrict__ ptile,
                                                State* st, int record) {
    __shared__ int s_range[6]; // i0, j0, k0, ni, nj, nk
    __shared__ unsigned s_nsurv, s_nfirst, s_nrec, s_base[2];
    __shared__ unsigned s_surv[kSBPerWg];
    __shared__ int s_first[kSBPerWg * kSBVol];
    __shared__ unsigned short s_fpos[kSBPerWg * kSBVol]; // band (3 bits) | rank within the workgroup's entries of that band << 3
    __shared__ unsigned s_bcnt[kBands], s_bbase[kBands];
    __shared__ int s_rslot[kSBPerWg * kSBVol];
    __shared__ unsigned long long s_rcand[kSBPerWg * kSBVol];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup -> (frame, slot within the frame).  Consecutive workgroups land on consecutive XCDs, each with its own 4 MiB
    // L2, and a candidate's 8 corner probes gather from its frame's 2.4 MB packed image.
    // The batch is nf x 8 work units (a frame's chunks c with c % 8 == q); XCD x takes units [x nf, (x + 1) nf) in frame-major
    // order, i.e. exactly nf / 8 frames' worth whatever nf is, and walks them frame after frame (dispatch order ~ j), so that its L2 holds ONE
    // 2.4 MB image at a time.  (Whole frames per XCD -- frames x, x + 8, ... -- left some XCDs with two frames and others with one whenever
    // nf is not a multiple of 8: a 14-frame batch took as long as a 16-frame one.)
    int f, wslot, wstride;
    {
        const int nf = (int)gridDim.y, id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int per_unit = (int)gridDim.x >> 3;        // workgroups per unit
        const int x = id & 7, j = id >> 3;               // j = 0 .. nf * per_unit - 1 on this XCD
        const int u = x * nf + j / per_unit;             // global unit
        f = u >> 3;
        wslot = (u & 7) + 8 * (j % per_unit);            // 0 .. grid.x - 1; slot 0 of a frame also publishes the frame's statistics
        wstride = (int)gridDim.x;
    }
    const float* M = B.f[f].m;
    const uint2* img = pimg + (size_t)f * C.width * C.height;
    const int tw = tiles_w(C.width), th = tiles_h(C.height);
    const float2* tiles = ptile + (size_t)f * tw * th;

    if (wave == 0) frame_candidate_range(st, C, f, lane, wslot == 0, s_range);
    if (tid == 0) {
        s_nsurv = 0u; s_nfirst = 0u; s_nrec = 0u;
    }
    if (tid < kBands) s_bcnt[tid] = 0u;
    __syncthreads();
    // full batches file a block under the image band (eighths of the image height) it is first seen in; see kBands
    const bool bands = KC_BANDS != 0 && (int)gridDim.y >= KC_STEAL_MIN_FRAMES && record == 0;
    const float band_scale = 8.0f / (float)(C.width * C.height);
    const int i0 = s_range[0], j0 = s_range[1], k0 = s_range[2];
    int ni = s_range[3], nj = s_range[4], nk = s_range[5];
    unsigned long long ncand = (unsigned long long)((long long)ni * nj * nk);
    if (ni > 4096 || nj > 4096 || nk > 4096) { // > 160 m at 5 mm: treat as a bad frame, select nothing
        if (wslot == 0 && tid == 0) atomicOr(&st->overflow, 4u);
        ncand = 0; ni = nj = nk = 0;
    }
    if (wslot == 0 && tid == 0) st->n_cand[f] = ncand;

    const float cube_res = C.res * 8.0f; // CubeHandler.cpp:164
    const float half = C.res / 2;        // VoxelCube.h:47
    const float o_lo = 0.0f * C.res + half, o_hi = 7.0f * C.res + half; // VoxelCentroidOffSet of x = 0 / 7
    const bmask_t fbit = (bmask_t)1 << f;
    // super-block grid of the frame's range (<= 1024^3 < 2^32 entries)
    const unsigned nsi = (unsigned)(ni + kSB - 1) / kSB, nsj = (unsigned)(nj + kSB - 1) / kSB, nsk = (unsigned)(nk + kSB - 1) / kSB;
    const unsigned n_super = nsi * nsj * nsk;

    for (unsigned chunk = (unsigned)wslot; (unsigned long long)chunk * kSBPerWg < n_super; chunk += (unsigned)wstride) {
        // ---- coarse test: 8 lanes per super-block (one per corner), waves 0 and 1
        if (tid < kSBPerWg * 8) {
            const unsigned sb = chunk * kSBPerWg + (unsigned)(tid >> 3);
            const int corner = tid & 7;
            bool survive = false;
            if (sb < n_super) {
                const unsigned q1 = sb / nsk, q2 = q1 / nsj;
                const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
                // first and last block of the super-block inside the range, per axis
                const int bi0 = i0 + si * kSB, bj0 = j0 + sj * kSB, bk0 = k0 + sk * kSB;
                const int bi1 = min(bi0 + kSB - 1, i0 + ni - 1), bj1 = min(bj0 + kSB - 1, j0 + nj - 1), bk1 = min(bk0 + kSB - 1, k0 + nk - 1);
                survive = superblock_survives(C, M, tiles, tw, corner, bi0, bi1, bj0, bj1, bk0, bk1, cube_res, o_lo, o_hi);
            }
            if (survive && corner == 0) s_surv[atomicAdd(&s_nsurv, 1u)] = chunk * kSBPerWg + (unsigned)(tid >> 3);
        }
        __syncthreads();
        const unsigned nsurv = s_nsurv;
        // ---- exact test: one wave per surviving super-block, one lane per block
        for (unsigned sv = (unsigned)wave; sv < nsurv; sv += 4u) {
            const unsigned sb = s_surv[sv];
            const unsigned q1 = sb / nsk, q2 = q1 / nsj;
            const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
            const int ci = si * kSB + (lane >> 4), cj = sj * kSB + ((lane >> 2) & 3), ck = sk * kSB + (lane & 3); // position in the range
            bool first = false, rec = false;
            int pool_idx = -1, band = 0;
            if (ci < ni && cj < nj && ck < nk) {
                const int bi = i0 + ci, bj = j0 + cj, bk = k0 + ck;
                const float bx = (float)bi * cube_res, by = (float)bj * cube_res, bz = (float)bk * cube_res;
                int pmax; // the lowest on-image corner (largest pixel index): files the block under an image band below
                if (block_selected<FAST>(C, M, img, bx, by, bz, o_lo, o_hi, pmax)) {
                    if (!key_in_range(bi, bj, bk)) {
                        atomicOr(&st->overflow, 8u);
                    } else {
                        bool created;
                        pool_idx = table_claim(V, st, bi, bj, bk, &created); // table slot; KC translates it
                        if (pool_idx >= 0) {
                            first = atomicOr(&V.bmask[pool_idx], fbit) == (bmask_t)0;
                            rec = record != 0;
                            // the image row of the block's lowest on-image corner (pixel index / pixels per band; a heuristic, any band is correct)
                            if (bands) band = min(kBands - 1, (int)((float)max(pmax, 0) * band_scale));
                        }
                    }
                }
            }
            // wave-aggregated appends to the workgroup's lists: batch list (first selection in this batch) and record list
            const unsigned long long m_a = __ballot(first), m_b = __ballot(rec);
            const unsigned long long below = (1ULL << lane) - 1ULL;
            if (m_a) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_nfirst, (unsigned)__popcll(m_a));
                base = __shfl(base, 0, 64);
                if (first) {
                    const unsigned k = base + __popcll(m_a & below);
                    s_first[k] = pool_idx;
                    s_fpos[k] = (unsigned short)((unsigned)band | ((bands ? atomicAdd(&s_bcnt[band], 1u) : k) << 3));
                }
            }
            if (m_b) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_nrec, (unsigned)__popcll(m_b));
                base = __shfl(base, 0, 64);
                if (rec) {
                    const unsigned r = base + __popcll(m_b & below);
                    s_rslot[r] = pool_idx;
                    // candidate rank == position in the reference's i, j, k loop nest, k fastest (CubeHandler.cpp:170-173)
                    s_rcand[r] = ((unsigned long long)ci * (unsigned long long)nj + (unsigned long long)cj) * (unsigned long long)nk + (unsigned long long)ck;
                }
            }
        }
        __syncthreads();
        // ---- one global append per workgroup and list
        const unsigned nfirst = s_nfirst, nrec = s_nrec;
        if (tid < kBands) { // one global append per list
            const unsigned c = bands ? s_bcnt[tid] : (tid == 0 ? nfirst : 0u);
            s_bbase[tid] = c ? atomicAdd(&st->n_list[tid], c) : 0u;
        }
        if (tid == 0) {
            s_base[1] = nrec ? atomicAdd(&st->n_rec, nrec) : 0u;
            s_nsurv = 0u;
        }
        __syncthreads();
        for (unsigned k = (unsigned)tid; k < nfirst; k += 256u) {
            const unsigned fp = s_fpos[k], b = fp & 7u;
            const unsigned pos = s_bbase[b] + (fp >> 3);
            if (pos < V.max_blocks) V.blist[(size_t)b * V.max_blocks + pos] = s_first[k];
        }
        for (unsigned k = (unsigned)tid; k < nrec; k += 256u) {
            const unsigned pos = s_base[1] + k;
            if (pos < V.max_blocks) { V.sel_list[pos] = s_rslot[k]; V.sel_cand[pos] = s_rcand[k]; }
        }
        __syncthreads(); // the lists are reused by the next chunk
        if (tid == 0) { s_nfirst = 0u; s_nrec = 0u; }
        if (tid < kBands) s_bcnt[tid] = 0u;
        // (the next chunk's coarse test does not touch s_nfirst / s_nrec / s_bcnt; its __syncthreads orders the reset before their next use)
    }
}

// ---------------------------------------------------------------------------------------------
// KB for a batch of several frames, in two steps.  Consecutive frames select nearly the same blocks -- a block of the bench scene is selected by ~25 of a
// batch's 32 frames -- and in k_select every one of those selections is a hash probe + a returning atomicOr on the block's batch mask + a list append by
// whichever frame came first, ~650 k dependent round trips per batch, with the lists in LDS that force four barriers on every chunk of super-blocks.
// k_select_vote only RECORDS a frame's selections: one 64-bit word per super-block of its range (bit = lane = block; 0 for a super-block the coarse
// test dropped), plain stores into sbits[f][super-block].  Nothing is shared between the waves of a workgroup any more, so every wave walks chunks of 8
// super-blocks on its own -- coarse test (8 lanes per super-block), then the exact test of each survivor (one lane per block) -- without a barrier.
// The super-blocks are aligned to absolute block coordinates (block >> 2), so that the frames of a batch cut space into the SAME super-blocks: the
// first and last super-block of an axis may be partly outside the frame's range (bits of blocks outside it stay 0; a block is a candidate of frame f
// iff it lies in f's range, as in k_select).  k_select_merge then ORs the frames' words per super-block and claims every selected block once.
// A frame whose range has more super-blocks than a row of sbits holds (kVoteCap) claims directly, like k_select; the two mix freely (both OR into bmask).
// ---------------------------------------------------------------------------------------------
#ifndef KB_VOTE
#define KB_VOTE 1            // 0: every batch goes through k_select
#endif
#ifndef KB_VOTE_MIN_FRAMES
#define KB_VOTE_MIN_FRAMES 20 // per batch, k_select against k_select_vote + k_select_merge (profiles/r04_ab_kb_select.txt): 24 / 33 us at 4 frames, 33 / 39 at 8, 51 / 51 at 16, 82 / 75 at 32
#endif
#ifndef KB_VOTE_WGS
#define KB_VOTE_WGS 1792     // workgroups of a k_select_vote launch (all resident: 7 per CU), shared out among the frames
#endif
static_assert(kSB == 4, "k_select_vote / k_select_merge: super-block = block >> 2");

#ifdef KB_TRACE // development aid (make EXTRA=-DKB_TRACE, tools/kb_trace.sh): per-wave phase times of the last k_select_vote launch, dumped by op_volume_destroy
__device__ unsigned long long g_kb_trace[kSelectGrid * kMaxBatch * 4 * 8];
#define KB_T(K) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr_[K] += now_ - tr_last_; tr_last_ = now_; } while (0)
#define KB_N(K, V) do { tr_[K] += (V); } while (0)
#else
#define KB_T(K) do { } while (0)
#define KB_N(K, V) do { } while (0)
#endif
template <bool FAST>
__global__ __launch_bounds__(256, KB_MINWAVES) void k_select_vote(BatchInv B, CamParams C, VolView V, const uint2* __restrict__ pimg, const float2* __restrict__ ptile,
                                                                  State* st, unsigned long long* __restrict__ sbits, unsigned vote_cap) {
    __shared__ int s_range[6]; // i0, j0, k0, ni, nj, nk
    __shared__ unsigned s_vn[3], s_vsb[3][32]; // the survivors of three consecutive rounds (one barrier per round)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) s_vn[tid] = 0u;
    int f, wslot, wstride; // workgroup -> (frame, slot within the frame): whole frames per XCD, as in k_select
    {
        const int nf = (int)gridDim.y, id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int per_unit = (int)gridDim.x >> 3;
        const int x = id & 7, j = id >> 3;
        const int u = x * nf + j / per_unit;
        f = u >> 3;
        wslot = (u & 7) + 8 * (j % per_unit);
        wstride = (int)gridDim.x;
    }
    const float* M = B.f[f].m;
    const uint2* img = pimg + (size_t)f * C.width * C.height;
    const int tw = tiles_w(C.width), th = tiles_h(C.height);
    const float2* tiles = ptile + (size_t)f * tw * th;
#ifdef KB_TRACE
    unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last_ = __builtin_amdgcn_s_memtime();
#endif
    if (wave == 0) frame_candidate_range(st, C, f, lane, wslot == 0, s_range);
    __syncthreads();
    KB_T(0);
    const int i0 = s_range[0], j0 = s_range[1], k0 = s_range[2];
    int ni = s_range[3], nj = s_range[4], nk = s_range[5];
    unsigned long long ncand = (unsigned long long)((long long)ni * nj * nk);
    if (ni > 4096 || nj > 4096 || nk > 4096) { // > 160 m at 5 mm: treat as a bad frame, select nothing
        if (wslot == 0 && tid == 0) atomicOr(&st->overflow, 4u);
        ncand = 0; ni = nj = nk = 0;
    }
    if (wslot == 0 && tid == 0) st->n_cand[f] = ncand;
    const float cube_res = C.res * 8.0f; // CubeHandler.cpp:164
    const float half = C.res / 2;        // VoxelCube.h:47
    const float o_lo = 0.0f * C.res + half, o_hi = 7.0f * C.res + half; // VoxelCentroidOffSet of x = 0 / 7
    const bmask_t fbit = (bmask_t)1 << f;
    const int oi = (i0 >> 2) * kSB, oj = (j0 >> 2) * kSB, ok = (k0 >> 2) * kSB; // first block of super-block 0 (<= the first block of the range)
    const unsigned nsi = ni > 0 ? (unsigned)(i0 + ni - oi + kSB - 1) / kSB : 0u, nsj = nj > 0 ? (unsigned)(j0 + nj - oj + kSB - 1) / kSB : 0u,
                   nsk = nk > 0 ? (unsigned)(k0 + nk - ok + kSB - 1) / kSB : 0u;
    const unsigned n_super = nsi * nsj * nsk; // (<= 1026^3 < 2^32)
    const bool vote = n_super <= vote_cap; // (vote_cap <= kVoteCap, a row of sbits)
    unsigned long long* bits = sbits + (size_t)f * kVoteCap;
    if (wslot == 0 && tid == 0) { // for k_select_merge: the super-blocks this frame's words are laid out over (extent 0: it has none)
        int* r = st->sel_rng[f];
        r[0] = oi >> 2; r[1] = oj >> 2; r[2] = ok >> 2;
        r[3] = vote ? (int)nsi : 0; r[4] = vote ? (int)nsj : 0; r[5] = vote ? (int)nsk : 0;
        r[6] = !vote && n_super != 0u; // this frame claims directly: the merge step must expect batch masks that are already set
    }
    // A round = 32 super-blocks: every wave runs the coarse test of 8 of them (8 lanes per super-block, one per corner), the survivors of the four waves
    // are pooled in LDS and dealt out again for the exact test (one wave per super-block, one lane per block) -- a wave's own 8 super-blocks hold anything
    // from 0 to 8 survivors.  The pool of round r + 2 is emptied while round r runs, so one barrier per round is enough.
    // The 32 super-blocks of a round are spread evenly over the range (slot s of round r = super-block s * n_rounds + r), not adjacent: survivors come in
    // clusters -- a round of 32 neighbours has anything from 0 to 32 of them, and the busiest workgroup decided the kernel's length.
    // (Rounds drawn from a per-frame counter instead of the fixed stride: measured, no gain -- a workgroup has two rounds, the draw for the second is
    // made before the first one's weight is known.)
    const unsigned n_rounds = (n_super + 31u) / 32u;
    const float inv_nsk = 1.0f / (float)nsk, inv_nsj = 1.0f / (float)nsj;
    auto div_small = [](unsigned a, unsigned d, float inv_d) { // floor(a / d) for a < 2^22: the float quotient is off by one at most
        unsigned q = (unsigned)((float)a * inv_d);
        const unsigned r = q * d;
        if (r > a) --q; else if (a - r >= d) ++q;
        return q;
    };
    unsigned vc = 0;
    for (unsigned round = (unsigned)wslot; round < n_rounds; round += (unsigned)wstride, vc = vc == 2u ? 0u : vc + 1u) {
        {
            const unsigned sb = (unsigned)(tid >> 3) * n_rounds + round;
            const int corner = lane & 7;
            bool survive = false;
            if (sb < n_super) {
                const unsigned q1 = vote ? div_small(sb, nsk, inv_nsk) : sb / nsk, q2 = vote ? div_small(q1, nsj, inv_nsj) : q1 / nsj;
                const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
                // first and last block of the super-block inside the range, per axis
                const int bi0 = max(oi + si * kSB, i0), bj0 = max(oj + sj * kSB, j0), bk0 = max(ok + sk * kSB, k0);
                const int bi1 = min(oi + si * kSB + kSB - 1, i0 + ni - 1), bj1 = min(oj + sj * kSB + kSB - 1, j0 + nj - 1), bk1 = min(ok + sk * kSB + kSB - 1, k0 + nk - 1);
                survive = superblock_survives(C, M, tiles, tw, corner, bi0, bi1, bj0, bj1, bk0, bk1, cube_res, o_lo, o_hi);
                if (corner == 0) {
                    if (survive) s_vsb[vc][atomicAdd(&s_vn[vc], 1u)] = sb;
                    else if (vote) bits[sb] = 0ull; // dropped as a whole: no block of it is selected
                }
            }
        }
        KB_T(1); KB_N(4, 1);
        __syncthreads();
        const unsigned n_todo = s_vn[vc];
        if (tid == 0) s_vn[vc == 0u ? 2u : vc - 1u] = 0u; // the pool of the round after the next (its last readers have passed the barrier above)
        KB_T(3);
        // ---- exact test: one lane per block of a surviving super-block
        for (unsigned sv = (unsigned)wave; sv < n_todo; sv += 4u) {
            KB_N(5, 1);
            const unsigned sb = s_vsb[vc][sv];
            const unsigned q1 = vote ? div_small(sb, nsk, inv_nsk) : sb / nsk, q2 = vote ? div_small(q1, nsj, inv_nsj) : q1 / nsj;
            const int sk = (int)(sb - q1 * nsk), sj = (int)(q1 - q2 * nsj), si = (int)q2;
            const int bi = oi + si * kSB + (lane >> 4), bj = oj + sj * kSB + ((lane >> 2) & 3), bk = ok + sk * kSB + (lane & 3);
            bool selected = false;
            if (bi >= i0 && bi < i0 + ni && bj >= j0 && bj < j0 + nj && bk >= k0 && bk < k0 + nk) { // a candidate of this frame
                int pmax;
                selected = block_selected<FAST>(C, M, img, (float)bi * cube_res, (float)bj * cube_res, (float)bk * cube_res, o_lo, o_hi, pmax);
            }
            if (vote) { // the frame's word for this super-block
                const unsigned long long word = __ballot(selected);
                if (lane == 0) bits[sb] = word;
                KB_T(2);
                continue;
            }
            // (a range too large for sbits: claim directly)
            int slot = -1;
            if (selected) {
                if (!key_in_range(bi, bj, bk)) {
                    atomicOr(&st->overflow, 8u);
                } else {
                    bool created;
                    const int ts = table_claim(V, st, bi, bj, bk, &created);
                    if (ts >= 0 && atomicOr(&V.bmask[ts], fbit) == (bmask_t)0) slot = ts;
                }
            }
            const unsigned long long got = __ballot(slot >= 0);
            unsigned base = 0;
            if (lane == 0 && got) base = atomicAdd(&st->n_list[0], (unsigned)__popcll(got));
            base = __shfl(base, 0, 64);
            if (slot >= 0) {
                const unsigned pos = base + (unsigned)__popcll(got & ((1ULL << lane) - 1ULL));
                if (pos < V.max_blocks) V.blist[pos] = slot;
            }
        }
    }
#ifdef KB_TRACE
    if (lane == 0) for (int k = 0; k < 8; ++k) g_kb_trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + k] = tr_[k];
#endif
}

// ---------------------------------------------------------------------------------------------
// KB, second step of a voting batch.  The frames' words are laid out over their own ranges, but on the same absolute super-block grid, so the words of
// different frames for one super-block can be put side by side: one wave per super-block S, lane e fetches frame e's word for S (0 where S is outside e's
// range) -- one round trip for the whole batch.  The 32 x 64 bit matrix is transposed with one ballot per selected block (lane = block gets the mask of the
// frames that selected it), each block is claimed once, its complete mask ORed into its batch mask, and -- unless a frame on the direct path listed it
// first -- it joins the batch list.  The walk covers the bounding range of the frames' ranges; if the frames lie so far apart that this has more
// super-blocks than the frames' words together, the words are walked instead (frame by frame) and the wave of the LOWEST frame whose word for S is not 0
// deals with S.  A workgroup handles kMergeWords consecutive super-blocks and appends their blocks as ONE segment in super-block / lane order: the batch
// list comes out in runs of spatially adjacent blocks (super-blocks k fastest), which k_integrate rewards -- its workgroups draw consecutive entries, and
// neighbours gather from the same image lines at the same time.
// ---------------------------------------------------------------------------------------------
#ifndef KB_MERGE_WORDS
#define KB_MERGE_WORDS 8
#endif
constexpr int kMergeWords = KB_MERGE_WORDS; // = waves per workgroup
#ifndef KB_MERGE_GRID
#define KB_MERGE_GRID 2048
#endif
__global__ __launch_bounds__(64 * kMergeWords) void k_select_merge(VolView V, State* st, const unsigned long long* __restrict__ sbits, int nf) {
    __shared__ int s_r[kMaxBatch][6];          // first super-block (absolute) and extent in super-blocks of every frame's words
    __shared__ unsigned s_pre[kMaxBatch + 1];  // words before frame f (only for the walk over the words)
    __shared__ unsigned s_wc[kMergeWords], s_wp[kMergeWords], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this lane's frame (lane < nf; every wave loads the 32 rows itself: no barrier in the usual case), for the gathers below
    int e0 = 0, e1 = 0, e2 = 0, en0 = 0, en1 = 0, en2 = 0, direct = 0;
    if (lane < nf) {
        const int* r = st->sel_rng[lane];
        e0 = r[0]; e1 = r[1]; e2 = r[2]; en0 = r[3]; en1 = r[4]; en2 = r[5]; direct = r[6];
    }
    const bool any_direct = __ballot(direct != 0) != 0ull;
    unsigned total = (unsigned)en0 * (unsigned)en1 * (unsigned)en2; // this frame's words (<= kVoteCap), then all frames' (<= 64 x 2^18)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    // the bounding range of the frames' words (every wave computes it: six 64-lane reductions)
    const bool has_words = en0 > 0 && en1 > 0 && en2 > 0;
    int lo0 = has_words ? e0 : INT_MAX, lo1 = has_words ? e1 : INT_MAX, lo2 = has_words ? e2 : INT_MAX;
    int hi0 = has_words ? e0 + en0 : INT_MIN, hi1 = has_words ? e1 + en1 : INT_MIN, hi2 = has_words ? e2 + en2 : INT_MIN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo0 = min(lo0, __shfl_xor(lo0, o, 64)); lo1 = min(lo1, __shfl_xor(lo1, o, 64)); lo2 = min(lo2, __shfl_xor(lo2, o, 64));
        hi0 = max(hi0, __shfl_xor(hi0, o, 64)); hi1 = max(hi1, __shfl_xor(hi1, o, 64)); hi2 = max(hi2, __shfl_xor(hi2, o, 64));
    }
    const unsigned un0 = (unsigned)hi0 - (unsigned)lo0, un1 = (unsigned)hi1 - (unsigned)lo1, un2 = (unsigned)hi2 - (unsigned)lo2; // (meaningless without words: total == 0)
    const unsigned long long n_union = total ? (unsigned long long)un0 * (unsigned long long)un1 * (unsigned long long)un2 : 0ull;
    const bool by_union = n_union <= (unsigned long long)total; // (the usual case: consecutive frames of one camera)
    const unsigned n_units = by_union ? (unsigned)n_union : total;
    if (!by_union) { // the walk over the words needs to know where a frame's words start
        if (tid < nf) {
#pragma unroll
            for (int c = 0; c < 6; ++c) s_r[tid][c] = st->sel_rng[tid][c];
        }
        __syncthreads();
        if (tid == 0) {
            s_pre[0] = 0u;
            for (int f = 0; f < nf; ++f) s_pre[f + 1] = s_pre[f] + (unsigned)s_r[f][3] * (unsigned)s_r[f][4] * (unsigned)s_r[f][5];
        }
        __syncthreads();
    }

    for (unsigned run = blockIdx.x; (unsigned long long)run * kMergeWords < n_units; run += gridDim.x) {
        const unsigned g = run * kMergeWords + (unsigned)wave;
        int slot = -1;
        if (g < n_units) {
            int S0, S1, S2, f = -1; // the super-block (absolute); per-word walk: the frame whose word this is
            if (by_union) {
                const unsigned q1 = g / un2, q2 = q1 / un1;
                S0 = lo0 + (int)q2; S1 = lo1 + (int)(q1 - q2 * un1); S2 = lo2 + (int)(g - q1 * un2);
            } else {
                f = (int)__popcll(__ballot(lane < nf && s_pre[lane + 1] <= g)); // frames whose words end at or before g (kMaxBatch <= 64 lanes)
                const unsigned sb = g - s_pre[f];
                const int* r = s_r[f];
                const unsigned nsj = (unsigned)r[4], nsk = (unsigned)r[5];
                const unsigned q1 = sb / nsk, q2 = q1 / nsj;
                S0 = r[0] + (int)q2; S1 = r[1] + (int)(q1 - q2 * nsj); S2 = r[2] + (int)(sb - q1 * nsk);
            }
            // lane e: frame e's word for S
            unsigned long long w = 0ull;
            const int d0 = S0 - e0, d1 = S1 - e1, d2 = S2 - e2;
            if ((unsigned)d0 < (unsigned)en0 && (unsigned)d1 < (unsigned)en1 && (unsigned)d2 < (unsigned)en2)
                w = sbits[(size_t)lane * kVoteCap + (((unsigned)d0 * (unsigned)en1 + (unsigned)d1) * (unsigned)en2 + (unsigned)d2)];
            const unsigned long long voters = __ballot(w != 0ull);
            if (voters != 0ull && (by_union || (int)__builtin_ctzll(voters) == f)) {
                // transpose: which blocks are selected at all, then one ballot per selected block
                unsigned long long any = w;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o, 64);
                bmask_t m = (bmask_t)0;
                for (unsigned long long rem = any; rem != 0ull; rem &= rem - 1ull) {
                    const int l = (int)__builtin_ctzll(rem);
                    const unsigned long long col = __ballot((w >> l) & 1ull);
                    if (lane == l) m = (bmask_t)col;
                }
                if (m != (bmask_t)0) {
                    const int bi = S0 * kSB + (lane >> 4), bj = S1 * kSB + ((lane >> 2) & 3), bk = S2 * kSB + (lane & 3);
                    if (!key_in_range(bi, bj, bk)) {
                        atomicOr(&st->overflow, 8u);
                    } else {
                        bool created;
                        const int ts = table_claim(V, st, bi, bj, bk, &created); // table slot; KC translates it
                        if (ts >= 0) {
                            // this wave is the only one that sees this block -- unless a frame claims directly, then the batch mask tells who listed it
                            if (!any_direct) { V.bmask[ts] = m; slot = ts; }
                            else if (atomicOr(&V.bmask[ts], m) == (bmask_t)0) slot = ts;
                        }
                    }
                }
            }
        }
        const unsigned long long got = __ballot(slot >= 0);
        if (lane == 0) s_wc[wave] = (unsigned)__popcll(got);
        if (__syncthreads_or(got != 0ull) == 0) continue; // nothing selected in these super-blocks
        if (tid == 0) {
            unsigned n = 0;
            for (int w = 0; w < kMergeWords; ++w) { s_wp[w] = n; n += s_wc[w]; }
            s_base = atomicAdd(&st->n_list[0], n);
        }
        __syncthreads();
        if (slot >= 0) {
            const unsigned pos = s_base + s_wp[wave] + (unsigned)__popcll(got & ((1ULL << lane) - 1ULL));
            if (pos < V.max_blocks) V.blist[pos] = slot;
        }
    }
}

} // namespace

namespace opv {

void launch_prepare_frames(op_volume* v, const BatchFwd& F, int nf, const CamParams& C, const BatchPtrs& Q, unsigned seq) {
    const int g1 = ka_grid(C.width, C.height);
    for (int f0 = 0; f0 < nf; f0 += kKaFrames) {
        KaFwd A;
        const int na = nf - f0 < kKaFrames ? nf - f0 : kKaFrames;
        std::memcpy(A.f, F.f + f0, sizeof(PoseFwd) * (size_t)na);
        hipLaunchKernelGGL(k_prepare_frames, dim3(g1, na), dim3(256), 0, v->stream, A, f0, C, Q, v->pimg, v->ptile, v->partial, v->state, seq,
                           (const unsigned*)v->n_blocks, v->hstat_dev);
    }
}

int launch_select(op_volume* v, const BatchInv& I, const CamParams& C, int nf, bool record, const int* cube_keys, unsigned n_cubes) {
    const VolView V = v->view();
    if (cube_keys)
        hipLaunchKernelGGL(k_mark_cubes, dim3((n_cubes + 255u) / 256u), dim3(256), 0, v->stream, V, v->state, cube_keys, n_cubes);
    else if (KB_VOTE && KC_BANDS == 0 && !record && nf >= (v->select_mode > 0 ? 2 : KB_VOTE_MIN_FRAMES) && v->select_mode != OP_VOLUME_SELECT_DIRECT) { // (an explicit limit: every batch of >= 2 frames) // several frames: they record their selections, one pass claims every block once
        // the frames' voting words (64 MB): only a volume that takes this path ever has them (PrepareCubes / ComputeBounding volumes, short batches, the
        // many small sub-map volumes of a DenseSlam run do not)
        if (!v->sbits) OP_HIP(op::cached_malloc((void**)&v->sbits, (size_t)kMaxBatch * kVoteCap * sizeof(unsigned long long)));
        const unsigned vote_cap = v->select_mode > 0 ? (unsigned)v->select_mode : kVoteCap;
        const int per_frame = std::max(8, std::min(kSelectGrid, (KB_VOTE_WGS / nf + 7) / 8 * 8)); // a multiple of 8: whole frames per XCD
        if (C.fast_px)
            hipLaunchKernelGGL(k_select_vote<true>, dim3(per_frame, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile, v->state, v->sbits, vote_cap);
        else
            hipLaunchKernelGGL(k_select_vote<false>, dim3(per_frame, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile, v->state, v->sbits, vote_cap);
        hipLaunchKernelGGL(k_select_merge, dim3(KB_MERGE_GRID), dim3(64 * kMergeWords), 0, v->stream, V, v->state, (const unsigned long long*)v->sbits, nf);
    } else if (C.fast_px)
        hipLaunchKernelGGL(k_select<true>, dim3(kSelectGrid, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile,
                           v->state, record ? 1 : 0);
    else
        hipLaunchKernelGGL(k_select<false>, dim3(kSelectGrid, nf), dim3(256), 0, v->stream, I, C, V, (const uint2*)v->pimg, (const float2*)v->ptile,
                           v->state, record ? 1 : 0);
    return OP_OK;
}

void launch_finish_select(op_volume* v) { hipLaunchKernelGGL(k_finish_select, dim3(256), dim3(256), 0, v->stream, v->view(), v->state); }

void kb_trace_dump(op_volume* v) {
#ifdef KB_TRACE
    if (hipSetDevice(v->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
        const int nw = std::max(8, std::min(kSelectGrid, (KB_VOTE_WGS / kMaxBatch + 7) / 8 * 8)) * kMaxBatch * 4; // waves of a full batch's launch
        std::vector<unsigned long long> t((size_t)nw * 8);
        if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_kb_trace), t.size() * 8) == hipSuccess) {
            double sum[8] = {0}, mx[8] = {0}, tot_max = 0, tot_sum = 0;
            std::vector<double> tots;
            for (int w = 0; w < nw; ++w) {
                double tot = 0;
                for (int k = 0; k < 8; ++k) { const double d = (double)t[(size_t)w * 8 + k]; sum[k] += d; mx[k] = std::max(mx[k], d); if (k < 4) tot += d; }
                tot_max = std::max(tot_max, tot); tot_sum += tot; tots.push_back(tot);
            }
            std::sort(tots.begin(), tots.end());
            const double n = nw;
            fprintf(stderr, "kb trace (shader cycles, %d waves; mean/max): total %.0f/%.0f (median %.0f, 90%% %.0f, 99%% %.0f) setup %.0f/%.0f coarse %.0f/%.0f exact %.0f/%.0f barrier %.0f/%.0f | rounds %.2f/%.0f exact tests %.2f/%.0f\n",
                    nw, tot_sum / n, tot_max, tots[tots.size() / 2], tots[tots.size() * 9 / 10], tots[tots.size() * 99 / 100], sum[0] / n, mx[0], sum[1] / n, mx[1], sum[2] / n, mx[2], sum[3] / n, mx[3], sum[4] / n, mx[4], sum[5] / n, mx[5]);
        }
    }
#else
    (void)v;
#endif
}

} // namespace opv
