// volume.hip -- device-resident voxel-block-hashed TSDF volume for gfx950 (MI355X) and the C-ABI
// entry points op_volume_* declared in include/onepiece_hip.h.
//
// What it replaces (file:line under /root/reference/src):
//   integration::CubeHandler::IntegrateImage      Integration/CubeHandler.cpp:197-210
//   CubeHandler::ComputeBounding  (kernel K1)     Integration/CubeHandler.cpp:116-145
//   CubeHandler::PrepareCubes     (kernel K2)     Integration/CubeHandler.cpp:147-196
//   Integrator::GetSDF                            Integration/Integrator.cpp:8-35
//   Integrator::IntegrateImage    (kernel K3)     Integration/Integrator.cpp:36-94
//   TSDFVoxel::operator+                          Integration/TSDFVoxel.h:24-39
//   CubeHandler::Merge            (kernel K4*)    Integration/CubeHandler.h:145-167
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   pool   : max_blocks x [5 planes x 512 floats]; plane order sdf, weight, c0, c1, c2; in-plane
//            index = the reference's voxel id x + 8y + 64z.  A wave64 therefore owns one z-slice
//            and reads/writes 256 contiguous bytes per plane -- fully coalesced, unlike the
//            reference's 20-byte AoS TSDFVoxel.
//   keys   : max_blocks x int32[3] (block id), indexed by pool slot.
//   table  : open-addressing hash table of int4 {x, y, z, pool slot}; the probe start is the low
//            bits of the reference's 64-bit VoxelGridHasher value (Geometry/Geometry.h:101-112).
//   frame lists (sel_list / sel_cand) : pool slots + candidate ranks of the blocks PrepareCubes
//            selected for the current frame.
// The whole per-frame path (K1 -> K2 -> K3) is enqueued without any host synchronisation: every
// kernel is launched with a fixed grid and reads its trip counts from device memory.
//
// Floating point: compiled with -ffp-contract=off; every expression keeps the reference's operand
// order and intermediate types (see oracle/onepiece_oracle.c, which this must match bit for bit on
// block selection and to rounding on voxel values -- in practice also bit for bit).
#include <cfloat>
#include <climits>
#include <algorithm>
#include <numeric>
#include <vector>

#include "common.hpp"
#include "host_math.hpp"

namespace op {
thread_local char g_last_error[512] = "";
}

namespace {

using op::fail;

constexpr int kVox = 512;            // voxels per block (CUBE_SIZE^3, VoxelCube.h:4)
constexpr int kBlockFloats = 5 * kVox;
constexpr int kEmpty = -1;           // table slot never used
constexpr int kLocked = -2;          // slot reserved by an in-flight insert of another key
constexpr int kDead = -3;            // reserved but the pool was full
constexpr int kKeySentinel = INT_MIN;
constexpr int kPixPerWg = 1024;      // K1: pixels per workgroup (256 threads x 4)
constexpr int kSelectGrid = 1024;    // K2 persistent grid (256-thread workgroups)
constexpr int kIntegrateGrid = 1024; // K3 persistent grid (512-thread workgroups, 4 per CU)

struct FrameParams {
    float pose[16];
    float pose_inv[16];
    float planes[24]; // top, left, right, bottom, near, far
    float fx, fy, cx, cy, depth_scale, res, trunc;
    int width, height, depth_u16;
};

struct FrameState {
    unsigned n_sel;    // length of this frame's cube_id_list
    unsigned overflow; // bit0: pool full, bit1: table full, bit2: candidate range too large
    unsigned long long n_cand;
    unsigned long long stat_frames, stat_sel;
    float bbox[6]; // max xyz, min xyz of the last ComputeBounding
    unsigned n_inside, pad;
};

struct VolView {
    int4* table;
    unsigned table_mask;
    int* keys;
    float* pool;
    unsigned max_blocks;
    unsigned* n_blocks;
    int* sel_list;
    unsigned long long* sel_cand;
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long hash_key_dev(int x, int y, int z) {
    return ((unsigned long long)(long long)x * 73856093ULL) ^ ((unsigned long long)(long long)y * 19349663ULL) ^
           ((unsigned long long)(long long)z * 83492791ULL);
}

// Eigen's 3-term reduction order a0 + (a1 + a2).
__device__ __forceinline__ float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }

// Integrator.cpp:20-21: (((f*X)/Z) + 0.5) + c in double, truncated toward zero; values that do not
// fit an int (UB on the CPU, INT_MIN from cvttsd2si) are mapped to INT_MIN and fail the bounds test.
__device__ __forceinline__ int project_px(float f, float X, float Z, float c) {
    const double t = (double)((f * X) / Z) + 0.5 + (double)c;
    if (!(t > -2147483649.0 && t < 2147483648.0)) return INT_MIN;
    return (int)t;
}

__device__ __forceinline__ float depth_at(const void* depth, int is_u16, float depth_scale, size_t idx) {
    if (!is_u16) return ((const float*)depth)[idx];
    return (float)((const unsigned short*)depth)[idx] / depth_scale; // Integrator.cpp:29
}

// Integrator::GetSDF with pose_inv precomputed (Integrator.cpp:8-35).
__device__ __forceinline__ float get_sdf(const FrameParams& P, const void* depth, float px, float py, float pz) {
    const float* M = P.pose_inv;
    const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    const float q1 = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    const float q2 = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    const int u = project_px(P.fx, q0, q2, P.cx);
    const int v = project_px(P.fy, q1, q2, P.cy);
    if (v < 0 || v >= P.height || u < 0 || u >= P.width) return 999.0f;
    const float d = depth_at(depth, P.depth_u16, P.depth_scale, (size_t)v * P.width + u);
    if (d <= 0) return 999.0f;
    return d - q2;
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Looks `key` up; if absent reserves a slot (state kLocked) for it.  Returns the pool slot (>= 0)
// when the block exists, otherwise -1 and *reserved = table slot (or -1 when the table is full).
// Concurrent callers in one launch always carry distinct keys, so a locked slot belongs to some
// other key and can be skipped.
__device__ int table_find_or_reserve(const VolView& V, int x, int y, int z, int* reserved) {
    unsigned s = (unsigned)hash_key_dev(x, y, z) & V.table_mask;
    *reserved = -1;
    for (unsigned probe = 0; probe <= V.table_mask; ++probe, s = (s + 1) & V.table_mask) {
        int4 e = V.table[s];
        if (e.w == kEmpty) {
            const int old = atomicCAS(&((int*)&V.table[s])[3], kEmpty, kLocked);
            if (old == kEmpty) { *reserved = (int)s; return -1; }
            continue; // somebody else (another key) took it
        }
        if (e.w >= 0 && e.x == x && e.y == y && e.z == z) return e.w;
    }
    return -1;
}

__device__ int table_find(const VolView& V, int x, int y, int z) {
    unsigned s = (unsigned)hash_key_dev(x, y, z) & V.table_mask;
    for (unsigned probe = 0; probe <= V.table_mask; ++probe, s = (s + 1) & V.table_mask) {
        const int4 e = V.table[s];
        if (e.w == kEmpty) return -1;
        if (e.w >= 0 && e.x == x && e.y == y && e.z == z) return e.w;
    }
    return -1;
}

__device__ __forceinline__ void table_publish(const VolView& V, int slot, int x, int y, int z, int pool_idx) {
    int* e = (int*)&V.table[slot];
    e[0] = x; e[1] = y; e[2] = z;
    __threadfence();
    atomicExch(&e[3], pool_idx);
}

// ---------------------------------------------------------------------------------------------
// pool / table maintenance
// ---------------------------------------------------------------------------------------------
__global__ void k_fill_pool(float* pool, size_t first_block, size_t n_blocks) {
    // default TSDFVoxel {sdf 999, weight 0, color (-1,-1,-1)} (TSDFVoxel.h:79-81)
    const size_t total = n_blocks * kBlockFloats;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int plane = (int)((i % kBlockFloats) / kVox);
        pool[first_block * kBlockFloats + i] = plane == 0 ? 999.0f : (plane == 1 ? 0.0f : -1.0f);
    }
}

__global__ void k_clear_table(int4* table, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        table[i] = make_int4(kKeySentinel, kKeySentinel, kKeySentinel, kEmpty);
}

// ---------------------------------------------------------------------------------------------
// K1: ComputeBounding (CubeHandler.cpp:116-145): back-project, transform, frustum test, min/max.
// One partial result per workgroup (no atomics); K2 reduces the partials.
// partial layout: [max x,y,z, min x,y,z, inside (as uint bits), pad] per workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bounding(FrameParams P, const void* __restrict__ depth, float* __restrict__ partial,
                                                  FrameState* st) {
    __shared__ float s_red[4][6];
    __shared__ unsigned s_cnt[4];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) st->n_sel = 0; // new frame: empty cube_id_list
    float mx0 = -FLT_MAX, mx1 = -FLT_MAX, mx2 = -FLT_MAX, mn0 = FLT_MAX, mn1 = FLT_MAX, mn2 = FLT_MAX;
    unsigned inside = 0;
    const int npix = P.width * P.height;
#pragma unroll
    for (int r = 0; r < kPixPerWg / 256; ++r) {
        const int pix = blockIdx.x * kPixPerWg + r * 256 + tid;
        if (pix >= npix) continue;
        const float z = depth_at(depth, P.depth_u16, P.depth_scale, (size_t)pix);
        if (!(z > 0)) continue;
        const int i = pix / P.width, j = pix - i * P.width;
        const float x = ((float)j - P.cx) * z / P.fx; // PointCloud.cpp:90-93
        const float y = ((float)i - P.cy) * z / P.fy;
        const float* M = P.pose;                       // Geometry.cpp:19-27
        const float q0 = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * 1.0f;
        const float q1 = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * 1.0f;
        const float q2 = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * 1.0f;
        const float q3 = ((M[12] * x + M[13] * y) + M[14] * z) + M[15] * 1.0f;
        const float p0 = q0 / q3, p1 = q1 / q3, p2 = q2 / q3;
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
    mx0 = wave_max(mx0); mx1 = wave_max(mx1); mx2 = wave_max(mx2);
    mn0 = wave_min(mn0); mn1 = wave_min(mn1); mn2 = wave_min(mn2);
    inside = wave_sum(inside);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        s_red[wave][0] = mx0; s_red[wave][1] = mx1; s_red[wave][2] = mx2;
        s_red[wave][3] = mn0; s_red[wave][4] = mn1; s_red[wave][5] = mn2;
        s_cnt[wave] = inside;
    }
    __syncthreads();
    if (tid < 6) {
        float v = s_red[0][tid];
        for (int w = 1; w < 4; ++w) v = tid < 3 ? fmaxf(v, s_red[w][tid]) : fminf(v, s_red[w][tid]);
        partial[blockIdx.x * 8 + tid] = v;
    } else if (tid == 6) {
        ((unsigned*)partial)[blockIdx.x * 8 + 6] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

// ---------------------------------------------------------------------------------------------
// K2: PrepareCubes (CubeHandler.cpp:147-196).  One thread per candidate block of the bbox +-1
// range; 8 corner-voxel GetSDF probes; selected blocks are looked up / inserted in the hash table
// and appended to the frame list.  List append and pool allocation are aggregated per workgroup
// chunk (one atomic each per 256 candidates).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_select(FrameParams P, VolView V, const void* __restrict__ depth,
                                                const float* __restrict__ partial, int n_partial, FrameState* st) {
    __shared__ float s_red[4][6];
    __shared__ unsigned s_cnt[4];
    __shared__ int s_range[6]; // i0, j0, k0, ni, nj, nk
    __shared__ unsigned s_wsel[4], s_wnew[4], s_base[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // -- finish ComputeBounding: reduce the K1 partials (every workgroup does it redundantly)
    float mx0 = -FLT_MAX, mx1 = -FLT_MAX, mx2 = -FLT_MAX, mn0 = FLT_MAX, mn1 = FLT_MAX, mn2 = FLT_MAX;
    unsigned inside = 0;
    for (int g = tid; g < n_partial; g += 256) {
        const float* p = partial + g * 8;
        mx0 = fmaxf(mx0, p[0]); mx1 = fmaxf(mx1, p[1]); mx2 = fmaxf(mx2, p[2]);
        mn0 = fminf(mn0, p[3]); mn1 = fminf(mn1, p[4]); mn2 = fminf(mn2, p[5]);
        inside += ((const unsigned*)p)[6];
    }
    mx0 = wave_max(mx0); mx1 = wave_max(mx1); mx2 = wave_max(mx2);
    mn0 = wave_min(mn0); mn1 = wave_min(mn1); mn2 = wave_min(mn2);
    inside = wave_sum(inside);
    if (lane == 0) {
        s_red[wave][0] = mx0; s_red[wave][1] = mx1; s_red[wave][2] = mx2;
        s_red[wave][3] = mn0; s_red[wave][4] = mn1; s_red[wave][5] = mn2;
        s_cnt[wave] = inside;
    }
    __syncthreads();
    if (tid == 0) {
        float b[6];
        for (int c = 0; c < 6; ++c) {
            float v = s_red[0][c];
            for (int w = 1; w < 4; ++w) v = c < 3 ? fmaxf(v, s_red[w][c]) : fminf(v, s_red[w][c]);
            b[c] = v;
        }
        const unsigned tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (tot == 0) {
            for (int c = 0; c < 6; ++c) s_range[c] = 0;
        } else {
            for (int c = 0; c < 3; ++c) {
                // GetCubeID (VoxelCube.h:63-74): floor(p/res) in float -> int, then
                // floor((pb + 0.0)/8) in double == arithmetic shift by 3.
                const int hi = ((int)floorf(b[c] / P.res)) >> 3;
                const int lo = ((int)floorf(b[3 + c] / P.res)) >> 3;
                s_range[c] = lo - 1;
                s_range[3 + c] = hi - lo + 3;
            }
        }
        if (blockIdx.x == 0) {
            for (int c = 0; c < 6; ++c) st->bbox[c] = b[c];
            st->n_inside = tot;
        }
    }
    __syncthreads();
    const int i0 = s_range[0], j0 = s_range[1], k0 = s_range[2];
    const long long ni = s_range[3], nj = s_range[4], nk = s_range[5];
    unsigned long long ncand = (unsigned long long)(ni * nj * nk);
    if (ni > 4096 || nj > 4096 || nk > 4096) { // > 160 m at 5 mm: treat as a bad frame, select nothing
        if (blockIdx.x == 0 && tid == 0) atomicOr(&st->overflow, 4u);
        ncand = 0;
    }
    if (blockIdx.x == 0 && tid == 0) st->n_cand = ncand;

    const float cube_res = P.res * 8.0f; // CubeHandler.cpp:164
    const float half = P.res / 2;        // VoxelCube.h:47
    const float o_lo = 0.0f * P.res + half, o_hi = 7.0f * P.res + half; // VoxelCentroidOffSet of x = 0 / 7

    for (unsigned long long chunk = blockIdx.x; chunk * 256ULL < ncand; chunk += gridDim.x) {
        const unsigned long long c = chunk * 256ULL + tid;
        bool sel = false, is_new = false;
        int bi = 0, bj = 0, bk = 0, pool_idx = -1, slot = -1;
        if (c < ncand) {
            bk = k0 + (int)(c % (unsigned long long)nk);
            bj = j0 + (int)((c / (unsigned long long)nk) % (unsigned long long)nj);
            bi = i0 + (int)(c / (unsigned long long)(nk * nj));
            const float bx = (float)bi * cube_res, by = (float)bj * cube_res, bz = (float)bk * cube_res;
            float min_sdf = FLT_MAX;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) { // voxel ids {0,7,56,63,448,455,504,511}
                const float px = bx + ((corner & 1) ? o_hi : o_lo);
                const float py = by + ((corner & 2) ? o_hi : o_lo);
                const float pz = bz + ((corner & 4) ? o_hi : o_lo);
                const float a = fabsf(get_sdf(P, depth, px, py, pz));
                if (min_sdf > a) min_sdf = a;
            }
            sel = min_sdf < P.trunc;
            if (sel) {
                pool_idx = table_find_or_reserve(V, bi, bj, bk, &slot);
                is_new = pool_idx < 0;
                if (is_new && slot < 0) { atomicOr(&st->overflow, 2u); sel = false; is_new = false; }
            }
        }
        // workgroup-aggregated list append / pool allocation
        const unsigned long long m_sel = __ballot(sel), m_new = __ballot(is_new);
        const unsigned long long below = (1ULL << lane) - 1ULL;
        unsigned r_sel = __popcll(m_sel & below), r_new = __popcll(m_new & below);
        if (lane == 0) { s_wsel[wave] = __popcll(m_sel); s_wnew[wave] = __popcll(m_new); }
        __syncthreads();
        if (tid == 0) {
            const unsigned tsel = s_wsel[0] + s_wsel[1] + s_wsel[2] + s_wsel[3];
            const unsigned tnew = s_wnew[0] + s_wnew[1] + s_wnew[2] + s_wnew[3];
            s_base[0] = tsel ? atomicAdd(&st->n_sel, tsel) : 0u;
            s_base[1] = tnew ? atomicAdd(V.n_blocks, tnew) : 0u;
        }
        __syncthreads();
        for (int w = 0; w < wave; ++w) { r_sel += s_wsel[w]; r_new += s_wnew[w]; }
        if (sel) {
            if (is_new) {
                const unsigned idx = s_base[1] + r_new;
                if (idx < V.max_blocks) {
                    pool_idx = (int)idx;
                    V.keys[3 * idx] = bi; V.keys[3 * idx + 1] = bj; V.keys[3 * idx + 2] = bk;
                    table_publish(V, slot, bi, bj, bk, pool_idx);
                } else {
                    atomicOr(&st->overflow, 1u);
                    atomicExch(&((int*)&V.table[slot])[3], kDead);
                    pool_idx = -1;
                }
            }
            const unsigned pos = s_base[0] + r_sel;
            if (pos < V.max_blocks) {
                V.sel_list[pos] = pool_idx;
                V.sel_cand[pos] = c;
            }
        }
        __syncthreads(); // s_wsel / s_base are reused by the next chunk
    }
}

// ---------------------------------------------------------------------------------------------
// K3: Integrator::IntegrateImage (Integrator.cpp:36-94) over the frame list.  One 512-thread
// workgroup per block (thread = voxel, wave = z-slice), persistent grid striding over the list.
// Block ownership is exclusive, so the read-modify-write needs no atomics.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_integrate(FrameParams P, VolView V, const void* __restrict__ depth,
                                                   const unsigned char* __restrict__ rgb, FrameState* st,
                                                   unsigned long long* __restrict__ upd_partial) {
    __shared__ unsigned s_upd[8];
    const unsigned n = st->n_sel < V.max_blocks ? st->n_sel : V.max_blocks;
    const int vid = threadIdx.x;
    const float half = P.res / 2;
    // VoxelCentroidOffSet[vid] (VoxelCube.h:48-61): x*res + half with x = vid & 7 etc.
    const float ox = (float)(vid & 7) * P.res + half;
    const float oy = (float)((vid >> 3) & 7) * P.res + half;
    const float oz = (float)(vid >> 6) * P.res + half;
    const float* M = P.pose_inv;
    unsigned upd = 0;
    for (unsigned b = blockIdx.x; b < n; b += gridDim.x) {
        const int idx = V.sel_list[b];
        if (idx < 0) continue; // pool overflow (reported through st->overflow)
        const int kx = V.keys[3 * idx], ky = V.keys[3 * idx + 1], kz = V.keys[3 * idx + 2];
        // GetGlobalPoint (VoxelCube.h:75-80): Point3(id) * CUBE_SIZE * VoxelResolution + offset
        const float px = ((float)kx * 8.0f) * P.res + ox;
        const float py = ((float)ky * 8.0f) * P.res + oy;
        const float pz = ((float)kz * 8.0f) * P.res + oz;
        const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
        const float q1 = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
        const float q2 = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
        const int u = project_px(P.fx, q0, q2, P.cx);
        const int v = project_px(P.fy, q1, q2, P.cy);
        if (v < 0 || v >= P.height || u < 0 || u >= P.width) continue;
        const size_t pix = (size_t)v * P.width + u;
        const float d = depth_at(depth, P.depth_u16, P.depth_scale, pix);
        if (d <= 0) continue;
        const float new_sdf = d - q2;
        if (fabsf(new_sdf) < P.trunc) {
            ++upd;
            const float c0 = (float)rgb[3 * pix] / 255.0f, c1 = (float)rgb[3 * pix + 1] / 255.0f,
                        c2 = (float)rgb[3 * pix + 2] / 255.0f;
            float* vox = V.pool + (size_t)idx * kBlockFloats + vid;
            const float s = vox[0], w = vox[kVox];
            if (!(s >= 1 || w <= 0)) { // TSDFVoxel::IsValid (TSDFVoxel.h:75-78)
                // TSDFVoxel::operator+ with other = (new_sdf, 1.0, c) (TSDFVoxel.h:24-39)
                const float wsum = w + 1.0f;
                const float o0 = vox[2 * kVox], o1 = vox[3 * kVox], o2 = vox[4 * kVox];
                vox[0] = (w * s + 1.0f * new_sdf) / wsum;
                vox[kVox] = wsum;
                vox[2 * kVox] = (w * o0 + 1.0f * c0) / wsum;
                vox[3 * kVox] = (w * o1 + 1.0f * c1) / wsum;
                vox[4 * kVox] = (w * o2 + 1.0f * c2) / wsum;
            } else {
                vox[0] = new_sdf; vox[kVox] = 1.0f;
                vox[2 * kVox] = c0; vox[3 * kVox] = c1; vox[4 * kVox] = c2;
            }
        }
    }
    // per-workgroup update counter (each workgroup owns its slot: no atomics)
    upd = wave_sum(upd);
    if ((vid & 63) == 0) s_upd[vid >> 6] = upd;
    __syncthreads();
    if (vid == 0) {
        unsigned t = 0;
        for (int w = 0; w < 8; ++w) t += s_upd[w];
        upd_partial[blockIdx.x] += t;
        if (blockIdx.x == 0) { st->stat_frames += 1; st->stat_sel += n; }
    }
}

// ---------------------------------------------------------------------------------------------
// export / import / merge kernels
// ---------------------------------------------------------------------------------------------
// SoA pool -> AoS {sdf,w,c0,c1,c2} x 512 for blocks [first, first+count)
__global__ __launch_bounds__(512) void k_export_aos(const float* __restrict__ pool, size_t first, float* __restrict__ out) {
    const size_t b = blockIdx.x;
    const float* src = pool + (first + b) * kBlockFloats + threadIdx.x;
    float* dst = out + (b * kVox + threadIdx.x) * 5;
#pragma unroll
    for (int p = 0; p < 5; ++p) dst[p] = src[p * kVox];
}

// insert (distinct) keys; slots[i] receives the pool slot of key i (existing or new)
__global__ void k_insert_keys(VolView V, const int* __restrict__ keys, size_t n, int* __restrict__ slots, FrameState* st) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
    int slot;
    int idx = table_find_or_reserve(V, x, y, z, &slot);
    if (idx < 0) {
        if (slot < 0) { atomicOr(&st->overflow, 2u); slots[i] = -1; return; }
        const unsigned nb = atomicAdd(V.n_blocks, 1u);
        if (nb >= V.max_blocks) {
            atomicOr(&st->overflow, 1u);
            atomicExch(&((int*)&V.table[slot])[3], kDead);
            slots[i] = -1;
            return;
        }
        idx = (int)nb;
        V.keys[3 * idx] = x; V.keys[3 * idx + 1] = y; V.keys[3 * idx + 2] = z;
        table_publish(V, slot, x, y, z, idx);
        slots[i] = -(idx + 2); // encoded "newly created": <= -2
        return;
    }
    slots[i] = idx;
}

// AoS voxels -> pool planes for the given slots (SetCubeMap / AddCube + assignment)
__global__ __launch_bounds__(512) void k_import_aos(float* __restrict__ pool, const int* __restrict__ slots,
                                                    const float* __restrict__ in) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    if (idx <= -2) idx = -(idx + 2);
    const float* src = in + ((size_t)blockIdx.x * kVox + threadIdx.x) * 5;
    float* dst = pool + (size_t)idx * kBlockFloats + threadIdx.x;
#pragma unroll
    for (int p = 0; p < 5; ++p) dst[p * kVox] = src[p];
}

// CubeHandler::Merge (CubeHandler.h:145-167): dst block (slots) += src block (TSDFVoxel::operator+,
// general weights), or plain copy when the block was just created in dst.
__global__ __launch_bounds__(512) void k_merge_blocks(float* __restrict__ dpool, const float* __restrict__ spool,
                                                      const int* __restrict__ slots) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    const bool fresh = idx <= -2;
    if (fresh) idx = -(idx + 2);
    const float* a = spool + (size_t)blockIdx.x * kBlockFloats + threadIdx.x; // src block i lives in src pool slot i
    float* t = dpool + (size_t)idx * kBlockFloats + threadIdx.x;
    const float bs = a[0], bw = a[kVox], b0 = a[2 * kVox], b1 = a[3 * kVox], b2 = a[4 * kVox];
    const float tw = t[kVox];
    if (fresh || tw == 0) { // copy / "weight == 0 -> return other"
        t[0] = bs; t[kVox] = bw; t[2 * kVox] = b0; t[3 * kVox] = b1; t[4 * kVox] = b2;
        return;
    }
    if (bw == 0) return;
    const float w = tw + bw;
    if (w != 0) {
        const float ts = t[0], t0 = t[2 * kVox], t1 = t[3 * kVox], t2 = t[4 * kVox];
        t[0] = (tw * ts + bw * bs) / w;
        t[2 * kVox] = (tw * t0 + bw * b0) / w;
        t[3 * kVox] = (tw * t1 + bw * b1) / w;
        t[4 * kVox] = (tw * t2 + bw * b2) / w;
    } else {
        t[0] = 999.0f; t[2 * kVox] = t[3 * kVox] = t[4 * kVox] = -1.0f;
    }
    t[kVox] = w;
}

// K4a: sum-form pack for the RCCL reduce: [w*sdf, w, w*c0, w*c1, w*c2] planes per union key.
__global__ __launch_bounds__(512) void k_pack_sum(VolView V, const int* __restrict__ ukeys, float* __restrict__ out) {
    __shared__ int s_idx;
    if (threadIdx.x == 0) s_idx = table_find(V, ukeys[3 * blockIdx.x], ukeys[3 * blockIdx.x + 1], ukeys[3 * blockIdx.x + 2]);
    __syncthreads();
    const int idx = s_idx;
    float* o = out + (size_t)blockIdx.x * kBlockFloats + threadIdx.x;
    float s = 0, w = 0, c0 = 0, c1 = 0, c2 = 0;
    if (idx >= 0) {
        const float* t = V.pool + (size_t)idx * kBlockFloats + threadIdx.x;
        w = t[kVox];
        if (w > 0) { s = w * t[0]; c0 = w * t[2 * kVox]; c1 = w * t[3 * kVox]; c2 = w * t[4 * kVox]; }
        else w = 0;
    }
    o[0] = s; o[kVox] = w; o[2 * kVox] = c0; o[3 * kVox] = c1; o[4 * kVox] = c2;
}

// K4b: normalise the reduced sums back to mean form into the (re-keyed) volume.
__global__ __launch_bounds__(512) void k_unpack_sum(float* __restrict__ pool, const int* __restrict__ slots,
                                                    const float* __restrict__ sum) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    if (idx <= -2) idx = -(idx + 2);
    const float* a = sum + (size_t)blockIdx.x * kBlockFloats + threadIdx.x;
    float* t = pool + (size_t)idx * kBlockFloats + threadIdx.x;
    const float w = a[kVox];
    if (w > 0) {
        t[0] = a[0] / w; t[kVox] = w; t[2 * kVox] = a[2 * kVox] / w; t[3 * kVox] = a[3 * kVox] / w; t[4 * kVox] = a[4 * kVox] / w;
    } else {
        t[0] = 999.0f; t[kVox] = 0.0f; t[2 * kVox] = t[3 * kVox] = t[4 * kVox] = -1.0f;
    }
}

unsigned next_pow2(unsigned long long v) {
    unsigned long long p = 1;
    while (p < v) p <<= 1;
    return (unsigned)p;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// op_volume: host object
// ---------------------------------------------------------------------------------------------
struct op_volume {
    int device = 0;
    hipStream_t stream = nullptr;
    op_camera cam{};
    float res = 0.01f, trunc = 0.1f, far_d = 5.0f, near_d = 0.5f;
    unsigned max_blocks = 0;
    unsigned table_size = 0;
    // device memory
    int4* table = nullptr;
    int* keys = nullptr;
    float* pool = nullptr;
    unsigned* n_blocks = nullptr;
    int* sel_list = nullptr;
    unsigned long long* sel_cand = nullptr;
    FrameState* state = nullptr;
    float* partial = nullptr;
    int n_partial_cap = 0;
    unsigned long long* upd_partial = nullptr;
    // optional per-kernel HIP-event timing (op_volume_profile_*): every `prof_every`-th frame gets
    // four events on the volume's stream (before K1, after K1, after K2, after K3)
    int prof_every = 0;
    uint64_t prof_frame = 0;
    std::vector<hipEvent_t> prof_events; // 4 per sampled frame
    void* img_depth = nullptr; // staging for host images
    unsigned char* img_rgb = nullptr;
    size_t img_cap_px = 0;

    VolView view() const {
        VolView V;
        V.table = table; V.table_mask = table_size - 1; V.keys = keys; V.pool = pool;
        V.max_blocks = max_blocks; V.n_blocks = n_blocks; V.sel_list = sel_list; V.sel_cand = sel_cand;
        return V;
    }
};

namespace {

int vol_reset(op_volume* v) {
    hipLaunchKernelGGL(k_clear_table, dim3(1024), dim3(256), 0, v->stream, v->table, (size_t)v->table_size);
    OP_HIP(hipMemsetAsync(v->n_blocks, 0, sizeof(unsigned), v->stream));
    OP_HIP(hipMemsetAsync(v->state, 0, sizeof(FrameState), v->stream));
    OP_HIP(hipMemsetAsync(v->upd_partial, 0, sizeof(unsigned long long) * kIntegrateGrid, v->stream));
    OP_HIP(hipGetLastError());
    return OP_OK;
}

// Raises OP_ERR_CAPACITY if a previous kernel flagged an overflow.  Synchronises.
int vol_check(op_volume* v) {
    OP_HIP(hipStreamSynchronize(v->stream));
    unsigned of = 0;
    OP_HIP(hipMemcpy(&of, &v->state->overflow, sizeof(of), hipMemcpyDeviceToHost));
    if (of & 1u) return fail(OP_ERR_CAPACITY, "block pool exhausted (max_blocks = %u); create the volume with a larger max_blocks", v->max_blocks);
    if (of & 2u) return fail(OP_ERR_CAPACITY, "hash table exhausted (size %u)", v->table_size);
    if (of & 4u) return fail(OP_ERR_INVALID, "frame bounding box spans more than 4096 blocks on an axis");
    return OP_OK;
}

int vol_block_count(op_volume* v, unsigned* n) {
    OP_TRY(vol_check(v));
    OP_HIP(hipMemcpy(n, v->n_blocks, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (*n > v->max_blocks) *n = v->max_blocks;
    return OP_OK;
}

int vol_stage_images(op_volume* v, const void** depth, int depth_fmt, const unsigned char** rgb, int mem) {
    if (mem == OP_MEM_DEVICE) return OP_OK;
    const size_t npx = (size_t)v->cam.width * v->cam.height;
    if (npx > v->img_cap_px) {
        if (v->img_depth) OP_HIP(hipFree(v->img_depth));
        if (v->img_rgb) OP_HIP(hipFree(v->img_rgb));
        OP_HIP(hipMalloc(&v->img_depth, npx * 4));
        OP_HIP(hipMalloc((void**)&v->img_rgb, npx * 3));
        v->img_cap_px = npx;
    }
    OP_HIP(hipMemcpyAsync(v->img_depth, *depth, npx * (depth_fmt == OP_DEPTH_U16 ? 2 : 4), hipMemcpyHostToDevice, v->stream));
    *depth = v->img_depth;
    if (rgb && *rgb) {
        OP_HIP(hipMemcpyAsync(v->img_rgb, *rgb, npx * 3, hipMemcpyHostToDevice, v->stream));
        *rgb = v->img_rgb;
    }
    return OP_OK;
}

void make_params(const op_volume* v, const float pose[16], const float* pose_inv, int depth_fmt, FrameParams* P) {
    std::memcpy(P->pose, pose, sizeof(P->pose));
    if (pose_inv) std::memcpy(P->pose_inv, pose_inv, sizeof(P->pose_inv));
    else op_host::mat4_inverse(pose, P->pose_inv);
    op_host::CameraPOD c{v->cam.fx, v->cam.fy, v->cam.cx, v->cam.cy, v->cam.width, v->cam.height, v->cam.depth_scale};
    op_host::frustum_planes(c, pose, v->far_d, v->near_d, P->planes);
    P->fx = v->cam.fx; P->fy = v->cam.fy; P->cx = v->cam.cx; P->cy = v->cam.cy;
    P->depth_scale = v->cam.depth_scale; P->res = v->res; P->trunc = v->trunc;
    P->width = v->cam.width; P->height = v->cam.height; P->depth_u16 = depth_fmt == OP_DEPTH_U16;
}

int vol_ensure_partials(op_volume* v, int n) {
    if (n <= v->n_partial_cap) return OP_OK;
    if (v->partial) OP_HIP(hipFree(v->partial));
    OP_HIP(hipMalloc((void**)&v->partial, (size_t)n * 8 * sizeof(float)));
    v->n_partial_cap = n;
    return OP_OK;
}

// enqueue K1 + K2 (+ K3) for one frame whose images are already on the device
int vol_enqueue_frame(op_volume* v, const FrameParams& P, const void* d_depth, const unsigned char* d_rgb, bool integrate) {
    const int npix = P.width * P.height;
    const int g1 = (npix + kPixPerWg - 1) / kPixPerWg;
    OP_TRY(vol_ensure_partials(v, g1));
    const VolView V = v->view();
    const bool sample = integrate && v->prof_every > 0 && (v->prof_frame++ % (uint64_t)v->prof_every) == 0 &&
                        v->prof_events.size() < 4 * 65536;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (sample) {
        for (auto& e : ev) OP_HIP(hipEventCreate(&e));
        OP_HIP(hipEventRecord(ev[0], v->stream));
    }
    hipLaunchKernelGGL(k_bounding, dim3(g1), dim3(256), 0, v->stream, P, d_depth, v->partial, v->state);
    if (sample) OP_HIP(hipEventRecord(ev[1], v->stream));
    hipLaunchKernelGGL(k_select, dim3(kSelectGrid), dim3(256), 0, v->stream, P, V, d_depth, (const float*)v->partial, g1, v->state);
    if (sample) OP_HIP(hipEventRecord(ev[2], v->stream));
    if (integrate)
        hipLaunchKernelGGL(k_integrate, dim3(kIntegrateGrid), dim3(512), 0, v->stream, P, V, d_depth, d_rgb, v->state, v->upd_partial);
    if (sample) {
        OP_HIP(hipEventRecord(ev[3], v->stream));
        for (auto e : ev) v->prof_events.push_back(e);
    }
    OP_HIP(hipGetLastError());
    return OP_OK;
}

int check_cam(const op_camera* cam) {
    if (!cam || cam->width <= 0 || cam->height <= 0 || (long long)cam->width * cam->height > (1LL << 30))
        return fail(OP_ERR_INVALID, "invalid camera");
    return OP_OK;
}

} // namespace

extern "C" {

int op_abi_version(void) { return OP_ABI_VERSION; }
const char* op_last_error(void) { return op::g_last_error; }

int op_device_count(int* count) {
    if (!count) return fail(OP_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = e == hipSuccess ? n : 0;
    return OP_OK;
}

int op_camera_preset(int type, op_camera* out) {
    if (!out) return fail(OP_ERR_INVALID, "null camera");
    if (type == 0) *out = op_camera{517.3f, 516.5f, 318.6f, 255.3f, 640, 480, 5000.0f};          // Camera.h:78-92
    else if (type == 1) *out = op_camera{514.817f, 515.375f, 318.771f, 238.447f, 640, 480, 1000.0f}; // Camera.h:94-104
    else return fail(OP_ERR_INVALID, "unknown camera preset %d", type);
    return OP_OK;
}

int op_mat4_inverse(const float m[16], float out[16]) {
    if (!m || !out) return fail(OP_ERR_INVALID, "null matrix");
    op_host::mat4_inverse(m, out);
    return OP_OK;
}

uint64_t op_hash_key(int32_t x, int32_t y, int32_t z) { return op_host::hash_key(x, y, z); }

int op_frustum_planes(const op_camera* cam, const float pose[16], float far_dist, float near_dist, float planes[24]) {
    OP_TRY(check_cam(cam));
    if (!pose || !planes) return fail(OP_ERR_INVALID, "null argument");
    op_host::CameraPOD c{cam->fx, cam->fy, cam->cx, cam->cy, cam->width, cam->height, cam->depth_scale};
    op_host::frustum_planes(c, pose, far_dist, near_dist, planes);
    return OP_OK;
}

int op_se3_exp(const float x[6], float T[16]) {
    if (!x || !T) return fail(OP_ERR_INVALID, "null argument");
    op_host::se3_exp(x, T);
    return OP_OK;
}

int op_volume_create(const op_camera* cam, float voxel_res, float truncation, float far_dist, float near_dist, int device,
                     uint64_t max_blocks, op_volume** out) {
    if (!out) return fail(OP_ERR_INVALID, "null out");
    *out = nullptr;
    OP_TRY(check_cam(cam));
    if (!(voxel_res > 0) || !(truncation > 0)) return fail(OP_ERR_INVALID, "voxel_res and truncation must be > 0");
    OP_TRY(op::use_device(device));
    if (max_blocks == 0) max_blocks = 1u << 18;
    if (max_blocks > (1ull << 27)) return fail(OP_ERR_INVALID, "max_blocks too large");
    op_volume* v = new op_volume();
    v->device = device; v->cam = *cam; v->res = voxel_res; v->trunc = truncation; v->far_d = far_dist; v->near_d = near_dist;
    v->max_blocks = (unsigned)max_blocks;
    v->table_size = next_pow2(2ull * max_blocks);
    auto cleanup = [&](int rc) { op_volume_destroy(v); return rc; };
#define OP_HIP_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cleanup(fail(OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
    OP_HIP_C(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    OP_HIP_C(hipMalloc((void**)&v->table, sizeof(int4) * (size_t)v->table_size));
    OP_HIP_C(hipMalloc((void**)&v->keys, sizeof(int) * 3 * (size_t)v->max_blocks));
    OP_HIP_C(hipMalloc((void**)&v->pool, sizeof(float) * kBlockFloats * (size_t)v->max_blocks));
    OP_HIP_C(hipMalloc((void**)&v->n_blocks, sizeof(unsigned)));
    OP_HIP_C(hipMalloc((void**)&v->sel_list, sizeof(int) * (size_t)v->max_blocks));
    OP_HIP_C(hipMalloc((void**)&v->sel_cand, sizeof(unsigned long long) * (size_t)v->max_blocks));
    OP_HIP_C(hipMalloc((void**)&v->state, sizeof(FrameState)));
    OP_HIP_C(hipMalloc((void**)&v->upd_partial, sizeof(unsigned long long) * kIntegrateGrid));
#undef OP_HIP_C
    hipLaunchKernelGGL(k_fill_pool, dim3(4096), dim3(256), 0, v->stream, v->pool, (size_t)0, (size_t)v->max_blocks);
    int rc = vol_reset(v);
    if (rc != OP_OK) return cleanup(rc);
    if (hipStreamSynchronize(v->stream) != hipSuccess) return cleanup(fail(OP_ERR_HIP, "volume initialisation failed"));
    *out = v;
    return OP_OK;
}

int op_volume_destroy(op_volume* v) {
    if (!v) return OP_OK;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    for (auto e : v->prof_events) (void)hipEventDestroy(e);
    void* ptrs[] = {v->table, v->keys, v->pool, v->n_blocks, v->sel_list, v->sel_cand, v->state, v->partial,
                    v->upd_partial, v->img_depth, v->img_rgb};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
    return OP_OK;
}

#define OP_VOL(v)                                              \
    if (!(v)) return fail(OP_ERR_INVALID, "null volume");      \
    OP_HIP(hipSetDevice((v)->device))

int op_volume_set_resolution(op_volume* v, float voxel_res) {
    OP_VOL(v);
    if (!(voxel_res > 0)) return fail(OP_ERR_INVALID, "voxel_res must be > 0");
    v->res = voxel_res;
    return OP_OK;
}
int op_volume_set_truncation(op_volume* v, float truncation) {
    OP_VOL(v);
    v->trunc = truncation;
    return OP_OK;
}
int op_volume_set_camera(op_volume* v, const op_camera* cam) {
    OP_VOL(v);
    OP_TRY(check_cam(cam));
    v->cam = *cam;
    return OP_OK;
}
int op_volume_set_near_far(op_volume* v, float near_dist, float far_dist) {
    OP_VOL(v);
    v->near_d = near_dist; v->far_d = far_dist;
    return OP_OK;
}

int op_volume_clear(op_volume* v) {
    OP_VOL(v);
    unsigned n = 0;
    OP_HIP(hipStreamSynchronize(v->stream));
    OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
    if (n > v->max_blocks) n = v->max_blocks;
    if (n) hipLaunchKernelGGL(k_fill_pool, dim3(4096), dim3(256), 0, v->stream, v->pool, (size_t)0, (size_t)n);
    OP_TRY(vol_reset(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    return OP_OK;
}

int op_volume_sync(op_volume* v) {
    OP_VOL(v);
    return vol_check(v);
}

int op_volume_stream(op_volume* v, void** stream) {
    OP_VOL(v);
    if (!stream) return fail(OP_ERR_INVALID, "null stream out");
    *stream = (void*)v->stream;
    return OP_OK;
}

int op_volume_block_count(op_volume* v, size_t* n) {
    OP_VOL(v);
    if (!n) return fail(OP_ERR_INVALID, "null n");
    unsigned c = 0;
    OP_TRY(vol_block_count(v, &c));
    *n = c;
    return OP_OK;
}

int op_volume_compute_bounding(op_volume* v, const void* depth, int depth_fmt, int mem, const float pose[16], float max_pos[3],
                               float min_pos[3], size_t* n_inside) {
    OP_VOL(v);
    if (!depth || !pose) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, nullptr, mem));
    FrameParams P;
    make_params(v, pose, nullptr, depth_fmt, &P);
    const int npix = P.width * P.height, g1 = (npix + kPixPerWg - 1) / kPixPerWg;
    OP_TRY(vol_ensure_partials(v, g1));
    hipLaunchKernelGGL(k_bounding, dim3(g1), dim3(256), 0, v->stream, P, depth, v->partial, v->state);
    OP_HIP(hipGetLastError());
    OP_HIP(hipStreamSynchronize(v->stream));
    std::vector<float> part((size_t)g1 * 8);
    OP_HIP(hipMemcpy(part.data(), v->partial, part.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    size_t inside = 0;
    for (int g = 0; g < g1; ++g) {
        for (int c = 0; c < 3; ++c) {
            mx[c] = std::max(mx[c], part[g * 8 + c]);
            mn[c] = std::min(mn[c], part[g * 8 + 3 + c]);
        }
        unsigned cnt;
        std::memcpy(&cnt, &part[g * 8 + 6], 4);
        inside += cnt;
    }
    if (max_pos) std::memcpy(max_pos, mx, sizeof(mx));
    if (min_pos) std::memcpy(min_pos, mn, sizeof(mn));
    if (n_inside) *n_inside = inside;
    return OP_OK;
}

int op_volume_prepare_cubes(op_volume* v, const void* depth, int depth_fmt, int mem, const float pose[16], const float* pose_inv,
                            int32_t* ids_xyz, size_t cap, size_t* n, size_t* n_candidates) {
    OP_VOL(v);
    if (!depth || !pose) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, nullptr, mem));
    FrameParams P;
    make_params(v, pose, pose_inv, depth_fmt, &P);
    OP_TRY(vol_enqueue_frame(v, P, depth, nullptr, false));
    OP_TRY(vol_check(v));
    FrameState st;
    OP_HIP(hipMemcpy(&st, v->state, sizeof(st), hipMemcpyDeviceToHost));
    const size_t ns = st.n_sel;
    if (n) *n = ns;
    if (n_candidates) *n_candidates = (size_t)st.n_cand;
    if (ids_xyz && ns) {
        std::vector<int> list(ns);
        std::vector<unsigned long long> cand(ns);
        OP_HIP(hipMemcpy(list.data(), v->sel_list, ns * sizeof(int), hipMemcpyDeviceToHost));
        OP_HIP(hipMemcpy(cand.data(), v->sel_cand, ns * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned nb = 0;
        OP_HIP(hipMemcpy(&nb, v->n_blocks, sizeof(nb), hipMemcpyDeviceToHost));
        std::vector<int> keys((size_t)nb * 3);
        OP_HIP(hipMemcpy(keys.data(), v->keys, keys.size() * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<size_t> order(ns);
        std::iota(order.begin(), order.end(), (size_t)0);
        // candidate rank == position in the reference's i,j,k loop nest -> cube_id_list order
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cand[a] < cand[b]; });
        for (size_t i = 0; i < ns && i < cap; ++i) {
            const int idx = list[order[i]];
            for (int c = 0; c < 3; ++c) ids_xyz[3 * i + c] = keys[(size_t)idx * 3 + c];
        }
    }
    return OP_OK;
}

int op_volume_integrate(op_volume* v, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, const float pose[16],
                        const float* pose_inv) {
    OP_VOL(v);
    if (!depth || !rgb || !pose) return fail(OP_ERR_INVALID, "null argument");
    const unsigned char* c = rgb;
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, &c, mem));
    FrameParams P;
    make_params(v, pose, pose_inv, depth_fmt, &P);
    return vol_enqueue_frame(v, P, depth, c, true);
}

int op_volume_integrate_sequence(op_volume* v, const void* depth, size_t depth_stride_bytes, int depth_fmt, const uint8_t* rgb,
                                 size_t rgb_stride_bytes, const float* poses, size_t n_frames) {
    OP_VOL(v);
    if (!depth || !rgb || !poses) return fail(OP_ERR_INVALID, "null argument");
    FrameParams P;
    for (size_t f = 0; f < n_frames; ++f) {
        make_params(v, poses + 16 * f, nullptr, depth_fmt, &P);
        OP_TRY(vol_enqueue_frame(v, P, (const char*)depth + f * depth_stride_bytes, rgb + f * rgb_stride_bytes, true));
    }
    return OP_OK;
}

int op_volume_stats(op_volume* v, uint64_t* frames, uint64_t* blocks_selected, uint64_t* voxels_visited, uint64_t* voxels_updated) {
    OP_VOL(v);
    OP_TRY(vol_check(v));
    FrameState st;
    OP_HIP(hipMemcpy(&st, v->state, sizeof(st), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> part(kIntegrateGrid);
    OP_HIP(hipMemcpy(part.data(), v->upd_partial, part.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long upd = 0;
    for (auto p : part) upd += p;
    if (frames) *frames = st.stat_frames;
    if (blocks_selected) *blocks_selected = st.stat_sel;
    if (voxels_visited) *voxels_visited = st.stat_sel * (uint64_t)kVox;
    if (voxels_updated) *voxels_updated = upd;
    return OP_OK;
}

int op_volume_profile_enable(op_volume* v, int sample_every) {
    OP_VOL(v);
    OP_HIP(hipStreamSynchronize(v->stream));
    for (auto e : v->prof_events) (void)hipEventDestroy(e);
    v->prof_events.clear();
    v->prof_every = sample_every > 0 ? sample_every : 0;
    v->prof_frame = 0;
    return OP_OK;
}

int op_volume_profile_read(op_volume* v, double ms_sum[3], uint64_t* n_samples) {
    OP_VOL(v);
    if (!ms_sum || !n_samples) return fail(OP_ERR_INVALID, "null argument");
    OP_HIP(hipStreamSynchronize(v->stream));
    ms_sum[0] = ms_sum[1] = ms_sum[2] = 0.0;
    const size_t n = v->prof_events.size() / 4;
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            float ms = 0.0f;
            OP_HIP(hipEventElapsedTime(&ms, v->prof_events[4 * i + k], v->prof_events[4 * i + k + 1]));
            ms_sum[k] += ms;
        }
    *n_samples = n;
    return OP_OK;
}

int op_volume_has_cube(op_volume* v, int32_t x, int32_t y, int32_t z, int* present) {
    OP_VOL(v);
    if (!present) return fail(OP_ERR_INVALID, "null present");
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    std::vector<int> keys((size_t)nb * 3);
    if (nb) OP_HIP(hipMemcpy(keys.data(), v->keys, keys.size() * sizeof(int), hipMemcpyDeviceToHost));
    *present = 0;
    for (unsigned i = 0; i < nb; ++i)
        if (keys[3 * i] == x && keys[3 * i + 1] == y && keys[3 * i + 2] == z) { *present = 1; break; }
    return OP_OK;
}

int op_volume_download(op_volume* v, int32_t* keys_xyz, float* voxels_aos, size_t cap, size_t* n) {
    OP_VOL(v);
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    if (n) *n = nb;
    const size_t take = std::min((size_t)nb, cap);
    if (keys_xyz && take) OP_HIP(hipMemcpy(keys_xyz, v->keys, take * 3 * sizeof(int), hipMemcpyDeviceToHost));
    if (voxels_aos && take) {
        const size_t chunk = 8192; // 80 MiB of staging
        float* stage = nullptr;
        OP_HIP(hipMalloc((void**)&stage, std::min(chunk, take) * kBlockFloats * sizeof(float)));
        for (size_t first = 0; first < take; first += chunk) {
            const size_t cnt = std::min(chunk, take - first);
            hipLaunchKernelGGL(k_export_aos, dim3((unsigned)cnt), dim3(512), 0, v->stream, (const float*)v->pool, first, stage);
            hipError_t e = hipStreamSynchronize(v->stream);
            if (e == hipSuccess)
                e = hipMemcpy(voxels_aos + first * kBlockFloats, stage, cnt * kBlockFloats * sizeof(float), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { (void)hipFree(stage); return fail(OP_ERR_HIP, "download failed: %s", hipGetErrorString(e)); }
        }
        OP_HIP(hipFree(stage));
    }
    return OP_OK;
}

int op_volume_upload(op_volume* v, const int32_t* keys_xyz, const float* voxels_aos, size_t n) {
    OP_VOL(v);
    if (n == 0) return OP_OK;
    if (!keys_xyz || !voxels_aos) return fail(OP_ERR_INVALID, "null argument");
    // later duplicates override earlier ones, like repeated map assignment; the device insert needs distinct keys
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return std::lexicographical_compare(keys_xyz + 3 * a, keys_xyz + 3 * a + 3, keys_xyz + 3 * b, keys_xyz + 3 * b + 3);
    });
    std::vector<size_t> uniq;
    for (size_t i = 0; i < n; ++i) {
        const bool last = i + 1 == n || !std::equal(keys_xyz + 3 * order[i], keys_xyz + 3 * order[i] + 3, keys_xyz + 3 * order[i + 1]);
        if (last) uniq.push_back(order[i]);
    }
    const size_t chunk = 8192;
    int *d_keys = nullptr, *d_slots = nullptr;
    float* d_vox = nullptr;
    OP_HIP(hipMalloc((void**)&d_keys, chunk * 3 * sizeof(int)));
    OP_HIP(hipMalloc((void**)&d_slots, chunk * sizeof(int)));
    OP_HIP(hipMalloc((void**)&d_vox, chunk * kBlockFloats * sizeof(float)));
    std::vector<int> hk(chunk * 3);
    std::vector<float> hv(chunk * kBlockFloats);
    int rc = OP_OK;
    for (size_t first = 0; first < uniq.size() && rc == OP_OK; first += chunk) {
        const size_t cnt = std::min(chunk, uniq.size() - first);
        for (size_t i = 0; i < cnt; ++i) {
            std::memcpy(&hk[3 * i], keys_xyz + 3 * uniq[first + i], 3 * sizeof(int));
            std::memcpy(&hv[i * kBlockFloats], voxels_aos + uniq[first + i] * kBlockFloats, kBlockFloats * sizeof(float));
        }
        hipError_t e = hipMemcpy(d_keys, hk.data(), cnt * 3 * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_vox, hv.data(), cnt * kBlockFloats * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_insert_keys, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, v->stream, v->view(), (const int*)d_keys, cnt, d_slots, v->state);
            hipLaunchKernelGGL(k_import_aos, dim3((unsigned)cnt), dim3(512), 0, v->stream, v->pool, (const int*)d_slots, (const float*)d_vox);
            e = hipStreamSynchronize(v->stream);
        }
        if (e != hipSuccess) rc = fail(OP_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_keys); (void)hipFree(d_slots); (void)hipFree(d_vox);
    if (rc != OP_OK) return rc;
    return vol_check(v);
}

int op_volume_merge(op_volume* dst, op_volume* src) {
    OP_VOL(dst);
    if (!src) return fail(OP_ERR_INVALID, "null src");
    if (dst->device != src->device) return fail(OP_ERR_INVALID, "op_volume_merge needs both volumes on one device; use pack_sum/unpack_sum across devices");
    if (dst->res != src->res) // CubeHandler.h:147-151: warn and leave dst untouched
        return fail(OP_ERR_MISMATCH, "[Warning]::[MergeVoxelHash]::Voxel resolution is not identical.");
    if (dst == src) return fail(OP_ERR_INVALID, "cannot merge a volume into itself");
    unsigned ns = 0;
    OP_TRY(vol_block_count(src, &ns));
    OP_TRY(vol_check(dst));
    if (!ns) return OP_OK;
    int* d_slots = nullptr;
    OP_HIP(hipMalloc((void**)&d_slots, (size_t)ns * sizeof(int)));
    hipLaunchKernelGGL(k_insert_keys, dim3((ns + 255) / 256), dim3(256), 0, dst->stream, dst->view(), (const int*)src->keys, (size_t)ns, d_slots, dst->state);
    hipLaunchKernelGGL(k_merge_blocks, dim3(ns), dim3(512), 0, dst->stream, dst->pool, (const float*)src->pool, (const int*)d_slots);
    hipError_t e = hipStreamSynchronize(dst->stream);
    (void)hipFree(d_slots);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "merge failed: %s", hipGetErrorString(e));
    return vol_check(dst);
}

int op_volume_keys_device(op_volume* v, int32_t* d_keys, size_t cap, size_t* n) {
    OP_VOL(v);
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    if (n) *n = nb;
    const size_t take = std::min((size_t)nb, cap);
    if (d_keys && take) OP_HIP(hipMemcpy(d_keys, v->keys, take * 3 * sizeof(int), hipMemcpyDeviceToDevice));
    return OP_OK;
}

int op_volume_pack_sum(op_volume* v, const int32_t* d_union_keys, size_t n_union, float* d_out) {
    OP_VOL(v);
    if (n_union == 0) return OP_OK;
    if (!d_union_keys || !d_out) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_check(v));
    hipLaunchKernelGGL(k_pack_sum, dim3((unsigned)n_union), dim3(512), 0, v->stream, v->view(), (const int*)d_union_keys, d_out);
    OP_HIP(hipGetLastError());
    OP_HIP(hipStreamSynchronize(v->stream));
    return OP_OK;
}

int op_volume_unpack_sum(op_volume* v, const int32_t* d_union_keys, size_t n_union, const float* d_sum) {
    OP_VOL(v);
    OP_TRY(op_volume_clear(v));
    if (n_union == 0) return OP_OK;
    if (!d_union_keys || !d_sum) return fail(OP_ERR_INVALID, "null argument");
    if (n_union > v->max_blocks) return fail(OP_ERR_CAPACITY, "union of %zu blocks exceeds max_blocks %u", n_union, v->max_blocks);
    int* d_slots = nullptr;
    OP_HIP(hipMalloc((void**)&d_slots, n_union * sizeof(int)));
    hipLaunchKernelGGL(k_insert_keys, dim3((unsigned)((n_union + 255) / 256)), dim3(256), 0, v->stream, v->view(), (const int*)d_union_keys, n_union, d_slots, v->state);
    hipLaunchKernelGGL(k_unpack_sum, dim3((unsigned)n_union), dim3(512), 0, v->stream, v->pool, (const int*)d_slots, d_sum);
    hipError_t e = hipStreamSynchronize(v->stream);
    (void)hipFree(d_slots);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "unpack failed: %s", hipGetErrorString(e));
    return vol_check(v);
}

} // extern "C"
