// volume_ops.hip -- the volume operations next to fusion (volume_core.hpp lists the translation units); file:line under /root/reference/src:
//   CubeHandler::GetCubeMap / SetCubeMap / AddCube      Integration/CubeHandler.h:185-198,347-356   (k_export_aos, k_insert_keys, k_import_aos)
//   CubeHandler::Merge                                  Integration/CubeHandler.h:145-167           (k_merge_blocks; K4 k_pack_sum / k_unpack_sum for the multi-GPU form)
//   CubeHandler::Transform / TransformNearest           Integration/CubeHandler.h:199-338, VoxelCube.cpp:6-50
//   CubeHandler::GetPointCloud                          Integration/CubeHandler.cpp:45-69
//   CubeHandler::ExtractTriangleMesh / GenerateMeshByCube + MarchingCube   Integration/CubeHandler.cpp:9-114, MarchingCube.cpp:8-74
//   CubeHandler::WriteToFile / ReadFromFile / ReadFromFileFloat            Integration/CubeHandler.h:40-128
#include "volume_core.hpp"

namespace {

__global__ __launch_bounds__(512) void k_export_aos(const float* __restrict__ pool, size_t first, float* __restrict__ out) {
    const size_t b = blockIdx.x;
    const float* src = pool + (first + b) * kBlockFloats + threadIdx.x;
    float* dst = out + (b * kVox + threadIdx.x) * 5;
#pragma unroll
    for (int p = 0; p < 5; ++p) dst[p] = src[p * kVox];
}

// insert keys; slots[i] receives the TABLE slot of key i, encoded -(slot+2) when newly created; the
// consumers below translate it to the pool slot through tvals (next kernel => visible)
__global__ void k_insert_keys(VolView V, const int* __restrict__ keys, size_t n, int* __restrict__ slots, State* st) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
    if (!key_in_range(x, y, z)) { atomicOr(&st->overflow, 8u); slots[i] = -1; return; }
    bool created;
    const int slot = table_claim(V, st, x, y, z, &created);
    slots[i] = slot < 0 ? -1 : (created ? -(slot + 2) : slot);
}

// AoS voxels -> pool planes for the given slots (SetCubeMap / AddCube + assignment)
__global__ __launch_bounds__(512) void k_import_aos(float* __restrict__ pool, const int* __restrict__ slots,
                                                    const int* __restrict__ tvals, const float* __restrict__ in) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    if (idx <= -2) idx = -(idx + 2);
    idx = tvals[idx]; // table slot -> pool slot
    if (idx < 0) return;
    const float* src = in + ((size_t)blockIdx.x * kVox + threadIdx.x) * 5;
    float* dst = pool + (size_t)idx * kBlockFloats + threadIdx.x;
#pragma unroll
    for (int p = 0; p < 5; ++p) dst[p * kVox] = src[p];
}

// CubeHandler::Merge (CubeHandler.h:145-167): dst block (slots) += src block (TSDFVoxel::operator+,
// general weights), or plain copy when the block was just created in dst.
__global__ __launch_bounds__(512) void k_merge_blocks(float* __restrict__ dpool, const float* __restrict__ spool,
                                                      const int* __restrict__ slots, const int* __restrict__ tvals) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    const bool fresh = idx <= -2;
    if (fresh) idx = -(idx + 2);
    idx = tvals[idx]; // table slot -> pool slot
    if (idx < 0) return;
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
                                                    const int* __restrict__ tvals, const float* __restrict__ sum) {
    int idx = slots[blockIdx.x];
    if (idx == -1) return;
    if (idx <= -2) idx = -(idx + 2);
    idx = tvals[idx]; // table slot -> pool slot
    if (idx < 0) return;
    const float* a = sum + (size_t)blockIdx.x * kBlockFloats + threadIdx.x;
    float* t = pool + (size_t)idx * kBlockFloats + threadIdx.x;
    const float w = a[kVox];
    if (w > 0) {
        t[0] = a[0] / w; t[kVox] = w; t[2 * kVox] = a[2 * kVox] / w; t[3 * kVox] = a[3 * kVox] / w; t[4 * kVox] = a[4 * kVox] / w;
    } else {
        t[0] = 999.0f; t[kVox] = 0.0f; t[2 * kVox] = t[3 * kVox] = t[4 * kVox] = -1.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// Volume resampling (CubeHandler::Transform / TransformNearest, CubeHandler.h:199-338) and
// GetPointCloud (CubeHandler.cpp:45-69)
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ Vox5 default_voxel() { return Vox5{999.0f, 0.0f, -1.0f, -1.0f, -1.0f}; }

// cube_map.find(GetCubeID(p)) + GetVoxel(GetVoxelID(p)) (VoxelCube.h:63-67,81-86); default voxel if absent
__device__ Vox5 fetch_voxel(const VolView& S, int px, int py, int pz) {
    const int cx = px >> 3, cy = py >> 3, cz = pz >> 3; // floor((p + 0.0) / 8)
    const int idx = table_find(S, cx, cy, cz);
    if (idx < 0) return default_voxel();
    const int vid = (px - cx * 8) + (py - cy * 8) * 8 + (pz - cz * 8) * 64;
    const float* t = S.pool + (size_t)idx * kBlockFloats + vid;
    return Vox5{t[0], t[kVox], t[2 * kVox], t[3 * kVox], t[4 * kVox]};
}
// TSDFVoxel::operator*(float) (TSDFVoxel.h:56-67)
__device__ __forceinline__ Vox5 vox_scale(const Vox5& a, float wgt) {
    if (wgt == 0 || a.w == 0) return default_voxel();
    return Vox5{a.s * wgt, a.w * wgt, a.c0 * wgt, a.c1 * wgt, a.c2 * wgt};
}
// TSDFVoxel::add (TSDFVoxel.h:40-51)
__device__ __forceinline__ Vox5 vox_add_direct(const Vox5& a, const Vox5& b) {
    if (a.w == 0) return b;
    if (b.w == 0) return a;
    return Vox5{a.s + b.s, a.w + b.w, a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2};
}
// one stage of ReadVoxelInterpolate (VoxelCube.cpp:17-20):
// ((a * (1 - t)).add(b * t)) / ((1 - t) * (a.weight != 0) + t * (b.weight != 0))
__device__ __forceinline__ Vox5 interp_stage(const Vox5& a, const Vox5& b, float t) {
    if (!(a.w != 0 || b.w != 0)) return default_voxel();
    const Vox5 sum = vox_add_direct(vox_scale(a, 1 - t), vox_scale(b, t));
    const float d = (1 - t) * (float)(a.w != 0) + t * (float)(b.w != 0);
    return vox_scale(sum, 1 / d); // operator/(w) = operator*(1 / w) (TSDFVoxel.h:68-71)
}

// workgroup-wide minimum / maximum of an int (512 threads = 8 waves), result in every thread
__device__ __forceinline__ void wg_min_max(int v_lo, int v_hi, int* s_red, int* out_lo, int* out_hi) {
    for (int o = 32; o > 0; o >>= 1) { v_lo = min(v_lo, __shfl_xor(v_lo, o, 64)); v_hi = max(v_hi, __shfl_xor(v_hi, o, 64)); }
    const int wave = threadIdx.x >> 6;
    __syncthreads(); // (s_red may still be read from a previous call)
    if ((threadIdx.x & 63) == 0) { s_red[wave] = v_lo; s_red[8 + wave] = v_hi; }
    __syncthreads();
    int lo = s_red[0], hi = s_red[8];
#pragma unroll
    for (int k = 1; k < 8; ++k) { lo = min(lo, s_red[k]); hi = max(hi, s_red[8 + k]); }
    *out_lo = lo; *out_hi = hi;
}

// pass 1: AddTransformedCube / AddTransformedCubeNearest (CubeHandler.h:199-241), executed with the
// RESULT's CubePara (alloc_res).  One workgroup per source block.  A block's 512 voxels land in a handful of result blocks: the
// workgroup gathers the distinct ids in a small LDS set and claims each ONCE in the result's hash table (a voxel-by-voxel claim
// was ~2 000 hash probes per source block, most of them for ids another voxel had just entered).
constexpr int kClaimSet = 128; // slots of the set (a power of two); a workgroup whose voxels reach more distinct blocks claims the overflow directly
template <bool NEAREST>
__global__ __launch_bounds__(512) void k_transform_alloc(VolView S, VolView D, State* dst_state, Mat4 T, float alloc_res) {
    __shared__ unsigned long long s_set[kClaimSet];
    const int b = blockIdx.x, vid = threadIdx.x;
    if (vid < kClaimSet) s_set[vid] = kEmptyKey;
    __syncthreads();
    const int kx = S.keys[3 * b], ky = S.keys[3 * b + 1], kz = S.keys[3 * b + 2];
    const float half = alloc_res / 2;
    const float px = ((float)kx * 8.0f) * alloc_res + ((float)(vid & 7) * alloc_res + half);
    const float py = ((float)ky * 8.0f) * alloc_res + ((float)((vid >> 3) & 7) * alloc_res + half);
    const float pz = ((float)kz * 8.0f) * alloc_res + ((float)(vid >> 6) * alloc_res + half);
    const float* M = T.m;
    const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    const float q1 = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    const float q2 = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    const float q3 = ((M[12] * px + M[13] * py) + M[14] * pz) + M[15] * 1.0f;
    const float n0 = NEAREST ? q0 / q3 : q0 / q3 - half, n1 = NEAREST ? q1 / q3 : q1 / q3 - half,
                n2 = NEAREST ? q2 / q3 : q2 / q3 - half;
    const int p0 = (int)floorf(n0 / alloc_res), p1 = (int)floorf(n1 / alloc_res), p2 = (int)floorf(n2 / alloc_res);
    int lx = INT_MIN, ly = INT_MIN, lz = INT_MIN;
#pragma unroll
    for (int k = 0; k < (NEAREST ? 1 : 8); ++k) {
        const int cx = (p0 + (k & 1)) >> 3, cy = (p1 + ((k >> 1) & 1)) >> 3, cz = (p2 + ((k >> 2) & 1)) >> 3;
        if (cx == lx && cy == ly && cz == lz) continue;
        lx = cx; ly = cy; lz = cz;
        if (!key_in_range(cx, cy, cz)) { atomicOr(&dst_state->overflow, 8u); continue; }
        // enter the id into the workgroup's set; a full set (never, for a rigid motion) falls back to the direct claim
        const unsigned long long key = pack_key(cx, cy, cz);
        unsigned sl = (unsigned)(hash_key_dev(cx, cy, cz) * 0x9E3779B97F4A7C15ULL >> 57) & (kClaimSet - 1);
        bool placed = false;
        for (int probe = 0; probe < kClaimSet; ++probe, sl = (sl + 1) & (kClaimSet - 1)) {
            const unsigned long long cur = s_set[sl];
            if (cur == key) { placed = true; break; }
            if (cur == kEmptyKey) {
                const unsigned long long old = atomicCAS(&s_set[sl], kEmptyKey, key);
                if (old == kEmptyKey || old == key) { placed = true; break; }
            }
        }
        if (!placed) { bool created; table_claim(D, dst_state, cx, cy, cz, &created); }
    }
    __syncthreads();
    if (vid < kClaimSet && s_set[vid] != kEmptyKey) {
        const unsigned long long key = s_set[vid];
        bool created;
        table_claim(D, dst_state, (int)(key >> 42) - kCoordLimit, (int)((key >> 21) & 0x1FFFFFull) - kCoordLimit, (int)(key & 0x1FFFFFull) - kCoordLimit, &created); // AddCube
    }
}

// pass 2: every voxel of the result reads the source through trans^-1 (CubeHandler.h:257-294 /
// :312-334) with the SOURCE's CubePara (this->c_para).  One workgroup per result block; result
// voxels are still default, so `voxels[voxel_id] += result` stores `result` (weight == 0 -> other).
// The taps of a block's 512 voxels fall into a few source blocks (a rotated 8^3 cube spans at most three per axis): the workgroup
// takes the bounding box of its taps in block coordinates, looks every block of the box up ONCE (up to kSrcBox of them) and the taps
// index that LDS table -- eight hash probes per voxel become none.  A box beyond kSrcBox blocks (a projective `trans`, NaN) probes per tap.
constexpr int kSrcBox = 64;
__device__ __forceinline__ Vox5 load_voxel(const VolView& S, int idx, int px, int py, int pz) {
    if (idx < 0) return default_voxel();
    const float* t = S.pool + (size_t)idx * kBlockFloats + ((px & 7) + (py & 7) * 8 + (pz & 7) * 64);
    return Vox5{t[0], t[kVox], t[2 * kVox], t[3 * kVox], t[4 * kVox]};
}
template <bool NEAREST>
__global__ __launch_bounds__(512) void k_transform_fill(VolView S, VolView D, Mat4 Tinv, float src_res) {
    __shared__ int s_red3[3][16];
    __shared__ int s_slot[kSrcBox];
    const int b = blockIdx.x, vid = threadIdx.x;
    const int kx = D.keys[3 * b], ky = D.keys[3 * b + 1], kz = D.keys[3 * b + 2];
    const float half = src_res / 2;
    const float px = ((float)kx * 8.0f) * src_res + ((float)(vid & 7) * src_res + half);
    const float py = ((float)ky * 8.0f) * src_res + ((float)((vid >> 3) & 7) * src_res + half);
    const float pz = ((float)kz * 8.0f) * src_res + ((float)(vid >> 6) * src_res + half);
    const float* M = Tinv.m;
    const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
    const float q1 = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
    const float q2 = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
    const float q3 = ((M[12] * px + M[13] * py) + M[14] * pz) + M[15] * 1.0f;
    const float n0 = NEAREST ? q0 / q3 : q0 / q3 - half, n1 = NEAREST ? q1 / q3 : q1 / q3 - half,
                n2 = NEAREST ? q2 / q3 : q2 / q3 - half;
    const int p0 = (int)floorf(n0 / src_res), p1 = (int)floorf(n1 / src_res), p2 = (int)floorf(n2 / src_res);
    // the source blocks this workgroup's taps touch (taps at p and p + 1; the nearest form has ONE tap per voxel and probes directly: measured, the
    // box costs it more than it saves).  An int overflow of p + 1 only ever widens the box: fallback.
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0, z0 = 0, z1 = 0;
    if (!NEAREST) { // the three axes in ONE pass over the workgroup (two barriers instead of six)
        int lo[3] = {p0 >> 3, p1 >> 3, p2 >> 3};
        int hi[3] = {(int)(((long long)p0 + 1) >> 3), (int)(((long long)p1 + 1) >> 3), (int)(((long long)p2 + 1) >> 3)};
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64)); }
        const int wave = vid >> 6;
        if ((vid & 63) == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { s_red3[a][wave] = lo[a]; s_red3[a][8 + wave] = hi[a]; }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int l = s_red3[a][0], h = s_red3[a][8];
#pragma unroll
            for (int k = 1; k < 8; ++k) { l = min(l, s_red3[a][k]); h = max(h, s_red3[a][8 + k]); }
            lo[a] = l; hi[a] = h;
        }
        x0 = lo[0]; x1 = hi[0]; y0 = lo[1]; y1 = hi[1]; z0 = lo[2]; z1 = hi[2];
    }
    const long long ex = (long long)x1 - x0 + 1, ey = (long long)y1 - y0 + 1, ez = (long long)z1 - z0 + 1;
    const bool boxed = !NEAREST && ex * ey <= kSrcBox && ex * ey * ez <= kSrcBox; // (uniform)
    if (boxed) {
        if (vid < (int)(ex * ey * ez)) s_slot[vid] = table_find(S, x0 + vid % (int)ex, y0 + (vid / (int)ex) % (int)ey, z0 + vid / (int)(ex * ey));
        __syncthreads();
    }
    auto fetch = [&](int tx, int ty, int tz) -> Vox5 { // cube_map.find(GetCubeID(p)) + GetVoxel(GetVoxelID(p)); default voxel if absent
        if (!boxed) return fetch_voxel(S, tx, ty, tz);
        return load_voxel(S, s_slot[((tx >> 3) - x0) + (int)ex * (((ty >> 3) - y0) + (int)ey * ((tz >> 3) - z0))], tx, ty, tz);
    };
    Vox5 r;
    if (NEAREST) {
        r = fetch(p0, p1, p2);
    } else {
        Vox5 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fetch(p0 + (k & 1), p1 + ((k >> 1) & 1), p2 + ((k >> 2) & 1));
        // ReadVoxelInterpolate (VoxelCube.cpp:6-50)
        const float xw = (n0 - (float)p0 * src_res) / src_res, yw = (n1 - (float)p1 * src_res) / src_res,
                    zw = (n2 - (float)p2 * src_res) / src_res;
        const Vox5 z1v = interp_stage(interp_stage(v[0], v[1], xw), interp_stage(v[2], v[3], xw), yw);
        const Vox5 z2v = interp_stage(interp_stage(v[4], v[5], xw), interp_stage(v[6], v[7], xw), yw);
        r = interp_stage(z1v, z2v, zw);
    }
    float* t = D.pool + (size_t)b * kBlockFloats + vid;
    t[0] = r.s; t[kVox] = r.w; t[2 * kVox] = r.c0; t[3 * kVox] = r.c1; t[4 * kVox] = r.c2;
}

// The trilinear form as ONE WAVE per result block (eight z-slices of 64 voxels in turn).  The 512-thread form above is a chain of dependent
// phases -- key, positions, box reduction, hash probes, taps, stores -- with workgroup barriers between them and only four workgroups per CU to
// overlap one block's waiting with another's work (measured: neither staging the tap box in LDS nor halving the number of tap loads moved it).
// A wave needs no barriers: the box is a shuffle reduction, the probed slots sit in the wave's own 64 LDS words, and 32 independent blocks are in
// flight per CU.  Same positions, same taps, same arithmetic as the form above.  2.82 -> 2.61 ms on the 164 k-block room volume.
// (Also measured, on top of this form, and not kept: the result blocks in Morton order of their keys with one compact region per XCD -- the kernel
// fetches 5.8 GB for a 1.7 GB source at 40 % L2 hits, so locality looked like the lever -- 2.69 ms, and 1.39 against 1.20 ms for the nearest form.)
__global__ __launch_bounds__(64) void k_transform_fill_wave(VolView S, VolView D, Mat4 Tinv, float src_res) {
    __shared__ int s_slot[kSrcBox];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int kx = D.keys[3 * b], ky = D.keys[3 * b + 1], kz = D.keys[3 * b + 2];
    const float half = src_res / 2;
    const float* M = Tinv.m;
    float n0[8], n1[8], n2[8];
    int p0[8], p1[8], p2[8];
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
#pragma unroll
    for (int zs = 0; zs < 8; ++zs) {
        const int vid = lane + 64 * zs;
        const float px = ((float)kx * 8.0f) * src_res + ((float)(vid & 7) * src_res + half);
        const float py = ((float)ky * 8.0f) * src_res + ((float)((vid >> 3) & 7) * src_res + half);
        const float pz = ((float)kz * 8.0f) * src_res + ((float)(vid >> 6) * src_res + half);
        const float q0 = ((M[0] * px + M[1] * py) + M[2] * pz) + M[3] * 1.0f;
        const float q1 = ((M[4] * px + M[5] * py) + M[6] * pz) + M[7] * 1.0f;
        const float q2 = ((M[8] * px + M[9] * py) + M[10] * pz) + M[11] * 1.0f;
        const float q3 = ((M[12] * px + M[13] * py) + M[14] * pz) + M[15] * 1.0f;
        n0[zs] = q0 / q3 - half; n1[zs] = q1 / q3 - half; n2[zs] = q2 / q3 - half;
        p0[zs] = (int)floorf(n0[zs] / src_res); p1[zs] = (int)floorf(n1[zs] / src_res); p2[zs] = (int)floorf(n2[zs] / src_res);
        lo[0] = min(lo[0], p0[zs] >> 3); hi[0] = max(hi[0], (int)(((long long)p0[zs] + 1) >> 3));
        lo[1] = min(lo[1], p1[zs] >> 3); hi[1] = max(hi[1], (int)(((long long)p1[zs] + 1) >> 3));
        lo[2] = min(lo[2], p2[zs] >> 3); hi[2] = max(hi[2], (int)(((long long)p2[zs] + 1) >> 3));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64)); }
    const int x0 = lo[0], y0 = lo[1], z0 = lo[2];
    const long long ex = (long long)hi[0] - x0 + 1, ey = (long long)hi[1] - y0 + 1, ez = (long long)hi[2] - z0 + 1;
    const bool boxed = ex * ey <= kSrcBox && ex * ey * ez <= kSrcBox; // (uniform)
    if (boxed) {
        if (lane < (int)(ex * ey * ez)) s_slot[lane] = table_find(S, x0 + lane % (int)ex, y0 + (lane / (int)ex) % (int)ey, z0 + lane / (int)(ex * ey));
        __syncthreads(); // (one wave: orders the LDS writes before the reads below)
    }
    auto fetch = [&](int tx, int ty, int tz) -> Vox5 {
        if (!boxed) return fetch_voxel(S, tx, ty, tz);
        return load_voxel(S, s_slot[((tx >> 3) - x0) + (int)ex * (((ty >> 3) - y0) + (int)ey * ((tz >> 3) - z0))], tx, ty, tz);
    };
#pragma unroll 2
    for (int zs = 0; zs < 8; ++zs) {
        Vox5 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fetch(p0[zs] + (k & 1), p1[zs] + ((k >> 1) & 1), p2[zs] + ((k >> 2) & 1));
        // ReadVoxelInterpolate (VoxelCube.cpp:6-50)
        const float xw = (n0[zs] - (float)p0[zs] * src_res) / src_res, yw = (n1[zs] - (float)p1[zs] * src_res) / src_res,
                    zw = (n2[zs] - (float)p2[zs] * src_res) / src_res;
        const Vox5 z1v = interp_stage(interp_stage(v[0], v[1], xw), interp_stage(v[2], v[3], xw), yw);
        const Vox5 z2v = interp_stage(interp_stage(v[4], v[5], xw), interp_stage(v[6], v[7], xw), yw);
        const Vox5 r = interp_stage(z1v, z2v, zw);
        float* t = D.pool + (size_t)b * kBlockFloats + (lane + 64 * zs);
        t[0] = r.s; t[kVox] = r.w; t[2 * kVox] = r.c0; t[3 * kVox] = r.c1; t[4 * kVox] = r.c2;
    }
}

// GetPointCloud: voxels with weight != 0 and |sdf| < truncation, in the reference's x,y,z loop order
// inside a block.  counts == nullptr: emit using offsets; else count only.
__global__ __launch_bounds__(512) void k_point_cloud(VolView V, float res, float trunc, unsigned* __restrict__ counts,
                                                     const unsigned* __restrict__ offsets, float* __restrict__ xyz,
                                                     float* __restrict__ col) {
    __shared__ unsigned s_w[8];
    const int b = blockIdx.x, o = threadIdx.x;
    const int x = o >> 6, y = (o >> 3) & 7, z = o & 7; // loop nest: x outer, y, z inner
    const int vid = x + y * 8 + z * 64;
    const float* t = V.pool + (size_t)b * kBlockFloats + vid;
    const float sdf = t[0], w = t[kVox];
    const bool ok = w != 0 && fabsf(sdf) < trunc;
    const unsigned long long m = __ballot(ok);
    const int lane = o & 63, wave = o >> 6;
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    if (counts) {
        if (o == 0) { unsigned tot = 0; for (int k = 0; k < 8; ++k) tot += s_w[k]; counts[b] = tot; }
        return;
    }
    if (!ok) return;
    unsigned rank = __popcll(m & ((1ULL << lane) - 1ULL));
    for (int k = 0; k < wave; ++k) rank += s_w[k];
    const size_t pos = (size_t)offsets[b] + rank;
    const float cube_res = 8.0f * res, half = res / 2; // VoxelCube.h:150, :47
    const float f = fabsf(sdf) / trunc;
    xyz[3 * pos] = (float)V.keys[3 * b] * cube_res + ((float)x * res + half);
    xyz[3 * pos + 1] = (float)V.keys[3 * b + 1] * cube_res + ((float)y * res + half);
    xyz[3 * pos + 2] = (float)V.keys[3 * b + 2] * cube_res + ((float)z * res + half);
    col[3 * pos] = f; col[3 * pos + 1] = f; col[3 * pos + 2] = f;
}


// ---------------------------------------------------------------------------------------------
// Mesh extraction: CubeHandler::ExtractTriangleMesh / GenerateMeshByCube (CubeHandler.cpp:9-114) +
// MarchingCube (MarchingCube.cpp:8-74).  One workgroup per block, one thread per voxel in the reference's
// x, y, z loop order; the 7 neighbour blocks a voxel on the +x/+y/+z faces needs are looked up once per
// workgroup.  The 256 x 16 triangle table and the 12 x 2 edge table are the CALLER'S data (the reference
// keeps them in MarchingCubePredefined.h; its shim passes them through the C-ABI); the triangle table arrives
// packed to one signed byte per entry (4 KB: it stays in the L1 / L2 of every CU instead of being copied into
// 16 KB of LDS by each of the volume's workgroups).
// Two passes with the same kernel: counts (triangles per block) and, after a scan, the ordered emit of
// three unshared vertices per triangle, exactly as MarchingCube() pushes them.
// The block's sdf values and IsValid flags (and the +x/+y/+z layer of its neighbours) are staged ONCE in a 9^3 LDS
// tile: the case of a voxel comes from 8 LDS reads; positions are arithmetic; colours -- needed only by the
// emit pass, and there only by blocks that hold triangles -- are staged the same way when they are.  (Rounds 1-4 loaded
// 8 corners x 5 planes from global memory for every voxel in both passes: 3.3 ms per pass on the 164 k-block
// room volume, profiles/r05_volume_ops.kernel_stats.csv.)
// ---------------------------------------------------------------------------------------------
constexpr int kMeshEdge = 9, kMeshTile = kMeshEdge * kMeshEdge * kMeshEdge;
// the pool slots of a listed block and of its 7 upper neighbours (HasCube(neighbor_cube_id); -1: absent), one lane per lookup: both passes of k_mesh then
// start on their voxels one load after reading them (the dependent chain keys -> hash probe -> voxels was what a 512-thread workgroup waited for)
__global__ __launch_bounds__(256) void k_mesh_neighbours(VolView V, const unsigned* __restrict__ blocks, unsigned n, int* __restrict__ nbs) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n * 8u) return;
    const unsigned e = i >> 3, j = i & 7u;
    const int b = (int)blocks[e];
    nbs[i] = j == 0 ? b : table_find(V, V.keys[3 * b] + (int)(j & 1u), V.keys[3 * b + 1] + (int)((j >> 1) & 1u), V.keys[3 * b + 2] + (int)((j >> 2) & 1u));
}
__global__ __launch_bounds__(512) void k_mesh(VolView V, float res, const signed char* __restrict__ tri8, const int* __restrict__ edge_pairs,
                                              const unsigned* __restrict__ blocks, const int* __restrict__ nbs, unsigned* __restrict__ counts,
                                              const unsigned* __restrict__ offsets, float* __restrict__ pts, float* __restrict__ col) {
    __shared__ float s_sdf[kMeshTile];
    __shared__ float s_col[3][kMeshTile];         // emit pass, blocks with triangles: the colour planes of the same 9^3 voxels
    __shared__ unsigned char s_ok[kMeshTile + 3]; // 1: the voxel exists and IsValid (sdf < 1, weight > 0: TSDFVoxel.h:75-78)
    __shared__ int s_edge[24];
    __shared__ int s_nb[8];
    __shared__ unsigned s_w[8];
    const int b = (int)blocks[blockIdx.x], o = threadIdx.x;
    if (o < 24) s_edge[o] = edge_pairs[o];
    const int kx = V.keys[3 * b], ky = V.keys[3 * b + 1], kz = V.keys[3 * b + 2];
    if (o < 8) s_nb[o] = nbs[8 * (size_t)blockIdx.x + o];
    __syncthreads();
    {   // the tile: own 512 voxels (thread = voxel id, coalesced plane rows), then the 217 voxels of the +x / +y / +z layer
        const float* own = V.pool + (size_t)b * kBlockFloats;
        const float sd = own[o], w = own[kVox + o];
        const int ti = (o & 7) + kMeshEdge * ((o >> 3) & 7) + kMeshEdge * kMeshEdge * (o >> 6);
        s_sdf[ti] = sd; s_ok[ti] = (sd >= 1 || w <= 0) ? 0 : 1;
        if (o < 217) {
            int x, y, z;
            if (o < 81) { x = o % 9; y = o / 9; z = 8; }
            else if (o < 153) { const int j = o - 81; x = j % 9; y = 8; z = j / 9; }
            else { const int j = o - 153; x = 8; y = j & 7; z = j >> 3; }
            const int nb = s_nb[(x >> 3) | ((y >> 3) << 1) | ((z >> 3) << 2)];
            float sd2 = 0.0f;
            unsigned char ok2 = 0;
            if (nb >= 0) {
                const float* t = V.pool + (size_t)nb * kBlockFloats + ((x & 7) + (y & 7) * 8 + (z & 7) * 64);
                sd2 = t[0];
                ok2 = (sd2 >= 1 || t[kVox] <= 0) ? 0 : 1;
            }
            const int tj = x + kMeshEdge * y + kMeshEdge * kMeshEdge * z;
            s_sdf[tj] = sd2; s_ok[tj] = ok2;
        }
    }
    __syncthreads();
    const int x = o >> 6, y = (o >> 3) & 7, z = o & 7;    // loop nest: x outer, y, z inner
    float cs[8];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int xo = (i == 1 || i == 2 || i == 5 || i == 6), yo = (i == 2 || i == 3 || i == 6 || i == 7), zo = i >= 4; // CornerXYZOffset, VoxelCube.h:45-47
        const int tq = (x + xo) + kMeshEdge * (y + yo) + kMeshEdge * kMeshEdge * (z + zo);
        cs[i] = s_sdf[tq];
        ok = ok && s_ok[tq] != 0;
    }
    int ci = 0;
    unsigned ntri = 0;
    uint4 rw = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu); // the case's table row: sixteen signed bytes, -1 terminated
    auto entry = [&](int i) -> int { const unsigned wd = i < 4 ? rw.x : (i < 8 ? rw.y : (i < 12 ? rw.z : rw.w)); return (int)(signed char)((wd >> (8 * (i & 3))) & 0xffu); };
    if (ok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ci |= cs[i] > 0 ? 1 << i : 0;  // DetermineCase
        rw = *reinterpret_cast<const uint4*>(tri8 + 16 * ci);
#pragma unroll
        for (int i = 0; i < 16; i += 3) ntri += entry(i) != -1 ? 1u : 0u; // rows are well formed: whole triangles, then -1s (validated by the caller side: i == 15 is -1)
    }
    // exclusive scan of ntri over the workgroup in thread (= reference loop) order
    unsigned incl = ntri;
    const int lane = o & 63, wave = o >> 6;
    for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    if (counts) {
        if (o == 0) { unsigned tot = 0; for (int k = 0; k < 8; ++k) tot += s_w[k]; counts[blockIdx.x] = tot; }
        return;
    }
    {   // emit pass: a block without triangles is done; otherwise its colour planes are staged like the sdf tile (coalesced plane rows + the
        // neighbours' layer) -- a gather per emitted vertex fetched a 32-byte sector for every 4 bytes it used (2.3 GB per pass on the 164 k-block volume)
        unsigned tot = 0;
        for (int k = 0; k < 8; ++k) tot += s_w[k];
        if (tot == 0) return;                      // (uniform over the workgroup)
        const float* own = V.pool + (size_t)b * kBlockFloats;
        const int ti = (o & 7) + kMeshEdge * ((o >> 3) & 7) + kMeshEdge * kMeshEdge * (o >> 6);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) s_col[c3][ti] = own[(2 + c3) * kVox + o];
        if (o < 217) {
            int x2, y2, z2;
            if (o < 81) { x2 = o % 9; y2 = o / 9; z2 = 8; }
            else if (o < 153) { const int j = o - 81; x2 = j % 9; y2 = 8; z2 = j / 9; }
            else { const int j = o - 153; x2 = 8; y2 = j & 7; z2 = j >> 3; }
            const int nb = s_nb[(x2 >> 3) | ((y2 >> 3) << 1) | ((z2 >> 3) << 2)];
            const int tj = x2 + kMeshEdge * y2 + kMeshEdge * kMeshEdge * z2;
            if (nb >= 0) {
                const float* t = V.pool + (size_t)nb * kBlockFloats + ((x2 & 7) + (y2 & 7) * 8 + (z2 & 7) * 64);
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) s_col[c3][tj] = t[(2 + c3) * kVox];
            }
        }
        __syncthreads();
    }
    if (!ntri) return;
    unsigned first = incl - ntri;
    for (int k = 0; k < wave; ++k) first += s_w[k];
    size_t vtx = ((size_t)offsets[blockIdx.x] + first) * 3;
    const float cube_res = 8.0f * res, half = res / 2;    // VoxelCube.h:149-153, :48-61
    for (int i = 0; i < 16 && entry(i) != -1; i += 3)
        for (int j = 0; j < 3; ++j, ++vtx) {
            const int e = entry(i + j);
            // InterpolateEdgeVetex (MarchingCube.cpp:8-16) between corners s_edge[2e] and s_edge[2e + 1]
            float sv[2], pv[2][3], cv[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cn = s_edge[2 * e + q];
                const int xo = (cn == 1 || cn == 2 || cn == 5 || cn == 6), yo = (cn == 2 || cn == 3 || cn == 6 || cn == 7), zo = cn >= 4;
                const int gx = x + xo, gy = y + yo, gz = z + zo;                     // 0 .. 8
                const int vx = gx & 7, vy = gy & 7, vz = gz & 7, bxo = gx >> 3, byo = gy >> 3, bzo = gz >> 3;
                const int tq = gx + kMeshEdge * gy + kMeshEdge * kMeshEdge * gz;
                sv[q] = s_sdf[tq];
                pv[q][0] = (float)(kx + bxo) * cube_res + ((float)vx * res + half);
                pv[q][1] = (float)(ky + byo) * cube_res + ((float)vy * res + half);
                pv[q][2] = (float)(kz + bzo) * cube_res + ((float)vz * res + half);
                cv[q][0] = s_col[0][tq]; cv[q][1] = s_col[1][tq]; cv[q][2] = s_col[2][tq];
            }
            const float sdf_diff = sv[1] - sv[0];
            const float t = sv[0] / sdf_diff;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pts[3 * vtx + k] = pv[0][k] - t * (pv[1][k] - pv[0][k]);
                col[3 * vtx + k] = (cv[0][k] + cv[1][k]) / 2.0f;  // (c1 + c2) / 2
            }
        }
}

} // namespace

extern "C" {

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
        OP_HIP(op::cached_malloc((void**)&stage, std::min(chunk, take) * kBlockFloats * sizeof(float)));
        for (size_t first = 0; first < take; first += chunk) {
            const size_t cnt = std::min(chunk, take - first);
            hipLaunchKernelGGL(k_export_aos, dim3((unsigned)cnt), dim3(512), 0, v->stream, (const float*)v->pool, first, stage);
            hipError_t e = hipStreamSynchronize(v->stream);
            if (e == hipSuccess)
                e = hipMemcpy(voxels_aos + first * kBlockFloats, stage, cnt * kBlockFloats * sizeof(float), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { op::cached_free(stage); return fail(OP_ERR_HIP, "download failed: %s", hipGetErrorString(e)); }
        }
        op::cached_free(stage);
    }
    return OP_OK;
}

int op_volume_upload(op_volume* v, const int32_t* keys_xyz, const float* voxels_aos, size_t n) {
    OP_VOL(v);
    if (n == 0) return OP_OK;
    if (!keys_xyz || !voxels_aos) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_flush(v)); // frames queued by op_volume_integrate come BEFORE the upload, as the caller issued them
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
    {   // room for every new block up front (the pool grows; nothing can overflow below)
        unsigned nb = 0;
        OP_TRY(vol_block_count(v, &nb));
        OP_TRY(vol_reserve(v, (unsigned long long)nb + uniq.size()));
    }
    { unsigned nb = 0; OP_TRY(vol_block_count(v, &nb)); vol_mark_foreign(v, (unsigned long long)nb + uniq.size()); } // caller-supplied voxel data in every block that exists after this call
    const size_t chunk = 8192;
    int *d_keys = nullptr, *d_slots = nullptr;
    float* d_vox = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_keys, chunk * 3 * sizeof(int)));
    OP_HIP(op::cached_malloc((void**)&d_slots, chunk * sizeof(int)));
    OP_HIP(op::cached_malloc((void**)&d_vox, chunk * kBlockFloats * sizeof(float)));
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
            hipLaunchKernelGGL(k_import_aos, dim3((unsigned)cnt), dim3(512), 0, v->stream, v->pool, (const int*)d_slots, (const int*)v->tvals, (const float*)d_vox);
            e = hipStreamSynchronize(v->stream);
        }
        if (e != hipSuccess) rc = fail(OP_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
    }
    op::cached_free(d_keys); op::cached_free(d_slots); op::cached_free(d_vox);
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
    unsigned nd = 0;
    OP_TRY(vol_block_count(dst, &nd));
    if (!ns) return OP_OK;
    OP_TRY(vol_reserve(dst, (unsigned long long)nd + ns)); // worst case: no block in common
    vol_mark_foreign(dst, (unsigned long long)nd + ns); // merged means: general weights in the blocks that exist after this call
    int* d_slots = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_slots, (size_t)ns * sizeof(int)));
    hipLaunchKernelGGL(k_insert_keys, dim3((ns + 255) / 256), dim3(256), 0, dst->stream, dst->view(), (const int*)src->keys, (size_t)ns, d_slots, dst->state);
    hipLaunchKernelGGL(k_merge_blocks, dim3(ns), dim3(512), 0, dst->stream, dst->pool, (const float*)src->pool, (const int*)d_slots, (const int*)dst->tvals);
    hipError_t e = hipStreamSynchronize(dst->stream);
    op::cached_free(d_slots);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "merge failed: %s", hipGetErrorString(e));
    return vol_check(dst);
}

int op_volume_keys_device(op_volume* v, int32_t* d_keys, size_t cap, size_t* n) {
    OP_VOL(v);
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    if (n) *n = nb;
    const size_t take = std::min((size_t)nb, cap);
    if (d_keys && take) { // on the volume's stream and complete on return (a device-to-device hipMemcpy does not block the host)
        OP_HIP(hipMemcpyAsync(d_keys, v->keys, take * 3 * sizeof(int), hipMemcpyDeviceToDevice, v->stream));
        OP_HIP(hipStreamSynchronize(v->stream));
    }
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

// The root's side of the merge in two steps, so that a caller can normalise slices of the union while later slices are
// still in the reduce: _begin validates, makes room (the pool grows if the union needs it), drops the volume's own
// content and enters all union keys; _chunk writes the normalised voxels of union blocks [first, first + count).
int op_volume_unpack_sum_begin(op_volume* v, const int32_t* d_union_keys, size_t n_union) {
    OP_VOL(v);
    // validate BEFORE the volume's own content is dropped: a refused unpack must leave the locally fused volume intact
    if (n_union && !d_union_keys) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_check(v));
    OP_TRY(vol_reserve(v, n_union)); // grows the root's pool if the union needs it; a refusal leaves the volume as it was
    OP_TRY(op_volume_clear(v));
    if (v->unpack_slots) { op::cached_free(v->unpack_slots); v->unpack_slots = nullptr; }
    v->unpack_n = n_union;
    if (n_union == 0) return OP_OK;
    vol_mark_foreign(v, n_union); // normalised sums of several ranks
    OP_HIP(op::cached_malloc((void**)&v->unpack_slots, n_union * sizeof(int)));
    hipLaunchKernelGGL(k_insert_keys, dim3((unsigned)((n_union + 255) / 256)), dim3(256), 0, v->stream, v->view(), (const int*)d_union_keys, n_union, v->unpack_slots, v->state);
    OP_HIP(hipGetLastError());
    OP_TRY(vol_check(v));
    v->unpack_gen = v->generation;
    return OP_OK;
}

int op_volume_unpack_sum_chunk(op_volume* v, size_t first, size_t count, const float* d_sum_chunk) {
    OP_VOL(v);
    if (count == 0) return OP_OK;
    if (!d_sum_chunk) return fail(OP_ERR_INVALID, "null argument");
    if (!v->unpack_slots || first + count > v->unpack_n) return fail(OP_ERR_INVALID, "op_volume_unpack_sum_chunk: range outside the union given to _begin");
    if (v->unpack_gen != v->generation) // growth re-hashes the table, clear drops it, fusion may do either: the slots of _begin are stale
        return fail(OP_ERR_INVALID, "op_volume_unpack_sum_chunk: the volume was cleared, grown or fused into since op_volume_unpack_sum_begin");
    hipLaunchKernelGGL(k_unpack_sum, dim3((unsigned)count), dim3(512), 0, v->stream, v->pool, (const int*)(v->unpack_slots + first), (const int*)v->tvals, d_sum_chunk);
    OP_HIP(hipGetLastError());
    OP_HIP(hipStreamSynchronize(v->stream));
    return OP_OK;
}

int op_volume_unpack_sum(op_volume* v, const int32_t* d_union_keys, size_t n_union, const float* d_sum) {
    if (v && n_union && !d_sum) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(op_volume_unpack_sum_begin(v, d_union_keys, n_union));
    return op_volume_unpack_sum_chunk(v, 0, n_union, d_sum);
}

int op_volume_transform(op_volume* src, const float T[16], const float* T_inv, int nearest, uint64_t max_blocks, op_volume** out) {
    OP_VOL(src);
    if (!T || !out) return fail(OP_ERR_INVALID, "null argument");
    *out = nullptr;
    unsigned ns = 0;
    OP_TRY(vol_block_count(src, &ns));
    // Transform copies c_para into the result (CubeHandler.h:249); TransformNearest does not
    // (CubeHandler.h:301-305), so its result keeps the default resolution 0.01 (VoxelCube.h:27)
    const float dst_res = nearest ? 0.01f : src->res;
    // a rigid motion turns ns blocks into ~1.25 ns (164 k -> 205 k on the room volume); the pool starts at twice the source and the
    // idempotent allocation pass below runs again after a growth if a thin, badly aligned shell needs more (8 ns was 13 GB for that volume)
    if (max_blocks == 0) max_blocks = std::max<uint64_t>(2ull * ns + 4096ull, 1ull << 14);
    op_volume* dst = nullptr;
    OP_TRY(op_volume_create(&src->cam, dst_res, src->trunc, src->far_d, src->near_d, src->device, max_blocks, &dst));
    vol_mark_foreign(dst, max_blocks); // resampled values (the reference's own divisions may even leave NaN / inf in them); the bound is tightened below
    Mat4 M, Mi;
    std::memcpy(M.m, T, sizeof(M.m));
    if (T_inv) std::memcpy(Mi.m, T_inv, sizeof(Mi.m));
    else op_host::mat4_inverse(T, Mi.m); // trans.inverse() (CubeHandler.h:265,320)
    int rc = OP_OK;
    if (ns) {
        unsigned nd = 0;
        for (;;) { // if the result outgrows its pool, vol_block_count grows it and the (idempotent) allocation pass runs again
            const unsigned cap_before = dst->max_blocks;
            if (nearest) hipLaunchKernelGGL(k_transform_alloc<true>, dim3(ns), dim3(512), 0, dst->stream, src->view(), dst->view(), dst->state, M, dst_res);
            else hipLaunchKernelGGL(k_transform_alloc<false>, dim3(ns), dim3(512), 0, dst->stream, src->view(), dst->view(), dst->state, M, dst_res);
            rc = vol_block_count(dst, &nd);
            if (rc != OP_OK || dst->max_blocks == cap_before) break;
        }
        if (rc == OP_OK && nd) {
            if (nearest) hipLaunchKernelGGL(k_transform_fill<true>, dim3(nd), dim3(512), 0, dst->stream, src->view(), dst->view(), Mi, src->res);
            else hipLaunchKernelGGL(k_transform_fill_wave, dim3(nd), dim3(64), 0, dst->stream, src->view(), dst->view(), Mi, src->res);
            rc = vol_check(dst);
        }
    }
    if (rc != OP_OK) { op_volume_destroy(dst); return rc; }
    { unsigned nd = 0; if (vol_block_count(dst, &nd) == OP_OK) dst->plain_from = nd; } // exactly the resampled blocks
    *out = dst;
    return OP_OK;
}

int op_volume_resolution(op_volume* v, float* voxel_res) {
    OP_VOL(v);
    if (!voxel_res) return fail(OP_ERR_INVALID, "null argument");
    *voxel_res = v->res;
    return OP_OK;
}

int op_volume_point_cloud(op_volume* v, float* xyz, float* colors, size_t cap, size_t* n) {
    OP_VOL(v);
    if (!n) return fail(OP_ERR_INVALID, "null n");
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    *n = 0;
    if (!nb) return OP_OK;
    unsigned *d_counts = nullptr, *d_offsets = nullptr;
    float *d_xyz = nullptr, *d_col = nullptr;
    int rc = OP_OK;
    hipError_t e = op::cached_malloc((void**)&d_counts, nb * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_offsets, nb * sizeof(unsigned));
    std::vector<unsigned> cnt(nb), off(nb);
    size_t total = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_point_cloud, dim3(nb), dim3(512), 0, v->stream, v->view(), v->res, v->trunc, d_counts, (const unsigned*)nullptr,
                           (float*)nullptr, (float*)nullptr);
        e = hipStreamSynchronize(v->stream);
        if (e == hipSuccess) e = hipMemcpy(cnt.data(), d_counts, nb * sizeof(unsigned), hipMemcpyDeviceToHost);
        for (unsigned b = 0; b < nb; ++b) { off[b] = (unsigned)total; total += cnt[b]; }
    }
    *n = total;
    if (e == hipSuccess && xyz && colors && total) {
        if (total > cap) rc = fail(OP_ERR_CAPACITY, "point cloud has %zu points, buffer holds %zu", total, cap);
        else {
            e = hipMemcpy(d_offsets, off.data(), nb * sizeof(unsigned), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_xyz, total * 12);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_col, total * 12);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_point_cloud, dim3(nb), dim3(512), 0, v->stream, v->view(), v->res, v->trunc, (unsigned*)nullptr,
                                   (const unsigned*)d_offsets, d_xyz, d_col);
                e = hipStreamSynchronize(v->stream);
            }
            if (e == hipSuccess) e = hipMemcpy(xyz, d_xyz, total * 12, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(colors, d_col, total * 12, hipMemcpyDeviceToHost);
        }
    }
    void* ptrs[] = {d_counts, d_offsets, d_xyz, d_col};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "point cloud failed: %s", hipGetErrorString(e));
    return rc;
}


int op_volume_extract_mesh(op_volume* v, const int32_t* tri_table, const int32_t* edge_pairs, const int32_t* only_block, float* points,
                           float* colors, size_t cap_vertices, size_t* n_vertices) {
    OP_VOL(v);
    if (!tri_table || !edge_pairs || !n_vertices) return fail(OP_ERR_INVALID, "null argument");
    for (int c = 0; c < 256; ++c)
        for (int i = 0; i < 16; ++i) {
            const int e = tri_table[16 * c + i];
            if (e < -1 || e > 11) return fail(OP_ERR_INVALID, "tri_table[%d][%d] = %d is not an edge id or -1", c, i, e);
            if (i == 15 && e != -1) return fail(OP_ERR_INVALID, "tri_table row %d is not -1 terminated", c);
        }
    for (int i = 0; i < 24; ++i)
        if (edge_pairs[i] < 0 || edge_pairs[i] > 7) return fail(OP_ERR_INVALID, "edge_pairs[%d] = %d is not a corner id", i, edge_pairs[i]);
    unsigned nb = 0;
    OP_TRY(vol_block_count(v, &nb));
    *n_vertices = 0;
    if (!nb) return OP_OK;
    // block list: every block in pool order, or the one requested (GenerateMeshByCube)
    std::vector<unsigned> list;
    if (only_block) {
        std::vector<int> keys((size_t)nb * 3);
        OP_HIP(hipMemcpy(keys.data(), v->keys, keys.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (unsigned b = 0; b < nb; ++b)
            if (keys[3 * b] == only_block[0] && keys[3 * b + 1] == only_block[1] && keys[3 * b + 2] == only_block[2]) list.push_back(b);
        if (list.empty()) return OP_OK;
    } else {
        list.resize(nb);
        for (unsigned b = 0; b < nb; ++b) list[b] = b;
    }
    const unsigned nl = (unsigned)list.size();
    unsigned *d_list = nullptr, *d_counts = nullptr, *d_offsets = nullptr;
    signed char* d_tri = nullptr;
    int *d_edge = nullptr, *d_nbs = nullptr;
    float *d_pts = nullptr, *d_col = nullptr;
    int rc = OP_OK;
    signed char tri8[256 * 16];
    for (int i = 0; i < 256 * 16; ++i) tri8[i] = (signed char)tri_table[i]; // (validated above: -1 .. 11)
    hipError_t e = op::cached_malloc((void**)&d_list, nl * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_counts, nl * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_offsets, nl * sizeof(unsigned));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_tri, 256 * 16);
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_edge, 24 * sizeof(int));
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_nbs, (size_t)nl * 8 * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_list, list.data(), nl * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_tri, tri8, 256 * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_edge, edge_pairs, 24 * sizeof(int), hipMemcpyHostToDevice);
    std::vector<unsigned> cnt(nl), off(nl);
    size_t total_tri = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_mesh_neighbours, dim3((nl * 8u + 255u) / 256u), dim3(256), 0, v->stream, v->view(), (const unsigned*)d_list, nl, d_nbs);
        hipLaunchKernelGGL(k_mesh, dim3(nl), dim3(512), 0, v->stream, v->view(), v->res, (const signed char*)d_tri, (const int*)d_edge, (const unsigned*)d_list,
                           (const int*)d_nbs, d_counts, (const unsigned*)nullptr, (float*)nullptr, (float*)nullptr);
        e = hipStreamSynchronize(v->stream);
        if (e == hipSuccess) e = hipMemcpy(cnt.data(), d_counts, nl * sizeof(unsigned), hipMemcpyDeviceToHost);
        for (unsigned b = 0; b < nl; ++b) { off[b] = (unsigned)total_tri; total_tri += cnt[b]; }
    }
    const size_t total = total_tri * 3;
    *n_vertices = total;
    if (e == hipSuccess && points && colors && total) {
        if (total > cap_vertices) rc = fail(OP_ERR_CAPACITY, "mesh has %zu vertices, buffer holds %zu", total, cap_vertices);
        else if (total_tri > 0xffffffffull / 3) rc = fail(OP_ERR_CAPACITY, "mesh too large");
        else {
            e = hipMemcpy(d_offsets, off.data(), nl * sizeof(unsigned), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_pts, total * 12);
            if (e == hipSuccess) e = op::cached_malloc((void**)&d_col, total * 12);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_mesh, dim3(nl), dim3(512), 0, v->stream, v->view(), v->res, (const signed char*)d_tri, (const int*)d_edge,
                                   (const unsigned*)d_list, (const int*)d_nbs, (unsigned*)nullptr, (const unsigned*)d_offsets, d_pts, d_col);
                e = hipStreamSynchronize(v->stream);
            }
            if (e == hipSuccess) e = hipMemcpy(points, d_pts, total * 12, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(colors, d_col, total * 12, hipMemcpyDeviceToHost);
        }
    }
    void* ptrs[] = {d_list, d_counts, d_offsets, d_tri, d_edge, d_nbs, d_pts, d_col};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "mesh extraction failed: %s", hipGetErrorString(e));
    return rc;
}

int op_volume_write_file(op_volume* v, const char* path) {
    OP_VOL(v);
    if (!path) return fail(OP_ERR_INVALID, "null path");
    size_t n = 0;
    OP_TRY(op_volume_block_count(v, &n));
    std::vector<int32_t> keys(3 * n);
    std::unique_ptr<float[]> vox(new float[std::max<size_t>(n, 1) * (size_t)kBlockFloats]);
    if (n) OP_TRY(op_volume_download(v, keys.data(), vox.get(), n, &n));
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(OP_ERR_INVALID, "cannot open %s for writing", path);
    // CubeHandler::WriteToFile (CubeHandler.h:113-128): the block count's raw bits sit in a float slot; then per block
    // VoxelCube::WriteToBuffer (VoxelCube.h:128-148): id, {i, sdf, w, c0, c1, c2} of every voxel with |sdf| < 1 and
    // w != 0, terminator -2.  Two passes: per-block record counts -> offsets, then the blocks are formatted in parallel.
    const float* vx = vox.get();
    std::vector<size_t> off(n + 1, 0);
    for_block_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b) {
            size_t c = 0;
            for (int i = 0; i < kVox; ++i) {
                const float* t = &vx[(b * kVox + i) * 5];
                c += (std::fabs(t[0]) < 1 && t[1] != 0) ? 1 : 0;
            }
            off[b + 1] = 4 + 6 * c;
        }
    });
    off[0] = 1;
    for (size_t b = 0; b < n; ++b) off[b + 1] += off[b];
    const size_t total = off[n];
    std::unique_ptr<float[]> buffer(new float[total]);
    const unsigned int size = (unsigned int)n;
    std::memcpy(&buffer[0], &size, 4);
    float* out = buffer.get();
    for_block_ranges(n, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b) {
            float* o = out + off[b];
            for (int c = 0; c < 3; ++c) *o++ = (float)keys[3 * b + c];
            for (int i = 0; i < kVox; ++i) {
                const float* t = &vx[(b * kVox + i) * 5];
                if (std::fabs(t[0]) < 1 && t[1] != 0) {
                    *o++ = (float)i;
                    for (int k = 0; k < 5; ++k) *o++ = t[k];
                }
            }
            *o++ = -2.0f;
        }
    });
    const bool ok = std::fwrite(buffer.get(), sizeof(float), total, f) == total;
    std::fclose(f);
    return ok ? OP_OK : fail(OP_ERR_INVALID, "short write to %s", path);
}

int op_volume_read_file(op_volume* v, const char* path, int legacy_float_format) {
    OP_VOL(v);
    if (!path) return fail(OP_ERR_INVALID, "null path");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(OP_ERR_INVALID, "cannot open %s", path);
    std::fseek(f, 0, SEEK_END);
    const long len = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const size_t nfl = (size_t)len / sizeof(float);
    std::unique_ptr<float[]> buffer_mem(new float[nfl + 1]);
    float* buffer = buffer_mem.get();
    buffer[nfl] = 0.0f;
    const bool ok = std::fread(buffer, sizeof(float), nfl, f) == nfl;
    std::fclose(f);
    if (!ok || nfl < 2) return fail(OP_ERR_INVALID, "cannot read %s", path);
    unsigned int count = 0;
    size_t ptr = 0;
    if (legacy_float_format) { count = (unsigned int)buffer[1]; ptr = 2; } // CubeHandler.h:91-94
    else { std::memcpy(&count, &buffer[0], 4); ptr = 1; }                   // CubeHandler.h:51-53
    // pass 1 (sequential, cheap): where every block's record starts -- the stream is only delimited by its terminators
    std::vector<size_t> start;
    start.reserve((size_t)count + 1);
    for (unsigned int c = 0; c < count && ptr + 3 <= nfl; ++c) {
        start.push_back(ptr);
        ptr += 3;
        if (!legacy_float_format) {                  // VoxelCube::ReadFromBuffer (VoxelCube.h:153-166): {i, 5 floats}* -2
            while (ptr < nfl && buffer[ptr] != -2.0f) ptr += 6;
            ptr++;
        } else {                                     // VoxelCube::ReadFromBufferFloat (VoxelCube.h:168-193)
            ptr++;
            while (ptr < nfl && buffer[ptr] != -2.0f) ptr += 3;
            ptr++;
            const size_t cnt = ptr < nfl ? (size_t)buffer[ptr++] : 0;
            ptr += 5 * cnt;
        }
        if (ptr > nfl + 1) return fail(OP_ERR_INVALID, "corrupt .map file %s", path);
    }
    const size_t nb = start.size();
    start.push_back(ptr < nfl ? ptr : nfl);
    // pass 2 (parallel over blocks): cube_map[cube_id] = VoxelCube(cube_id) (default voxels), then the stored voxels
    std::vector<int32_t> keys(3 * nb);
    std::unique_ptr<float[]> vox_mem(new float[std::max<size_t>(nb, 1) * (size_t)kBlockFloats]);
    float* vox = vox_mem.get();
    std::atomic<int> bad{0};
    for_block_ranges(nb, [&](size_t lo, size_t hi) {
        for (size_t b = lo; b < hi; ++b) {
            size_t q = start[b];
            const size_t end = start[b + 1];
            for (int c = 0; c < 3; ++c) keys[3 * b + c] = (int32_t)buffer[q + c];
            q += 3;
            float* blk = vox + b * (size_t)kBlockFloats;
            for (int i = 0; i < kVox; ++i) { blk[5 * i] = 999.0f; blk[5 * i + 1] = 0.0f; blk[5 * i + 2] = blk[5 * i + 3] = blk[5 * i + 4] = -1.0f; }
            if (!legacy_float_format) {
                while (q < end && buffer[q] != -2.0f) {
                    const int i = (int)buffer[q++];
                    if (i < 0 || i >= kVox || q + 5 > nfl) { bad = 1; break; }
                    for (int k = 0; k < 5; ++k) blk[5 * i + k] = buffer[q++];
                }
            } else {
                q++;
                while (q < end && buffer[q] != -2.0f) {
                    const int i = (int)buffer[q++];
                    if (i < 0 || i >= kVox || q + 2 > nfl) { bad = 1; break; }
                    blk[5 * i] = buffer[q++]; blk[5 * i + 1] = buffer[q++];
                }
                q++;
                const size_t cnt = q < nfl ? (size_t)buffer[q++] : 0;
                for (size_t k = 0; k < cnt && q + 5 <= nfl; ++k) {
                    const int i = (int)buffer[q++];
                    if (i < 0 || i >= kVox) { bad = 1; break; }
                    float* t = &blk[5 * i];
                    t[2] = (float)(buffer[q++] / 255.0); t[3] = (float)(buffer[q++] / 255.0); t[4] = (float)(buffer[q++] / 255.0);
                    const float cw = buffer[q++];
                    t[2] = t[2] / cw; t[3] = t[3] / cw; t[4] = t[4] / cw;
                }
            }
        }
    });
    if (bad) return fail(OP_ERR_INVALID, "corrupt .map file %s", path);
    OP_TRY(op_volume_clear(v)); // cube_map.clear() (CubeHandler.h:42)
    return op_volume_upload(v, keys.data(), vox, nb);
}

} // extern "C"
