// raycast.hip -- the raycaster north_star names (volume_core.hpp lists the translation units).
#include "volume_core.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// Ray casting (north_star "integrate/raycast").  The reference has NO raycast; the
// definition is this implementation's own and is validated against the analytic synthetic scene:
// march every pixel ray from near to far through the voxel-block hash, sample the sdf
// trilinearly over the 8 surrounding voxel centres (all 8 must be observed, weight > 0), step one
// voxel inside allocated blocks (valid sample or not: IntegrateImage only writes |sdf| < truncation, so the
// free space in front of a surface is unobserved voxels INSIDE allocated blocks and must not be leapt over), jump
// to the exit face of a block that is absent from the hash, and report the first
// + -> - crossing by linear interpolation as z-depth.  Normal = normalised central difference of the
// trilinear sdf (+-res/2), colour = trilinear colour at the hit.  16x16 pixel tiles per workgroup
// keep neighbouring rays -- which walk the same blocks -- on one CU.
// ---------------------------------------------------------------------------------------------
struct BlockCache { int cx, cy, cz, idx; };

template <bool COL>
__device__ __forceinline__ bool rc_fetch(const VolView& V, BlockCache& bc, int px, int py, int pz, Vox5* out) {
    const int cx = px >> 3, cy = py >> 3, cz = pz >> 3;
    if (!(cx == bc.cx && cy == bc.cy && cz == bc.cz)) { bc.cx = cx; bc.cy = cy; bc.cz = cz; bc.idx = table_find(V, cx, cy, cz); }
    if (bc.idx < 0) return false;
    const int vid = (px - cx * 8) + (py - cy * 8) * 8 + (pz - cz * 8) * 64;
    const float* t = V.pool + (size_t)bc.idx * kBlockFloats + vid;
    out->s = t[0]; out->w = t[kVox];
    if (COL) { out->c0 = t[2 * kVox]; out->c1 = t[3 * kVox]; out->c2 = t[4 * kVox]; } // colour planes only at the hit
    else { out->c0 = out->c1 = out->c2 = 0.0f; }
    return out->w > 0;
}

template <bool COL>
__device__ bool rc_sample_t(const VolView& V, BlockCache& bc, float res, float x, float y, float z, float* sdf, float* col) {
    const float gx = x / res - 0.5f, gy = y / res - 0.5f, gz = z / res - 0.5f;
    const float fx0 = floorf(gx), fy0 = floorf(gy), fz0 = floorf(gz);
    const int ix = (int)fx0, iy = (int)fy0, iz = (int)fz0;
    const float fx = gx - fx0, fy = gy - fy0, fz = gz - fz0;
    float acc = 0, a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        Vox5 t;
        if (!rc_fetch<COL>(V, bc, ix + (k & 1), iy + ((k >> 1) & 1), iz + ((k >> 2) & 1), &t)) return false;
        const float wx = (k & 1) ? fx : 1.0f - fx, wy = (k & 2) ? fy : 1.0f - fy, wz = (k & 4) ? fz : 1.0f - fz;
        const float w = (wx * wy) * wz;
        acc += w * t.s; a0 += w * t.c0; a1 += w * t.c1; a2 += w * t.c2;
    }
    *sdf = acc;
    if (COL) { col[0] = a0; col[1] = a1; col[2] = a2; }
    return true;
}
// marching and normal samples read only the sdf and weight planes (2 of the 5)
__device__ __forceinline__ bool rc_sample(const VolView& V, BlockCache& bc, float res, float x, float y, float z, float* sdf, float* col) {
    return col ? rc_sample_t<true>(V, bc, res, x, y, z, sdf, col) : rc_sample_t<false>(V, bc, res, x, y, z, sdf, nullptr);
}

__global__ __launch_bounds__(256) void k_raycast(VolView V, op_camera cam, Mat4 P, float res, float near_d, float far_d,
                                                 float* __restrict__ depth_out, float* __restrict__ normals_out, float* __restrict__ colors_out) {
    const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= cam.width || py >= cam.height) return;
    const size_t pix = (size_t)py * cam.width + px;
    const float dcx = ((float)px - cam.cx) / cam.fx, dcy = ((float)py - cam.cy) / cam.fy;
    const float* M = P.m;
    const float d0 = (M[0] * dcx + M[1] * dcy) + M[2], d1 = (M[4] * dcx + M[5] * dcy) + M[6], d2 = (M[8] * dcx + M[9] * dcy) + M[10];
    const float o0 = M[3], o1 = M[7], o2 = M[11];
    const float fine = res, coarse = res * 8.0f;
    BlockCache bc{INT_MIN, INT_MIN, INT_MIN, -1};
    float t = near_d, t_prev = 0, s_prev = 0, hit = 0;
    bool have_prev = false;
    while (t <= far_d) {
        float sdf;
        if (rc_sample(V, bc, res, o0 + t * d0, o1 + t * d1, o2 + t * d2, &sdf, nullptr)) {
            if (have_prev && s_prev > 0 && sdf <= 0) { hit = t_prev + (t - t_prev) * (s_prev / (s_prev - sdf)); break; }
            have_prev = true; s_prev = sdf; t_prev = t;
            t += fine;
        } else {
            have_prev = false;
            // an invalid sample inside an allocated block is an unobserved voxel: step one voxel.  Only a block that
            // is absent from the hash is skipped, up to its exit face (no valid sample can lie in it: all 8 voxel
            // centres around a point of an absent block cannot be observed)
            const float p0 = o0 + t * d0, p1 = o1 + t * d1, p2 = o2 + t * d2;
            const float b0 = floorf(p0 / coarse), b1 = floorf(p1 / coarse), b2 = floorf(p2 / coarse);
            const int bx = (int)b0, by = (int)b1, bz = (int)b2;
            if (!(bx == bc.cx && by == bc.cy && bz == bc.cz)) { bc.cx = bx; bc.cy = by; bc.cz = bz; bc.idx = table_find(V, bx, by, bz); }
            float step = fine;
            if (bc.idx < 0) {
                float t_exit = FLT_MAX;
                if (d0 > 0) t_exit = fminf(t_exit, ((b0 + 1.0f) * coarse - p0) / d0); else if (d0 < 0) t_exit = fminf(t_exit, (b0 * coarse - p0) / d0);
                if (d1 > 0) t_exit = fminf(t_exit, ((b1 + 1.0f) * coarse - p1) / d1); else if (d1 < 0) t_exit = fminf(t_exit, (b1 * coarse - p1) / d1);
                if (d2 > 0) t_exit = fminf(t_exit, ((b2 + 1.0f) * coarse - p2) / d2); else if (d2 < 0) t_exit = fminf(t_exit, (b2 * coarse - p2) / d2);
                if (t_exit < FLT_MAX) step = fmaxf(fine, t_exit + 0.01f * res);
            }
            t += step;
        }
    }
    depth_out[pix] = hit;
    float n[3] = {0, 0, 0}, c[3] = {0, 0, 0};
    if (hit > 0 && (normals_out || colors_out)) {
        const float x = o0 + hit * d0, y = o1 + hit * d1, z = o2 + hit * d2, h = 0.5f * res;
        float s0;
        if (!rc_sample(V, bc, res, x, y, z, &s0, c)) { c[0] = c[1] = c[2] = 0; }
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float sp = 0, sm = 0;
            if (ok) ok = rc_sample(V, bc, res, x + (a == 0 ? h : 0.0f), y + (a == 1 ? h : 0.0f), z + (a == 2 ? h : 0.0f), &sp, nullptr) &&
                         rc_sample(V, bc, res, x - (a == 0 ? h : 0.0f), y - (a == 1 ? h : 0.0f), z - (a == 2 ? h : 0.0f), &sm, nullptr);
            n[a] = sp - sm;
        }
        const float l2 = sum3(n[0] * n[0], n[1] * n[1], n[2] * n[2]);
        if (ok && l2 > 0) { const float l = sqrtf(l2); n[0] /= l; n[1] /= l; n[2] /= l; } else { n[0] = n[1] = n[2] = 0; }
    }
    if (normals_out) { normals_out[3 * pix] = n[0]; normals_out[3 * pix + 1] = n[1]; normals_out[3 * pix + 2] = n[2]; }
    if (colors_out) { colors_out[3 * pix] = c[0]; colors_out[3 * pix + 1] = c[1]; colors_out[3 * pix + 2] = c[2]; }
}

} // namespace

extern "C" {

int op_volume_raycast(op_volume* v, const op_camera* cam, const float pose[16], float* depth_out, float* normals_out, float* colors_out, int mem) {
    OP_VOL(v);
    if (!pose || !depth_out) return fail(OP_ERR_INVALID, "null argument");
    const op_camera c = cam ? *cam : v->cam;
    OP_TRY(check_cam(&c));
    OP_TRY(vol_check(v));
    const size_t npx = (size_t)c.width * c.height;
    float *d_depth = depth_out, *d_nrm = normals_out, *d_col = colors_out;
    if (mem == OP_MEM_HOST) {
        d_depth = d_nrm = d_col = nullptr;
        OP_HIP(op::cached_malloc((void**)&d_depth, npx * 4));
        if (normals_out) OP_HIP(op::cached_malloc((void**)&d_nrm, npx * 12));
        if (colors_out) OP_HIP(op::cached_malloc((void**)&d_col, npx * 12));
    }
    Mat4 P;
    std::memcpy(P.m, pose, sizeof(P.m));
    hipLaunchKernelGGL(k_raycast, dim3((c.width + 15) / 16, (c.height + 15) / 16), dim3(256), 0, v->stream, v->view(), c, P, v->res, v->near_d,
                       v->far_d, d_depth, d_nrm, d_col);
    hipError_t e = hipStreamSynchronize(v->stream);
    if (mem == OP_MEM_HOST) {
        if (e == hipSuccess) e = hipMemcpy(depth_out, d_depth, npx * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && normals_out) e = hipMemcpy(normals_out, d_nrm, npx * 12, hipMemcpyDeviceToHost);
        if (e == hipSuccess && colors_out) e = hipMemcpy(colors_out, d_col, npx * 12, hipMemcpyDeviceToHost);
        op::cached_free(d_depth);
        if (d_nrm) op::cached_free(d_nrm);
        if (d_col) op::cached_free(d_col);
    }
    if (e != hipSuccess) return fail(OP_ERR_HIP, "raycast failed: %s", hipGetErrorString(e));
    return OP_OK;
}

} // extern "C"
