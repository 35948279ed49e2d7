// icp_grid.hip -- the registration context and its search structure (icp_core.hpp lists the translation units): the uniform grid over the target (bbox ->
// cell counts -> exclusive scan -> counting sort into float4 records), op_icp_create / destroy / set_source / options -- and the other users of that grid and
// of the scan: PointCloud::EstimateNormals (exact k-NN over the grid + PCA) and PointCloud::LoadFromDepth / LoadFromRGBD (compaction of valid pixels).
#include "icp_core.hpp"

namespace {

// ---- target grid build ----------------------------------------------------------------------
__global__ void k_bbox(const float* __restrict__ xyz, size_t m, unsigned* __restrict__ box /*max3,min3*/) {
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) {
            const float v = xyz[3 * i + c];
            if (fabsf(v) <= FLT_MAX) { mx[c] = fmaxf(mx[c], v); mn[c] = fminf(mn[c], v); } // NaN and +-inf never enter the grid
        }
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
        }
    }
    // one atomic pair per workgroup and component (six hot addresses: per-wave atomics serialise badly)
    __shared__ float s_mx[4][3], s_mn[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) { s_mx[wave][c] = mx[c]; s_mn[wave][c] = mn[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        float a = s_mx[0][c], b = s_mn[0][c];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { a = fmaxf(a, s_mx[w][c]); b = fminf(b, s_mn[w][c]); }
        atomicMax(&box[c], enc_f(a));
        atomicMin(&box[3 + c], enc_f(b));
    }
}

__global__ void k_cell_count(const float* __restrict__ xyz, size_t m, Grid g, unsigned* __restrict__ count) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX)) return;
    const int cx = cell_coord(x, g.ox, g.inv_cell, g.gx), cy = cell_coord(y, g.oy, g.inv_cell, g.gy),
              cz = cell_coord(z, g.oz, g.inv_cell, g.gz);
    atomicAdd(&count[((size_t)cz * g.gy + cy) * g.gx + cx], 1u);
}

// Exclusive scan of the cell counts in cell order, so that x-adjacent cells own adjacent ranges of
// the sorted target (the 27-cell scan then touches 9 contiguous runs).  Three small kernels:
// per-workgroup totals -> scan of totals (one workgroup) -> per-element offsets.
__global__ __launch_bounds__(256) void k_scan_totals(const unsigned* __restrict__ count, size_t n, unsigned* __restrict__ totals) {
    __shared__ unsigned s[4];
    const size_t base = (size_t)blockIdx.x * kScanWg + threadIdx.x * 4;
    unsigned t = 0;
    for (int k = 0; k < 4; ++k) t += base + k < n ? count[base + k] : 0u;
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) totals[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(1024) void k_scan_of_totals(unsigned* __restrict__ totals, size_t n) {
    // single workgroup, sequential over tiles of 1024
    __shared__ unsigned s[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < n; base += 1024) {
        const size_t i = base + threadIdx.x;
        const unsigned v = i < n ? totals[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
            const unsigned add = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n) totals[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_apply(const unsigned* __restrict__ count, size_t n, const unsigned* __restrict__ totals,
                                                    unsigned* __restrict__ start) {
    __shared__ unsigned s[4];
    const size_t base = (size_t)blockIdx.x * kScanWg + threadIdx.x * 4;
    unsigned c[4], t = 0;
    for (int k = 0; k < 4; ++k) { c[k] = base + k < n ? count[base + k] : 0u; t += c[k]; }
    // exclusive scan of t across the workgroup
    unsigned incl = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s[wave] = incl;
    __syncthreads();
    unsigned off = totals[blockIdx.x] + incl - t;
    for (int w = 0; w < wave; ++w) off += s[w];
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) start[base + k] = off;
        off += c[k];
    }
}

__global__ void k_cell_scatter(const float* __restrict__ xyz, size_t m, Grid g,
                               const unsigned* __restrict__ start, unsigned* __restrict__ left, float4* __restrict__ sorted) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX)) return;
    const int cx = cell_coord(x, g.ox, g.inv_cell, g.gx), cy = cell_coord(y, g.oy, g.inv_cell, g.gy),
              cz = cell_coord(z, g.oz, g.inv_cell, g.gz);
    const size_t c = ((size_t)cz * g.gy + cy) * g.gx + cx;
    // `left` is the cell's count from k_cell_count, counted down: the order inside a cell is arbitrary either way (the search
    // orders candidates by (distance, original index)), and no second per-cell array has to be allocated and zeroed
    const unsigned pos = start[c] + (atomicSub(&left[c], 1u) - 1u);
    sorted[pos] = make_float4(x, y, z, __int_as_float((int)i));
}

// ---- EstimateNormals: exact k-NN over the cell grid + PCA plane fit ----------------------------
// PointCloud::EstimateNormals (PointCloud.cpp:102-144): knn nearest points (nanoflann order:
// ascending squared distance), the prefix with SQUARED distance <= radius (KDTree.h:245-251),
// geometry::FitPlane (Geometry.cpp:172-218).  One thread per (cell-sorted) point; the k best are
// kept sorted in LDS (one column per thread); cells are scanned in growing Chebyshev rings until
// the k-th distance is provably final: every unscanned point is farther than ring * cell.
constexpr int kNrmThreads = 128;
constexpr int kNrmMaxK = 32;

__device__ __forceinline__ void sym3_smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double v[3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off < 1e-300 || off <= 1e-34 * diag) break; // converged to the limit of double
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), sn = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double kp = A[k][p], kq = A[k][q]; A[k][p] = c * kp - sn * kq; A[k][q] = sn * kp + c * kq; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double pk = A[p][k], qk = A[q][k]; A[p][k] = c * pk - sn * qk; A[q][k] = sn * pk + c * qk; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double kp = V[k][p], kq = V[k][q]; V[k][p] = c * kp - sn * kq; V[k][q] = sn * kp + c * kq; }
            }
    }
    // column of the smallest eigenvalue, selected without dynamic indexing (keeps A, V in registers)
    const bool m1 = A[1][1] < A[0][0];
    const double e01 = m1 ? A[1][1] : A[0][0];
    const bool m2 = A[2][2] < e01;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = m2 ? V[k][2] : (m1 ? V[k][1] : V[k][0]);
}

__global__ __launch_bounds__(kNrmThreads) void k_estimate_normals(Grid g, const unsigned* __restrict__ cell_start, const float4* __restrict__ pts,
                                                                  size_t m, int knn, float radius, float cell, float* __restrict__ normals) {
    // The k best candidates of a lane live in its LDS column, UNSORTED while the search runs: a candidate is compared with
    // the worst one kept (its key and slot are in registers) and, if better, overwrites it, after which the column is
    // rescanned for the new worst -- k reads.  Keeping the column sorted instead costs a shift loop per accepted
    // candidate whose trip count is the maximum over the 64 lanes, and some lane accepts almost every candidate: 72 k
    // LDS operations per wave against 11 k here (1.03 -> 0.4 ms at 307 200 points).  Keys are (bits of the squared
    // distance, original index): non-negative floats order like their bit patterns, so "nearer, ties to the smaller
    // index" (nanoflann's order, KDTree.h:245-251) is one unsigned 64-bit compare.  The column is sorted once at the end.
    __shared__ unsigned long long s_key[kNrmMaxK][kNrmThreads];
    __shared__ int s_p[kNrmMaxK][kNrmThreads]; // sorted position of the neighbour (its record is pts[pos])
    const int tid = threadIdx.x;
    const size_t q = blockIdx.x * (size_t)blockDim.x + tid;
    if (q >= m) return;
    const float4 me = pts[q];
    const int cx = cell_coord(me.x, g.ox, g.inv_cell, g.gx), cy = cell_coord(me.y, g.oy, g.inv_cell, g.gy),
              cz = cell_coord(me.z, g.oz, g.inv_cell, g.gz);
    int cnt = 0, worst_slot = 0;
    unsigned long long worst = 0ull;
    const int max_ring = max(g.gx, max(g.gy, g.gz));
    auto offer = [&](const float4& c, unsigned p) __attribute__((always_inline)) {
        const float dx = me.x - c.x, dy = me.y - c.y, dz = me.z - c.z;
        const float d = dx * dx + dy * dy + dz * dz;
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(c.w);
        if (cnt < knn) {
            s_key[cnt][tid] = key; s_p[cnt][tid] = (int)p;
            if (cnt == 0 || key > worst) { worst = key; worst_slot = cnt; }
            ++cnt;
        } else if (key < worst) {
            s_key[worst_slot][tid] = key; s_p[worst_slot][tid] = (int)p;
            worst = 0ull;
            for (int k = 0; k < knn; ++k) {
                const unsigned long long kk = s_key[k][tid];
                if (kk >= worst) { worst = kk; worst_slot = k; }
            }
        }
    };
    for (int ring = 0; ring <= max_ring; ++ring) {
        for (int z = cz - ring; z <= cz + ring; ++z) {
            if (z < 0 || z >= g.gz) continue;
            for (int y = cy - ring; y <= cy + ring; ++y) {
                if (y < 0 || y >= g.gy) continue;
                const bool shell_row = (z == cz - ring || z == cz + ring || y == cy - ring || y == cy + ring);
                // on a shell row scan the whole x run, otherwise only the two x end cells of the ring
                for (int part = 0; part < (shell_row ? 1 : 2); ++part) {
                    int x_lo, x_hi;
                    if (shell_row) { x_lo = cx - ring; x_hi = cx + ring; }
                    else { x_lo = x_hi = part == 0 ? cx - ring : cx + ring; if (ring == 0 && part == 1) continue; }
                    x_lo = max(x_lo, 0); x_hi = min(x_hi, g.gx - 1);
                    if (x_lo > x_hi) continue;
                    const size_t row = ((size_t)z * g.gy + y) * g.gx;
                    const unsigned beg = cell_start[row + x_lo], end = cell_start[row + x_hi + 1]; // exclusive scan incl. the total
                    for (unsigned p = beg; p < end; p += 4) { // four candidates in flight per trip
                        float4 c[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) c[k] = pts[min(p + k, end - 1)];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (p + k < end) offer(c[k], p + k);
                    }
                }
            }
        }
        // every point outside the scanned cube is farther than ring * cell from the query
        const float reach = (float)ring * cell;
        if (cnt == knn && __uint_as_float((unsigned)(worst >> 32)) <= reach * reach) break;
        if (cnt == (int)min((size_t)knn, m) && ring >= max_ring) break;
    }
    // ascending (distance, index): insertion sort of the lane's column
    for (int i = 1; i < cnt; ++i) {
        const unsigned long long key = s_key[i][tid];
        const int pp = s_p[i][tid];
        int j = i;
        while (j > 0) {
            const unsigned long long prev = s_key[j - 1][tid];
            if (!(key < prev)) break;
            s_key[j][tid] = prev; s_p[j][tid] = s_p[j - 1][tid];
            --j;
        }
        s_key[j][tid] = key; s_p[j][tid] = pp;
    }
    int used = 0;
    while (used < cnt && !(__uint_as_float((unsigned)(s_key[used][tid] >> 32)) > radius)) ++used; // squared distance vs radius, as the reference does
    float nx = 0, ny = 0, nz = 0;
    if (used >= 3) {
        float s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < used; ++k) { const float4 c = pts[s_p[k][tid]]; s0 += c.x; s1 += c.y; s2 += c.z; }
        const float m0 = s0 / (float)used, m1 = s1 / (float)used, m2 = s2 / (float)used;
        float w00 = 0, w01 = 0, w02 = 0, w11 = 0, w12 = 0, w22 = 0, w10 = 0, w20 = 0, w21 = 0;
        for (int k = 0; k < used; ++k) {
            const float4 c = pts[s_p[k][tid]];
            const float d0 = c.x - m0, d1 = c.y - m1, d2 = c.z - m2;
            w00 += d0 * d0; w01 += d0 * d1; w02 += d0 * d2; w10 += d1 * d0; w11 += d1 * d1; w12 += d1 * d2;
            w20 += d2 * d0; w21 += d2 * d1; w22 += d2 * d2;
        }
        const float fn = (float)used;
        double v[3];
        sym3_smallest_eigvec((double)(w00 / fn), 0.5 * ((double)(w01 / fn) + (double)(w10 / fn)), 0.5 * ((double)(w02 / fn) + (double)(w20 / fn)),
                             (double)(w11 / fn), 0.5 * ((double)(w12 / fn) + (double)(w21 / fn)), (double)(w22 / fn), v);
        nx = (float)v[0]; ny = (float)v[1]; nz = (float)v[2];
        const float z2 = sum3(nx * nx, ny * ny, nz * nz);
        if (z2 > 0) { const float l = sqrtf(z2); nx /= l; ny /= l; nz /= l; }
    }
    const int orig = __float_as_int(me.w);
    normals[3 * (size_t)orig] = nx; normals[3 * (size_t)orig + 1] = ny; normals[3 * (size_t)orig + 2] = nz;
}

// ---- LoadFromDepth with order-preserving compaction --------------------------------------------
__global__ __launch_bounds__(256) void k_depth_count(const void* __restrict__ depth, int is_u16, float depth_scale, size_t npix,
                                                     unsigned* __restrict__ count) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float z = is_u16 ? (float)((const unsigned short*)depth)[i] / depth_scale : ((const float*)depth)[i];
    count[i] = z > 0 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_depth_scatter(const void* __restrict__ depth, int is_u16, op_camera cam, size_t npix,
                                                       const unsigned* __restrict__ start, float* __restrict__ xyz,
                                                       const unsigned char* __restrict__ rgb, float* __restrict__ colors) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float z = is_u16 ? (float)((const unsigned short*)depth)[i] / cam.depth_scale : ((const float*)depth)[i];
    if (!(z > 0)) return;
    const int r = (int)(i / cam.width), c = (int)(i - (size_t)r * cam.width);
    const unsigned p = start[i];
    xyz[3 * p] = ((float)c - cam.cx) * z / cam.fx; // PointCloud.cpp:90-93
    xyz[3 * p + 1] = ((float)r - cam.cy) * z / cam.fy;
    xyz[3 * p + 2] = z;
    if (colors) { // LoadFromRGBD (PointCloud.cpp:40-42): Point3(b0,b1,b2) / 255.0f in stored channel order
        colors[3 * p] = (float)rgb[3 * i] / 255.0f;
        colors[3 * p + 1] = (float)rgb[3 * i + 1] / 255.0f;
        colors[3 * p + 2] = (float)rgb[3 * i + 2] / 255.0f;
    }
}

} // namespace

namespace opi {

int device_exclusive_scan(const unsigned* d_count, size_t n, unsigned* d_start, hipStream_t stream, unsigned* total_out) {
    const size_t nwg = (n + kScanWg - 1) / kScanWg;
    unsigned* d_tot = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_tot, (nwg + 1) * sizeof(unsigned)));
    hipLaunchKernelGGL(k_scan_totals, dim3((unsigned)nwg), dim3(256), 0, stream, d_count, n, d_tot);
    hipLaunchKernelGGL(k_scan_of_totals, dim3(1), dim3(1024), 0, stream, d_tot, nwg);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nwg), dim3(256), 0, stream, d_count, n, (const unsigned*)d_tot, d_start);
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess && total_out) {
        unsigned last_start = 0, last_count = 0;
        e = hipMemcpy(&last_start, d_start + (n - 1), 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&last_count, d_count + (n - 1), 4, hipMemcpyDeviceToHost);
        *total_out = last_start + last_count;
    }
    op::cached_free(d_tot);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "scan failed: %s", hipGetErrorString(e));
    return OP_OK;
}

void scan_launch(const unsigned* d_count, size_t n, unsigned* d_tot, unsigned* d_start, hipStream_t stream) {
    const size_t nwg = (n + kScanWg - 1) / kScanWg;
    hipLaunchKernelGGL(k_scan_totals, dim3((unsigned)nwg), dim3(256), 0, stream, d_count, n, d_tot);
    hipLaunchKernelGGL(k_scan_of_totals, dim3(1), dim3(1024), 0, stream, d_tot, nwg);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nwg), dim3(256), 0, stream, d_count, n, (const unsigned*)d_tot, d_start);
}

} // namespace opi

extern "C" {

// extent_divisor > 0: the cell is the largest extent of the bounding box / extent_divisor instead of the threshold
// (EstimateNormals' k-NN grid; the box comes from the device either way)
static int icp_create(const float* tgt_xyz, const float* tgt_normals, size_t m, double threshold, double extent_divisor, int mem, int device,
                      op_icp** out) {
    if (!out) return fail(OP_ERR_INVALID, "null out");
    *out = nullptr;
    if (!tgt_xyz && m) return fail(OP_ERR_INVALID, "null target");
    if (!(threshold > 0) && !(extent_divisor > 0)) return fail(OP_ERR_INVALID, "threshold must be > 0");
    if (m >= kMaxPoints) return fail(OP_ERR_INVALID, "target too large (at most %zu points)", kMaxPoints - 1);
    OP_TRY(op::use_device(device));
    op_icp* c = new op_icp();
    c->device = device; c->m = m; c->threshold = threshold; c->has_normals = tgt_normals != nullptr;
    auto bail = [&](int rc) { op_icp_destroy(c); return rc; };
#define OP_HIP_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
    OP_HIP_C(op::cached_stream(&c->stream));
    const size_t m1 = m ? m : 1;
    OP_HIP_C(op::cached_malloc((void**)&c->tgt_orig, m1 * 3 * sizeof(float)));
    OP_HIP_C(op::cached_malloc((void**)&c->tgt, (m + 1) * sizeof(float4))); // + the dummy record of the neighbour scan
    {
        const float inf = std::numeric_limits<float>::infinity();
        const float dummy[4] = {inf, inf, inf, 0.0f};
        OP_HIP_C(hipMemcpy(c->tgt + m, dummy, sizeof(dummy), hipMemcpyHostToDevice));
    }
    float* d_nrm = nullptr;
    const hipMemcpyKind kind = mem == OP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (m) { // device sources: ordered on the context's stream (a device-to-device hipMemcpy does not block the host)
        if (mem == OP_MEM_DEVICE) OP_HIP_C(hipMemcpyAsync(c->tgt_orig, tgt_xyz, m * 3 * sizeof(float), kind, c->stream));
        else OP_HIP_C(hipMemcpy(c->tgt_orig, tgt_xyz, m * 3 * sizeof(float), kind));
    }
    if (c->has_normals) {
        OP_HIP_C(op::cached_malloc((void**)&d_nrm, m1 * 3 * sizeof(float)));
        c->nrm_orig = d_nrm; // owned by the context from here on (freed by op_icp_destroy)
        if (m) {
            if (mem == OP_MEM_DEVICE) OP_HIP_C(hipMemcpyAsync(d_nrm, tgt_normals, m * 3 * sizeof(float), kind, c->stream));
            else OP_HIP_C(hipMemcpy(d_nrm, tgt_normals, m * 3 * sizeof(float), kind));
        }
    }
    // bounding box -> grid
    unsigned* d_box = nullptr;
    OP_HIP_C(op::cached_malloc((void**)&d_box, 6 * sizeof(unsigned)));
    unsigned init[6] = {0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    OP_HIP_C(hipMemcpy(d_box, init, sizeof(init), hipMemcpyHostToDevice));
    if (m) hipLaunchKernelGGL(k_bbox, dim3(128), dim3(256), 0, c->stream, (const float*)c->tgt_orig, m, d_box);
    OP_HIP_C(hipStreamSynchronize(c->stream));
    unsigned box[6];
    OP_HIP_C(hipMemcpy(box, d_box, sizeof(box), hipMemcpyDeviceToHost));
    op::cached_free(d_box);
    float mx[3], mn[3];
    for (int k = 0; k < 3; ++k) { mx[k] = dec_f(box[k]); mn[k] = dec_f(box[3 + k]); }
    if (!m || !(mx[0] >= mn[0])) { for (int k = 0; k < 3; ++k) { mx[k] = 0; mn[k] = 0; } }
    // cell >= threshold (slightly larger so that float rounding of the cell index cannot hide a
    // neighbour closer than threshold); grow it if the grid would exceed kMaxCells
    double cell = threshold * 1.001;
    if (extent_divisor > 0) {
        const float ext = std::max(mx[0] - mn[0], std::max(mx[1] - mn[1], mx[2] - mn[2]));
        cell = ext > 0 ? (double)ext / extent_divisor : 1.0;
        c->threshold = cell;
    }
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(mx[k]) || !std::isfinite(mn[k]) || !std::isfinite((double)mx[k] - (double)mn[k]))
            return bail(fail(OP_ERR_INVALID, "target bounding box is not finite"));
    for (int grow = 0;; ++grow) { // bounded: the extent is finite, so cell *= 1.26 reaches it within ~400 steps of doubles
        double tot = 1;
        for (int k = 0; k < 3; ++k) tot *= std::floor(((double)mx[k] - (double)mn[k]) / cell) + 2.0;
        if (tot <= (double)kMaxCells) break;
        if (grow > 4096 || !std::isfinite(cell)) return bail(fail(OP_ERR_INVALID, "cannot size the search grid (threshold %g)", threshold));
        cell *= 1.26;
    }
    c->grid.ox = mn[0]; c->grid.oy = mn[1]; c->grid.oz = mn[2];
    c->grid.inv_cell = (float)(1.0 / cell);
    c->grid.gx = (int)std::floor((mx[0] - mn[0]) / cell) + 2;
    c->grid.gy = (int)std::floor((mx[1] - mn[1]) / cell) + 2;
    c->grid.gz = (int)std::floor((mx[2] - mn[2]) / cell) + 2;
    c->ncell = (size_t)c->grid.gx * c->grid.gy * c->grid.gz;
    // cell_start = exclusive scan of the per-cell counts over ncell + 4 entries (the padding holds the total), so a
    // run of x-adjacent cells is [cell_start[first], cell_start[last + 1]) and one 16-byte load sees both ends
    const size_t n_tab = c->ncell + 4;
    OP_HIP_C(op::cached_malloc((void**)&c->cell_start, n_tab * sizeof(unsigned)));
    unsigned* d_count = nullptr;
    OP_HIP_C(op::cached_malloc((void**)&d_count, n_tab * sizeof(unsigned)));
    auto drop = [&]() { op::cached_free(d_count); };
    hipError_t e = hipMemsetAsync(d_count, 0, n_tab * sizeof(unsigned), c->stream);
    if (e != hipSuccess) { drop(); return bail(fail(OP_ERR_HIP, "grid build failed: %s", hipGetErrorString(e))); }
    if (m) hipLaunchKernelGGL(k_cell_count, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->tgt_orig, m, c->grid, d_count);
    int rc = device_exclusive_scan(d_count, n_tab, c->cell_start, c->stream, nullptr);
    if (rc != OP_OK) { drop(); return bail(rc); }
    if (m) hipLaunchKernelGGL(k_cell_scatter, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->tgt_orig, m,
                              c->grid, (const unsigned*)c->cell_start, d_count, c->tgt);
    e = hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) return bail(fail(OP_ERR_HIP, "grid build failed: %s", hipGetErrorString(e)));
    OP_HIP_C(op::cached_malloc((void**)&c->sync, (kGroups + 1) * sizeof(unsigned)));
    OP_HIP_C(hipMemset(c->sync, 0, (kGroups + 1) * sizeof(unsigned)));
    OP_HIP_C(op::cached_malloc((void**)&c->result, kNSums * sizeof(double)));
    OP_HIP_C(op::cached_malloc((void**)&c->T_dev, 16 * sizeof(float)));
    OP_HIP_C(op::cached_malloc((void**)&c->stage, (size_t)kGroups * kNSums * sizeof(double)));
    OP_HIP_C(op::cached_host_malloc((void**)&c->result_host, (size_t)kGroups * kNSums * sizeof(double)));
    OP_HIP_C(hipHostGetDevicePointer((void**)&c->result_host_dev, c->result_host, 0));
    std::memset(c->result_host, 0, (size_t)kGroups * kNSums * sizeof(double));
    for (hipEvent_t& ev : c->chunk_ev) OP_HIP_C(op::cached_event(&ev));
#undef OP_HIP_C
    *out = c;
    return OP_OK;
}

int op_icp_create(const float* tgt_xyz, const float* tgt_normals, size_t m, double threshold, int mem, int device, op_icp** out) {
    if (!(threshold > 0)) return fail(OP_ERR_INVALID, "threshold must be > 0");
    OP_TRY(icp_create(tgt_xyz, tgt_normals, m, threshold, 0.0, mem, device, out));
    // OP_RUNTIME_OPT_ICP_DEFAULT_SUMS: the reference's own sequential float32 sums unless the process opted into the fp64 reduction (the mode that is
    // within north_star's 1e-4 of the CPU path on every pair is the default of the drop-in surface)
    (*out)->sums = op::runtime_options().icp_default_sums.load();
    return OP_OK;
}

int op_icp_destroy(op_icp* c) {
    if (c && c->worker_active) { c->worker.join(); c->worker_active = false; }
    if (!c) return OP_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    icp_trace_dump(c); // (-DICP_TRACE development builds: per-wave phase times of the last launch)
    void* ptrs[] = {c->tgt_orig, c->tgt, c->sync, c->cell_start, c->src, c->nn, c->inl, c->partials, c->result,
                    c->T_dev, c->stage, c->nrm_orig, c->flag, c->start, c->scan_tot, c->rows_dev};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (c->result_host) op::cached_free(c->result_host);
    for (hipEvent_t ev : c->chunk_ev)
        op::release_event(ev, c->device);
    if (c->rows_host) op::cached_free(c->rows_host);
    if (c->seq_ev) op::release_event(c->seq_ev, c->device);
    if (c->seq_out) op::cached_free(c->seq_out);
    if (c->seq_total) op::cached_free(c->seq_total);
    if (c->seq_host) op::cached_free(c->seq_host);
    if (c->fin_aux) op::cached_free(c->fin_aux);
    if (c->fin_list) op::cached_free(c->fin_list);
    if (c->tgt_host) op::cached_free(c->tgt_host);
    if (c->nrm_host) op::cached_free(c->nrm_host);
    if (c->tie_count) op::cached_free(c->tie_count);
    if (c->tie_rec) op::cached_free(c->tie_rec);
    if (c->tie_patch) op::cached_free(c->tie_patch);
    op::release_stream(c->stream, c->device);
    delete c;
    return OP_OK;
}

int op_release_cached_memory(void) {
    op::release_cached_memory();
    return OP_OK;
}

int op_icp_set_option(op_icp* c, int option, int value) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_set_option");
    if (option == OP_ICP_OPT_FINISH && (value == OP_ICP_FINISH_REFERENCE || value == OP_ICP_FINISH_FP64)) { c->finish = value; return OP_OK; }
    if (option == OP_ICP_OPT_SUMS && (value == OP_ICP_SUMS_FP64 || value == OP_ICP_SUMS_REFERENCE_F32)) { c->sums = value; return OP_OK; }
    if (option == OP_ICP_OPT_TIES && (value == OP_ICP_TIES_LOWEST_INDEX || value == OP_ICP_TIES_REFERENCE)) { c->ties = value; return OP_OK; }
    return fail(OP_ERR_INVALID, "op_icp_set_option: unknown option %d / value %d", option, value);
}

int op_icp_tie_stats(op_icp* c, uint64_t* tied_queries, uint64_t* changed) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_tie_stats");
    if (tied_queries) *tied_queries = c->tie_queries;
    if (changed) *changed = c->tie_changed;
    return OP_OK;
}

int op_icp_final_stats(op_icp* c, uint64_t* redecided) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_final_stats");
    if (redecided) *redecided = c->fin_redecided;
    return OP_OK;
}

int op_icp_set_source(op_icp* c, const float* src_xyz, size_t n, int mem) {
    if (!c) return fail(OP_ERR_INVALID, "null icp");
    OP_ICP_NOT_BUSY(c, "op_icp_set_source");
    OP_HIP(hipSetDevice(c->device));
    if (!src_xyz && n) return fail(OP_ERR_INVALID, "null source");
    if (n >= kMaxPoints) return fail(OP_ERR_INVALID, "source too large (at most %zu points)", kMaxPoints - 1);
    if (n > c->src_cap) {
        void* old[] = {c->src, c->nn, c->inl, c->partials};
        for (void* p : old)
            if (p) op::cached_free(p);
        c->src = nullptr; c->nn = nullptr; c->inl = nullptr; c->partials = nullptr; c->partials_cap = 0;
        OP_HIP(op::cached_malloc((void**)&c->src, n * 3 * sizeof(float)));
        OP_HIP(op::cached_malloc((void**)&c->nn, n * sizeof(int)));
        OP_HIP(op::cached_malloc((void**)&c->inl, n * sizeof(int)));
        c->src_cap = n;
    }
    c->n = n;
    if (n) { // a device-to-device copy does not block the host: ordered on the context's stream, ahead of the kernels that read it
        if (mem == OP_MEM_DEVICE) OP_HIP(hipMemcpyAsync(c->src, src_xyz, n * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        else OP_HIP(hipMemcpy(c->src, src_xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    }
    int wg = (int)((n + kIterThreads - 1) / kIterThreads); // one source point per thread
    if (wg < 1) wg = 1;
    if (!c->partials || wg > c->partials_cap) {
        op::cached_free(c->partials);
        c->partials = nullptr;
        OP_HIP(op::cached_malloc((void**)&c->partials, (size_t)wg * kNSums * sizeof(double)));
        c->partials_cap = wg;
    }
    c->n_wg = wg;
    return OP_OK;
}

static int points_from_images(const op_camera* cam, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, int device, float* xyz_out,
                              float* colors_out, size_t* n) {
    if (!cam || !depth || !xyz_out || !n || ((rgb == nullptr) != (colors_out == nullptr))) return fail(OP_ERR_INVALID, "null argument");
    if (cam->width <= 0 || cam->height <= 0) return fail(OP_ERR_INVALID, "invalid camera");
    OP_TRY(op::use_device(device));
    const size_t npix = (size_t)cam->width * cam->height;
    const size_t dbytes = npix * (depth_fmt == OP_DEPTH_U16 ? 2 : 4);
    void* d_depth = nullptr;
    unsigned *d_count = nullptr, *d_start = nullptr;
    float *d_xyz = nullptr, *d_col = nullptr;
    unsigned char* d_rgb = nullptr;
    const unsigned char* rsrc = rgb;
    int rc = OP_OK;
    hipError_t e = hipSuccess;
    const void* dsrc = depth;
    if (mem == OP_MEM_HOST) {
        e = op::cached_malloc(&d_depth, dbytes);
        if (e == hipSuccess) e = hipMemcpy(d_depth, depth, dbytes, hipMemcpyHostToDevice);
        dsrc = d_depth;
    }
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_count, npix * 4);
    if (e == hipSuccess) e = op::cached_malloc((void**)&d_start, npix * 4);
    if (e == hipSuccess && mem == OP_MEM_HOST) e = op::cached_malloc((void**)&d_xyz, npix * 12);
    if (e == hipSuccess && mem == OP_MEM_HOST && rgb) {
        e = op::cached_malloc((void**)&d_col, npix * 12);
        if (e == hipSuccess) e = op::cached_malloc((void**)&d_rgb, npix * 3);
        if (e == hipSuccess) e = hipMemcpy(d_rgb, rgb, npix * 3, hipMemcpyHostToDevice);
        rsrc = d_rgb;
    }
    unsigned total = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_depth_count, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, nullptr, dsrc, depth_fmt == OP_DEPTH_U16,
                           cam->depth_scale, npix, d_count);
        rc = device_exclusive_scan(d_count, npix, d_start, nullptr, &total);
        if (rc == OP_OK) {
            float* dst = mem == OP_MEM_HOST ? d_xyz : xyz_out;
            float* cdst = rgb ? (mem == OP_MEM_HOST ? d_col : colors_out) : nullptr;
            hipLaunchKernelGGL(k_depth_scatter, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, nullptr, dsrc, depth_fmt == OP_DEPTH_U16, *cam,
                               npix, (const unsigned*)d_start, dst, rsrc, cdst);
            e = hipDeviceSynchronize();
            if (e == hipSuccess && mem == OP_MEM_HOST && total) e = hipMemcpy(xyz_out, d_xyz, (size_t)total * 12, hipMemcpyDeviceToHost);
            if (e == hipSuccess && mem == OP_MEM_HOST && total && rgb) e = hipMemcpy(colors_out, d_col, (size_t)total * 12, hipMemcpyDeviceToHost);
        }
    }
    if (d_depth) op::cached_free(d_depth);
    if (d_count) op::cached_free(d_count);
    if (d_start) op::cached_free(d_start);
    if (d_xyz) op::cached_free(d_xyz);
    if (d_col) op::cached_free(d_col);
    if (d_rgb) op::cached_free(d_rgb);
    if (rc != OP_OK) return rc;
    if (e != hipSuccess) return fail(OP_ERR_HIP, "points_from_depth failed: %s", hipGetErrorString(e));
    *n = total;
    return OP_OK;
}

int op_points_from_depth(const op_camera* cam, const void* depth, int depth_fmt, int mem, int device, float* xyz_out, size_t* n) {
    return points_from_images(cam, depth, depth_fmt, nullptr, mem, device, xyz_out, nullptr, n);
}

int op_points_from_rgbd(const op_camera* cam, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, int device, float* xyz_out,
                        float* colors_out, size_t* n) {
    if (!rgb || !colors_out) return fail(OP_ERR_INVALID, "null argument");
    return points_from_images(cam, depth, depth_fmt, rgb, mem, device, xyz_out, colors_out, n);
}

int op_estimate_normals(const float* xyz, size_t n, float radius, int knn, int mem, int device, float* normals_out) {
    if (!xyz || !normals_out) return fail(OP_ERR_INVALID, "null argument");
    if (knn < 1 || knn > kNrmMaxK) return fail(OP_ERR_INVALID, "knn must be in [1, %d]", kNrmMaxK);
    if (n == 0) return OP_OK;
    // grid cell = extent / 300: a 640x480 depth cloud of a 6 m room (4-10 mm spacing) gets 2 cm cells, and most points
    // find their 30 neighbours within the first ring (27 cells); measured 1.35 / 1.03 / 1.12 / 1.12 / 1.46 ms for
    // divisors 400 / 300 / 250 / 200 / 150 (tools/ab_normals_cell.sh)
    op_icp* c = nullptr;
    OP_TRY(icp_create(xyz, nullptr, n, 0.0, 300.0, mem, device, &c));
    float* d_nrm = nullptr;
    hipError_t e = op::cached_malloc((void**)&d_nrm, n * 12);
    if (e == hipSuccess) e = hipMemsetAsync(d_nrm, 0, n * 12, c->stream);
    if (e == hipSuccess) {
        const float cell = 1.0f / c->grid.inv_cell;
        hipLaunchKernelGGL(k_estimate_normals, dim3((unsigned)((n + kNrmThreads - 1) / kNrmThreads)), dim3(kNrmThreads), 0, c->stream, c->grid,
                           (const unsigned*)c->cell_start, (const float4*)c->tgt, n, knn, radius, cell, d_nrm);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess && mem == OP_MEM_DEVICE) { // ordered on the stream and finished before d_nrm goes back to the buffer cache
        e = hipMemcpyAsync(normals_out, d_nrm, n * 12, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    } else if (e == hipSuccess) {
        e = hipMemcpy(normals_out, d_nrm, n * 12, hipMemcpyDeviceToHost);
    }
    if (d_nrm) op::cached_free(d_nrm);
    op_icp_destroy(c);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "estimate_normals failed: %s", hipGetErrorString(e));
    return OP_OK;
}

} // extern "C"
