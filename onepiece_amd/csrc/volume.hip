// volume.hip -- device-resident voxel-block-hashed TSDF volume for gfx950 (MI355X): the host object (create, growth + replay,
// the host-image staging ring, the batching queue) and the C-ABI entry points op_volume_* of include/onepiece_hip.h that fuse
// frames.  The kernels live in one translation unit per family -- select.hip (KA, KB), integrate.hip (KC), volume_ops.hip
// (merge, resampling, point cloud, mesh, files), raycast.hip -- and share volume_core.hpp.
//
// What the volume replaces (file:line under /root/reference/src):
//   integration::CubeHandler::IntegrateImage      Integration/CubeHandler.cpp:197-210
//   CubeHandler::ComputeBounding  (kernel KA)     Integration/CubeHandler.cpp:116-145
//   CubeHandler::PrepareCubes     (kernel KB)     Integration/CubeHandler.cpp:147-196
//   Integrator::GetSDF                            Integration/Integrator.cpp:8-35
//   Integrator::IntegrateImage    (kernel KC)     Integration/Integrator.cpp:36-94
//   TSDFVoxel::operator+                          Integration/TSDFVoxel.h:24-39
//   CubeHandler::Merge            (k_merge_blocks, K4 k_pack_sum/k_unpack_sum)  Integration/CubeHandler.h:145-167
//   CubeHandler::Transform / TransformNearest / GetPointCloud                   Integration/CubeHandler.h:199-338, CubeHandler.cpp:45-69
//   CubeHandler::WriteToFile / ReadFromFile / ReadFromFileFloat                 Integration/CubeHandler.h:40-128
// plus a raycaster that north_star asks for and the reference does not have (raycast.hip).
//
// Data layout in HBM:
//   pool   : max_blocks x [5 planes x 512 floats]; plane order sdf, weight, c0, c1, c2; in-plane
//            index = the reference's voxel id x + 8y + 64z.  A wave64 therefore owns one z-slice
//            and reads/writes 256 contiguous bytes per plane -- fully coalesced, unlike the
//            reference's 20-byte AoS TSDFVoxel.
//   keys   : max_blocks x int32[3] (block id), indexed by pool slot.
//   tkeys / tvals : open-addressing hash table (packed 64-bit block id -> pool slot); the probe start
//            is the low bits of the reference's 64-bit VoxelGridHasher value (Geometry/Geometry.h:101-112).
//   bmask / blist : per table slot, the frames of the current batch that selected the block; the
//            list of slots the batch touched.
//   pimg   : the batch's frames packed as {depth, rgba} per pixel.
// Frames are fused in batches of up to kMaxBatch: KA (prepare + bounding) -> KB (select) -> KC
// (integrate), enqueued without any host synchronisation: every kernel is launched with a fixed
// grid and reads its trip counts from device memory.
//
// Floating point: compiled with -ffp-contract=off; every expression keeps the reference's operand
// order and intermediate types (the CPU restatement in oracle/ is what the tests compare against --
// bit for bit on block selection and, in practice, on every voxel value).
#include "volume_core.hpp"

namespace op {
thread_local char g_last_error[512] = "";
}

namespace op {
RuntimeOptions& runtime_options() { static RuntimeOptions o; return o; }
}

namespace {

// test hook: both forms of the projection for n operand triples (see op_debug_project_uv)
__global__ void k_debug_project_uv(CamParams C, const float* __restrict__ X, const float* __restrict__ Y,
                                   const float* __restrict__ Z, size_t n, int* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pix = C.fast_px ? project_pixel<true>(C, X[i], Y[i], Z[i]) : project_pixel<false>(C, X[i], Y[i], Z[i]);
    out[4 * i] = pix < 0 ? INT_MIN : pix % C.width;
    out[4 * i + 1] = pix < 0 ? INT_MIN : pix / C.width;
    out[4 * i + 2] = px_round_dp((C.fx * X[i]) / Z[i], C.cx);   // the reference's formula with the plain division
    out[4 * i + 3] = px_round_dp((C.fy * Y[i]) / Z[i], C.cy);
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

__global__ void k_clear_table(unsigned long long* tkeys, int* tvals, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        tkeys[i] = kEmptyKey;
        tvals[i] = kPending;
    }
}

// Growth: the n allocated blocks (distinct keys, pool slot = index) re-enter a freshly cleared, larger table.
__global__ void k_rehash(unsigned long long* __restrict__ tkeys, int* __restrict__ tvals, unsigned mask, const int* __restrict__ keys, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
    const unsigned long long key = pack_key(x, y, z);
    for (unsigned s = (unsigned)hash_key_dev(x, y, z) & mask;; s = (s + 1) & mask)
        if (atomicCAS(&tkeys[s], kEmptyKey, key) == kEmptyKey) { tvals[s] = (int)i; return; }
}

// After a select-only launch (PrepareCubes API): clear the batch masks again and translate the
// recorded table slots into pool slots.
__global__ void k_has_cube(VolView V, int x, int y, int z, int* out) { *out = table_find(V, x, y, z) >= 0 ? 1 : 0; }

} // namespace

namespace opv {

int vol_ring_send(op_volume* v, op_volume::RingSlot& r); // the DMA of the staged host frames that have not been sent yet

// Every block with a pool slot below `bound` may hold data k_integrate did not write (see its PLAIN comment).
void vol_mark_foreign(op_volume* v, unsigned long long bound) {
    v->plain = false;
    ++v->content_gen;
    const unsigned b = bound > 0xffffffffull ? 0xffffffffu : (unsigned)bound;
    if (b > v->plain_from) v->plain_from = b;
}

int vol_reset(op_volume* v) {
    ++v->generation; ++v->content_gen;
    hipLaunchKernelGGL(k_clear_table, dim3(1024), dim3(256), 0, v->stream, v->tkeys, v->tvals, (size_t)v->table_size);
    OP_HIP(hipMemsetAsync(v->n_blocks, 0, sizeof(unsigned), v->stream));
    OP_HIP(hipMemsetAsync(v->bmask, 0, sizeof(bmask_t) * (size_t)v->table_size, v->stream));
    OP_HIP(hipMemsetAsync(v->state, 0, sizeof(State), v->stream));
    OP_HIP(hipMemsetAsync(v->upd_partial, 0, sizeof(unsigned long long) * kPartialGrid, v->stream));
    OP_HIP(hipMemsetAsync(v->sel_partial, 0, sizeof(unsigned long long) * kPartialGrid, v->stream));
    OP_HIP(hipMemsetAsync(v->chg_partial, 0, sizeof(unsigned long long) * 2 * kPartialGrid, v->stream));
    OP_HIP(hipGetLastError());
    return OP_OK;
}


int vol_enqueue_batch(op_volume* v, const BatchFwd& F, const BatchInv& I, const BatchPtrs& Q, int nf, int depth_fmt, bool select_only, bool record,
                      const int* cube_keys = nullptr, unsigned n_cubes = 0);

// Grows the pool to new_max blocks (and the hash table to twice that), keeping the first n_valid blocks.  The stream
// must be idle.  The new buffers are allocated before the old ones are released, so a failed allocation leaves the
// volume intact (OP_ERR_CAPACITY).  CubeMap growth in the reference is std::unordered_map's (CubeHandler.h:22).
int vol_grow(op_volume* v, unsigned long long want, unsigned n_valid) {
    if (want > kHardMaxBlocks) want = kHardMaxBlocks;
    if (want <= v->max_blocks) return fail(OP_ERR_CAPACITY, "volume cannot grow beyond %u blocks", v->max_blocks);
    const unsigned new_max = (unsigned)want, new_table = next_pow2(2ull * new_max);
    float* pool = nullptr;
    int *keys = nullptr, *blist = nullptr, *sel_list = nullptr, *tvals = nullptr;
    unsigned long long *sel_cand = nullptr, *tkeys = nullptr;
    bmask_t* bmask = nullptr;
    hipError_t e = op::cached_malloc((void**)&pool, sizeof(float) * kBlockFloats * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&keys, sizeof(int) * 3 * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&blist, sizeof(int) * (size_t)(KC_BANDS ? kBands : 1) * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&sel_list, sizeof(int) * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&sel_cand, sizeof(unsigned long long) * (size_t)new_max);
    if (e == hipSuccess) e = op::cached_malloc((void**)&tkeys, sizeof(unsigned long long) * (size_t)new_table);
    if (e == hipSuccess) e = op::cached_malloc((void**)&tvals, sizeof(int) * (size_t)new_table);
    if (e == hipSuccess) e = op::cached_malloc((void**)&bmask, sizeof(bmask_t) * (size_t)new_table);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        void* got[] = {pool, keys, blist, sel_list, sel_cand, tkeys, tvals, bmask};
        for (void* q : got)
            if (q) op::cached_free(q);
        return fail(OP_ERR_CAPACITY, "cannot grow the volume to %u blocks: %s", new_max, hipGetErrorString(e));
    }
    if (n_valid) {
        OP_HIP(hipMemcpyAsync(pool, v->pool, sizeof(float) * kBlockFloats * (size_t)n_valid, hipMemcpyDeviceToDevice, v->stream));
        OP_HIP(hipMemcpyAsync(keys, v->keys, sizeof(int) * 3 * (size_t)n_valid, hipMemcpyDeviceToDevice, v->stream));
    }
    hipLaunchKernelGGL(k_fill_pool, dim3(4096), dim3(256), 0, v->stream, pool, (size_t)n_valid, (size_t)(new_max - n_valid));
    hipLaunchKernelGGL(k_clear_table, dim3(1024), dim3(256), 0, v->stream, tkeys, tvals, (size_t)new_table);
    OP_HIP(hipMemsetAsync(bmask, 0, sizeof(bmask_t) * (size_t)new_table, v->stream));
    if (n_valid) hipLaunchKernelGGL(k_rehash, dim3((n_valid + 255) / 256), dim3(256), 0, v->stream, tkeys, tvals, new_table - 1, (const int*)keys, n_valid);
    OP_HIP(hipMemcpyAsync(v->n_blocks, &n_valid, sizeof(unsigned), hipMemcpyHostToDevice, v->stream));
    OP_HIP(hipGetLastError());
    OP_HIP(hipStreamSynchronize(v->stream)); // n_valid is a stack variable; the old buffers are released next
    void* old[] = {v->pool, v->keys, v->blist, v->sel_list, v->sel_cand, v->tkeys, v->tvals, v->bmask};
    for (void* q : old)
        if (q) op::cached_free(q);
    v->pool = pool; v->keys = keys; v->blist = blist; v->sel_list = sel_list; v->sel_cand = sel_cand;
    v->tkeys = tkeys; v->tvals = tvals; v->bmask = bmask;
    v->max_blocks = new_max; v->table_size = new_table;
    ++v->n_grows; ++v->generation; ++v->content_gen;
    return OP_OK;
}

// Makes room for `need` blocks in total (upload / merge / unpack know their demand up front).  Synchronises when it grows.
int vol_reserve(op_volume* v, unsigned long long need) {
    if (need <= v->max_blocks) return OP_OK;
    if (need > kHardMaxBlocks) return fail(OP_ERR_CAPACITY, "%llu blocks exceed the limit of %u blocks per volume", need, kHardMaxBlocks);
    OP_HIP(hipStreamSynchronize(v->stream));
    unsigned n = 0;
    OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
    if (n > v->max_blocks) n = v->max_blocks;
    return vol_grow(v, std::max<unsigned long long>(next_pow2(need), 2ull * v->max_blocks), n);
}

// Retires log entries (and their staging slots) of batches the device has reported complete.  Never blocks.
void vol_retire(op_volume* v) {
    if (!v->hstat) return;
    const unsigned lo = __atomic_load_n(&v->hstat[0], __ATOMIC_ACQUIRE);
    // widen the 32-bit report: it can only lie at or behind the last launched batch, and less than 2^31 behind
    uint64_t done = (v->seq & ~0xffffffffull) | lo;
    if (done > v->seq) done -= 1ull << 32;
    while (!v->log.empty() && v->log.front().seq <= done) {
        const int rs = v->log.front().ring_slot;
        if (rs >= 0 && v->ring[rs].busy_seq == v->log.front().seq) v->ring[rs].busy_seq = 0;
        v->log.pop_front();
    }
}

// The stream is idle.  If a batch exhausted the pool or the table: grow, then replay everything from that batch on (the
// failing batch's KC and all later batches did nothing).  Loops until the log has gone through.  Other overflow bits
// (bad frames) are left for the caller to report.
int vol_recover(op_volume* v, unsigned* flags_out) {
    for (;;) {
        StateHead st; // the head only: State is ~20 KB, and a pageable device-to-host copy of 16 KB or more takes the runtime's pinned-staging path (ms)
        OP_HIP(hipMemcpy(&st, v->state, sizeof(st), hipMemcpyDeviceToHost));
        if (flags_out) *flags_out = st.overflow;
        if ((st.overflow & 3u) == 0u) {
            for (auto& r : v->log)
                if (r.ring_slot >= 0 && v->ring[r.ring_slot].busy_seq == r.seq) v->ring[r.ring_slot].busy_seq = 0;
            v->log.clear();
            return OP_OK;
        }
        unsigned n = 0;
        OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
        if (n > v->max_blocks) n = v->max_blocks; // claims beyond the pool were marked dead; the blocks below it are real
        // first failing batch: widen the device's 32-bit sequence number like vol_retire does
        uint64_t fail_seq = (v->seq & ~0xffffffffull) | st.fail_seq;
        if (fail_seq > v->seq) fail_seq -= 1ull << 32;
        int rc = vol_grow(v, 2ull * v->max_blocks, n);
        if (rc != OP_OK) { // cannot grow: report, drop the frames that cannot be fused
            v->log.clear();
            for (auto& r : v->ring) r.busy_seq = 0;
            const unsigned keep = st.overflow & ~3u;
            OP_HIP(hipMemcpy(&v->state->overflow, &keep, sizeof(keep), hipMemcpyHostToDevice));
            return rc;
        }
        const unsigned keep = st.overflow & ~3u;
        OP_HIP(hipMemcpy(&v->state->overflow, &keep, sizeof(keep), hipMemcpyHostToDevice));
        std::deque<op_volume::BatchRec> replay;
        replay.swap(v->log);
        while (!replay.empty() && replay.front().seq < fail_seq) replay.pop_front(); // those completed
        v->recovering = true;
        int rrc = OP_OK;
        for (auto& r : replay) {
            rrc = vol_enqueue_batch(v, r.F, r.I, r.P, r.nf, r.fmt, false, false); // assigns a new sequence number
            ++v->n_replayed;
            if (rrc != OP_OK) break;
            if (r.ring_slot >= 0) v->ring[r.ring_slot].busy_seq = v->seq;
            v->log.back().ring_slot = r.ring_slot;
        }
        v->recovering = false;
        OP_TRY(rrc);
        OP_HIP(hipStreamSynchronize(v->stream));
    }
}

// Flushes, synchronises, grows + replays if needed, and reports frames that cannot be fused at all.
int vol_check(op_volume* v) {
    OP_TRY(vol_flush(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    unsigned of = 0;
    OP_TRY(vol_recover(v, &of));
    if (of & 12u) { // report once, then clear: the offending frames selected nothing, everything else was fused
        const unsigned zero = 0;
        OP_HIP(hipMemcpy(&v->state->overflow, &zero, sizeof(zero), hipMemcpyHostToDevice));
    }
    if (of & 4u) return fail(OP_ERR_INVALID, "frame bounding box spans more than 4096 blocks on an axis");
    if (of & 8u) return fail(OP_ERR_INVALID, "block coordinate outside +-2^20 (not representable in the device hash key)");
    return OP_OK;
}

int vol_block_count(op_volume* v, unsigned* n) {
    OP_TRY(vol_check(v));
    OP_HIP(hipMemcpy(n, v->n_blocks, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (*n > v->max_blocks) *n = v->max_blocks;
    return OP_OK;
}

// Single-frame scratch of the synchronous calls (ComputeBounding / PrepareCubes): host images are copied to the device
// stream-ordered; device images are used in place.  (op_volume_integrate stages through the pinned ring below.)
int vol_stage_images(op_volume* v, const void** depth, int depth_fmt, const unsigned char** rgb, int mem, int slot = 0) {
    if (mem == OP_MEM_DEVICE) return OP_OK;
    const size_t npx = (size_t)v->cam.width * v->cam.height;
    if (npx > v->img_cap_px) {
        OP_TRY(vol_flush(v));
        OP_HIP(hipStreamSynchronize(v->stream));
        if (v->img_depth) op::cached_free(v->img_depth);
        if (v->img_rgb) op::cached_free(v->img_rgb);
        v->img_depth = nullptr; v->img_rgb = nullptr; v->img_cap_px = 0;
        OP_HIP(op::cached_malloc(&v->img_depth, npx * 4));
        OP_HIP(op::cached_malloc((void**)&v->img_rgb, npx * 3));
        v->img_cap_px = npx;
    }
    void* d = (char*)v->img_depth + (size_t)slot * npx * 4;
    OP_HIP(hipMemcpyAsync(d, *depth, npx * (depth_fmt == OP_DEPTH_U16 ? 2 : 4), hipMemcpyHostToDevice, v->stream));
    *depth = d;
    if (rgb && *rgb) {
        unsigned char* c = v->img_rgb + (size_t)slot * npx * 3;
        OP_HIP(hipMemcpyAsync(c, *rgb, npx * 3, hipMemcpyHostToDevice, v->stream));
        *rgb = c;
    }
    return OP_OK;
}

CamParams cam_params(const op_volume* v, int depth_fmt) {
    CamParams C;
    C.fx = v->cam.fx; C.fy = v->cam.fy; C.cx = v->cam.cx; C.cy = v->cam.cy;
    C.depth_scale = v->cam.depth_scale; C.res = v->res; C.trunc = v->trunc;
    C.width = v->cam.width; C.height = v->cam.height; C.depth_u16 = depth_fmt == OP_DEPTH_U16;
    C.ax = px_axis(C.cx, C.width); C.ay = px_axis(C.cy, C.height);
    C.fast_px = C.ax.exact && C.ay.exact;
    return C;
}

// per-frame host work of the path: frustum planes (Frustum.cpp:7-46) and pose^-1 (Integrator.cpp:48)
void frame_params(const op_volume* v, const float pose[16], const float* pose_inv, PoseFwd* fwd, PoseInv* inv) {
    std::memcpy(fwd->pose, pose, sizeof(fwd->pose));
    op_host::CameraPOD c{v->cam.fx, v->cam.fy, v->cam.cx, v->cam.cy, v->cam.width, v->cam.height, v->cam.depth_scale};
    op_host::frustum_planes(c, pose, v->far_d, v->near_d, fwd->planes);
    float full[16];
    if (pose_inv) std::memcpy(full, pose_inv, sizeof(full));
    else op_host::mat4_inverse(pose, full);
    std::memcpy(inv->m, full, sizeof(inv->m));
}

int vol_ensure_frame_buffers(op_volume* v) {
    const size_t npx = (size_t)v->cam.width * v->cam.height;
    if (npx <= v->pimg_px && v->cam.width == v->pimg_w && v->cam.height == v->pimg_h) return OP_OK; // (KA's grid and the tile grid depend on both)
    OP_HIP(hipStreamSynchronize(v->stream)); // released buffers go back to a cache and may be handed out at once
    if (v->pimg) op::cached_free(v->pimg);
    if (v->partial) op::cached_free(v->partial);
    if (v->ptile) op::cached_free(v->ptile);
    v->pimg = nullptr; v->partial = nullptr; v->ptile = nullptr; v->pimg_px = 0; v->pimg_w = v->pimg_h = 0;
    const size_t g1 = (size_t)ka_grid(v->cam.width, v->cam.height);
    OP_HIP(op::cached_malloc((void**)&v->pimg, (size_t)kMaxBatch * npx * sizeof(uint2)));
    OP_HIP(op::cached_malloc((void**)&v->partial, (size_t)kMaxBatch * g1 * 8 * sizeof(float)));
    OP_HIP(op::cached_malloc((void**)&v->ptile, (size_t)kMaxBatch * tiles_w(v->cam.width) * tiles_h(v->cam.height) * sizeof(float2)));
    v->pimg_px = npx; v->pimg_w = v->cam.width; v->pimg_h = v->cam.height;
    return OP_OK;
}

// Enqueue one batch (1..kMaxBatch frames whose images are on the device): KA, KB and, unless
// select_only, KC.  No host synchronisation.
// With cube_keys (device array of n_cubes ids, nf == 1): KB is replaced by k_mark_cubes -- the frame is fused into exactly
// those cubes (op_volume_integrate_cubes); the caller has reserved the room and synchronises, so the batch is not logged.
int vol_enqueue_batch(op_volume* v, const BatchFwd& F, const BatchInv& I, const BatchPtrs& Q, int nf, int depth_fmt, bool select_only, bool record,
                      const int* cube_keys, unsigned n_cubes) {
    OP_TRY(vol_ensure_frame_buffers(v));
    // Look at the device's lagging progress report (no synchronisation): retire confirmed batches and, when the pool is
    // about to run full, grow it NOW -- between batches -- instead of paying for a replay later.
    vol_retire(v);
    if (v->hstat && !v->recovering && !v->grow_refused && (unsigned long long)v->hstat[1] + v->max_blocks / 8u > v->max_blocks && v->max_blocks < kHardMaxBlocks) {
        OP_HIP(hipStreamSynchronize(v->stream));
        unsigned of = 0;
        OP_TRY(vol_recover(v, &of)); // also handles an overflow that has already happened
        unsigned n = 0;
        OP_HIP(hipMemcpy(&n, v->n_blocks, sizeof(n), hipMemcpyDeviceToHost));
        if ((unsigned long long)n + v->max_blocks / 8u > v->max_blocks) {
            const int rc = vol_grow(v, 2ull * v->max_blocks, n < v->max_blocks ? n : v->max_blocks);
            if (rc != OP_OK && rc != OP_ERR_CAPACITY) return rc; // out of memory is not fatal yet: the pool may still suffice
            if (rc == OP_ERR_CAPACITY) v->grow_refused = true;
        }
        v->hstat[1] = n;
    }
    const unsigned seq = (unsigned)(++v->seq);
    ++v->generation;
    if (select_only || !vol_fusion_keeps_summaries(v)) ++v->content_gen; // (otherwise k_integrate restates the raycaster's summaries of what it changes)
    if (!select_only && !cube_keys) v->log.push_back(op_volume::BatchRec{v->seq, F, I, Q, nf, depth_fmt, -1});
    const CamParams C = cam_params(v, depth_fmt);
    const bool sample = !select_only && v->prof_every > 0 && (v->prof_batch++ % (uint64_t)v->prof_every) == 0 &&
                        v->prof_events.size() < 4 * 65536;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (sample) {
        for (auto& e : ev) OP_HIP(hipEventCreate(&e));
        OP_HIP(hipEventRecord(ev[0], v->stream));
    }
    launch_prepare_frames(v, F, nf, C, Q, seq);
    if (sample) OP_HIP(hipEventRecord(ev[1], v->stream));
    OP_TRY(launch_select(v, I, C, nf, record, cube_keys, n_cubes));
    if (sample) OP_HIP(hipEventRecord(ev[2], v->stream));
    if (select_only) launch_finish_select(v);
    else launch_integrate(v, I, C, nf);
    if (sample) {
        OP_HIP(hipEventRecord(ev[3], v->stream));
        for (auto e : ev) v->prof_events.push_back(e);
        v->prof_frames.push_back(nf);
    }
    OP_HIP(hipGetLastError());
    return OP_OK;
}

// launch the frames queued by op_volume_integrate
int vol_flush(op_volume* v) {
    if (v->pend_n == 0) return OP_OK;
    const int nf = v->pend_n, rs = v->ring_cur;
    v->pend_n = 0;
    v->ring_cur = -1;
    if (rs >= 0) { // the batch's kernels start when its host images have arrived
        OP_TRY(vol_ring_send(v, v->ring[rs]));
        OP_HIP(hipEventRecord(v->ring[rs].copied, v->copy_stream));
        OP_HIP(hipStreamWaitEvent(v->stream, v->ring[rs].copied, 0));
    }
    OP_TRY(vol_enqueue_batch(v, v->pend_F, v->pend_I, v->pend_P, nf, v->pend_fmt, false, false));
    if (rs >= 0) { v->ring[rs].busy_seq = v->seq; v->log.back().ring_slot = rs; }
    return OP_OK;
}

// ---- host-image staging ring (op_volume_integrate with OP_MEM_HOST) ------------------------------------------------
// Helper threads for the pageable -> pinned copy: one host core moves ~10-15 GB/s, i.e. 2.1 MB frames at 5-7 k frames/s;
// the caller's thread plus two helpers share the chunks of a frame.  The helpers spin briefly after work (calls arrive
// every < 100 us while a sequence is being fused) and sleep on a condition variable when idle.
class CopyPool {
  public:
    static CopyPool& get() { static CopyPool p; return p; }
    void copy2(void* d0, const void* s0, size_t n0, void* d1, const void* s1, size_t n1) {
        if (helpers_.empty() || n0 + n1 < (1u << 18)) { std::memcpy(d0, s0, n0); if (n1) std::memcpy(d1, s1, n1); return; }
        std::lock_guard<std::mutex> call(call_mutex_); // one copy at a time (volumes on several threads share the pool)
        Job job[2] = {{(char*)d0, (const char*)s0, n0}, {(char*)d1, (const char*)s1, n1}};
        const size_t chunks = (n0 + kChunk - 1) / kChunk + (n1 + kChunk - 1) / kChunk;
        uint64_t epoch;
        {   // publish the job under the lock the helpers take their snapshot under: a helper sees all of it or none
            std::lock_guard<std::mutex> lk(m_);
            job_[0] = job[0]; job_[1] = job[1];
            total_ = chunks;
            epoch = ++epoch_;
            done_.store(0, std::memory_order_relaxed);
            next_.store((epoch & 0xffffffffull) << 32, std::memory_order_release); // chunk counter tagged with the epoch
            epoch_hint_.store(epoch_, std::memory_order_release);
        }
        cv_.notify_all();
        work(epoch, job, chunks);
        // Every chunk is DRAWN with a compare-exchange on the epoch-tagged counter and counted in done_ after its memcpy, so
        // done_ == chunks means nobody is still copying.  A helper that wakes up late holds a snapshot of THIS job and this
        // epoch: its first draw fails against the next job's tag and it leaves without touching anything.
        while (done_.load(std::memory_order_acquire) < chunks) __builtin_ia32_pause();
    }
  private:
    static constexpr size_t kChunk = 1u << 17;
    struct Job { char* d; const char* s; size_t n; };
    CopyPool() {
        int n = 2;
        n = op::runtime_options().copy_threads.load(); // OP_RUNTIME_OPT_COPY_THREADS
        if ((int)std::thread::hardware_concurrency() <= 2) n = 0;
        for (int i = 0; i < n && i < 8; ++i) helpers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++epoch_; epoch_hint_.store(epoch_, std::memory_order_release); }
        cv_.notify_all();
        for (auto& t : helpers_) t.join();
    }
    // copies chunks of the job of `epoch` (the caller's own copy of it) until none is left or the shared counter has moved on
    void work(uint64_t epoch, const Job job[2], size_t total) {
        const size_t c0 = (job[0].n + kChunk - 1) / kChunk;
        const uint64_t tag = (epoch & 0xffffffffull) << 32;
        for (;;) {
            uint64_t v = next_.load(std::memory_order_acquire);
            for (;;) {
                if ((v & 0xffffffff00000000ull) != tag || (v & 0xffffffffull) >= total) return;
                if (next_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) break;
            }
            const size_t c = (size_t)(v & 0xffffffffull);
            const Job& j = c < c0 ? job[0] : job[1];
            const size_t off = (c < c0 ? c : c - c0) * kChunk;
            std::memcpy(j.d + off, j.s + off, std::min(kChunk, j.n - off));
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    // takes a snapshot of a new epoch's job under the lock, or reports that there is none / that it is time to stop
    int poll(uint64_t& seen, Job job[2], size_t& total) {
        if (epoch_hint_.load(std::memory_order_acquire) == seen) return 0; // nothing new: do not touch the lock while spinning
        std::lock_guard<std::mutex> lk(m_);
        if (stop_) return -1;
        if (epoch_ == seen) return 0;
        seen = epoch_;
        job[0] = job_[0]; job[1] = job_[1]; total = total_;
        return 1;
    }
    void loop() {
        uint64_t seen = 0;
        Job job[2] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
        size_t total = 0;
        for (;;) {
            // spin for a while (a frame arrives every < 100 us during a sequence), then sleep
            int got = 0;
            for (int spin = 0; spin < 20000 && got == 0; ++spin) {
                got = poll(seen, job, total);
                if (got == 0) __builtin_ia32_pause();
            }
            if (got == 0) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                continue; // poll() takes the snapshot under the lock
            }
            if (got < 0) return;
            work(seen, job, total);
        }
    }
    std::vector<std::thread> helpers_;
    std::mutex m_, call_mutex_;
    std::condition_variable cv_;
    uint64_t epoch_ = 0;
    bool stop_ = false;
    Job job_[2];
    size_t total_ = 0;
    std::atomic<uint64_t> next_{0};       // (epoch & 0xffffffff) << 32 | next chunk index
    std::atomic<size_t> done_{0};
    std::atomic<uint64_t> epoch_hint_{0}; // lock-free mirror of epoch_ for the spinning helpers
};

// op_device_write: pageable host memory -> device memory.  A plain hipMemcpy from pageable memory makes the runtime pin the caller's pages for
// the transfer (0.5-0.7 ms for a 0.9 MB image); like the volume's staging ring this goes through a pinned buffer instead -- the parts are
// copied next to each other by the caller's thread and the CopyPool helpers, then ONE DMA moves the span they cover.
int write_staged(void* dst, size_t n_parts, const void* const* parts, const size_t* bytes, const size_t* offsets, int device) {
    struct Stage { std::mutex mu; void* pinned = nullptr; size_t cap = 0; hipStream_t stream = nullptr; };
    constexpr int kStageDevices = 64;
    static Stage stage[kStageDevices]; // one staging buffer and stream per device, keyed by the device's own number
    if (device < 0 || device >= kStageDevices) return fail(OP_ERR_INVALID, "op_device_write: device %d (0 .. %d)", device, kStageDevices - 1);
    Stage& S = stage[device];
    size_t lo = (size_t)-1, hi = 0;
    for (size_t i = 0; i < n_parts; ++i) {
        if (!parts[i] || !bytes[i]) return fail(OP_ERR_INVALID, "op_device_write: empty part");
        if (offsets[i] + bytes[i] < offsets[i]) return fail(OP_ERR_INVALID, "op_device_write: offset + size overflows");
        lo = std::min(lo, offsets[i]); hi = std::max(hi, offsets[i] + bytes[i]);
    }
    if (n_parts == 0 || n_parts > 2) return fail(OP_ERR_INVALID, "op_device_write: one or two parts per call");
    const size_t span = hi - lo;
    std::lock_guard<std::mutex> lock(S.mu);
    hipError_t e = hipSuccess;
    if (S.cap < span) {
        if (S.pinned) (void)hipHostFree(S.pinned);
        S.pinned = nullptr; S.cap = 0;
        const size_t want = span < (4u << 20) ? (4u << 20) : span;
        e = hipHostMalloc(&S.pinned, want, hipHostMallocDefault);
        if (e == hipSuccess) S.cap = want;
    }
    if (e == hipSuccess && !S.stream) e = hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        char* h = (char*)S.pinned;
        CopyPool::get().copy2(h + (offsets[0] - lo), parts[0], bytes[0], n_parts > 1 ? h + (offsets[1] - lo) : nullptr, n_parts > 1 ? parts[1] : nullptr,
                              n_parts > 1 ? bytes[1] : 0);
        e = hipMemcpyAsync((char*)dst + lo, S.pinned, span, hipMemcpyHostToDevice, S.stream); // (a gap between the parts carries staging leftovers: it is the caller's padding)
        if (e == hipSuccess) e = hipStreamSynchronize(S.stream); // complete on return: the staging buffer is free again, the data is visible to every stream
    }
    if (e != hipSuccess) return fail(OP_ERR_HIP, "op_device_write: %s", hipGetErrorString(e));
    return OP_OK;
}

int vol_ring_alloc(op_volume* v) {
    const size_t npx = (size_t)v->cam.width * v->cam.height;
    if (v->ring_px >= npx && v->copy_stream) return OP_OK;
    OP_TRY(vol_check(v)); // nothing in flight may still read the old slots
    if (!v->copy_stream) OP_HIP(op::cached_stream(&v->copy_stream));
    for (auto& r : v->ring) {
        if (r.d_depth) op::cached_free(r.d_depth);
        if (r.d_rgb) op::cached_free(r.d_rgb);
        if (r.h_depth) op::cached_free(r.h_depth);
        if (r.h_rgb) op::cached_free(r.h_rgb);
        r.d_depth = r.h_depth = nullptr; r.d_rgb = r.h_rgb = nullptr; r.busy_seq = 0;
        OP_HIP(op::cached_malloc(&r.d_depth, (size_t)kMaxBatch * npx * 4));
        OP_HIP(op::cached_malloc((void**)&r.d_rgb, (size_t)kMaxBatch * npx * 3));
        OP_HIP(op::cached_host_malloc(&r.h_depth, (size_t)kMaxBatch * npx * 4));
        OP_HIP(op::cached_host_malloc((void**)&r.h_rgb, (size_t)kMaxBatch * npx * 3));
        if (!r.copied) OP_HIP(hipEventCreateWithFlags(&r.copied, hipEventDisableTiming));
    }
    v->ring_px = npx;
    v->ring_cur = -1;
    return OP_OK;
}

// The DMA of the staged positions that have not been sent: consecutive frames go as ONE span per image kind -- a copy of ~1 MB runs at 33 GB/s over
// this PCIe link, one of >= 4 MB at 53 GB/s (tests/tools/pcie_probe.py).
#ifndef OP_RING_DMA_FRAMES
#define OP_RING_DMA_FRAMES 8
#endif
int vol_ring_send(op_volume* v, op_volume::RingSlot& r) {
    if (r.dma_hi <= r.dma_lo) return OP_OK;
    const size_t npx = (size_t)v->cam.width * v->cam.height, cbytes = npx * 3, n = (size_t)(r.dma_hi - r.dma_lo);
    OP_HIP(hipMemcpyAsync((char*)r.d_depth + (size_t)r.dma_lo * r.dma_dbytes, (const char*)r.h_depth + (size_t)r.dma_lo * r.dma_dbytes, n * r.dma_dbytes, hipMemcpyHostToDevice, v->copy_stream));
    OP_HIP(hipMemcpyAsync(r.d_rgb + (size_t)r.dma_lo * cbytes, r.h_rgb + (size_t)r.dma_lo * cbytes, n * cbytes, hipMemcpyHostToDevice, v->copy_stream));
    r.dma_lo = r.dma_hi;
    return OP_OK;
}

// Stages one host frame into position `pos` of the batch being assembled: pageable -> pinned on the host (parallel); the
// asynchronous DMA on the copy stream follows when OP_RING_DMA_FRAMES consecutive frames are waiting, or when the batch is launched.
// The caller's buffers are free again when this returns.  (A batch has one depth format: frames lie dbytes apart.)
int vol_ring_stage(op_volume* v, const void** depth, int depth_fmt, const unsigned char** rgb, int pos) {
    OP_TRY(vol_ring_alloc(v));
    if (v->ring_cur < 0) { // first host frame of this batch: take the next slot
        const int rs = (int)(v->ring_next++ % (unsigned)op_volume::kRing);
        vol_retire(v);
        if (v->ring[rs].busy_seq != 0) { // its batch is not confirmed yet (only when the GPU is the bottleneck)
            const int keep_n = v->pend_n;
            v->pend_n = 0;                 // vol_check must not flush the half-assembled batch
            const int rc = vol_check(v);
            v->pend_n = keep_n;
            OP_TRY(rc);
        }
        v->ring_cur = rs;
        v->ring[rs].dma_lo = v->ring[rs].dma_hi = 0;
    }
    op_volume::RingSlot& r = v->ring[v->ring_cur];
    const size_t npx = (size_t)v->cam.width * v->cam.height, dbytes = npx * (depth_fmt == OP_DEPTH_U16 ? 2 : 4), cbytes = npx * 3;
    if (r.dma_hi > r.dma_lo && (pos != r.dma_hi || dbytes != r.dma_dbytes)) OP_TRY(vol_ring_send(v, r)); // not adjacent to what is waiting (device frames in between)
    char* hd = (char*)r.h_depth + (size_t)pos * dbytes;
    unsigned char* hc = r.h_rgb + (size_t)pos * cbytes;
    CopyPool::get().copy2(hd, *depth, dbytes, hc, *rgb, cbytes);
    if (r.dma_hi == r.dma_lo) r.dma_lo = pos;
    r.dma_hi = pos + 1;
    r.dma_dbytes = dbytes;
    if (r.dma_hi - r.dma_lo >= OP_RING_DMA_FRAMES) OP_TRY(vol_ring_send(v, r));
    *depth = (char*)r.d_depth + (size_t)pos * dbytes;
    *rgb = r.d_rgb + (size_t)pos * cbytes;
    return OP_OK;
}

// The fusion kernels address a batch of kMaxBatch packed frames (8 B per pixel) with 32-bit byte offsets -- frame f starts at f * npix * 8
// (k_integrate's buffer resource) -- and multiply image rows with 24-bit integer multiplies: 2^24 pixels (4096 x 4096) with batches of
// up to 32 frames, 2^23 with the 64-frame build (-DOP_MAX_BATCH=64).
constexpr long long kMaxPixels = (1LL << 29) / kMaxBatch < (1LL << 24) ? (1LL << 29) / kMaxBatch : (1LL << 24);
static_assert((unsigned long long)kMaxBatch * (unsigned long long)kMaxPixels * 8ull <= (1ull << 32), "the last frame of a batch must start and end below 2^32 bytes");
int check_cam(const op_camera* cam) {
    if (!cam || cam->width <= 0 || cam->height <= 0 || (long long)cam->width * cam->height > kMaxPixels)
        return fail(OP_ERR_INVALID, "invalid camera (images of up to %lld pixels are supported)", kMaxPixels);
    return OP_OK;
}

} // namespace opv

extern "C" {

int op_abi_version(void) { return OP_ABI_VERSION; }

int op_device_alloc(size_t bytes, int device, void** device_ptr) {
    if (!device_ptr || bytes == 0) return fail(OP_ERR_INVALID, "op_device_alloc: null argument");
    *device_ptr = nullptr;
    OP_TRY(op::use_device(device));
    OP_HIP(op::cached_malloc(device_ptr, bytes));
    return OP_OK;
}

int op_device_write(void* device_ptr, size_t n_parts, const void* const* parts, const size_t* bytes, const size_t* offsets, int device) {
    if (!device_ptr || !parts || !bytes || !offsets) return fail(OP_ERR_INVALID, "op_device_write: null argument");
    OP_TRY(op::use_device(device));
    return write_staged(device_ptr, n_parts, parts, bytes, offsets, device);
}

int op_device_upload(const void* host, size_t bytes, int device, void** device_ptr) {
    if (!host || !device_ptr || bytes == 0) return fail(OP_ERR_INVALID, "op_device_upload: null argument");
    OP_TRY(op_device_alloc(bytes, device, device_ptr));
    const size_t zero = 0;
    const int rc = write_staged(*device_ptr, 1, &host, &bytes, &zero, device);
    if (rc != OP_OK) { op::cached_free(*device_ptr); *device_ptr = nullptr; }
    return rc;
}

int op_device_release(void* device_ptr, int device) {
    if (!device_ptr) return OP_OK;
    OP_TRY(op::use_device(device));
    op::cached_free(device_ptr);
    return OP_OK;
}

int op_runtime_hw_queues(int* requested) {
    if (!requested) return fail(OP_ERR_INVALID, "null argument");
    const char* e = std::getenv("GPU_MAX_HW_QUEUES"); // (reading the runtime's own variable to REPORT it; nothing of the library's behaviour depends on it)
    *requested = e ? std::atoi(e) : 4; // 4 = the runtime's own default
    return OP_OK;
}

int op_runtime_configure(int hw_queues) {
    if (hw_queues < 1 || hw_queues > 64) return fail(OP_ERR_INVALID, "op_runtime_configure: %d hardware queues", hw_queues);
    char buf[16];
    std::snprintf(buf, sizeof(buf), "%d", hw_queues);
    if (setenv("GPU_MAX_HW_QUEUES", buf, /*overwrite=*/0) != 0) return fail(OP_ERR_INVALID, "op_runtime_configure: setenv failed");
    op::runtime_options().hw_queues_requested.store(hw_queues);
    return OP_OK;
}

int op_runtime_set_option(int option, long long value) {
    op::RuntimeOptions& o = op::runtime_options();
    switch (option) {
        case OP_RUNTIME_OPT_MERGE_ALGORITHM:
            if (value != OP_MERGE_OWNER_EXCHANGE && value != OP_MERGE_DENSE_REDUCE) return fail(OP_ERR_INVALID, "op_runtime_set_option: unknown merge algorithm %lld", value);
            o.merge_algorithm.store((int)value); return OP_OK;
        case OP_RUNTIME_OPT_MERGE_SLICE_BLOCKS:
            if (value < 0) return fail(OP_ERR_INVALID, "op_runtime_set_option: slice size %lld", value);
            o.merge_slice_blocks.store(value); return OP_OK;
        case OP_RUNTIME_OPT_MERGE_FORCE_SINGLE_RANK: o.merge_force_single_rank.store(value != 0); return OP_OK;
        case OP_RUNTIME_OPT_TRACKER_GRAPH: o.tracker_graph.store(value != 0); return OP_OK;
        case OP_RUNTIME_OPT_COPY_THREADS:
            if (value < 0 || value > 8) return fail(OP_ERR_INVALID, "op_runtime_set_option: %lld copy threads (0 .. 8)", value);
            o.copy_threads.store((int)value); return OP_OK;
        case OP_RUNTIME_OPT_ICP_DEFAULT_SUMS:
            if (value != OP_ICP_SUMS_FP64 && value != OP_ICP_SUMS_REFERENCE_F32) return fail(OP_ERR_INVALID, "op_runtime_set_option: unknown ICP sums mode %lld", value);
            o.icp_default_sums.store((int)value); return OP_OK;
        case OP_RUNTIME_OPT_TRACKER_DEFAULT_SUMS:
            if (value != OP_TRACK_SUMS_FP64 && value != OP_TRACK_SUMS_REFERENCE_F32 && value != OP_TRACK_SUMS_REFERENCE_F32_HOST) return fail(OP_ERR_INVALID, "op_runtime_set_option: unknown tracker sums mode %lld", value);
            o.tracker_default_sums.store((int)value); return OP_OK;
        case OP_RUNTIME_OPT_TRACKER_BATCH_SUMS: o.tracker_batch_sums.store(value != 0); return OP_OK;
        case OP_RUNTIME_OPT_ICP_MANY_IN_FLIGHT:
            if (value < 1 || value > 1024) return fail(OP_ERR_INVALID, "op_runtime_set_option: %lld iterations in flight", value);
            o.icp_many_in_flight.store((int)value); return OP_OK;
        case OP_RUNTIME_OPT_MERGE_FAULT:
            if (value < 0) return fail(OP_ERR_INVALID, "op_runtime_set_option: merge fault %lld", value);
            o.merge_fault.store(value); return OP_OK;
        case OP_RUNTIME_OPT_CACHE_DEVICE_BYTES:
            if (value < 0) return fail(OP_ERR_INVALID, "op_runtime_set_option: cache limit %lld", value);
            o.cache_device_bytes.store(value); return OP_OK;
        default: break;
    }
    return fail(OP_ERR_INVALID, "op_runtime_set_option: unknown option %d", option);
}

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

int op_frustum_from_camera(const op_camera* cam, const float pose[16], float far_dist, float near_dist, float planes[24], float corners[24]) {
    OP_TRY(check_cam(cam));
    if (!pose || !planes) return fail(OP_ERR_INVALID, "null argument");
    op_host::CameraPOD c{cam->fx, cam->fy, cam->cx, cam->cy, cam->width, cam->height, cam->depth_scale};
    op_host::frustum_planes(c, pose, far_dist, near_dist, planes, corners);
    return OP_OK;
}

int op_frustum_from_vectors(const float forward[3], const float position[3], const float right[3], const float up[3], float far_dist, float near_dist,
                            float fov, float aspect, float planes[24], float corners[24]) {
    if (!forward || !position || !right || !up || !planes) return fail(OP_ERR_INVALID, "null argument");
    op_host::frustum_from_vectors(forward, position, right, up, far_dist, near_dist, fov, aspect, planes, corners);
    return OP_OK;
}

int op_get_sdf(const op_camera* cam, const float point[3], const float pose[16], const float* pose_inv, const void* depth, int depth_fmt, float* sdf) {
    OP_TRY(check_cam(cam));
    if (!point || !pose || !depth || !sdf) return fail(OP_ERR_INVALID, "null argument");
    float inv[16];
    if (pose_inv) std::memcpy(inv, pose_inv, sizeof(inv));
    else op_host::mat4_inverse(pose, inv);                          // Integrator.cpp:18
    // (pose_inv * (p, 1)).head<3>() accumulated column by column (:19)
    float q[3];
    for (int r = 0; r < 3; ++r) q[r] = ((inv[4 * r] * point[0] + inv[4 * r + 1] * point[1]) + inv[4 * r + 2] * point[2]) + inv[4 * r + 3] * 1.0f;
    const int u = px_round_dp((cam->fx * q[0]) / q[2], cam->cx);    // :20-21: float product and quotient, the sum in double, truncated
    const int w = px_round_dp((cam->fy * q[1]) / q[2], cam->cy);
    *sdf = 999.0f;
    if (w < 0 || w >= cam->height || u < 0 || u >= cam->width) return OP_OK;   // :23-24
    const size_t at = (size_t)w * cam->width + u;
    const float d = depth_fmt == OP_DEPTH_U16 ? (float)((const unsigned short*)depth)[at] / cam->depth_scale : ((const float*)depth)[at]; // :26-29
    if (d <= 0) return OP_OK;                                        // :30
    *sdf = d - q[2];
    return OP_OK;
}

int op_debug_project_px(float a, float c, int fast) {
    return fast ? px_round_sp(a, px_split(c)) : px_round_dp(a, c);
}

int op_debug_project_uv(float fx, float fy, float cx, float cy, const float* X, const float* Y, const float* Z, size_t n, int device, int32_t* out) {
    if (!X || !Y || !Z || !out) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(op::use_device(device));
    if (n == 0) return OP_OK;
    float* d_in = nullptr;
    int* d_out = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_in, 3 * n * sizeof(float)));
    hipError_t e = op::cached_malloc((void**)&d_out, 4 * n * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_in, X, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + n, Y, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + 2 * n, Z, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        CamParams C{};
        C.fx = fx; C.fy = fy; C.cx = cx; C.cy = cy; C.width = 640; C.height = 480;
        C.ax = px_axis(cx, C.width); C.ay = px_axis(cy, C.height);
        C.fast_px = C.ax.exact && C.ay.exact;
        hipLaunchKernelGGL(k_debug_project_uv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, C, (const float*)d_in,
                           (const float*)(d_in + n), (const float*)(d_in + 2 * n), n, d_out);
        e = hipMemcpy(out, d_out, 4 * n * sizeof(int), hipMemcpyDeviceToHost);
    }
    op::cached_free(d_in);
    if (d_out) op::cached_free(d_out);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "op_debug_project_uv failed: %s", hipGetErrorString(e));
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
    if (max_blocks == 0) max_blocks = 1u << 18; // initial capacity; the pool and the table grow on demand (vol_grow)
    if (max_blocks > kHardMaxBlocks) return fail(OP_ERR_INVALID, "max_blocks too large (limit %u)", kHardMaxBlocks);
    op_volume* v = new op_volume();
    v->device = device; v->cam = *cam; v->res = voxel_res; v->trunc = truncation; v->far_d = far_dist; v->near_d = near_dist;
    v->max_blocks = (unsigned)max_blocks;
    v->table_size = next_pow2(2ull * max_blocks);
    auto cleanup = [&](int rc) { op_volume_destroy(v); return rc; };
#define OP_HIP_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cleanup(fail(OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
    OP_HIP_C(op::cached_stream(&v->stream)); // creating and destroying a stream costs 1.5-2 ms each: kept like the buffers
    OP_HIP_C(op::cached_malloc((void**)&v->tkeys, sizeof(unsigned long long) * (size_t)v->table_size));
    OP_HIP_C(op::cached_malloc((void**)&v->tvals, sizeof(int) * (size_t)v->table_size));
    OP_HIP_C(op::cached_malloc((void**)&v->bmask, sizeof(bmask_t) * (size_t)v->table_size));
    OP_HIP_C(op::cached_malloc((void**)&v->blist, sizeof(int) * (size_t)(KC_BANDS ? kBands : 1) * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->sel_partial, sizeof(unsigned long long) * kPartialGrid));
    OP_HIP_C(op::cached_malloc((void**)&v->keys, sizeof(int) * 3 * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->pool, sizeof(float) * kBlockFloats * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->n_blocks, sizeof(unsigned)));
    OP_HIP_C(op::cached_malloc((void**)&v->sel_list, sizeof(int) * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->sel_cand, sizeof(unsigned long long) * (size_t)v->max_blocks));
    OP_HIP_C(op::cached_malloc((void**)&v->state, sizeof(State)));
    OP_HIP_C(op::cached_malloc((void**)&v->upd_partial, sizeof(unsigned long long) * kPartialGrid));
    OP_HIP_C(op::cached_malloc((void**)&v->chg_partial, sizeof(unsigned long long) * 2 * kPartialGrid));
    OP_HIP_C(op::cached_host_malloc((void**)&v->hstat, 2 * sizeof(unsigned)));
    v->hstat[0] = 0; v->hstat[1] = 0;
    OP_HIP_C(hipHostGetDevicePointer((void**)&v->hstat_dev, v->hstat, 0));
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
    kc_trace_dump(v); // (-DKC_TRACE / -DKB_TRACE development builds: per-wave phase times of the last launch)
    kb_trace_dump(v);
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    for (auto e : v->prof_events) (void)hipEventDestroy(e);
    void* ptrs[] = {v->tkeys, v->tvals, v->keys, v->pool, v->n_blocks, v->bmask, v->blist, v->sel_list, v->sel_cand, v->state,
                    v->partial, v->pimg, v->ptile, v->sbits, v->upd_partial, v->sel_partial, v->chg_partial, v->img_depth, v->img_rgb, v->unpack_slots, v->rc_list, v->rc_count, v->rc_hit, v->rc_sum, v->rc_order};
    for (void* p : ptrs)
        if (p) op::cached_free(p);
    if (v->copy_stream) (void)hipStreamSynchronize(v->copy_stream);
    for (auto& r : v->ring) {
        if (r.d_depth) op::cached_free(r.d_depth);
        if (r.d_rgb) op::cached_free(r.d_rgb);
        if (r.h_depth) op::cached_free(r.h_depth);
        if (r.h_rgb) op::cached_free(r.h_rgb);
        if (r.copied) (void)hipEventDestroy(r.copied);
    }
    if (v->hstat) op::cached_free(v->hstat);
    op::release_stream(v->copy_stream, v->device); // both were synchronised above
    op::release_stream(v->stream, v->device);
    delete v;
    return OP_OK;
}


// Queued and in-flight frames were accepted under the old setting: a CHANGE of resolution, truncation or camera waits until
// they are fused for good (vol_check: flush, synchronise, grow + replay if a batch ran out of pool).  A replay rebuilds its
// kernel parameters from the volume's current settings (BatchRec keeps poses, frustum planes and image pointers only), so no
// batch may still be replayable when a setting changes.
int op_volume_set_resolution(op_volume* v, float voxel_res) {
    OP_VOL(v);
    if (!(voxel_res > 0)) return fail(OP_ERR_INVALID, "voxel_res must be > 0");
    if (voxel_res == v->res) return OP_OK;
    OP_TRY(vol_check(v));
    v->res = voxel_res;
    return OP_OK;
}
int op_volume_set_truncation(op_volume* v, float truncation) {
    OP_VOL(v);
    if (truncation == v->trunc) return OP_OK;
    OP_TRY(vol_check(v));
    v->trunc = truncation;
    return OP_OK;
}
int op_volume_set_camera(op_volume* v, const op_camera* cam) {
    OP_VOL(v);
    OP_TRY(check_cam(cam));
    if (std::memcmp(cam, &v->cam, sizeof(op_camera)) == 0) return OP_OK;
    OP_TRY(vol_check(v));
    v->cam = *cam;
    return OP_OK;
}
int op_volume_set_near_far(op_volume* v, float near_dist, float far_dist) {
    OP_VOL(v);
    OP_TRY(vol_flush(v)); // queued frames were accepted under the old setting; launched batches carry their frustum planes with them
    v->near_d = near_dist; v->far_d = far_dist;
    return OP_OK;
}

int op_volume_set_option(op_volume* v, int option, int value) {
    OP_VOL(v);
    if (option == OP_VOLUME_OPT_SELECT) { // which form of the selection step batches take (results are identical; a tuning / test knob)
        if (value != OP_VOLUME_SELECT_AUTO && value != OP_VOLUME_SELECT_DIRECT && (value < 1 || value > (int)kVoteCap))
            return fail(OP_ERR_INVALID, "op_volume_set_option: OP_VOLUME_OPT_SELECT takes OP_VOLUME_SELECT_AUTO, OP_VOLUME_SELECT_DIRECT or 1..%u super-blocks per frame", kVoteCap);
        v->select_mode = value;
        return OP_OK;
    }
    if (option == OP_VOLUME_OPT_RAYCAST_PRUNE) { // results are identical either way: a measurement / test knob
        if (value != 0 && value != 1) return fail(OP_ERR_INVALID, "op_volume_set_option: OP_VOLUME_OPT_RAYCAST_PRUNE takes 0 or 1");
        v->rc_prune = value;
        return OP_OK;
    }
    if (option != OP_VOLUME_OPT_UPDATE) return fail(OP_ERR_INVALID, "op_volume_set_option: unknown option %d", option);
    if (value != OP_VOLUME_UPDATE_EXACT && value != OP_VOLUME_UPDATE_SUM_FORM) return fail(OP_ERR_INVALID, "op_volume_set_option: bad value %d", value);
    if (value == v->update_mode) return OP_OK;
    OP_TRY(vol_check(v)); // queued and replayable batches were accepted under the old setting (see op_volume_set_resolution)
    v->update_mode = value;
    return OP_OK;
}

int op_volume_clear(op_volume* v) {
    OP_VOL(v);
    v->pend_n = 0; // queued frames would be wiped anyway
    v->ring_cur = -1;
    unsigned n = 0;
    if (v->copy_stream) OP_HIP(hipStreamSynchronize(v->copy_stream));
    OP_HIP(hipStreamSynchronize(v->stream));
    v->log.clear(); // whatever was in flight (fused or poisoned) is wiped with the volume
    v->plain = true; v->plain_from = 0;
    for (auto& r : v->ring) r.busy_seq = 0;
    if (v->hstat) v->hstat[1] = 0;
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

int op_volume_flush(op_volume* v) {
    OP_VOL(v);
    return vol_flush(v);
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
    OP_TRY(vol_flush(v));
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, nullptr, mem));
    OP_TRY(vol_ensure_frame_buffers(v));
    BatchFwd F;
    BatchInv I;
    BatchPtrs Q{};
    frame_params(v, pose, nullptr, &F.f[0], &I.f[0]);
    Q.depth[0] = depth;
    const CamParams C = cam_params(v, depth_fmt);
    const int g1 = ka_grid(C.width, C.height);
    launch_prepare_frames(v, F, 1, C, Q, (unsigned)(++v->seq));
    OP_HIP(hipGetLastError());
    // no KB / KC follows to consume and zero the frame's bounding accumulators: do it here (the rows are read below)
    OP_HIP(hipMemsetAsync(reinterpret_cast<char*>(v->state) + offsetof(State, acc), 0, sizeof(State::acc), v->stream));
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
    OP_TRY(vol_flush(v));
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, nullptr, mem));
    BatchFwd F;
    BatchInv I;
    BatchPtrs Q{};
    frame_params(v, pose, pose_inv, &F.f[0], &I.f[0]);
    Q.depth[0] = depth;
    for (;;) { // a selection that outgrows the pool grows it (vol_check) and is then simply run again
        const unsigned cap_before = v->max_blocks;
        OP_TRY(vol_enqueue_batch(v, F, I, Q, 1, depth_fmt, /*select_only=*/true, /*record=*/true));
        OP_TRY(vol_check(v));
        if (v->max_blocks == cap_before) break;
    }
    StateHead st;
    OP_HIP(hipMemcpy(&st, v->state, sizeof(st), hipMemcpyDeviceToHost));
    unsigned long long n_cand0 = 0;
    OP_HIP(hipMemcpy(&n_cand0, reinterpret_cast<const char*>(v->state) + offsetof(State, n_cand), sizeof(n_cand0), hipMemcpyDeviceToHost));
    const size_t ns = std::min((size_t)st.n_rec, (size_t)v->max_blocks);
    if (n) *n = ns;
    if (n_candidates) *n_candidates = (size_t)n_cand0;
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
    // Single frames are queued and fused kMaxBatch at a time (one launch group per batch instead of
    // per frame); every accessor, setter and synchronising call flushes the queue first, so the
    // deferral cannot be observed.  Host images are copied to the staging ring right here (the
    // caller's buffers are only borrowed for the call); device images are used in place at flush.
    if (v->pend_n > 0 && v->pend_fmt != depth_fmt) OP_TRY(vol_flush(v));
    const unsigned char* c = rgb;
    // The first host image of a volume (or a larger one) allocates the staging ring, and that launches whatever is queued
    // (vol_ring_alloc -> vol_check -> vol_flush): the queue position is only valid afterwards.
    if (mem == OP_MEM_HOST) OP_TRY(vol_ring_alloc(v));
    const int slot = v->pend_n;
    if (mem == OP_MEM_HOST) OP_TRY(vol_ring_stage(v, &depth, depth_fmt, &c, slot));
    frame_params(v, pose, pose_inv, &v->pend_F.f[slot], &v->pend_I.f[slot]);
    v->pend_P.depth[slot] = depth;
    v->pend_P.rgb[slot] = c;
    v->pend_fmt = depth_fmt;
    ++v->frames_accepted;
    if (++v->pend_n == kMaxBatch) return vol_flush(v);
    return OP_OK;
}

int op_volume_integrate_cubes(op_volume* v, const void* depth, int depth_fmt, const uint8_t* rgb, int mem, const float pose[16], const float* pose_inv,
                              const int32_t* keys_xyz, size_t n) {
    OP_VOL(v);
    if (!depth || !rgb || !pose || (n && !keys_xyz)) return fail(OP_ERR_INVALID, "null argument");
    if (n == 0) return OP_OK;
    if (n > kHardMaxBlocks) return fail(OP_ERR_CAPACITY, "%zu cubes exceed the limit of %u blocks per volume", n, kHardMaxBlocks);
    unsigned have = 0;
    OP_TRY(vol_block_count(v, &have)); // flushes, synchronises, recovers
    OP_TRY(vol_reserve(v, (unsigned long long)have + n));
    const unsigned char* c = rgb;
    OP_TRY(vol_stage_images(v, &depth, depth_fmt, &c, mem));
    int* d_keys = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_keys, n * 3 * sizeof(int)));
    hipError_t e = hipMemcpyAsync(d_keys, keys_xyz, n * 3 * sizeof(int), hipMemcpyHostToDevice, v->stream);
    int rc = OP_OK;
    if (e == hipSuccess) {
        BatchFwd F;
        BatchInv I;
        BatchPtrs Q{};
        frame_params(v, pose, pose_inv, &F.f[0], &I.f[0]);
        Q.depth[0] = depth; Q.rgb[0] = c;
        rc = vol_enqueue_batch(v, F, I, Q, 1, depth_fmt, false, false, d_keys, (unsigned)n);
        if (rc == OP_OK) rc = vol_check(v);
    }
    (void)hipStreamSynchronize(v->stream);
    op::cached_free(d_keys);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "copying the cube list failed: %s", hipGetErrorString(e));
    return rc;
}

int op_volume_integrate_sequence(op_volume* v, const void* depth, size_t depth_stride_bytes, int depth_fmt, const uint8_t* rgb,
                                 size_t rgb_stride_bytes, const float* poses, size_t n_frames) {
    OP_VOL(v);
    if (!depth || !rgb || !poses) return fail(OP_ERR_INVALID, "null argument");
    // The frames join the queue op_volume_integrate fills: full batches are launched as they complete, the remainder waits for the next
    // frames (of this or the next call) or for the first call that flushes -- a stream of calls fuses kMaxBatch frames per launch
    // whatever the calls' lengths are.  (The device images must stay valid until the next synchronising call, as for op_volume_integrate.)
    if (v->pend_n > 0 && v->pend_fmt != depth_fmt) OP_TRY(vol_flush(v));
    for (size_t f = 0; f < n_frames; ++f) {
        const int slot = v->pend_n;
        frame_params(v, poses + 16 * f, nullptr, &v->pend_F.f[slot], &v->pend_I.f[slot]);
        v->pend_P.depth[slot] = (const char*)depth + f * depth_stride_bytes;
        v->pend_P.rgb[slot] = rgb + f * rgb_stride_bytes;
        v->pend_fmt = depth_fmt;
        ++v->frames_accepted;
        if (++v->pend_n == kMaxBatch) OP_TRY(vol_flush(v));
    }
    return OP_OK;
}

int op_volume_progress(op_volume* v, uint64_t* frames_accepted, uint64_t* frames_done) {
    OP_VOL(v);
    vol_retire(v); // looks at the device's progress report; never blocks
    uint64_t open = (uint64_t)v->pend_n;
    for (const auto& r : v->log) open += (uint64_t)r.nf;
    if (frames_accepted) *frames_accepted = v->frames_accepted;
    if (frames_done) *frames_done = v->frames_accepted >= open ? v->frames_accepted - open : 0;
    return OP_OK;
}

int op_volume_stats(op_volume* v, uint64_t* frames, uint64_t* blocks_selected, uint64_t* voxels_visited, uint64_t* voxels_updated) {
    OP_VOL(v);
    OP_TRY(vol_check(v));
    unsigned long long stat_frames = 0;
    OP_HIP(hipMemcpy(&stat_frames, reinterpret_cast<const char*>(v->state) + offsetof(State, stat_frames), sizeof(stat_frames), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> part(2 * kPartialGrid);
    OP_HIP(hipMemcpy(part.data(), v->upd_partial, kPartialGrid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    OP_HIP(hipMemcpy(part.data() + kPartialGrid, v->sel_partial, kPartialGrid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long upd = 0, sel = 0;
    for (int i = 0; i < kPartialGrid; ++i) { upd += part[i]; sel += part[kPartialGrid + i]; }
    if (frames) *frames = stat_frames;
    if (blocks_selected) *blocks_selected = sel;
    if (voxels_visited) *voxels_visited = sel * (uint64_t)kVox;
    if (voxels_updated) *voxels_updated = upd;
    return OP_OK;
}

int op_volume_stats_launches(op_volume* v, uint64_t* launches, uint64_t* blocks_read, uint64_t* voxels_written, uint64_t* shader_cycles) {
    OP_VOL(v);
    OP_TRY(vol_check(v));
    struct { unsigned long long stat_frames, stat_launches, kc_t[kKcTSlots], stat_kc_ticks; } st; // that stretch of State (2 KB), not all 20 KB of it
    static_assert(offsetof(State, stat_kc_ticks) - offsetof(State, stat_frames) + 8 == sizeof(st), "contiguous statistics fields of State");
    OP_HIP(hipMemcpy(&st, reinterpret_cast<const char*>(v->state) + offsetof(State, stat_frames), sizeof(st), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> part(2 * kPartialGrid);
    OP_HIP(hipMemcpy(part.data(), v->chg_partial, 2 * kPartialGrid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long chg = 0, blk = 0;
    for (int i = 0; i < kPartialGrid; ++i) { chg += part[i]; blk += part[kPartialGrid + i]; }
    if (shader_cycles) { // the last launch has not been folded by a following batch yet
        unsigned long long kc = 0;
        for (int x = 0; x < kKcTSlots; ++x)
            if (st.kc_t[x] > kc) kc = st.kc_t[x];
        *shader_cycles = st.stat_kc_ticks + kc;
    }
    if (launches) *launches = st.stat_launches;
    if (blocks_read) *blocks_read = blk;
    if (voxels_written) *voxels_written = chg;
    return OP_OK;
}

int op_volume_growth_stats(op_volume* v, uint64_t* max_blocks, uint64_t* grows, uint64_t* replayed_batches) {
    OP_VOL(v);
    if (max_blocks) *max_blocks = v->max_blocks;
    if (grows) *grows = v->n_grows;
    if (replayed_batches) *replayed_batches = v->n_replayed;
    return OP_OK;
}

int op_volume_profile_enable(op_volume* v, int sample_every) {
    OP_VOL(v);
    OP_TRY(vol_flush(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    for (auto e : v->prof_events) (void)hipEventDestroy(e);
    v->prof_events.clear();
    v->prof_frames.clear();
    v->prof_every = sample_every > 0 ? sample_every : 0;
    v->prof_batch = 0;
    return OP_OK;
}

int op_volume_profile_read(op_volume* v, double ms_sum[3], uint64_t* n_launches, uint64_t* n_frames) {
    OP_VOL(v);
    if (!ms_sum || !n_launches || !n_frames) return fail(OP_ERR_INVALID, "null argument");
    OP_TRY(vol_flush(v));
    OP_HIP(hipStreamSynchronize(v->stream));
    ms_sum[0] = ms_sum[1] = ms_sum[2] = 0.0;
    const size_t n = v->prof_events.size() / 4;
    uint64_t frames = 0;
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            float ms = 0.0f;
            OP_HIP(hipEventElapsedTime(&ms, v->prof_events[4 * i + k], v->prof_events[4 * i + k + 1]));
            ms_sum[k] += ms;
        }
        frames += (uint64_t)v->prof_frames[i];
    }
    *n_launches = n;
    *n_frames = frames;
    return OP_OK;
}

int op_volume_has_cube(op_volume* v, int32_t x, int32_t y, int32_t z, int* present) {
    OP_VOL(v);
    if (!present) return fail(OP_ERR_INVALID, "null present");
    OP_TRY(vol_check(v));
    int* d_out = nullptr;
    OP_HIP(op::cached_malloc((void**)&d_out, sizeof(int)));
    hipLaunchKernelGGL(k_has_cube, dim3(1), dim3(1), 0, v->stream, v->view(), x, y, z, d_out); // cube_map.find (CubeHandler.h:129-132)
    hipError_t e = hipMemcpyAsync(present, d_out, sizeof(int), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    op::cached_free(d_out);
    if (e != hipSuccess) return fail(OP_ERR_HIP, "has_cube failed: %s", hipGetErrorString(e));
    return OP_OK;
}

} // extern "C"
