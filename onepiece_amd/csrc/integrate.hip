// integrate.hip -- KC k_integrate: Integrator::IntegrateImage (Integration/Integrator.cpp:36-94) + TSDFVoxel::operator+ (TSDFVoxel.h:24-39) for all frames
// of a batch (volume_core.hpp lists the translation units).  file:line citations are relative to /root/reference/src.
#include "volume_core.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// KC: Integrator::IntegrateImage (Integrator.cpp:36-94) for all frames of the batch (k_integrate below).
// ---------------------------------------------------------------------------------------------
typedef unsigned int kc_v2u __attribute__((ext_vector_type(2)));

// PLAIN: the volume's content has only ever been written by this kernel since it was created / cleared (no upload, merge,
// resampling or file in between; the host tracks it).  Then every stored voxel is either the default or a running mean
// of finite in-band observations: weights are integers >= 1, colours are means of byte/255 values (non-negative, so their
// numerators w*c + n never cancel), and the update can be evaluated
//   * branch-free: an invalid voxel (TSDFVoxel::IsValid false) is the valid formula with weight 0 --
//     (0*s + new)/(0 + 1) = new exactly for finite s -- instead of a second code path with five selects;
//   * with ONE refined reciprocal of wsum shared by the four quotients (the compiler's own v_rcp + FMA sequence, spelled
//     out as in project_uv): bit-identical to the IEEE division whenever v_div_scale would not rescale, i.e. for
//     wsum in [1, 2^25] and a numerator that is 0 or >= 2^-100 in magnitude.  Colour numerators are 0 or >= 2^-32
//     (no cancellation); the sdf numerator CAN cancel to something tiny, so it alone is guarded: a lane whose
//     |w*s + new| is non-zero and below 2^-100 takes the plain division (never, in practice).
// Without PLAIN (arbitrary uploaded data: NaN, infinities, denormals, fractional weights) the update is the reference's
// two-branch form with four true divisions.
// ---------------------------------------------------------------------------------------------
// KC.  One workgroup per block of the batch list; the resident workgroups draw blocks from per-XCD counters.  Every voxel is
// read ONCE, every frame that selected the block is applied to it in frame order in registers (bit-identical to the
// reference's frame-by-frame running mean) and it is written once -- HBM traffic per voxel drops from 40 B per frame to
// 40 B per batch; block ownership is exclusive, so the read-modify-write needs no atomics.  A thread owns ZT voxels
// of one (x, y) column of the block (z = zg*ZT .. zg*ZT + ZT-1), a wave owns ZT z-slices, a workgroup of 8/ZT waves owns
// the block.  What that buys, per voxel and frame:
//   * everything that is uniform over the wave -- the frame's bit test, the three s_load_dwordx4 of its pose rows, the buffer
//     resource of its packed image, the loop control -- is paid once per ZT voxels instead of once per voxel (a third of the
//     issue slots of a one-voxel-per-thread kernel go to scalar and branch instructions, profiles/r03_issue_costs.json);
//   * the partial sums M[r][0]*px + M[r][1]*py of the three pose rows depend on x and y only and are shared by the ZT voxels
//     (the same two rounded products and one rounded sum the reference forms for each of them: bit-identical);
//   * ZT independent dependency chains per thread hide the VALU and gather latencies that eight waves per SIMD hid before,
//     so the kernel runs at a lower occupancy with a larger register budget.
// A plane row of a z-slice is still one 256-byte wave access.  Frames are applied in ascending order; the gathers of the
// NEXT selected frame are issued before the current frame's updates (two record sets, the frame loop unrolled by two).
// ---------------------------------------------------------------------------------------------
template <bool PLAIN>
__device__ __forceinline__ void voxel_update(float& s, float& w, float& c0, float& c1, float& c2, float new_sdf, unsigned rgba, const float* s_c255) {
    const float n0 = s_c255[rgba & 0xffu], n1 = s_c255[(rgba >> 8) & 0xffu], n2 = s_c255[(rgba >> 16) & 0xffu];
    if (PLAIN) {
        // TSDFVoxel::IsValid (TSDFVoxel.h:75-78) false -> weight 0 in the same formula (see the PLAIN comment above)
        const float wv = (s >= 1 || w <= 0) ? 0.0f : w;
        const float wsum = wv + 1.0f;
        float y = __builtin_amdgcn_rcpf(wsum);
        const float e = __builtin_fmaf(-wsum, y, 1.0f);
        y = __builtin_fmaf(e, y, y);
        const float ns = wv * s + 1.0f * new_sdf;
        const float m0 = wv * c0 + 1.0f * n0, m1 = wv * c1 + 1.0f * n1, m2 = wv * c2 + 1.0f * n2;
        float qs = div_shared_rcp(ns, wsum, y);
        const bool tiny = !(fabsf(ns) >= 0x1p-100f) && ns != 0.0f;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tiny) != 0ull, 0)) {
            if (tiny) qs = ns / wsum;
        }
        s = qs;
        c0 = div_shared_rcp(m0, wsum, y);
        c1 = div_shared_rcp(m1, wsum, y);
        c2 = div_shared_rcp(m2, wsum, y);
        w = wsum;
    } else if (!(s >= 1 || w <= 0)) { // TSDFVoxel::IsValid (TSDFVoxel.h:75-78)
        const float wsum = w + 1.0f;  // TSDFVoxel::operator+ with other = (new_sdf, 1.0, c) (TSDFVoxel.h:24-39)
        s = (w * s + 1.0f * new_sdf) / wsum;
        c0 = (w * c0 + 1.0f * n0) / wsum;
        c1 = (w * c1 + 1.0f * n1) / wsum;
        c2 = (w * c2 + 1.0f * n2) / wsum;
        w = wsum;
    } else {
        s = new_sdf; w = 1.0f; c0 = n0; c1 = n1; c2 = n2;
    }
}

// SUMF (opt-in, OP_VOLUME_UPDATE_SUM_FORM): the frames of the batch are not applied one by one.  Per voxel the kernel keeps the NUMBER of in-band
// observations of the batch, the sum of their sdf values and the sums of their colour bytes (exact integers), and forms the weighted mean with the
// stored voxel ONCE per batch: s' = (w s + sum sdf) / (w + n), c' = (w c + sum bytes / 255) / (w + n), w' = w + n -- TSDFVoxel::operator+
// (TSDFVoxel.h:24-39) applied n times in exact arithmetic.  Same blocks, same pixels, same weights (integers); sdf and colour differ from the
// frame-by-frame running mean by float rounding only (a few 1e-7 relative; north_star's bar is 1e-4).  Per voxel and frame the ~35 instructions
// of the exactly rounded update shrink to 7 (two selects, one float add, two byte-pair adds with their masks).
#ifdef KC_TRACE // development aid (make EXTRA=-DKC_TRACE): where a workgroup of the last k_integrate launch spent its time, per wave; dumped by op_volume_destroy
__device__ unsigned long long g_kc_trace[4096 * 4 * 8];
#define KC_T(K) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); kt_[K] += now_ - kt_last_; kt_last_ = now_; } while (0)
#define KC_N(K, V) do { kt_[K] += (V); } while (0)
#else
#define KC_T(K) do { } while (0)
#define KC_N(K, V) do { } while (0)
#endif
template <bool FAST, bool PLAIN, int ZT, bool SUMF = false>
__global__ __launch_bounds__(512 / ZT, (SUMF ? KC_SUM_MIN_WAVES : KC_COL_MIN_WAVES)) void k_integrate(BatchInv B, CamParams C, VolView V, const uint2* __restrict__ pimg, State* st,
                                                                            int n_frames, unsigned long long* __restrict__ upd_partial,
                                                                            unsigned long long* __restrict__ sel_partial, unsigned long long* __restrict__ chg_partial,
                                                                            unsigned plain_from, unsigned* __restrict__ rc_sum, unsigned rc_stamp) {
    constexpr int kWaves = 8 / ZT;            // waves per workgroup = z-groups per block
#ifdef KC_NO_SUMMARY // (A/B aid: the kernel without the raycaster's summaries)
    constexpr bool kSummaries = false;
#else
    constexpr bool kSummaries = !SUMF;
#endif
    __shared__ unsigned s_cnt[kWaves][2];
    __shared__ unsigned s_sign[2][kWaves];    // rc_sum: per wave, does its part of the block hold an observed sdf <= 0 (bit 0) / > 0 (bit 1) after the batch
    __shared__ float s_c255[256];             // (float)b / 255.0f for every byte (Integrator.cpp:78), correctly rounded once
    __shared__ unsigned s_next[2];
    const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#ifdef KC_TRACE
    unsigned long long kt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kt_last_ = t_in;
#endif
    // KB has consumed the frames' bounding accumulators: back to the identity for the next batch (also when poisoned)
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < (unsigned)(kMaxBatch * kAccSlots * 8); k += gridDim.x * blockDim.x) (&st->acc[0][0][0])[k] = 0u;
    if (st->overflow & 3u) return; // pool / table exhausted in this or an earlier batch: nothing is fused, the host replays
    const int tid = threadIdx.x, lane = tid & 63, zg = tid >> 6;
    if (!SUMF) for (int k = tid; k < 256; k += blockDim.x) s_c255[k] = (float)k / 255.0f;
    const unsigned npix = (unsigned)(C.width * C.height);
    const float half = C.res / 2;
    // VoxelCentroidOffSet (VoxelCube.h:48-61): x*res + half with x = lane & 7, y = lane >> 3
    const float ox = (float)(lane & 7) * C.res + half;
    const float oy = (float)(lane >> 3) * C.res + half;
    const float __attribute__((address_space(4)))* kargs =
        (const float __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr(); // BatchInv B = offset 0 of the kernarg segment
    (void)B;
    unsigned upd = 0, sel = 0, chg = 0, nblk = 0;
#ifndef KC_CHUNK_LOG2
#define KC_CHUNK_LOG2 5
#endif
    // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own 4 MiB L2; blocks that gather the same pixels should meet in one L2)
    // and dynamic scheduling (blocks differ in work: 1..32 frames touch them; the workgroups of an XCD DRAW list positions from one counter,
    // the next one before the current block is processed so that the atomic's round trip is hidden).  Two ways of dealing the batch's blocks
    // to the eight draw counters:
    //  * full batches (>= KC_STEAL_MIN_FRAMES frames): XCD x starts on the x-th contiguous eighth of the list (with -DKC_BANDS=1 on list x, see
    //    kBands); the eighths hold the same number of blocks but not the same work, so a workgroup whose share is exhausted reads all eight
    //    counters (one round trip) and goes on with the share that has the most left.
    //    Per 32-frame launch: eighths alone 689 us, with stealing 627-631 us;
    //  * short batches: list 0, chunks of 32 blocks dealt round-robin (every XCD a sample of the whole list), no stealing: for ONE frame per
    //    launch, where the kernel is HBM-bound and the work per block uniform, 79 us against 95 us with stealing (its last look costs a short
    //    launch more than it can win).  Measured crossover (tools/prof_driver.bin batch=N under the tracer, stealing vs chunks): 8 frames
    //    207 vs 194 us, 16: 360 vs 356, 24: 524 vs 527, 32: 677 vs 687.
    const bool eighths = n_frames >= KC_STEAL_MIN_FRAMES;
    const bool lists = eighths && KC_BANDS != 0;              // one list per share
    constexpr unsigned kChunk = 1u << KC_CHUNK_LOG2;
    const unsigned n0 = st->n_list[0] < V.max_blocks ? st->n_list[0] : V.max_blocks;
    const unsigned n_chunks = (n0 + kChunk - 1u) >> KC_CHUNK_LOG2;
    const unsigned per0 = eighths ? (n0 + (unsigned)kKcShares - 1u) / (unsigned)kKcShares : ((n_chunks + (unsigned)kKcShares - 1u) / (unsigned)kKcShares) << KC_CHUNK_LOG2;
    unsigned xcd = blockIdx.x % (unsigned)kKcShares;          // the share this workgroup draws from: its own first
    for (;;) {
    unsigned* ctr = &st->kc_next[xcd * 16u];
    if (tid == 0) s_next[0] = atomicAdd(ctr, 1u);
    // positions j < per_xcd of share xcd; its blocks are list[j] (one list per share) or positions of list 0
    unsigned per_xcd = per0, n = n0;
    const int* list = V.blist;
    if (lists) {
        const unsigned nl = st->n_list[xcd];
        per_xcd = n = nl < V.max_blocks ? nl : V.max_blocks;
        list = V.blist + (size_t)xcd * V.max_blocks;
    }
    __syncthreads();
    unsigned slot = 0u;
    for (unsigned j = s_next[0]; j < per_xcd;) {
        if (tid == 0) s_next[slot ^ 1u] = atomicAdd(ctr, 1u);
        const unsigned b = lists ? j : (eighths ? xcd * per_xcd + j : (((j >> KC_CHUNK_LOG2) * (unsigned)kKcShares + xcd) << KC_CHUNK_LOG2) + (j & (kChunk - 1u)));
        KC_T(0);
        const int tslot = b < n ? list[b] : -1;
        const int idx = tslot >= 0 ? V.tvals[tslot] : -1; // idx < 0: pool overflow (reported through st->overflow)
        KC_N(6, 1);
        if (idx >= 0) {
            const bmask_t mask = V.bmask[tslot];
            if (zg == 0) { sel += mask_popc(mask); ++nblk; }
            const int kx = V.keys[3 * idx], ky = V.keys[3 * idx + 1], kz = V.keys[3 * idx + 2];
            float* vox = V.pool + (size_t)idx * kBlockFloats + (zg * ZT) * 64 + lane;
            float s[ZT], w[ZT], c0[ZT], c1[ZT], c2[ZT], pz[ZT];
#pragma unroll
            for (int z = 0; z < ZT; ++z) {
                // (the sum form needs the stored voxel only after the frames: it is loaded there, and the registers are free until then)
                if (!SUMF) { s[z] = vox[z * 64]; w[z] = vox[kVox + z * 64]; c0[z] = vox[2 * kVox + z * 64]; c1[z] = vox[3 * kVox + z * 64]; c2[z] = vox[4 * kVox + z * 64]; }
                // GetGlobalPoint (VoxelCube.h:75-80): Point3(id) * CUBE_SIZE * VoxelResolution + offset
                pz[z] = ((float)kz * 8.0f) * C.res + ((float)(zg * ZT + z) * C.res + half);
            }
            const float px = ((float)kx * 8.0f) * C.res + ox;
            const float py = ((float)ky * 8.0f) * C.res + oy;
            unsigned changed = 0u;
            // sum form: sdf sum, byte sums of colour channels 0 and 2 in the two halves of one word, of channel 1 in the low half of another whose
            // high half counts the observations (<= 64 frames x 255 < 2^16)
            float ssum[ZT];
            unsigned acc02[ZT], acc1n[ZT];
#pragma unroll
            for (int z = 0; z < ZT; ++z) { ssum[z] = 0.0f; acc02[z] = 0u; acc1n[z] = 0u; }
            // one selected frame: projections of the thread's ZT voxels and their {depth, rgba} gathers
            auto project = [&](int f, kc_v2u (&rec)[ZT], float (&zc)[ZT]) {
                int fo = f;
                asm volatile("" : "+s"(fo));
                const float __attribute__((address_space(4)))* M = kargs + fo * 12;
                // (frame base in 32 bits: check_cam keeps kMaxBatch x npix x 8 below 2^32)
                const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)pimg + (unsigned)fo * (npix * 8u)), 0, (int)(npix * 8u), 0x00020000);
                const float a0 = M[0] * px + M[1] * py, a1 = M[4] * px + M[5] * py, a2 = M[8] * px + M[9] * py;
#pragma unroll
                for (int z = 0; z < ZT; ++z) {
                    const float q0 = (a0 + M[2] * pz[z]) + M[3] * 1.0f;
                    const float q1 = (a1 + M[6] * pz[z]) + M[7] * 1.0f;
                    const float q2 = (a2 + M[10] * pz[z]) + M[11] * 1.0f;
                    zc[z] = q2;
                    const int pix = project_pixel<FAST>(C, q0, q1, q2); // off-image: pixel -1 = an offset the buffer answers with zeros
                    rec[z] = __builtin_amdgcn_raw_buffer_load_b64(frame, pix * 8, 0, 0);
                }
            };
            // PLAIN: every block of the volume was written by this kernel only.  Otherwise (the volume has seen an upload, a merge, a
            // sum-form unpack or a file): the blocks that existed then (pool slots below plain_from) hold arbitrary data and take the
            // general update; blocks allocated since are this kernel's own and keep the fast one.  Uniform per block.
            const bool plain_block = PLAIN || (unsigned)idx >= plain_from;
            auto apply = [&](const kc_v2u (&rec)[ZT], const float (&zc)[ZT], auto plain_c) {
                constexpr bool kPlain = decltype(plain_c)::value;
#pragma unroll
                for (int z = 0; z < ZT; ++z) {
                    const float d = __uint_as_float(rec[z].x); // off-image pixels carry d == 0 -> skipped like `continue`
                    const float new_sdf = d - zc[z];
                    // Integrator.cpp:70,74 (d > 0 and |sdf| < truncation) as ONE compare and one divergent region: an absent
                    // observation takes the place of an out-of-band one
                    const float band = d > 0 ? fabsf(new_sdf) : C.trunc;
                    const bool hit = band < C.trunc;
                    upd += hit ? 1u : 0u;                              // per lane; summed over the wave at the end
                    if (SUMF) { // branch-free: an observation that misses adds zeros
                        ssum[z] += hit ? new_sdf : 0.0f;
                        const unsigned t = hit ? rec[z].y : 0u;           // byte 3 of a packed pixel is 1 (k_prepare_frames): the count
                        acc02[z] += t & 0x00ff00ffu;
                        acc1n[z] += (t >> 8) & 0x00ff00ffu;
                    } else if (hit) {
                        changed |= 1u << z;
                        voxel_update<kPlain>(s[z], w[z], c0[z], c1[z], c2[z], new_sdf, rec[z].y, s_c255);
                    }
                }
            };
            kc_v2u recA[ZT], recB[ZT];
            float zcA[ZT], zcB[ZT];
#ifdef KC_TRACE
            { float keep_ = px + py + pz[0]; if (!SUMF) keep_ += s[0]; asm volatile("" :: "v"(keep_)); KC_T(1); } // (the block's metadata and voxels have arrived)
#endif
            auto frames = [&](auto plain_c) {
                bmask_t m = mask;                                 // wave-uniform
                if (!m) return;
                int f = mask_ctz(m); m &= m - 1u;
                project(f, recA, zcA);
                for (;;) {
                    const bool more1 = m != 0u;
                    if (more1) { f = mask_ctz(m); m &= m - 1u; project(f, recB, zcB); }
                    apply(recA, zcA, plain_c);
                    if (!more1) break;
                    const bool more2 = m != 0u;
                    if (more2) { f = mask_ctz(m); m &= m - 1u; project(f, recA, zcA); }
                    apply(recB, zcB, plain_c);
                    if (!more2) break;
                }
            };
            if (SUMF || PLAIN || plain_block) frames(std::true_type{}); else frames(std::false_type{});
            KC_T(2); KC_N(7, mask_popc(mask));
            if (SUMF) {
#pragma unroll
                for (int z = 0; z < ZT; ++z)
                    if (acc1n[z] >> 16) { s[z] = vox[z * 64]; w[z] = vox[kVox + z * 64]; c0[z] = vox[2 * kVox + z * 64]; c1[z] = vox[3 * kVox + z * 64]; c2[z] = vox[4 * kVox + z * 64]; }
#pragma unroll
                for (int z = 0; z < ZT; ++z) {
                    const unsigned cnt = acc1n[z] >> 16;
                    if (cnt) {
                        changed |= 1u << z;
                        // TSDFVoxel::operator+ (TSDFVoxel.h:24-39) for the batch's observations at once; an invalid voxel (IsValid false,
                        // :75-78) is replaced by their mean, as the first observation would have replaced it
                        const bool valid = !(s[z] >= 1 || w[z] <= 0);
                        const float wv = valid ? w[z] : 0.0f, nf = (float)cnt, wsum = wv + nf;
                        const float b0 = (float)(acc02[z] & 0xffffu) / 255.0f, b1 = (float)(acc1n[z] & 0xffffu) / 255.0f, b2 = (float)(acc02[z] >> 16) / 255.0f;
                        s[z] = ((valid ? wv * s[z] : 0.0f) + ssum[z]) / wsum;
                        c0[z] = ((valid ? wv * c0[z] : 0.0f) + b0) / wsum;
                        c1[z] = ((valid ? wv * c1[z] : 0.0f) + b1) / wsum;
                        c2[z] = ((valid ? wv * c2[z] : 0.0f) + b2) / wsum;
                        w[z] = wsum;
                    }
                }
            }
#pragma unroll
            for (int z = 0; z < ZT; ++z)
                if ((changed >> z) & 1u) { vox[z * 64] = s[z]; vox[kVox + z * 64] = w[z]; vox[2 * kVox + z * 64] = c0[z]; vox[3 * kVox + z * 64] = c1[z]; vox[4 * kVox + z * 64] = c2[z]; }
            chg += (unsigned)__popc(changed);
            // The raycaster's block summaries (raycast.hip: k_rc_neighbours drops blocks by them before loading a voxel) describe exactly what is in
            // registers here -- the block's 512 voxels after the batch: the kernel that changes a block restates its summary, so views between fusions
            // find every block they meet described instead of starting from nothing (the host does not invalidate the summaries for such a batch).
            if (kSummaries && rc_sum) {                    // (uniform)
                bool neg = false, pos = false;
#pragma unroll
                for (int z = 0; z < ZT; ++z) { const bool obs = w[z] > 0; neg |= obs && s[z] <= 0; pos |= obs && s[z] > 0; }
                const unsigned f = (__builtin_amdgcn_ballot_w64(neg) ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(pos) ? 2u : 0u);
                if (lane == 0) s_sign[slot][zg] = f;
            }
        }
        KC_T(3);
        __syncthreads();                                   // every wave has read the mask; s_next[slot ^ 1] is visible
        KC_T(4);
        if (tslot >= 0 && tid == 0) V.bmask[tslot] = (bmask_t)0; // the owner clears it for the next batch
        if (kSummaries && rc_sum && idx >= 0 && tid == 0) {     // (s_sign[slot] is written again two blocks on, behind the next barrier)
            unsigned f = 0u;
#pragma unroll
            for (int k = 0; k < kWaves; ++k) f |= s_sign[slot][k];
            rc_sum[idx] = (rc_stamp << 2) | f;
        }
        slot ^= 1u;
        j = s_next[slot];
    }
    if (!eighths) break;
    // the share is exhausted: look at all the draw counters at once (one round trip) and go on with the share that has the most left
    __syncthreads();                                          // everybody has read the last draw
    if (tid < 64) {
        unsigned left = 0u;
        if (tid < kKcShares) {
            const unsigned c = __hip_atomic_load(&st->kc_next[(unsigned)tid * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned len = per0;
            if (lists) { const unsigned nl = st->n_list[tid]; len = nl < V.max_blocks ? nl : V.max_blocks; }
            left = c < len ? len - c : 0u;
        }
        unsigned key = ((left < 0x7fffffu ? left : 0x7fffffu) << 8) | (unsigned)tid; // most left, ties to the higher share index (any fixed rule)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned x = __shfl_xor(key, o, 64); key = x > key ? x : key; }
        if (tid == 0) s_next[0] = key;
    }
    __syncthreads();
    const unsigned key = s_next[0];
    if ((key >> 8) == 0u) break;                              // nothing left anywhere
    xcd = key & 0xffu;
    __syncthreads();                                          // s_next[0] is written again at the top
    }
#ifdef KC_TRACE
    KC_T(5);
    if (lane == 0 && blockIdx.x < 4096) for (int k = 0; k < 8; ++k) g_kc_trace[((size_t)blockIdx.x * 4 + zg) * 8 + k] = kt_[k];
#endif
    // per-workgroup counters into kPartialGrid slots
    upd = wave_sum(upd); chg = wave_sum(chg);
    if (lane == 0) { s_cnt[zg][0] = upd; s_cnt[zg][1] = chg; }
    __syncthreads();
    if (tid == 0) {
        unsigned t = 0, c = 0;
        for (int k = 0; k < kWaves; ++k) { t += s_cnt[k][0]; c += s_cnt[k][1]; }
        const unsigned slot_c = blockIdx.x % (unsigned)kPartialGrid;
        atomicAdd(&upd_partial[slot_c], (unsigned long long)t);
        atomicAdd(&sel_partial[slot_c], (unsigned long long)sel);
        atomicAdd(&chg_partial[slot_c], (unsigned long long)c);
        atomicAdd(&chg_partial[kPartialGrid + slot_c], (unsigned long long)nblk);
        if (blockIdx.x == 0) { st->stat_frames += (unsigned long long)n_frames; st->stat_launches += 1ull; }
        atomicMax(&st->kc_t[blockIdx.x % (unsigned)kKcTSlots], (unsigned long long)__builtin_amdgcn_s_memtime() - t_in);
    }
}

// ---------------------------------------------------------------------------------------------
// export / import / merge kernels
// ---------------------------------------------------------------------------------------------
// SoA pool -> AoS {sdf,w,c0,c1,c2} x 512 for blocks [first, first+count)
} // namespace

namespace opv {

// A batch of the exact update leaves the summaries of the blocks it touches exact (k_integrate restates them) and touches nothing else: views of the
// volume after it may go on using what is known.  Needs the summary array (a raycast has run on this volume) in the pool's current size and epoch.
bool vol_fusion_keeps_summaries(const op_volume* v) {
    const bool sum_form = v->update_mode == OP_VOLUME_UPDATE_SUM_FORM && v->trunc < 1.0f;
    return !sum_form && v->rc_sum != nullptr && v->rc_cap >= v->max_blocks && (v->content_gen >> 30) == v->rc_sum_epoch;
}

void launch_integrate(op_volume* v, const BatchInv& I, const CamParams& C, int nf) {
    const VolView V = v->view();
#define OP_KC(FASTPX, PLAINV, SUMFV) hipLaunchKernelGGL((k_integrate<FASTPX, PLAINV, (SUMFV ? KC_ZT_SUM : KC_ZT), SUMFV>), dim3(SUMFV ? kColGridSum : kColGrid), dim3(512 / (SUMFV ? KC_ZT_SUM : KC_ZT)), 0, v->stream, I, C, V, (const uint2*)v->pimg, \
                                                 v->state, nf, v->upd_partial, v->sel_partial, v->chg_partial, v->plain_from, rc_sum, rc_stamp)
    // The sum form tests TSDFVoxel::IsValid on the STORED voxel once per batch where the reference re-tests it before every frame (Integrator.cpp:74-87,
    // TSDFVoxel.h:75-78): the two agree to rounding only while no OBSERVATION can itself be invalid, i.e. truncation < 1 (an observed sdf is < truncation;
    // a stored voxel of any origin gets the test).  With truncation >= 1 the exact update runs, whatever the option says.
    const bool sum_form = v->update_mode == OP_VOLUME_UPDATE_SUM_FORM && v->trunc < 1.0f;
    // the raycaster's summaries are kept current by this launch (vol_fusion_keeps_summaries: the host has then NOT advanced the content generation)
    unsigned* rc_sum = vol_fusion_keeps_summaries(v) ? v->rc_sum : nullptr;
    const unsigned rc_stamp = (unsigned)(v->content_gen & 0x3fffffffull);
    if (sum_form) { if (C.fast_px) OP_KC(true, true, true); else OP_KC(false, true, true); }
    else if (C.fast_px) { if (v->plain) OP_KC(true, true, false); else OP_KC(true, false, false); }
    else { if (v->plain) OP_KC(false, true, false); else OP_KC(false, false, false); }
#undef OP_KC
}

void kc_trace_dump(op_volume* v) {
#ifdef KC_TRACE
    if (hipSetDevice(v->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
        const int nw = kColGrid * (8 / KC_ZT);
        std::vector<unsigned long long> t((size_t)4096 * 4 * 8);
        if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_kc_trace), t.size() * 8) == hipSuccess) {
            double sum[8] = {0}, mx[8] = {0};
            for (int w = 0; w < nw; ++w)
                for (int k = 0; k < 8; ++k) { const double d = (double)t[(size_t)w * 8 + k]; sum[k] += d; mx[k] = std::max(mx[k], d); }
            const double n = nw;
            fprintf(stderr, "kc trace (shader cycles per wave of the last launch, %d waves; mean/max): draw+list %.0f/%.0f metadata+voxels %.0f/%.0f frames %.0f/%.0f stores %.0f/%.0f barrier %.0f/%.0f tail %.0f/%.0f | blocks %.1f/%.0f frames applied %.0f/%.0f\n",
                    nw, sum[0] / n, mx[0], sum[1] / n, mx[1], sum[2] / n, mx[2], sum[3] / n, mx[3], sum[4] / n, mx[4], sum[5] / n, mx[5], sum[6] / n, mx[6], sum[7] / n, mx[7]);
        }
    }
#else
    (void)v;
#endif
}

} // namespace opv
