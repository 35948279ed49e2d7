// seq_sums.hpp -- sequential float32 sums on the device: the reference adds long runs of products to float32 accumulators ONE AFTER THE OTHER
// (the dense tracker's J J^T / J r over the accepted pixels, DenseOdometryFunction.cpp:297-381; point-to-plane ICP's over the inliers, ICP.cpp:121-136),
// and the rounding of those 10^5..10^6 dependent additions is part of its result -- it moves a pose by up to 2e-4 in the tracker and, where the 6x6
// system sits at JacobiSVD's rank threshold, by up to 5e-2 in ICP (DESIGN.md sections 5, 7).  A sequential float sum cannot be re-associated; what CAN
// run side by side are its accumulators.  Included by odometry.hip and icp.hip (each translation unit gets its own copy of the kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <vector>

namespace {

constexpr int kSeqRows = 384;          // rows per tile
constexpr int kSeqStride = kSeqRows + 4; // floats between two accumulators' rows in LDS: 4 (mod 32) spreads the lanes' 16-byte reads over the banks
constexpr int kSeqProducers = 2 * kSeqRows; // 12 producer waves: producer p owns row p % kSeqRows and every second accumulator
constexpr int kSeqThreads = 1024;      // wave 0 sums; waves 4, 8 and 12 -- the ones that share its SIMD -- only keep the barriers company, the other 12 produce

// products of one row for the accumulators k = H, H + 2, ...: everything but the row's address is a compile-time constant
template <int NACC, int H>
__device__ __forceinline__ void seq_produce_row(const float* __restrict__ r, float* __restrict__ pd_row) {
    if (NACC == 42) {
        float J[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) J[i] = r[i];
#pragma unroll
        for (int k = H; k < 42; k += 2) pd_row[k * kSeqStride] = k < 36 ? J[k / 6] * J[k % 6] : J[k - 36] * J[6];
    } else {
        if (H < NACC) pd_row[H * kSeqStride] = r[H];
    }
}

// The consumer's schedule: a dependent v_add_f32 can issue every ~8 cycles, an instruction every 4 -- so the eight 16-byte LDS reads that refill one register set are
// issued one by one in the shadow of the adds that drain the other (one ds_read, then four adds: sched_group_barrier masks 0x100 = DS read, 0x2 = VALU) instead of in a
// burst in front of them, where their issue cycles add to the chain.  Measured: +2 % (ICP 586 -> 596 iterations/s, 1.4 ms of k_seq_sums per 3e5 rows either way): the
// chain itself runs at ~9.8 cycles per add in this kernel against 8.25 in the bare microbenchmark, and that is where the time is.  Scheduling only: same adds, same order.
#ifndef SEQ_NO_INTERLEAVE
#define SEQ_INTERLEAVE() do { _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 4, 0); } } while (0)
#else
#define SEQ_INTERLEAVE() do { } while (0)
#endif
// Sequential float32 sums of NACC accumulators over n_pix compacted pixels of NF floats each, RPP rows per pixel.
//   NACC 42 (NF 7 * RPP): row = {J[6], r}; accumulator a*6+b += J[a]*J[b] (a, b < 6), accumulator 36+a += J[a]*r   -- the order of
//                    op_host::track_sums_reference_order: per pixel row 0 then row 1, per accumulator one rounded product and one rounded add
//   NACC 2  (NF 2):  accumulator k += value k of the pixel (NormalizeIntensity's two sums)
// One workgroup.  Wave 0 sums: lane k owns accumulator k and reads four consecutive rows of it per ds_read_b128, the next 32
// rows always in flight (two register sets) so that its only cost per row is the dependent add -- 8.25 shader cycles on this chip
// (tools/valu_ubench.hip OP 41), the floor of any sequential float32 sum; measured here: ~10 per row.  Twelve waves on the other SIMDs produce: thread p owns row
// p % 384 of the NEXT tile and every second accumulator -- 7 LDS reads, 21 multiplies, 21 LDS writes at constant offsets -- and stages
// the rows of the tile after that (global loads in flight while it multiplies).  out[0 .. NACC-1] = the sums, ((unsigned*)out)[NACC] = n_pix.
// Rows beyond the last pixel are products of zeros: acc + (+0.0f) == acc for every acc this loop can hold (it starts at +0 and a float
// sum only yields -0 from -0 + -0), so every tile is summed over all of its 384 rows.
template <int NACC, int NF, int RPP>
__device__ __forceinline__ void seq_sums_body(const float* __restrict__ rows, const unsigned* __restrict__ n_pix_ptr, float* __restrict__ out) {
    extern __shared__ float seq_lds[];
    constexpr int P = kSeqRows / RPP;                        // pixels per tile
    constexpr int RF = NF / RPP;                             // floats per row
    constexpr int kStage = P * NF;                           // floats of one staged tile
    constexpr int kLoads = (kStage + kSeqProducers - 1) / kSeqProducers;
    float* prod = seq_lds;                                   // [2][NACC][kSeqStride]
    float* stage = seq_lds + 2 * NACC * kSeqStride;          // [2][kStage]
    const unsigned n_pix = *n_pix_ptr;
    const unsigned n_tiles = (n_pix + (unsigned)P - 1u) / (unsigned)P;
    const int tid = threadIdx.x;
    const bool consumer = tid < 64;
    const int wave = tid >> 6;
    const bool idle = wave != 0 && (wave & 3) == 0;          // same SIMD as the summing wave (waves go to the SIMDs round-robin): nothing may delay its adds
    const int pj = (wave - 1 - (wave >> 2)) * 64 + (tid & 63); // producer index 0 .. 767 (meaningless for wave 0 and the idle waves)
    const int prow = pj >= kSeqRows ? pj - kSeqRows : pj;    // its row of the tile
    auto load_tile = [&](unsigned tile, float (&reg)[kLoads]) { // global -> registers (zeros beyond the data)
        const size_t base = (size_t)tile * kStage, end = (size_t)n_pix * NF;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int e = pj + i * kSeqProducers;
            reg[i] = (e < kStage && base + (size_t)e < end) ? rows[base + (size_t)e] : 0.0f;
        }
    };
    auto store_tile = [&](int buf, const float (&reg)[kLoads]) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int e = pj + i * kSeqProducers;
            if (e < kStage) stage[buf * kStage + e] = reg[i];
        }
    };
    auto produce = [&](int buf) { // stage[buf] -> prod[buf]: RPP rows of RF floats per pixel, row * RF == pixel * NF + (row % RPP) * RF
        const float* r = stage + buf * kStage + prow * RF;
        float* pd_row = prod + buf * (NACC * kSeqStride) + prow;
        if (pj < kSeqRows) seq_produce_row<NACC, 0>(r, pd_row); // (wave-uniform: 6 waves per half)
        else seq_produce_row<NACC, 1>(r, pd_row);
    };
    float acc = 0.0f;
    float reg[kLoads];
    // prologue: tile 0 staged and produced, tile 1 staged
    const bool producer = !consumer && !idle;
    if (producer && n_tiles) { load_tile(0, reg); store_tile(0, reg); }
    __syncthreads();
    if (producer && n_tiles) { produce(0); load_tile(1, reg); store_tile(1, reg); }
    __syncthreads();
    if (consumer) __builtin_amdgcn_s_setprio(3);
    for (unsigned t = 0; t < n_tiles; ++t) {
        const int cur = (int)(t & 1u);
        if (consumer) {
            if (tid < NACC) {
                const float4* src = reinterpret_cast<const float4*>(prod + cur * (NACC * kSeqStride) + tid * kSeqStride);
                constexpr int kChunk = 8, kChunks = kSeqRows / 4 / kChunk; // 8 x 16 bytes = 32 rows per register set, 12 sets per tile
                static_assert(kChunks % 2 == 0, "two register sets alternate");
                float4 A[kChunk], B[kChunk];
#pragma unroll
                for (int i = 0; i < kChunk; ++i) A[i] = src[i];
#pragma unroll 1
                for (int c = 0; c < kChunks; c += 2) {
                    // (one basic block: the refill of A for the next trip is unconditional -- on the last trip it re-reads the tile's first rows and is discarded)
                    const int cn = c + 2 < kChunks ? c + 2 : 0;
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) B[i] = src[(c + 1) * kChunk + i];
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) { acc += A[i].x; acc += A[i].y; acc += A[i].z; acc += A[i].w; }
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) A[i] = src[cn * kChunk + i];
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) { acc += B[i].x; acc += B[i].y; acc += B[i].z; acc += B[i].w; }
                    SEQ_INTERLEAVE();
                }
            }
        } else if (producer) {
            if (t + 2 < n_tiles) load_tile(t + 2, reg);      // in flight while the products are formed
            if (t + 1 < n_tiles) produce(cur ^ 1);           // tile t + 1 from stage[cur ^ 1]
            if (t + 2 < n_tiles) store_tile(cur, reg);       // stage[cur] held tile t: consumed by produce() one iteration ago
        }
        __syncthreads();
    }
    if (consumer && tid < NACC) out[tid] = acc;
    if (tid == 0) reinterpret_cast<unsigned*>(out)[NACC] = n_pix;
}
template <int NACC, int NF, int RPP>
__global__ __launch_bounds__(kSeqThreads) void k_seq_sums(const float* __restrict__ rows, const unsigned* __restrict__ n_pix_ptr, float* __restrict__ out) {
    seq_sums_body<NACC, NF, RPP>(rows, n_pix_ptr, out);
}
// Several independent problems in ONE launch, a workgroup (= one summing wave + its producers, one CU) each.  Why it exists: kernels of different streams only run side by side when
// the streams sit on different HARDWARE queues, and the runtime maps all streams of the process onto GPU_MAX_HW_QUEUES of them (default 4; tools/queue_probe.hip: K
// one-workgroup kernels on K streams take ceil(K / queues) kernel times) -- K workgroups of one launch have no such limit (ICP's reference-order replicas: op_icp_run_many).
constexpr int kSeqBatchMax = 32;
struct SeqBatchTable { const float* rows[kSeqBatchMax]; const unsigned* n_pix[kSeqBatchMax]; float* out[kSeqBatchMax]; };
template <int NACC, int NF, int RPP>
__global__ __launch_bounds__(kSeqThreads) void k_seq_sums_many(SeqBatchTable t) {
    seq_sums_body<NACC, NF, RPP>(t.rows[blockIdx.x], t.n_pix[blockIdx.x], t.out[blockIdx.x]);
}
constexpr size_t seq_lds_bytes(int nacc, int nf, int rpp) { return sizeof(float) * (2 * (size_t)nacc * kSeqStride + 2 * (size_t)(kSeqRows / rpp) * nf); }

// ---- several host threads take their sequential sums TOGETHER ------------------------------------------------------------------------------------------------------
// Every caller has its own stream (an ICP context's, a tracker's) and a host thread that needs the sums before it can go on.  K such threads launching k_seq_sums on K
// streams scale to the number of hardware queues of the process and no further (see k_seq_sums_many; two streams on one queue take turns).  With MINP or more participants they meet instead, once per iteration: a thread records "my rows are in
// place" on its stream (the request's event) and waits; the last one to arrive launches k_seq_sums_many -- a workgroup per waiting request -- on the rendezvous's own
// stream behind those events, copies the NACC + 1 numbers of every request to its pinned buffer, synchronises and releases everybody.  A participant that has nothing
// to sum in a round says so (pass), one that is done leaves; a waiter that is not released within a few milliseconds launches what is pending itself, so progress never
// depends on the count being right.  Below MINP participants submit() answers hipErrorNotReady and the caller launches its own kernel as before (that many
// independent launches run side by side on their own hardware queues; meeting only costs then).  Results do not depend on who sums with whom: the kernel body and its inputs are the stand-alone launch's.
// Measured (307 200-point ICP pairs, reference-order mode, profiles/r06_icp_hw_queues.txt): with 16 hardware queues 16 independent runs reach 4.1 k iterations/s, the
// rendezvous 5.3-5.7 k (8 contexts: 3.9 k alone, 3.1-3.6 k together -- hence MINP 9 for ICP); with 4 queues 2.0 k against 6.5 k.  For the tracker it does not pay (odometry.hip).  A variant without rounds (a free "lane" takes whatever is pending) was
// slower at every depth: the first arrival of a wave launches alone and the rest wait a whole kernel for the next lane.
struct SeqRequest { const float* rows; const unsigned* n_pix; float* out; float* host_out; hipEvent_t ready; hipError_t status; };
template <int NACC, int NF, int RPP, int MINP>
struct SeqRendezvous {
    std::mutex mu;
    std::condition_variable cv;
    int participants = 0, arrived = 0, device = -1;
    unsigned long long generation = 0;
    hipStream_t stream = nullptr;
    bool ready = false;
    std::vector<SeqRequest*> pending;
    // false: cannot be used on this device (no stream / no LDS opt-in, or bound to another device): the caller sums alone
    bool usable(int dev) {
        std::lock_guard<std::mutex> lk(mu);
        if (ready) return device == dev;
        if (device >= 0) return false; // (a failed attempt is not repeated)
        device = dev;
        if (hipSetDevice(dev) != hipSuccess) return false;
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seq_sums_many<NACC, NF, RPP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(NACC, NF, RPP)) == hipSuccess;
        if (ok) ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        ready = ok;
        return ok;
    }
    void flush_locked() { // mu held
        if (!pending.empty()) {
            hipError_t e = hipSetDevice(device);
            SeqBatchTable t{};
            const size_t n = pending.size();
            for (size_t i = 0; i < n && e == hipSuccess; ++i) {
                e = hipStreamWaitEvent(stream, pending[i]->ready, 0);
                t.rows[i] = pending[i]->rows; t.n_pix[i] = pending[i]->n_pix; t.out[i] = pending[i]->out;
            }
            if (e == hipSuccess) {
                hipLaunchKernelGGL((k_seq_sums_many<NACC, NF, RPP>), dim3((unsigned)n), dim3(kSeqThreads), seq_lds_bytes(NACC, NF, RPP), stream, t);
                e = hipGetLastError();
            }
            for (size_t i = 0; i < n && e == hipSuccess; ++i) e = hipMemcpyAsync(pending[i]->host_out, pending[i]->out, (NACC + 1) * sizeof(float), hipMemcpyDeviceToHost, stream);
            const hipError_t es = hipStreamSynchronize(stream); // (also after a failure: nothing enqueued may outlive the callers' buffers)
            if (e == hipSuccess) e = es;
            for (size_t i = 0; i < n; ++i) pending[i]->status = e; // every request of the launch learns how it went
            pending.clear();
        }
        arrived = 0;
        ++generation;
        cv.notify_all();
    }
    void join() { std::lock_guard<std::mutex> lk(mu); ++participants; }
    void pass() { // this participant has nothing to sum in this round
        std::lock_guard<std::mutex> lk(mu);
        if (++arrived >= participants) flush_locked();
    }
    void leave() { // this participant's loop is over
        std::lock_guard<std::mutex> lk(mu);
        --participants;
        if (participants > 0 && arrived >= participants) flush_locked();
        if (participants <= 0) { participants = 0; arrived = 0; }
    }
    // hipSuccess: host_out holds the sums.  hipErrorNotReady: fewer than MINP participants -- the caller launches its own kernel (the rows and n_pix are on `stream`).
    hipError_t submit(const float* rows, const unsigned* n_pix, float* out, float* host_out, hipEvent_t ev, hipStream_t stream_of_rows) {
        SeqRequest req{rows, n_pix, out, host_out, ev, hipSuccess};
        std::unique_lock<std::mutex> lk(mu);
        if (participants < MINP) { // too few to be worth meeting: counts as "nothing from me this round" for whoever does wait
            if (++arrived >= participants) flush_locked();
            return hipErrorNotReady;
        }
        const hipError_t er = hipEventRecord(ev, stream_of_rows);
        if (er != hipSuccess) { if (++arrived >= participants) flush_locked(); return er; }
        pending.push_back(&req);
        if (++arrived >= participants || pending.size() >= (size_t)kSeqBatchMax) flush_locked();
        else {
            const unsigned long long g = generation;
            while (generation == g)
                if (cv.wait_for(lk, std::chrono::milliseconds(5)) == std::cv_status::timeout && generation == g) flush_locked(); // (safety valve)
        }
        return req.status;
    }
};

} // namespace
