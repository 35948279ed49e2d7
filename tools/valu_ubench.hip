// valu_ubench.hip -- issue cost of the VALU operations k_integrate is made of, measured on the device it runs on.
// One wave per SIMD (grid = CUs x 4 waves of 64 threads, 256-thread workgroups), each running a long unrolled chain of
// INDEPENDENT instances of one operation (8 accumulators, so dependent-issue latency is hidden); s_memtime around the
// loop gives shader cycles per wave-instruction.  With 2 or 4 waves per SIMD the figure per SIMD stays the same when the
// operation is issue-bound.  Output: cycles per wave64 instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_ubench.hip -o tools/valu_ubench.bin && tools/valu_ubench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OP>
__global__ __launch_bounds__(256) void k_bench(float* out, unsigned long long* cyc, float a, float b, int iters) {
    float x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2 p0 = make_float2(x0, x1), p1 = make_float2(x2, x3), p2 = make_float2(x4, x5), p3 = make_float2(x6, x7), pb = make_float2(b, b);
    __shared__ float s_lds[256];
    s_lds[threadIdx.x] = a;
    const unsigned lds_off = threadIdx.x * 4u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(pb)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(pb)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(pb)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p3) : "v"(pb));) }
        if (OP == 3) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(x0)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x1)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x2)); asm volatile("v_rcp_f32 %0, %0" : "+v"(x3));) }
        if (OP == 4) { REP16(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x0) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x1) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x2) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x3) : "v"(b) : "vcc");) }
        if (OP == 5) { REP16(asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(b) : "vcc"); asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b) : "vcc"); asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(b) : "vcc"); asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(b) : "vcc");) }
        if (OP == 6) { REP16(asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(b)); asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b)); asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(b)); asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 7) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x1) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x2) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x3) : "v"(b));) }
        if (OP == 8) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x0), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x1), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x2), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x3), "v"(b) : "vcc");) }
        if (OP == 9) { REP16(asm volatile("v_mov_b32 %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 11) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x0) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x1) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x2) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x3) : "v"(b));) }
        if (OP == 12) { REP16(asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 13) { REP16(asm volatile("v_min_f32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 14) { REP16(asm volatile("v_max_u32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 15) { REP16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 16) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 17) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(x4)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x1) : "v"(x5)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x2) : "v"(x6)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x3) : "v"(x7));) }
        if (OP == 18) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 19) { REP16(asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p0), "v"(pb) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p1), "v"(pb) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p2), "v"(pb) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p3), "v"(pb) : "vcc");) }
        if (OP == 20) { REP16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x1)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x2) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 21) { REP16(asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc"); asm volatile("s_add_u32 s21, s21, 1" ::: "s21", "scc"); asm volatile("s_add_u32 s22, s22, 1" ::: "s22", "scc"); asm volatile("s_add_u32 s23, s23, 1" ::: "s23", "scc");) }
        if (OP == 22) { REP16(asm volatile("s_and_b64 s[20:21], s[20:21], exec" ::: "s20", "s21", "scc"); asm volatile("s_or_b64 s[22:23], s[22:23], exec" ::: "s22", "s23", "scc"); asm volatile("s_and_b64 s[24:25], s[24:25], exec" ::: "s24", "s25", "scc"); asm volatile("s_or_b64 s[26:27], s[26:27], exec" ::: "s26", "s27", "scc");) }
        if (OP == 23) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc"); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("s_add_u32 s21, s21, 1" ::: "s21", "scc");) }
        if (OP == 24) { REP16(asm volatile("s_nop 0"); asm volatile("s_nop 0"); asm volatile("s_nop 0"); asm volatile("s_nop 0");) }
        if (OP == 25) { REP16(asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x0) : "v"(b)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x1) : "v"(b)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x2) : "v"(b)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 26) { REP16(asm volatile("v_trunc_f32 %0, %0" : "+v"(x0)); asm volatile("v_trunc_f32 %0, %0" : "+v"(x1)); asm volatile("v_trunc_f32 %0, %0" : "+v"(x2)); asm volatile("v_trunc_f32 %0, %0" : "+v"(x3));) }
        if (OP == 27) { REP16(asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x0)); asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x1)); asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x2)); asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x3));) }
        if (OP == 28) { REP16(asm volatile("v_max_i32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(x1) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(x3) : "v"(b));) }
        if (OP == 29) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x0)); asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x1)); asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x2)); asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x3));) }
        if (OP == 30) { REP16(asm volatile("s_mul_i32 s20, s20, 3" ::: "s20"); asm volatile("s_mul_i32 s21, s21, 3" ::: "s21"); asm volatile("s_mul_i32 s22, s22, 3" ::: "s22"); asm volatile("s_mul_i32 s23, s23, 3" ::: "s23");) }
        if (OP == 31) { REP16(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");) }
        if (OP == 32) { asm volatile("s_cmp_eq_u32 s20, s20" ::: "scc"); REP16(asm volatile("s_cbranch_scc0 0" ::: ); asm volatile("s_cbranch_scc0 0" ::: ); asm volatile("s_cbranch_scc0 0" ::: ); asm volatile("s_cbranch_scc0 0" ::: );) }
        if (OP == 33) { REP16(asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(x0) : "s20"); asm volatile("v_readlane_b32 s21, %0, 3" :: "v"(x1) : "s21"); asm volatile("v_readlane_b32 s22, %0, 3" :: "v"(x2) : "s22"); asm volatile("v_readlane_b32 s23, %0, 3" :: "v"(x3) : "s23");) }
        if (OP == 34) { REP16(asm volatile("ds_read_b32 %0, %1" : "=v"(x4) : "v"(lds_off)); asm volatile("ds_read_b32 %0, %1" : "=v"(x5) : "v"(lds_off)); asm volatile("ds_read_b32 %0, %1" : "=v"(x6) : "v"(lds_off)); asm volatile("ds_read_b32 %0, %1" : "=v"(x7) : "v"(lds_off));) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 35) { REP16(asm volatile("s_load_dword s20, %0, 0x0" :: "s"(cyc) : "s20"); asm volatile("s_load_dword s21, %0, 0x0" :: "s"(cyc) : "s21"); asm volatile("s_load_dword s22, %0, 0x0" :: "s"(cyc) : "s22"); asm volatile("s_load_dword s23, %0, 0x0" :: "s"(cyc) : "s23");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 40) { // k_integrate's instruction-class ratio per voxel-frame wave (~95 VALU : 17 SALU : 6 branches : 3 SMEM), interleaved the way
                        // the kernel's frame loop is: six groups of 16 vector instructions, each followed by scalar bookkeeping and a not-taken branch
#define MIX_VALU16 \
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(b)); \
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x3) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x4) : "v"(b)); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x5), "v"(b) : "vcc"); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x6) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x7) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(b)); \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(x2) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x3) : "v"(b)); \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x4) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x5) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x6) : "v"(b)); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x7) : "v"(b));
#define MIX_SALU3 asm volatile("s_add_u32 s22, s22, 1" ::: "s22", "scc"); asm volatile("s_and_b64 s[24:25], s[24:25], exec" ::: "s24", "s25", "scc"); asm volatile("s_add_u32 s23, s23, 1" ::: "s23", "scc");
#define MIX_BRANCH asm volatile("s_cmp_eq_u32 s26, s26" ::: "scc"); asm volatile("s_cbranch_scc0 0" :::);
            MIX_VALU16 MIX_SALU3 MIX_BRANCH
            MIX_VALU16 MIX_SALU3 MIX_BRANCH asm volatile("s_load_dword s27, %0, 0x0" :: "s"(cyc) : "s27");
            MIX_VALU16 MIX_SALU3 MIX_BRANCH
            MIX_VALU16 MIX_SALU3 MIX_BRANCH asm volatile("s_load_dword s28, %0, 0x0" :: "s"(cyc) : "s28");
            MIX_VALU16 MIX_SALU3 MIX_BRANCH
            MIX_VALU16 MIX_SALU3 MIX_BRANCH asm volatile("s_load_dword s29, %0, 0x0" :: "s"(cyc) : "s29");
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (OP == 41) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(b));) } // ONE dependent chain: the latency a sequential float sum pays per term
        if (OP == 10) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(pb));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// s_memtime ticks per microsecond of wall time: the same wave reads s_memtime and s_memrealtime (constant 100 MHz) around a
// ~1 ms busy loop, so the per-instruction figures above convert to time (and to a fraction of a kernel's duration)
__global__ void k_clock(unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < 100000ull) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0; out[1] = r1 - r0;
}

template <int OP>
double run(const char* name, int waves_per_simd, float* d_out, unsigned long long* d_cyc, int cus, int iters) {
    const int wgs = cus * waves_per_simd; // 256 threads = 4 waves = one wave per SIMD of a CU (per resident workgroup)
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { hipEventCreate(&e0); hipEventCreate(&e1); }
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bench<OP>, dim3(wgs), dim3(256), 0, 0, d_out, d_cyc, 1.0f, 1.0000001f, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c((size_t)wgs * 4);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : c) s += (double)v;
    if (OP == 40) return s / c.size() / iters / waves_per_simd; // shader cycles per loop iteration and SIMD-issue slot
    const double per = s / c.size() / (64.0 * iters); // iters iterations x 64 instructions per wave
    std::printf("%-18s OP %2d waves/SIMD %d iters %d: %.2f s_memtime ticks per wave-instruction -> %.2f per SIMD-issue slot; kernel %.1f us = %.3f ns per instruction and SIMD (%.0f ticks per us while it ran)\n", name, OP, waves_per_simd, iters, per, per / waves_per_simd,
                ms * 1e3, ms * 1e6 / (64.0 * iters * waves_per_simd), (s / c.size()) / (ms * 1e3));
    return per / waves_per_simd;
}

int main(int argc, char** argv) {
    // argv[1]: loop iterations per kernel (64 instructions each; default 64).  With a large count (e.g. 1024) and ONE wave configuration
    // (argv[2] = waves per SIMD) the run is meant for `rocprofv3 --pmc GRBM_GUI_ACTIVE`: kernel cycles / (iters x 64 x waves) is then the
    // issue cost in the SAME clock the fusion kernels' GRBM_GUI_ACTIVE is counted in (tools/issue_model.py).
    const int iters = argc > 1 ? std::atoi(argv[1]) : 64;
    const int only_w = argc > 2 ? std::atoi(argv[2]) : 0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::printf("%s, %d CUs, clock %d kHz; s_memtime ticks at a constant 100 MHz on gfx9 -- ratios between rows are what matter\n", p.gcnArchName, cus, p.clockRate);
    float* d_out; unsigned long long* d_cyc;
    hipMalloc((void**)&d_out, (size_t)cus * 16 * 256 * 4);
    hipMalloc((void**)&d_cyc, (size_t)cus * 16 * 4 * 8);
    {
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, d_cyc);
        hipDeviceSynchronize();
        unsigned long long c[2];
        hipMemcpy(c, d_cyc, 16, hipMemcpyDeviceToHost);
        std::printf("clock: %llu s_memtime ticks in %llu s_memrealtime ticks (100 MHz) -> %.2f s_memtime ticks per microsecond\n", c[0], c[1], (double)c[0] / ((double)c[1] / 100.0));
    }
    for (int w : {1, 4, 8}) {
        if (only_w && w != only_w) continue;
        run<0>("v_mul_f32", w, d_out, d_cyc, cus, iters);
        run<1>("v_fma_f32", w, d_out, d_cyc, cus, iters);
        run<2>("v_pk_mul_f32", w, d_out, d_cyc, cus, iters);
        run<10>("v_pk_fma_f32", w, d_out, d_cyc, cus, iters);
        run<3>("v_rcp_f32", w, d_out, d_cyc, cus, iters);
        run<4>("v_div_scale_f32", w, d_out, d_cyc, cus, iters);
        run<5>("v_div_fmas_f32", w, d_out, d_cyc, cus, iters);
        run<6>("v_div_fixup_f32", w, d_out, d_cyc, cus, iters);
        run<7>("v_cndmask_b32", w, d_out, d_cyc, cus, iters);
        run<8>("v_cmp_lt_f32", w, d_out, d_cyc, cus, iters);
        run<9>("v_mov_b32", w, d_out, d_cyc, cus, iters);
        run<11>("v_cndmask e64 sgpr", w, d_out, d_cyc, cus, iters);
        run<17>("v_cndmask vcc 2src", w, d_out, d_cyc, cus, iters);
        run<20>("sub/mul/cndmask/add", w, d_out, d_cyc, cus, iters);
        run<12>("v_bfi_b32", w, d_out, d_cyc, cus, iters);
        run<13>("v_min_f32", w, d_out, d_cyc, cus, iters);
        run<14>("v_max_u32", w, d_out, d_cyc, cus, iters);
        run<15>("v_and_b32", w, d_out, d_cyc, cus, iters);
        run<16>("v_add_f32", w, d_out, d_cyc, cus, iters);
        run<18>("v_add_u32", w, d_out, d_cyc, cus, iters);
        run<19>("v_cmp_lt_u64", w, d_out, d_cyc, cus, iters);
        run<21>("s_add_u32", w, d_out, d_cyc, cus, iters);
        run<22>("s_and/or_b64", w, d_out, d_cyc, cus, iters);
        run<23>("v_mul + s_add mixed", w, d_out, d_cyc, cus, iters);
        run<24>("s_nop 0", w, d_out, d_cyc, cus, iters);
        run<25>("v_fmac_f32", w, d_out, d_cyc, cus, iters);
        run<26>("v_trunc_f32", w, d_out, d_cyc, cus, iters);
        run<27>("v_cvt_i32_f32", w, d_out, d_cyc, cus, iters);
        run<28>("v_max_i32", w, d_out, d_cyc, cus, iters);
        run<29>("v_lshlrev_b32", w, d_out, d_cyc, cus, iters);
        run<30>("s_mul_i32", w, d_out, d_cyc, cus, iters);
        run<31>("s_waitcnt (idle)", w, d_out, d_cyc, cus, iters);
        run<32>("s_cbranch not taken", w, d_out, d_cyc, cus, iters);
        run<33>("v_readlane_b32", w, d_out, d_cyc, cus, iters);
        run<34>("ds_read_b32", w, d_out, d_cyc, cus, iters);
        run<35>("s_load_dword", w, d_out, d_cyc, cus, iters);
    }
    run<41>("v_add_f32 dependent", 1, d_out, d_cyc, cus, iters);
    {   // Is the ADDITIVE issue model (tools/issue_model.py: every instruction class charged its own back-to-back issue cost, all into ONE budget per
        // SIMD) right for a kernel that mixes the classes?  One loop with k_integrate's class ratio -- 96 VALU : 18 SALU : 6 + 6 compare-and-branch :
        // 3 SMEM per iteration, 8 waves per SIMD -- is timed and compared with what the model predicts from the single-class rows measured above in
        // this very run (same clocks), and with the VALU-only prediction (scalar instructions of other waves co-issue with vector ones).
        const int w = 8;
        const double c_mul = run<0>("v_mul_f32", w, d_out, d_cyc, cus, iters), c_fma = run<1>("v_fma_f32", w, d_out, d_cyc, cus, iters),
                     c_add = run<16>("v_add_f32", w, d_out, d_cyc, cus, iters), c_cmp = run<8>("v_cmp_lt_f32", w, d_out, d_cyc, cus, iters),
                     c_cnd = run<11>("v_cndmask e64 sgpr", w, d_out, d_cyc, cus, iters), c_and = run<15>("v_and_b32", w, d_out, d_cyc, cus, iters),
                     c_sadd = run<21>("s_add_u32", w, d_out, d_cyc, cus, iters), c_sand = run<22>("s_and/or_b64", w, d_out, d_cyc, cus, iters),
                     c_br = run<32>("s_cbranch not taken", w, d_out, d_cyc, cus, iters);
        const double measured = run<40>("mixed", w, d_out, d_cyc, cus, iters);
        // per iteration: 6 x (5 mul, 4 fma, 3 add, 1 cmp, 2 cndmask, 1 and) vector; 6 x (2 s_add, 1 s_and) + 6 s_cmp scalar; 6 branches; 3 s_load (priced as a scalar slot, like the model)
        const double valu = 6 * (5 * c_mul + 4 * c_fma + 3 * c_add + 1 * c_cmp + 2 * c_cnd + 1 * c_and);
        const double salu = 6 * (2 * c_sadd + 1 * c_sand) + 6 * c_sadd + 3 * c_sadd, branch = 6 * c_br;
        std::printf("MIXED kernel (96 VALU : 24 SALU : 6 branch : 3 SMEM per iteration, %d waves/SIMD): measured %.1f shader cycles per iteration and SIMD-issue slot; "
                    "additive model %.1f (VALU %.1f + SALU/SMEM %.1f + branch %.1f) = %.3f x measured; VALU-only %.1f = %.3f x measured\n",
                    w, measured, valu + salu + branch, valu, salu, branch, (valu + salu + branch) / measured, valu, valu / measured);
    }
    return 0;
}
