// Per-instruction VALU throughput on one SIMD (gfx950), from kernel wall time: 4 waves per SIMD, each running ITER x 64
// copies of one instruction on four independent register chains.  Output: shader clocks per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "memory", "scc", "vcc", "s20", "s21", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"
#define KERNEL(name, body4)                                                                   \
    __global__ __launch_bounds__(256) void name(unsigned long long *out, float *sink) {       \
        float f = sink[threadIdx.x & 63];                                                     \
        unsigned long long t0, t1;                                                            \
        asm volatile("v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.5\n\tv_mov_b32 v22, 2.0\n\tv_mov_b32 v23, 3.0\n\tv_mov_b32 v24, 1.25\n\tv_mov_b32 v25, 1.0\n\tv_mov_b32 v26, 0.5\n\tv_mov_b32 v27, 0x7fc00000\n\t" \
                     "v_mov_b32 v28, 1.5\n\tv_mov_b32 v29, 1.5\n\tv_mov_b32 v30, 3.0\n\tv_mov_b32 v31, 3.0\n\tv_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 1.0\n\tv_mov_b32 v34, 1.0\n\tv_mov_b32 v35, 1.0\n\t" ::: CLOB); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                      \
        for (int i = 0; i < ITER; ++i) { asm volatile(REP16(body4) ::: CLOB); }               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                            \
        sink[threadIdx.x & 63] = f;                                                           \
    }
#define I4(op, tail) op " v32, " tail "\n\t" op " v33, " tail "\n\t" op " v34, " tail "\n\t" op " v35, " tail "\n\t"
KERNEL(k_add_vop2, I4("v_add_f32", "v28, v30"))
KERNEL(k_mul_vop2, I4("v_mul_f32", "v28, v30"))
KERNEL(k_mov, I4("v_mov_b32", "v28"))
KERNEL(k_fmac_vop2, I4("v_fmac_f32", "v28, v30"))
KERNEL(k_fma_vop3, I4("v_fma_f32", "v28, v30, v24"))
KERNEL(k_fma_neg_const, I4("v_fma_f32", "-v28, v30, 1.0"))
KERNEL(k_add_vop3_abs, I4("v_add_f32_e64", "v28, |v30|"))
KERNEL(k_div_scale, "v_div_scale_f32 v32, s[20:21], v28, v28, v30\n\tv_div_scale_f32 v33, s[20:21], v29, v29, v31\n\tv_div_scale_f32 v34, s[20:21], v28, v28, v31\n\tv_div_scale_f32 v35, s[20:21], v29, v29, v30\n\t")
KERNEL(k_div_scale_vcc, "v_div_scale_f32 v32, vcc, v30, v28, v30\n\tv_div_scale_f32 v33, vcc, v31, v29, v31\n\tv_div_scale_f32 v34, vcc, v30, v29, v30\n\tv_div_scale_f32 v35, vcc, v31, v28, v31\n\t")
KERNEL(k_div_fmas, I4("v_div_fmas_f32", "v28, v30, v24"))
KERNEL(k_div_fixup, I4("v_div_fixup_f32", "v28, v30, v24"))
KERNEL(k_rcp, I4("v_rcp_f32", "v28"))
KERNEL(k_cmp, "v_cmp_neq_f32 vcc, 0, v28\n\tv_cmp_neq_f32 vcc, 0, v29\n\tv_cmp_neq_f32 vcc, 0, v30\n\tv_cmp_neq_f32 vcc, 0, v31\n\t")
KERNEL(k_cndmask, I4("v_cndmask_b32", "v27, v28, vcc"))
KERNEL(k_sub_vop2, I4("v_sub_f32", "v28, v30"))
KERNEL(k_add_u32, I4("v_add_u32", "s20, v28"))
KERNEL(k_add_sgpr, I4("v_add_f32", "s20, v28"))
// packed fp32: two rows per instruction (even-aligned register pairs)
KERNEL(k_pk_add, "v_pk_add_f32 v[32:33], v[28:29], v[30:31]\n\tv_pk_add_f32 v[34:35], v[28:29], v[30:31]\n\tv_pk_add_f32 v[36:37], v[28:29], v[30:31]\n\tv_pk_add_f32 v[38:39], v[28:29], v[30:31]\n\t")
KERNEL(k_pk_mul, "v_pk_mul_f32 v[32:33], v[28:29], v[30:31]\n\tv_pk_mul_f32 v[34:35], v[28:29], v[30:31]\n\tv_pk_mul_f32 v[36:37], v[28:29], v[30:31]\n\tv_pk_mul_f32 v[38:39], v[28:29], v[30:31]\n\t")
KERNEL(k_pk_fma, "v_pk_fma_f32 v[32:33], v[28:29], v[30:31], v[24:25]\n\tv_pk_fma_f32 v[34:35], v[28:29], v[30:31], v[24:25]\n\tv_pk_fma_f32 v[36:37], v[28:29], v[30:31], v[24:25]\n\tv_pk_fma_f32 v[38:39], v[28:29], v[30:31], v[24:25]\n\t")
KERNEL(k_pk_mov, "v_pk_mov_b32 v[32:33], v[28:29], v[30:31]\n\tv_pk_mov_b32 v[34:35], v[28:29], v[30:31]\n\tv_pk_mov_b32 v[36:37], v[28:29], v[30:31]\n\tv_pk_mov_b32 v[38:39], v[28:29], v[30:31]\n\t")
// VGPR indexing (the interpreter's operand stack): the same adds with s_set_gpr_idx_on active
#define KERNEL_IDX(name, mode, body4)                                                         \
    __global__ __launch_bounds__(256) void name(unsigned long long *out, float *sink) {       \
        float f = sink[threadIdx.x & 63];                                                     \
        unsigned long long t0, t1;                                                            \
        asm volatile("v_mov_b32 v28, 1.5\n\tv_mov_b32 v29, 1.5\n\tv_mov_b32 v30, 3.0\n\tv_mov_b32 v31, 3.0\n\tv_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 1.0\n\tv_mov_b32 v34, 1.0\n\tv_mov_b32 v35, 1.0\n\t" ::: CLOB); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                      \
        asm volatile("s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, " mode ::: CLOB, "m0");      \
        for (int i = 0; i < ITER; ++i) { asm volatile(REP16(body4) ::: CLOB); }               \
        asm volatile("s_set_gpr_idx_off" ::: CLOB, "m0");                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                            \
        sink[threadIdx.x & 63] = f;                                                           \
    }
KERNEL_IDX(k_add_idx_src0, "0x1", I4("v_add_f32", "v28, v30"))
KERNEL_IDX(k_add_idx_dst, "0x8", I4("v_add_f32", "v28, v30"))
KERNEL_IDX(k_add_idx_s0s1d, "0xb", I4("v_add_f32", "v28, v30"))
KERNEL_IDX(k_mov_idx_dst, "0x8", I4("v_mov_b32", "v28"))
KERNEL_IDX(k_fixup_idx_dst, "0x8", I4("v_div_fixup_f32", "v28, v30, v24"))
// the full division row as generated (dependent chain), 4 rows
#define DIVROW(x, y, q) \
    "v_cmp_neq_f32 vcc, 0, " y "\n\tv_cndmask_b32 " x ", v27, " x ", vcc\n\tv_div_scale_f32 v36, s[20:21], " y ", " y ", " x "\n\tv_rcp_f32 v37, v36\n\tv_div_scale_f32 v38, vcc, " x ", " y ", " x "\n\t" \
    "v_fma_f32 v39, -v36, v37, 1.0\n\tv_fmac_f32 v37, v39, v37\n\tv_mul_f32 v39, v38, v37\n\tv_fma_f32 v20, -v36, v39, v38\n\tv_fmac_f32 v39, v20, v37\n\tv_fma_f32 v36, -v36, v39, v38\n\tv_div_fmas_f32 " q ", v36, v37, v39\n\t"
KERNEL(k_divrow4, DIVROW("v30", "v28", "v32") DIVROW("v31", "v29", "v33") DIVROW("v30", "v29", "v34") DIVROW("v31", "v28", "v35"))

// integer / bit work of the program compilers and of tree_generate (round 6): is anything but the fp32 VOP2 forms double rate?
KERNEL(k_and_vvv, I4("v_and_b32", "v28, v30"))
KERNEL(k_and_lit, I4("v_and_b32", "0xffe00000, v30"))
KERNEL(k_and_inl, I4("v_and_b32", "7, v30"))
KERNEL(k_addu_vvv, I4("v_add_u32", "v28, v30"))
KERNEL(k_addu_inl, I4("v_add_u32", "1, v30"))
KERNEL(k_lshl_vvv, I4("v_lshlrev_b32", "v28, v30"))
KERNEL(k_lshl_inl, I4("v_lshlrev_b32", "3, v30"))
KERNEL(k_xor_vvv, I4("v_xor_b32", "v28, v30"))
KERNEL(k_bfe, I4("v_bfe_u32", "v28, 3, 5"))
KERNEL(k_and_or, I4("v_and_or_b32", "v28, v30, v24"))
KERNEL(k_lshl_add, I4("v_lshl_add_u32", "v28, 2, v24"))
KERNEL(k_cmp_u32, "v_cmp_eq_u32 vcc, v28, v30\n\tv_cmp_eq_u32 vcc, v29, v30\n\tv_cmp_eq_u32 vcc, v28, v31\n\tv_cmp_eq_u32 vcc, v29, v31\n\t")
KERNEL(k_cndmask_s, I4("v_cndmask_b32_e64", "v27, v28, s[20:21]"))
KERNEL(k_mul_lo, I4("v_mul_lo_u32", "v28, v30"))
KERNEL(k_mul_u24, I4("v_mul_u32_u24", "v28, v30"))
KERNEL(k_mov_dpp, I4("v_mov_b32_dpp", "v28 wave_shl:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_mbcnt, I4("v_mbcnt_lo_u32_b32", "s20, v28"))
KERNEL(k_cvt_f32_u32, I4("v_cvt_f32_u32", "v28"))
KERNEL(k_add_f32_inl, I4("v_add_f32", "1.0, v30"))
KERNEL(k_mul_f32_lit, I4("v_mul_f32", "0x2f800000, v30"))
KERNEL(k_max_f32, I4("v_max_f32", "v28, v30"))
KERNEL(k_min_u32, I4("v_min_u32", "v28, v30"))
KERNEL(k_readlane, "v_readlane_b32 s20, v28, 3\n\tv_readlane_b32 s21, v29, 3\n\tv_readlane_b32 s20, v30, 5\n\tv_readlane_b32 s21, v31, 7\n\t")
KERNEL(k_salu, "s_add_u32 s20, s20, 1\n\ts_and_b32 s21, s21, 7\n\ts_add_u32 s20, s20, 1\n\ts_and_b32 s21, s21, 7\n\t")

// v_cndmask forms (the VOP2 form read 21.9 ticks in round 1's log: is it the form, the NaN source, or vcc never written?)
KERNEL(k_cndmask_v24, I4("v_cndmask_b32", "v24, v28, vcc"))
KERNEL(k_cndmask_e64vcc, I4("v_cndmask_b32_e64", "v24, v28, vcc"))
KERNEL(k_cndmask_wr, "v_cmp_neq_f32 vcc, 0, v28\n\ts_nop 1\n\tv_cndmask_b32 v32, v24, v28, vcc\n\tv_cndmask_b32 v33, v24, v28, vcc\n\tv_cndmask_b32 v34, v24, v28, vcc\n\t")
KERNEL(k_cndmask_inl, I4("v_cndmask_b32", "0, v28, vcc"))
KERNEL(k_or_vvv, I4("v_or_b32", "v28, v30"))
KERNEL(k_subu_vvv, I4("v_sub_u32", "v28, v30"))
KERNEL(k_lshr_inl, I4("v_lshrrev_b32", "3, v30"))
KERNEL(k_bfi, I4("v_bfi_b32", "v28, v30, v24"))
KERNEL(k_and_sgpr, I4("v_and_b32", "s20, v30"))
KERNEL(k_cmp_e64, "v_cmp_eq_u32_e64 s[20:21], v28, v30\n\tv_cmp_eq_u32_e64 s[20:21], v29, v30\n\tv_cmp_eq_u32_e64 s[20:21], v28, v31\n\tv_cmp_eq_u32_e64 s[20:21], v29, v31\n\t")

template <typename K> static void run(const char *name, K kern, int instrs_per_rep, unsigned long long *dout, float *dsink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, dsink);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, dsink);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ticks = 0; (void)hipMemcpy(&ticks, dout, 8, hipMemcpyDeviceToHost);
    const double n = (double)ITER * 16 * instrs_per_rep;     // instructions per wave
    const double ghz = ticks / (ms * 1e6);                    // wave 0 ran for ~ the whole kernel
    printf("%-18s kernel %.3f ms  ticks/instr/wave %6.2f  clocks/instr/SIMD %6.2f  (clock %.2f GHz)\n", name, ms, ticks / n, ticks / n / 4.0, ghz);
}
int main() {
    unsigned long long *dout; float *dsink;
    (void)hipMalloc(&dout, 64); (void)hipMalloc(&dsink, 256); (void)hipMemset(dsink, 0, 256);
#define RUN(k, n) run(#k, k, n, dout, dsink)
    RUN(k_add_vop2, 4); RUN(k_mul_vop2, 4); RUN(k_sub_vop2, 4); RUN(k_mov, 4); RUN(k_add_sgpr, 4); RUN(k_add_u32, 4); RUN(k_fmac_vop2, 4); RUN(k_fma_vop3, 4); RUN(k_fma_neg_const, 4);
    RUN(k_add_vop3_abs, 4); RUN(k_div_scale, 4); RUN(k_div_scale_vcc, 4); RUN(k_div_fmas, 4); RUN(k_div_fixup, 4); RUN(k_rcp, 4); RUN(k_cmp, 4); RUN(k_cndmask, 4);
    RUN(k_pk_add, 4); RUN(k_pk_mul, 4); RUN(k_pk_fma, 4); RUN(k_pk_mov, 4);
    RUN(k_add_idx_src0, 4); RUN(k_add_idx_dst, 4); RUN(k_add_idx_s0s1d, 4); RUN(k_mov_idx_dst, 4); RUN(k_fixup_idx_dst, 4);
    RUN(k_divrow4, 48);
    RUN(k_add_vop2, 4); RUN(k_pk_add, 4); RUN(k_add_vop2, 4); RUN(k_pk_fma, 4); RUN(k_fma_vop3, 4);
    RUN(k_and_vvv, 4); RUN(k_and_lit, 4); RUN(k_and_inl, 4); RUN(k_addu_vvv, 4); RUN(k_addu_inl, 4); RUN(k_lshl_vvv, 4); RUN(k_lshl_inl, 4); RUN(k_xor_vvv, 4); RUN(k_bfe, 4);
    RUN(k_and_or, 4); RUN(k_lshl_add, 4); RUN(k_cmp_u32, 4); RUN(k_cndmask_s, 4); RUN(k_mul_lo, 4); RUN(k_mul_u24, 4); RUN(k_mov_dpp, 4); RUN(k_mbcnt, 4); RUN(k_cvt_f32_u32, 4);
    RUN(k_add_f32_inl, 4); RUN(k_mul_f32_lit, 4); RUN(k_max_f32, 4); RUN(k_min_u32, 4); RUN(k_readlane, 4); RUN(k_salu, 4); RUN(k_add_vop2, 4);
    RUN(k_cndmask, 4); RUN(k_cndmask_v24, 4); RUN(k_cndmask_e64vcc, 4); RUN(k_cndmask_wr, 4); RUN(k_cndmask_inl, 4); RUN(k_or_vvv, 4); RUN(k_subu_vvv, 4); RUN(k_lshr_inl, 4); RUN(k_bfi, 4); RUN(k_and_sgpr, 4); RUN(k_cmp_e64, 4);
    return 0;
}
