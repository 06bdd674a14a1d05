// Issue-model micro-benchmarks (gfx950): how do VALU / SALU / branch instructions of SEVERAL waves on one SIMD
// share the issue slots?  Every kernel times ITER iterations of a pattern with s_memtime, one wave per block;
// blocks = 1024 * W puts W waves on every SIMD.  Output: shader-clock ticks per pattern per wave, and
// "per SIMD" = that divided by W (aggregate cost of one pattern instance on a SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 issue_model.hip -o issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 1000
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "memory", "scc", "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

#define KERNEL(name, setup, body, ninstr)                                                     \
    __global__ void name(unsigned long long *out, float *sink) {                              \
        __shared__ float lds[1024];                                                           \
        lds[threadIdx.x] = sink[threadIdx.x]; lds[threadIdx.x + 64] = 1.0f;                   \
        float f = sink[threadIdx.x];                                                          \
        unsigned long long t0, t1;                                                            \
        asm volatile("v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v22, 1.0\n\tv_mov_b32 v23, 1.0\n\tv_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 1.0\n\tv_mov_b32 v26, 1.0\n\tv_mov_b32 v27, 1.0\n\t" \
                     "v_mov_b32 v28, 1.5\n\tv_mov_b32 v29, 1.5\n\tv_mov_b32 v30, 3.0\n\tv_mov_b32 v31, 3.0\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t" \
                     "s_mov_b32 s20, 0\n\ts_mov_b32 s21, 0\n\ts_mov_b32 s24, 0\n\ts_mov_b32 s25, 0\n\ts_mov_b32 s26, 0\n\ts_mov_b32 s27, 0\n\t" setup ::: CLOB); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                      \
        for (int i = 0; i < ITER; ++i) {                                                      \
            asm volatile(body ::: CLOB);                                                      \
        }                                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                      \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                      \
        sink[threadIdx.x] = f + lds[threadIdx.x];                                             \
    }                                                                                         \
    static const int name##_n = ninstr;

#define V4 "v_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v24\n\tv_add_f32 v22, v22, v24\n\tv_add_f32 v23, v23, v24\n\t"
#define S1 "s_add_u32 s20, s20, 1\n\t"
#define V1a "v_add_f32 v20, v20, v24\n\t"
#define V1b "v_add_f32 v21, v21, v24\n\t"

KERNEL(k_valu16, "", REP4(V4), 16)
KERNEL(k_salu16, "", REP16(S1), 16)
KERNEL(k_mix_1v1s, "", REP8(V1a S1), 16)
KERNEL(k_mix_1v3s, "", REP4(V1a S1 S1 S1), 16)
KERNEL(k_mix_3v1s, "", REP4(V1a V1b V1a S1), 16)
KERNEL(k_mix_4v4s, "", REP2(V4 S1 S1 S1 S1), 16)
KERNEL(k_mix_8v8s, "", V4 V4 REP8(S1), 16)
KERNEL(k_branch16, "", REP16("s_branch 1f\n\t1:\n\t"), 16)
KERNEL(k_v4_branch, "", REP4(V4 "s_branch 1f\n\t1:\n\t"), 20)
KERNEL(k_v8_branch, "", REP2(V4 V4 "s_branch 1f\n\t1:\n\t"), 18)
KERNEL(k_cbr_nt16, "s_cmp_eq_u32 s20, 77\n\t", REP16("s_cbranch_scc1 1f\n\t1:\n\t"), 16)
KERNEL(k_bitcmp_cbr8, "s_mov_b32 s22, 0\n\ts_mov_b32 s23, 0\n\t", REP8("s_bitcmp1_b64 s[22:23], s21\n\ts_cbranch_scc1 1f\n\t1:\n\t"), 16)
KERNEL(k_pk16, "", REP4("v_pk_add_f32 v[20:21], v[20:21], v[24:25]\n\tv_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\tv_pk_mul_f32 v[26:27], v[26:27], v[24:25]\n\tv_pk_add_f32 v[32:33], v[32:33], v[24:25]\n\t"), 16)
KERNEL(k_rcp8, "", REP2("v_rcp_f32 v32, v28\n\tv_rcp_f32 v33, v29\n\tv_rcp_f32 v34, v30\n\tv_rcp_f32 v35, v31\n\t"), 8)
KERNEL(k_rcp4_v12, "", "v_rcp_f32 v32, v28\n\t" V4 "v_rcp_f32 v33, v29\n\t" V4 "v_rcp_f32 v34, v30\n\t" V4 "v_rcp_f32 v35, v31\n\t", 16)
// the IEEE division sequence, x = v30 (a), y = v28 (b): one row, then four rows interleaved by the assembler order
#define DIV1(x, y, d3, d4, d6, d7, d8) \
    "v_div_scale_f32 " d3 ", s[26:27], " y ", " y ", " x "\n\t" \
    "v_rcp_f32 " d4 ", " d3 "\n\t" \
    "v_div_scale_f32 " d6 ", vcc, " x ", " y ", " x "\n\t" \
    "v_fma_f32 " d7 ", -" d3 ", " d4 ", 1.0\n\t" \
    "v_fmac_f32 " d4 ", " d7 ", " d4 "\n\t" \
    "v_mul_f32 " d7 ", " d6 ", " d4 "\n\t" \
    "v_fma_f32 " d8 ", -" d3 ", " d7 ", " d6 "\n\t" \
    "v_fmac_f32 " d7 ", " d8 ", " d4 "\n\t" \
    "v_fma_f32 " d3 ", -" d3 ", " d7 ", " d6 "\n\t" \
    "v_div_fmas_f32 " d3 ", " d3 ", " d4 ", " d7 "\n\t" \
    "v_div_fixup_f32 " d8 ", " d3 ", " y ", " x "\n\t" \
    "v_cmp_neq_f32 vcc, 0, " y "\n\t" \
    "v_cndmask_b32 " d8 ", v27, " d8 ", vcc\n\t"
KERNEL(k_div1, "", DIV1("v30", "v28", "v32", "v33", "v34", "v35", "v36"), 13)
KERNEL(k_div4, "", DIV1("v30", "v28", "v32", "v33", "v34", "v35", "v36") DIV1("v31", "v29", "v37", "v38", "v39", "v40", "v41") DIV1("v30", "v29", "v32", "v33", "v34", "v35", "v42") DIV1("v31", "v28", "v37", "v38", "v39", "v40", "v43"), 52)
// fast division candidate: rcp + 2 NR on r, q, residual, correction (no scaling)
#define FDIV(x, y, r, e, q) \
    "v_rcp_f32 " r ", " y "\n\t" \
    "v_fma_f32 " e ", -" y ", " r ", 1.0\n\t" \
    "v_fmac_f32 " r ", " e ", " r "\n\t" \
    "v_mul_f32 " q ", " x ", " r "\n\t" \
    "v_fma_f32 " e ", -" y ", " q ", " x "\n\t" \
    "v_fmac_f32 " q ", " e ", " r "\n\t"
KERNEL(k_fdiv4, "", FDIV("v30", "v28", "v32", "v33", "v34") FDIV("v31", "v29", "v35", "v36", "v37") FDIV("v30", "v29", "v38", "v39", "v40") FDIV("v31", "v28", "v41", "v42", "v43"), 24)
KERNEL(k_readlane16, "", REP16("v_readlane_b32 s22, v32, s21\n\t"), 16)
KERNEL(k_readlane_v4, "", REP4("v_readlane_b32 s22, v32, s21\n\t" V4), 20)
KERNEL(k_rl_dep_salu, "", REP8("v_readlane_b32 s22, v32, s21\n\ts_add_u32 s23, s22, 1\n\t"), 16)
KERNEL(k_setpc4_v4, "", REP4(V4 "s_getpc_b64 s[22:23]\n\ts_add_u32 s22, s22, 1f-.\n\ts_addc_u32 s23, s23, 0\n\ts_setpc_b64 s[22:23]\n\t1:\n\t"), 32)
KERNEL(k_setpc4_v8, "", REP4(V4 V4 "s_getpc_b64 s[22:23]\n\ts_add_u32 s22, s22, 1f-.\n\ts_addc_u32 s23, s23, 0\n\ts_setpc_b64 s[22:23]\n\t1:\n\t"), 48)
KERNEL(k_idx_add4, "s_mov_b32 s21, 0\n\t", REP4("s_set_gpr_idx_on s21, gpr_idx(SRC0,SRC1,DST)\n\t" V4 "s_set_gpr_idx_off\n\t"), 24)
KERNEL(k_idx_pk4, "s_mov_b32 s21, 0\n\t", REP4("s_set_gpr_idx_on s21, gpr_idx(SRC0,SRC1,DST)\n\tv_pk_add_f32 v[20:21], v[20:21], v[24:25]\n\tv_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\ts_set_gpr_idx_off\n\t"), 16)
KERNEL(k_lds_b128, "v_mov_b32 v44, 0\n\t", REP4("ds_read_b128 v[32:35], v44\n\tds_read_b128 v[36:39], v44 offset:16\n\t") "s_waitcnt lgkmcnt(0)\n\t", 9)
KERNEL(k_lds_v, "v_mov_b32 v44, 0\n\t", REP4("ds_read_b128 v[32:35], v44\n\t" V4) "s_waitcnt lgkmcnt(0)\n\t", 21)
// a complete v2-style ADD handler at K=4 and K=8: fetch, window, adds, 4 not-taken dispatch tests, one taken branch
#define DISP "s_add_u32 s20, s20, 1\n\ts_bitcmp1_b64 s[24:25], s20\n\ts_cbranch_scc1 1f\n\ts_bitcmp1_b64 s[24:25], s20\n\ts_cbranch_scc1 1f\n\ts_branch 1f\n\t1:\n\t"
KERNEL(k_hadd_k4, "s_mov_b32 s21, 0\n\t", REP4("s_set_gpr_idx_on s21, gpr_idx(SRC0,SRC1,DST)\n\t" V4 "s_set_gpr_idx_off\n\ts_add_u32 s21, s21, 0\n\t" DISP), 52)
KERNEL(k_hadd_k8, "s_mov_b32 s21, 0\n\t", REP4("s_set_gpr_idx_on s21, gpr_idx(SRC0,SRC1,DST)\n\t" V4 V4 "s_set_gpr_idx_off\n\ts_add_u32 s21, s21, 0\n\t" DISP), 68)
KERNEL(k_hadd_k16, "s_mov_b32 s21, 0\n\t", REP4("s_set_gpr_idx_on s21, gpr_idx(SRC0,SRC1,DST)\n\t" V4 V4 V4 V4 "s_set_gpr_idx_off\n\ts_add_u32 s21, s21, 0\n\t" DISP), 100)

template <typename K>
static void run(const char *name, K kern, int ninstr, int W, unsigned long long *dout, float *dsink) {
    const int blocks = W > 0 ? 1024 * W : 1;
    std::vector<unsigned long long> h(blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, dout, dsink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, dout, dsink);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dout, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per = s / blocks / ITER;
    const int w = W > 0 ? W : 1;
    printf("%-14s W=%d  instrs %3d  ticks/pattern/wave %8.1f  per-SIMD %8.1f  ticks/instr/SIMD %6.2f  kernel %.3f ms\n", name, W, ninstr, per, per / w, per / w / ninstr, ms);
}

int main(int argc, char **argv) {
    unsigned long long *dout; float *dsink;
    hipMalloc(&dout, 65536 * sizeof(unsigned long long));
    hipMalloc(&dsink, 64 * sizeof(float));
    hipMemset(dsink, 0, 64 * sizeof(float));
    for (int W : {0, 1, 2, 4, 8}) {
        printf("---- W = %d waves per SIMD%s\n", W, W == 0 ? " (a single wave on the whole chip)" : "");
#define RUN(k) run(#k, k, k##_n, W, dout, dsink)
        RUN(k_valu16); RUN(k_salu16); RUN(k_mix_1v1s); RUN(k_mix_1v3s); RUN(k_mix_3v1s); RUN(k_mix_4v4s); RUN(k_mix_8v8s);
        RUN(k_branch16); RUN(k_v4_branch); RUN(k_v8_branch); RUN(k_cbr_nt16); RUN(k_bitcmp_cbr8);
        RUN(k_pk16); RUN(k_rcp8); RUN(k_rcp4_v12); RUN(k_div1); RUN(k_div4); RUN(k_fdiv4);
        RUN(k_readlane16); RUN(k_readlane_v4); RUN(k_rl_dep_salu); RUN(k_setpc4_v4); RUN(k_setpc4_v8);
        RUN(k_idx_add4); RUN(k_idx_pk4); RUN(k_lds_b128); RUN(k_lds_v);
        RUN(k_hadd_k4); RUN(k_hadd_k8); RUN(k_hadd_k16);
    }
    return 0;
}
