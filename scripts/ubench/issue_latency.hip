// Micro-benchmarks of the instruction patterns of the threaded-code interpreter (gfx950).
// Each kernel times ITER iterations of a REP-times repeated pattern with s_memtime; one wave per block.
// Build: hipcc --offload-arch=gfx950 -O3 issue_latency.hip -o issue_latency ; run: ./issue_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2000
#define STR2(x) #x
#define STR(x) STR2(x)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(name, setup, body16, nper)                                                     \
    __global__ void name(unsigned long long *out, float *sink) {                              \
        float f = sink[threadIdx.x];                                                          \
        unsigned long long t0, t1;                                                            \
        asm volatile(setup ::: "s20", "s21", "s22", "s23", "s24", "s25", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                      \
        for (int i = 0; i < ITER; ++i) {                                                      \
            asm volatile(body16 ::: "memory", "scc", "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"); \
        }                                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                      \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                      \
        sink[threadIdx.x] = f;                                                                \
    }                                                                                         \
    static const int name##_n = nper;

// A: dependent SALU chain
KERNEL(k_salu_dep, "s_mov_b32 s20, 0\n\t", REP16("s_add_u32 s20, s20, 1\n\t"), 16)
// B: dependent VALU chain
KERNEL(k_valu_dep, "v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.0\n\t", REP16("v_add_f32 v20, v20, v21\n\t"), 16)
// B2: independent VALU (4 chains)
KERNEL(k_valu_ind, "v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v22, 1.0\n\tv_mov_b32 v23, 1.0\n\tv_mov_b32 v24, 1.0\n\t",
       REP4("v_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v24\n\tv_add_f32 v22, v22, v24\n\tv_add_f32 v23, v23, v24\n\t"), 16)
// C: readlane -> salu -> readlane (lane select) round trip
KERNEL(k_readlane_rt, "s_mov_b32 s21, 0\n\tv_mov_b32 v20, 0\n\t", REP16("v_readlane_b32 s20, v20, s21\n\ts_add_u32 s21, s20, 0\n\t"), 16)
// C2: readlane with independent consumer
KERNEL(k_readlane_ind, "s_mov_b32 s21, 0\n\tv_mov_b32 v20, 0\n\t", REP16("v_readlane_b32 s20, v20, s21\n\t"), 16)
// D: gpr-index window around one v_mov
KERNEL(k_gpridx1, "s_mov_b32 s21, 1\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\t", REP16("s_set_gpr_idx_on s21, gpr_idx(SRC0)\n\tv_mov_b32 v20, v21\n\ts_set_gpr_idx_off\n\t"), 16)
// D4: gpr-index window around four v_mov
KERNEL(k_gpridx4, "s_mov_b32 s21, 1\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\t",
       REP16("s_set_gpr_idx_on s21, gpr_idx(SRC0)\n\tv_mov_b32 v20, v30\n\tv_mov_b32 v21, v31\n\tv_mov_b32 v22, v32\n\tv_mov_b32 v23, v33\n\ts_set_gpr_idx_off\n\t"), 16)
// D0: idx on/off only
KERNEL(k_gpridx0, "s_mov_b32 s21, 1\n\t", REP16("s_set_gpr_idx_on s21, gpr_idx(SRC0)\n\ts_set_gpr_idx_off\n\t"), 16)
// Dd: indexed source used directly by v_add (one window, four adds)
KERNEL(k_gpridx_add, "s_mov_b32 s21, 1\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t",
       REP16("s_set_gpr_idx_on s21, gpr_idx(SRC1)\n\tv_add_f32 v20, v20, v30\n\tv_add_f32 v21, v21, v31\n\tv_add_f32 v22, v22, v32\n\tv_add_f32 v23, v23, v33\n\ts_set_gpr_idx_off\n\t"), 16)
// E: computed jump to the next instruction
KERNEL(k_setpc_near, "", REP16("s_getpc_b64 s[22:23]\n\ts_add_u32 s22, s22, 1f-.\n\ts_addc_u32 s23, s23, 0\n\ts_setpc_b64 s[22:23]\n\t1:\n\t"), 16)
// E2: computed jump 512 bytes ahead (same as the handler slots); 4 per body
KERNEL(k_setpc_far, "", REP4("s_getpc_b64 s[22:23]\n\ts_add_u32 s22, s22, 1f-.\n\ts_addc_u32 s23, s23, 0\n\ts_setpc_b64 s[22:23]\n\t.p2align 9\n\t1:\n\t"), 4)
// F: taken short branch
KERNEL(k_branch, "", REP16("s_branch 1f\n\t1:\n\t"), 16)
// F2: taken branch over a 256-byte gap
KERNEL(k_branch_far, "", REP4("s_branch 1f\n\t.p2align 8\n\t1:\n\t"), 4)
// G: the ADD handler as generated (pop 4, add 4) with its fetch and a near computed jump
KERNEL(k_handler_add, "s_mov_b32 s20, 0\n\ts_mov_b32 s21, 4\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\t",
       REP4("s_add_u32 s20, s20, 0\n\tv_readlane_b32 s24, v36, s20\n\ts_sub_u32 s21, s21, 0\n\ts_set_gpr_idx_on s21, gpr_idx(SRC0)\n\tv_mov_b32 v24, v30\n\tv_mov_b32 v25, v31\n\tv_mov_b32 v26, v32\n\tv_mov_b32 v27, v33\n\ts_set_gpr_idx_off\n\t"
            "v_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v25\n\tv_add_f32 v22, v22, v26\n\tv_add_f32 v23, v23, v27\n\ts_getpc_b64 s[22:23]\n\ts_add_u32 s22, s22, 1f-.\n\ts_add_u32 s22, s22, s24\n\ts_addc_u32 s23, s23, 0\n\ts_setpc_b64 s[22:23]\n\t.p2align 9\n\t1:\n\t"), 4)

template <typename K>
static void run(const char *name, K kern, int nper, int blocks, unsigned long long *dout, float *dsink) {
    std::vector<unsigned long long> h(blocks);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, dout, dsink);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, dout, dsink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dout, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-16s blocks %5d  cycles per pattern %8.2f\n", name, blocks, s / blocks / ITER / nper);
}

int main() {
    unsigned long long *dout; float *dsink;
    hipMalloc(&dout, 65536 * sizeof(unsigned long long));
    hipMalloc(&dsink, 64 * sizeof(float));
    hipMemset(dsink, 0, 64 * sizeof(float));
    for (int blocks : {1, 256 * 4, 256 * 4 * 3, 256 * 4 * 6}) {
        printf("---- %d waves (%d per SIMD)\n", blocks, blocks >= 1024 ? blocks / 1024 : 0);
#define RUN(k) run(#k, k, k##_n, blocks, dout, dsink)
        RUN(k_salu_dep); RUN(k_valu_dep); RUN(k_valu_ind); RUN(k_readlane_rt); RUN(k_readlane_ind);
        RUN(k_gpridx0); RUN(k_gpridx1); RUN(k_gpridx4); RUN(k_gpridx_add);
        RUN(k_setpc_near); RUN(k_setpc_far); RUN(k_branch); RUN(k_branch_far); RUN(k_handler_add);
    }
    return 0;
}
