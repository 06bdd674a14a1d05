// Which operand positions does VGPR indexing (M0[15:12] = {SRC0, SRC1, SRC2, DST}_REL, M0[7:0] = index) move for the
// instruction kinds the in-place division handlers of gen_tc_asm.py use?  v20.. = 1, 2, 3, ...; index 2, all four bits on.
// build: hipcc --offload-arch=gfx950 -O1 -o gpr_index_modes gpr_index_modes.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(float *out) {
    float r0, r1, r2, r3, r4, r5;
    asm volatile(
        "v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 2.0\n\tv_mov_b32 v22, 3.0\n\tv_mov_b32 v23, 4.0\n\tv_mov_b32 v24, 5.0\n\t"
        "v_mov_b32 v25, 6.0\n\tv_mov_b32 v26, 7.0\n\tv_mov_b32 v27, 8.0\n\t"
        "v_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\t"
        "s_mov_b32 s20, 2\n\t"
        "s_set_gpr_idx_on s20, 0xf\n\t"
        "v_fma_f32 v30, v20, v21, v22\n\t"            // all relative: v32 = v22 * v23 + v24 = 3 * 4 + 5 = 17
        "v_div_scale_f32 v31, vcc, v20, v20, v21\n\t"   // all relative: v33 = scale of v22 = 3 (no scaling)
        "v_div_fixup_f32 v32, v20, v21, v22\n\t"        // v34 = fixup(q = v22 = 3, y = v23 = 4, x = v24 = 5) = 3
        "v_min3_f32 v33, v20, v21, v22\n\t"             // v35 = min(3, 4, 5) = 3
        "v_mul_f32 v34, v20, v21\n\t"                   // v36 = 3 * 4 = 12
        "s_set_gpr_idx_off\n\t"
        "v_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v34\n\tv_mov_b32 %3, v35\n\tv_mov_b32 %4, v36\n\tv_mov_b32 %5, v30\n\t"
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5)
        :: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "s20", "vcc");
    if (threadIdx.x == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; out[4] = r4; out[5] = r5; }
}

int main() {
    float *d, h[6];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("fma -> v32 = %g (17 if src0/1/2/dst all moved; 15 if src2 did not)   div_scale -> v33 = %g (3)   div_fixup -> v34 = %g (3)   min3 -> v35 = %g (3)   mul -> v36 = %g (12)   v30 = %g (0)\n",
           h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
