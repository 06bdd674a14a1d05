// How often do the shortened division sequences differ from the IEEE quotient?  2^32 random mantissa pairs (both operands in
// [1, 2), so that only the rounding of the quotient is exercised, not the range handling), three sequences:
//   fast  : rcp, q0 = x*r, e = fma(-y,q0,x), q = fma(e,r,q0)                          (no scaling, no fixup)
//   safe  : div_scale x2, rcp, q0, e, div_fmas(e,r,q0), div_fixup                      (range-safe, one correction)
//   ieee  : the compiler's x / y (div_scale x2, rcp, 2 fma refine, mul, 3 fma, div_fmas, div_fixup)
// build: hipcc --offload-arch=gfx950 -O3 -o div_faithful div_faithful.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ inline float div_fast(float x, float y) {
    float r = __builtin_amdgcn_rcpf(y);
    float q0 = x * r;
    float e = __builtin_fmaf(-y, q0, x);
    return __builtin_fmaf(e, r, q0);
}

__device__ inline float div_safe(float x, float y) {
    float d, n, q;
    asm volatile(
        "v_div_scale_f32 %0, vcc, %4, %4, %3\n\t"
        "v_rcp_f32 v200, %0\n\t"
        "v_div_scale_f32 %1, vcc, %3, %4, %3\n\t"
        "v_mul_f32 v201, %1, v200\n\t"
        "v_fma_f32 v202, -%0, v201, %1\n\t"
        "s_nop 1\n\t"
        "v_div_fmas_f32 %2, v202, v200, v201\n\t"
        "v_div_fixup_f32 %2, %2, %4, %3\n\t"
        : "=&v"(d), "=&v"(n), "=&v"(q) : "v"(x), "v"(y) : "vcc", "v200", "v201", "v202");
    return q;
}

__global__ void count(unsigned long long *out, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned bad_fast = 0, bad_safe = 0;
    for (uint32_t i = 0; i < 4096; ++i) {
        const uint32_t a = mix(tid * 4096u + i + seed), b = mix(a ^ 0x9e3779b9u);
        const float x = __uint_as_float(0x3f800000u | (a >> 9)), y = __uint_as_float(0x3f800000u | (b >> 9));
        const float ref = x / y;
        bad_fast += __float_as_uint(div_fast(x, y)) != __float_as_uint(ref);
        bad_safe += __float_as_uint(div_safe(x, y)) != __float_as_uint(ref);
    }
    atomicAdd(out, (unsigned long long)bad_fast);
    atomicAdd(out + 1, (unsigned long long)bad_safe);
}

int main() {
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, 16); hipMemset(d, 0, 16);
    const unsigned blocks = 4096, threads = 256;  // 2^20 threads x 2^12 pairs = 2^32
    hipLaunchKernelGGL(count, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("pairs 4294967296  fast!=ieee %llu  safe!=ieee %llu\n", h[0], h[1]);
    return 0;
}
