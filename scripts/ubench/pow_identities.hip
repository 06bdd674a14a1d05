// Which algebraic identities of pow does the DEVICE math library satisfy bit for bit?  The program compiler (csrc/sr_tc.hip,
// compile_general) replaces pow nodes with a constant exponent of 0, 1 or -1, or the constant base 1, by what they equal —
// but only identities that hold for the library's own powf on every one of the 2^32 operand bit patterns may be used: the
// register kernels call that function, and a tree must evaluate to the same bits whichever kernel takes it.
// NaN results count as equal whatever their payload (only the NaN class reaches a fitness value).
//   powf(x,  0) == 1      powf(x, 1) == x      powf(x, -1) == 1 / x  (IEEE division; the reference's b == 0 -> NaN rule does
//   powf(1,  y) == 1                                                  not apply: pow(0, -1) is +-inf, as is 1 / +-0)
// and the loose variant pow(|x|, c) with its (0, 0) -> 0 exception (forward.cu:195-200).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pow_identities pow_identities.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>

__device__ inline bool same(float a, float b) {
    return (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b);
}

__device__ inline unsigned ulps(float a, float b) {  // distance in units of the last place (same-sign finite values), else a large number
    if ((a != a && b != b) || __float_as_uint(a) == __float_as_uint(b)) return 0u;
    if (a != a || b != b || (__float_as_uint(a) ^ __float_as_uint(b)) >> 31) return 1u << 30;
    const uint32_t x = __float_as_uint(a) & 0x7FFFFFFFu, y = __float_as_uint(b) & 0x7FFFFFFFu;
    return x > y ? x - y : y - x;
}

__global__ void check(unsigned long long *bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned b0 = 0, b1 = 0, bm1 = 0, bone = 0, u1 = 0, um1 = 0, d1 = 0, dm1 = 0;
    for (uint32_t i = 0; i < 256; ++i) {
        const uint32_t bits = tid * 256u + i;
        const float x = __uint_as_float(bits);
        b0 += !same(powf(x, 0.0f), 1.0f);
        b1 += !same(powf(x, 1.0f), x);
        bm1 += !same(powf(x, -1.0f), 1.0f / x);
        bone += !same(powf(1.0f, x), 1.0f);
        const bool denorm = (bits & 0x7F800000u) == 0u && (bits & 0x007FFFFFu) != 0u;
        const unsigned e1 = ulps(powf(x, 1.0f), x), em1 = ulps(powf(x, -1.0f), 1.0f / x);
        if (denorm) { d1 += e1 != 0; dm1 += em1 != 0; }
        else { u1 = e1 > u1 ? e1 : u1; um1 = em1 > um1 ? em1 : um1; }
    }
    atomicMax((unsigned *)(bad + 4), u1); atomicMax((unsigned *)(bad + 5), um1);
    if (d1) atomicAdd(bad + 6, (unsigned long long)d1);
    if (dm1) atomicAdd(bad + 7, (unsigned long long)dm1);
    if (b0) atomicAdd(bad + 0, (unsigned long long)b0);
    if (b1) atomicAdd(bad + 1, (unsigned long long)b1);
    if (bm1) atomicAdd(bad + 2, (unsigned long long)bm1);
    if (bone) atomicAdd(bad + 3, (unsigned long long)bone);
}

int main() {
    unsigned long long *d, h[8];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    check<<<(1u << 24) / 256, 256>>>(d);   // 2^24 threads x 256 values = all 2^32 bit patterns
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mismatches over all 2^32 operands: pow(x,0)==1: %llu   pow(x,1)==x: %llu   pow(x,-1)==1/x: %llu   pow(1,y)==1: %llu\n", h[0], h[1], h[2], h[3]);
    printf("largest distance for a normal or special x (ulp; 2^30 = different class or sign): pow(x,1) vs x: %llu   pow(x,-1) vs 1/x: %llu;  denormal x that differ: %llu, %llu\n", h[4], h[5], h[6], h[7]);
    return 0;
}
