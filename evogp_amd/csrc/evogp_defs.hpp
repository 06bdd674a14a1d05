// evogp_defs.hpp — encodings, RNG and wave helpers shared by every gfx950 kernel.
//
// The numeric encodings are the wire format of the Forest tensors and therefore identical to the
// reference's (src/evogp/cuda/defs.h:5-57); everything else here is original CDNA4 code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evogp {

constexpr int kMaxStack = 1024;    // defs.h:5  upper bound on gp_len
constexpr int kMaxFullDepth = 10;  // defs.h:5  entries of depth2leaf_probs
constexpr int kNumFuncs = 29;      // defs.h:56 Function::END
constexpr float kDelta = 1e-9f;    // defs.h:7
constexpr float kMaxVal = 1e9f;    // defs.h:8
constexpr int kWave = 64;          // CDNA wavefront

enum NodeType : int { T_VAR = 0, T_CONST = 1, T_UFUNC = 2, T_BFUNC = 3, T_TFUNC = 4, T_MASK = 0x7F, T_OUT = 0x80 };

enum Func : int {
    F_IF = 0,
    F_ADD = 1, F_SUB, F_MUL, F_DIV, F_LOOSE_DIV, F_POW, F_LOOSE_POW, F_MAX, F_MIN, F_LT, F_GT, F_LE, F_GE,
    F_SIN = 14, F_COS, F_TAN, F_SINH, F_COSH, F_TANH, F_LOG, F_LOOSE_LOG, F_EXP, F_INV, F_LOOSE_INV, F_NEG, F_ABS,
    F_SQRT, F_LOOSE_SQRT,
    F_END = 29
};

// ---- per-tree RNG ---------------------------------------------------------------------------
// Seed hash: FNV-1a-64 over the 12 little-endian bytes of {n, k1, k2}, truncated to 32 bits
// (kernel.h:157-172).
__host__ __device__ inline uint32_t seed_hash(uint32_t n, uint32_t k1, uint32_t k2) {
    const uint32_t w[3] = {n, k1, k2};
    uint64_t h = 14695981039346656037ULL;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            h ^= (uint64_t)((w[i] >> (8 * b)) & 0xFFu);
            h *= 1099511628211ULL;
        }
    }
    return (uint32_t)h;
}

// L'Ecuyer's three-component Tausworthe generator ("taus88") as Thrust parameterises it
// (the reference's RandomEngine, kernel.h:20): all three LFSRs start from the raw seed.
struct Taus88 {
    uint32_t a, b, c;
    __host__ __device__ explicit Taus88(uint32_t s) : a(s), b(s), c(s) {}
    __host__ __device__ static inline uint32_t step(uint32_t z, int k, int q, int s) {
        const uint32_t t = ((z << q) ^ z) >> (k - s);
        return ((z & (0xFFFFFFFFu << (32 - k))) << s) ^ t;
    }
    __host__ __device__ inline uint32_t next() {
        a = step(a, 31, 13, 12);
        b = step(b, 29, 2, 4);
        c = step(c, 28, 3, 17);
        return a ^ b ^ c;
    }
    // thrust::uniform_real_distribution<float>(0,1): float(u32) * 2^-32, u32->f32 rounds to nearest
    // (so 1.0f is reachable).
    __host__ __device__ inline float uniform() { return (float)next() * 2.3283064365386963e-10f; }
};

// ---- counter-based random words (no counterpart in the reference, which draws with torch's generator) -----------------------------
// word k of item i of generation g under `seed` = the splitmix64 finaliser of (mix(seed * 1000003 + g) + (k << 40) + i), reduced to
// [0, 2^31 - 1) like torch.randint(0, 2^31 - 1): evogp_amd/parallel.py random_words is the same arithmetic in torch ops.  Rows in
// use: 0-5 the six words of offspring i (breed.hip), 7 the two generation keys (i = 0, 1), 16 + k contender k of tournament i.
__host__ __device__ inline unsigned long long mix64(unsigned long long x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline unsigned long long counter_base(long long seed, long long generation) {
    return mix64((unsigned long long)(seed * 1000003ll + generation));
}
__host__ __device__ inline unsigned counter_word(unsigned long long base, unsigned k, unsigned long long i) {
    const unsigned long long x = mix64(base + ((unsigned long long)k << 40) + i);
    return (unsigned)(((x >> 33) & 0x7FFFFFFFull) % 0x7FFFFFFFull);
}

__device__ inline float bits2f(uint32_t u) { return __uint_as_float(u); }
__device__ inline uint32_t f2bits(float f) { return __float_as_uint(f); }

// broadcast a wave-uniform value into an SGPR so the compiler keeps control flow scalar
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ---- wave-level scans and reductions on the DPP cross-lane path -------------------------------------
// `__shfl_*` lowers to ds_bpermute (an LDS-crossbar round trip of ~100 cycles per step); the DPP modifiers
// move data between lanes inside the VALU.  Inclusive scan over the 64 lanes: row_shr 1,2,4,8 scan
// each row of 16 lanes, row_bcast:15 and row_bcast:31 carry the row totals upward (the gfx9 sequence
// the AMDGPU atomic optimizer uses).  The last lane holds the reduction.
constexpr int kDppRowShr = 0x110, kDppBcast15 = 0x142, kDppBcast31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ inline int dpp_move(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ inline int wave_scan_incl(int v) {
    v += dpp_move<kDppRowShr | 1, 0xf>(0, v);
    v += dpp_move<kDppRowShr | 2, 0xf>(0, v);
    v += dpp_move<kDppRowShr | 4, 0xf>(0, v);
    v += dpp_move<kDppRowShr | 8, 0xf>(0, v);
    v += dpp_move<kDppBcast15, 0xa>(0, v);
    v += dpp_move<kDppBcast31, 0xc>(0, v);
    return v;
}

// wave-uniform reductions: the value of lane 63 of the scan, broadcast through an SGPR
__device__ inline int wave_max(int v) {
    const int lo = (int)0x80000000;
    v = max(v, dpp_move<kDppRowShr | 1, 0xf>(lo, v));
    v = max(v, dpp_move<kDppRowShr | 2, 0xf>(lo, v));
    v = max(v, dpp_move<kDppRowShr | 4, 0xf>(lo, v));
    v = max(v, dpp_move<kDppRowShr | 8, 0xf>(lo, v));
    v = max(v, dpp_move<kDppBcast15, 0xa>(lo, v));
    v = max(v, dpp_move<kDppBcast31, 0xc>(lo, v));
    return __builtin_amdgcn_readlane(v, 63);
}

// fixed-order sum of the 64 lanes (deterministic: the same association every run)
__device__ inline float wave_sum(float x) {
    int z = 0;
    float v = x;
    v += __int_as_float(dpp_move<kDppRowShr | 1, 0xf>(z, __float_as_int(v)));
    v += __int_as_float(dpp_move<kDppRowShr | 2, 0xf>(z, __float_as_int(v)));
    v += __int_as_float(dpp_move<kDppRowShr | 4, 0xf>(z, __float_as_int(v)));
    v += __int_as_float(dpp_move<kDppRowShr | 8, 0xf>(z, __float_as_int(v)));
    v += __int_as_float(dpp_move<kDppBcast15, 0xa>(z, __float_as_int(v)));
    v += __int_as_float(dpp_move<kDppBcast31, 0xc>(z, __float_as_int(v)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

} // namespace evogp
