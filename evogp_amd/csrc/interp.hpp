// interp.hpp — the wave-uniform stack-machine interpreter (gfx950).
//
// Semantics restate src/evogp/cuda/forward.cu:79-244 (_process_node) and :246-302
// (_treeGPEvalByStack); the execution model is new:
//
//   * one 64-lane wave interprets ONE tree; every lane works on K different input rows, so the
//     opcode, the operand-stack height and every branch are wave-uniform: decode runs on the scalar
//     unit (SGPR compares + s_cbranch) and never diverges, and its cost is shared by 64*K rows;
//   * the tree is loaded coalesced, one node per lane, pre-decoded into an {opcode, payload}
//     pair that stays in two VGPRs; the interpreter fetches instruction j with two v_readlane
//     — no memory access in the inner loop;
//   * the operand stack lives in VGPRs: K ext_vectors of DEPTH floats indexed with the uniform
//     stack height, which the gfx9 back end lowers to s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off;
//     the top of stack is cached in its own registers; the variables of the lane's input rows
//     live in VGPRs as well (another indexed vector per row).
//
// Trees are scanned in reverse prefix order (leaves push, functions pop), exactly as the reference
// does after reversing the arrays (forward.cu:281-296).
#pragma once
#include "evogp_defs.hpp"

#include <type_traits>

namespace evogp {

typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v12f __attribute__((ext_vector_type(12)));
typedef float v10f __attribute__((ext_vector_type(10)));
typedef float v9f __attribute__((ext_vector_type(9)));
// Register-tuple widths with a uniform dynamic index: 9..12, 16 and 32 dwords are lowered to
// s_set_gpr_idx_on + v_mov (vectors of <= 8 elements are expanded into compare/select chains by the
// back end, which is slower), so 9 is the smallest width used.
template <int N> struct VecOf;
template <> struct VecOf<9> { using type = v9f; };
template <> struct VecOf<10> { using type = v10f; };
template <> struct VecOf<12> { using type = v12f; };
template <> struct VecOf<16> { using type = v16f; };
template <> struct VecOf<32> { using type = v32f; };

constexpr int kMaxOutRegs = 16; // multi-output accumulators kept in registers

// ---- the Classification problem's prediction rule (src/evogp/problem/classification.py:62-67) --------------------------------
//     pred = torch.argmax(torch.clip(torch.softmax(outputs, dim), 1e-15, 1 - 1e-15))
// The arg-max of the soft-max is the arg-max of the raw outputs EXCEPT where two outputs are so close that their fp32 soft-max
// values round to the same float: exp(x_i - max) is 1 or 1 - 2^-24 only for max - x_i below ~1e-7, and a quotient by the same
// sum can then equal that of the maximum; torch.argmax returns the FIRST of equal values, i.e. possibly an index in front of
// the true maximum.  A row is AMBIGUOUS when an output in front of the first maximum lies within kSoftmaxTieMargin below it (1.25 x 2^-23:
// for d = max - x_i above it the true exp(-d) lies below 1 - 2^-23, so a faithfully rounded exp returns at most 1 - 2^-23, whose
// quotient by the sum differs from that of 1 by at least one unit in the last place: the two cannot round to the same float); the fast kernels take the raw arg-max and send trees with an ambiguous row to the recount kernel,
// which evaluates torch's arithmetic itself:
//   * aten's softmax_warp_forward (the kernel a row of <= 1024 fp32 elements gets): max over the row; e_i = std::exp(x_i - max)
//     (the device library's expf); the sum by an xor-butterfly over next_pow2(n) lanes holding one element each (lanes past
//     the row hold exp(-inf) = 0) -- every lane computes the same pairwise tree, halving the width level by level; e_i / sum
//     with the correctly rounded division;
//   * clip to [1e-15f, 1.0f] (1 - 1e-15 rounds to 1 in fp32); arg-max = first index of the largest value.
// A NaN output or an infinite maximum makes the whole soft-max row NaN (inf - inf), whose arg-max is index 0.
constexpr float kSoftmaxTieMargin = 1.4901161193847656e-07f;  // 1.25 * 2^-23 = 0x34200000

__device__ inline int argmax_raw(const float *x, int n, bool *ambiguous) {
    int best = 0;
    float m = x[0];
    bool poisoned = m != m;
    for (int o = 1; o < n; ++o) {
        const float v = x[o];
        poisoned |= v != v;
        if (v > m) { m = v; best = o; }
    }
    if (poisoned || __builtin_isinf(m)) { *ambiguous = false; return 0; }
    bool amb = false;
    const float thr = m - kSoftmaxTieMargin;
    for (int o = 0; o < best; ++o) amb |= x[o] >= thr;   // (x[o] < m for every o in front of the first maximum)
    *ambiguous = amb;
    return best;
}

__device__ inline int argmax_as_torch(const float *x, int n) {
    float m = x[0];
    bool poisoned = m != m;
    for (int o = 1; o < n; ++o) {
        poisoned |= x[o] != x[o];
        m = m < x[o] ? x[o] : m;
    }
    if (poisoned || __builtin_isinf(m)) return 0;
    float e[kMaxOutRegs];
    int width = 1;
    while (width < n) width <<= 1;
    for (int o = 0; o < kMaxOutRegs; ++o) e[o] = o < n ? expf(x[o] - m) : 0.0f;
    float part[kMaxOutRegs];
    for (int o = 0; o < kMaxOutRegs; ++o) part[o] = e[o];
    for (int off = width >> 1; off >= 1; off >>= 1)
        for (int o = 0; o < off; ++o) part[o] = part[o] + part[o + off];
    const float sum = part[0];
    int best = 0;
    float top = -1.0f;
    for (int o = 0; o < n; ++o) {
        float s = e[o] / sum;
        s = s < 1e-15f ? 1e-15f : (s > 1.0f ? 1.0f : s);
        if (s > top) { top = s; best = o; }
    }
    return best;
}

// Handler ids produced by the pre-decoder.  The six ids every default SR function set uses come
// first so the scalar dispatch reaches them in three compares.
enum : uint32_t {
    H_CONST = 0, H_VAR = 1, H_ADD = 2, H_SUB = 3, H_MUL = 4, H_DIV = 5,
    H_BIN_OTHER = 6,  // + (f - F_LOOSE_DIV): LOOSE_DIV .. GE  -> 6..14
    H_BIN_ZERO = 15,  // binary node with an id that is not a binary function: yields 0
    H_UN = 16,        // + (f - F_SIN): SIN .. LOOSE_SQRT -> 16..30
    H_UN_ZERO = 31,   // unary node with an unknown id: yields 0
    H_IF = 32         // ternary (any id)
};
constexpr uint32_t kNoOut = 0xFFFFFFFFu;

struct Decoded {
    uint32_t op;   // handler id
    uint32_t pay;  // CONST: value bits; VAR: variable index; function: output index or kNoOut
    int delta;     // change of stack height: 1 - arity
    bool heavy;    // transcendental / pow: only the FULL interpreter build carries these handlers
};

// Functions with long library expansions (range reduction, pow).  The LEAN interpreter build leaves
// them out so that its register budget and code size stay small; trees using them are routed to the
// FULL build.
__device__ inline bool is_heavy_func(uint32_t f) {
    return (f >= (uint32_t)F_SIN && f <= (uint32_t)F_EXP) || f == (uint32_t)F_POW || f == (uint32_t)F_LOOSE_POW;
}

// Pre-decode one node (forward.cu:86-115 for the type/value unpacking).  `multi` selects the
// multi-output conventions (type masked, OUT flag honoured); in single-output mode the raw type
// is compared, so any type outside 0..3 runs the ternary path as in the reference (:213-224).
__device__ inline Decoded decode_node(int type, float value, bool multi, int var_len, int out_len) {
    Decoded d;
    bool is_out = false;
    if (multi) { is_out = (type & T_OUT) != 0; type &= T_MASK; }
    d.pay = kNoOut;
    d.heavy = false;
    if (type == T_CONST) { d.op = H_CONST; d.pay = f2bits(value); d.delta = 1; return d; }
    if (type == T_VAR) {
        int v = (int)value;
        v = v < 0 ? 0 : (v >= var_len ? var_len - 1 : v); // the reference does not range-check (forward.cu:100-101)
        d.op = H_VAR; d.pay = (uint32_t)v; d.delta = 1; return d;
    }
    uint32_t f = (uint32_t)value;
    if (multi && is_out) {
        const uint32_t bits = f2bits(value);
        f = (uint32_t)(int)(int16_t)(bits & 0xFFFFu);
        const uint32_t oi = (uint32_t)(int)(int16_t)(bits >> 16);
        if (oi < (uint32_t)out_len) d.pay = oi;
    }
    if (type == T_UFUNC) {
        d.op = (f >= (uint32_t)F_SIN && f <= (uint32_t)F_LOOSE_SQRT) ? H_UN + (f - F_SIN) : H_UN_ZERO;
        d.heavy = d.op != H_UN_ZERO && is_heavy_func(f);
        d.delta = 0;
    } else if (type == T_BFUNC) {
        if (f >= (uint32_t)F_ADD && f <= (uint32_t)F_DIV) d.op = H_ADD + (f - F_ADD);
        else if (f >= (uint32_t)F_LOOSE_DIV && f <= (uint32_t)F_GE) d.op = H_BIN_OTHER + (f - F_LOOSE_DIV);
        else d.op = H_BIN_ZERO;
        d.heavy = d.op != H_BIN_ZERO && is_heavy_func(f);
        d.delta = -1;
    } else {
        d.op = H_IF; d.delta = -2;
    }
    return d;
}

// ---- node arithmetic (forward.cu:125-224) ------------------------------------------------------
// LEAN builds omit the heavy functions (their trees never reach a LEAN interpreter).
template <bool LEAN>
__device__ inline float op_unary(uint32_t h, float a) {
    switch (h) {
    case H_UN + (F_INV - F_SIN): return a == 0.0f ? __builtin_nanf("") : 1.0f / a;
    case H_UN + (F_LOOSE_INV - F_SIN): { const float d = fabsf(a) <= kDelta ? copysignf(kDelta, a) : a; return 1.0f / d; }
    case H_UN + (F_NEG - F_SIN): return -a;
    case H_UN + (F_ABS - F_SIN): return fabsf(a);
    case H_UN + (F_SQRT - F_SIN): return sqrtf(a);
    case H_UN + (F_LOOSE_SQRT - F_SIN): return sqrtf(fabsf(a));
    default: break;
    }
    if (!LEAN) {
        switch (h) {
        case H_UN + (F_SIN - F_SIN): return sinf(a);
        case H_UN + (F_COS - F_SIN): return cosf(a);
        case H_UN + (F_TAN - F_SIN): return tanf(a);
        case H_UN + (F_SINH - F_SIN): return sinhf(a);
        case H_UN + (F_COSH - F_SIN): return coshf(a);
        case H_UN + (F_TANH - F_SIN): return tanhf(a);
        case H_UN + (F_LOG - F_SIN): return logf(a);
        case H_UN + (F_LOOSE_LOG - F_SIN): return a == 0.0f ? -kMaxVal : logf(fabsf(a));
        case H_UN + (F_EXP - F_SIN): return expf(a);
        default: break;
        }
    }
    return 0.0f; // unknown ids leave the zero-initialised result (forward.cu:117)
}

template <bool LEAN>
__device__ inline float op_binary_other(uint32_t h, float a, float b) {
    switch (h) {
    case H_BIN_OTHER + (F_LOOSE_DIV - F_LOOSE_DIV): { const float d = fabsf(b) <= kDelta ? copysignf(kDelta, b) : b; return a / d; }
    case H_BIN_OTHER + (F_MAX - F_LOOSE_DIV): return a >= b ? a : b;
    case H_BIN_OTHER + (F_MIN - F_LOOSE_DIV): return a <= b ? a : b;
    case H_BIN_OTHER + (F_LT - F_LOOSE_DIV): return a < b ? 1.0f : -1.0f;
    case H_BIN_OTHER + (F_GT - F_LOOSE_DIV): return a > b ? 1.0f : -1.0f;
    case H_BIN_OTHER + (F_LE - F_LOOSE_DIV): return a <= b ? 1.0f : -1.0f;
    case H_BIN_OTHER + (F_GE - F_LOOSE_DIV): return a >= b ? 1.0f : -1.0f;
    default: break;
    }
    if (!LEAN) {
        if (h == H_BIN_OTHER + (F_POW - F_LOOSE_DIV)) return powf(a, b);
        if (h == H_BIN_OTHER + (F_LOOSE_POW - F_LOOSE_DIV)) return (a == 0.0f && b == 0.0f) ? 0.0f : powf(fabsf(a), b);
    }
    return 0.0f;
}

// Operand stack in registers for K input rows per lane.  Height h counts the elements; the top
// element is tos[k], element e (e < h-1) is s[k][e+1]; pushing at h == 0 parks the (meaningless)
// tos in s[k][0].  DEPTH is the largest height supported.
template <int K, int DEPTH>
struct RegStack {
    typename VecOf<DEPTH>::type s[K];
    float tos[K];
    int h; // wave-uniform
};

// Execute `n` pre-decoded instructions held one per lane in (opv, payv), lanes 0..n-1 in
// execution order.  vars[k] holds input row k of this lane; outs[k] the multi-output accumulators.
// The opcode of the NEXT instruction is fetched (v_readlane) before the current one is dispatched,
// so the VALU->SGPR latency of the fetch overlaps the handler; the payload is only fetched by the
// handlers that use it.
// Where a lane's input rows live: RegVars = K register tuples indexed with the (uniform) variable number,
// LdsVars = the wave's [variable][lane] block in LDS (wide inputs: more variables than a register tuple holds).
template <int VL, int K>
struct RegVars {
    const typename VecOf<VL>::type (&v)[K];
    __device__ inline float get(int k, uint32_t var) const { return v[k][var]; }
};
struct LdsVars {
    const float *base;  // the workgroup's [variable][row] block, already offset to this lane's first row
    int stride;         // rows per variable; a lane's K rows are consecutive
    __device__ inline float get(int k, uint32_t var) const { return base[var * stride + k]; }
};

template <bool MO, bool LEAN, int K, int DEPTH, class VA>
__device__ inline void run_chunk(uint32_t opv, uint32_t payv, int n, RegStack<K, DEPTH> &st, const VA &vars, v16f (&outs)[K]) {
    uint32_t next_op = (uint32_t)__builtin_amdgcn_readlane((int)opv, 0);
    for (int j = 0; j < n; ++j) {
        const uint32_t op = next_op;
        next_op = (uint32_t)__builtin_amdgcn_readlane((int)opv, (j + 1) & 63);
        if (op < H_ADD) { // leaf: push
            const uint32_t pay = (uint32_t)__builtin_amdgcn_readlane((int)payv, j);
            const int h = st.h;
            st.h = h + 1;
#pragma unroll
            for (int k = 0; k < K; ++k) st.s[k][h] = st.tos[k];
            if (op == H_CONST) {
#pragma unroll
                for (int k = 0; k < K; ++k) st.tos[k] = bits2f(pay);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) st.tos[k] = vars.get(k, pay);
            }
        } else if (op < H_UN) { // binary: a = top (left operand), b = next (right operand)
            const int h = st.h - 1;
            st.h = h;
            float b[K];
#pragma unroll
            for (int k = 0; k < K; ++k) b[k] = st.s[k][h];
            if (MO) {
                const uint32_t pay = (uint32_t)__builtin_amdgcn_readlane((int)payv, j);
                if (pay != kNoOut) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float a = st.tos[k];
                        const float r = op == H_ADD ? a + b[k] : op == H_SUB ? a - b[k] : op == H_MUL ? a * b[k]
                                        : op == H_DIV ? (b[k] == 0.0f ? __builtin_nanf("") : a / b[k])
                                                      : op_binary_other<LEAN>(op, a, b[k]);
                        outs[k][pay] += r;
                    }
                }
                // every function node hands its LAST popped operand to its parent (forward.cu:237-243)
#pragma unroll
                for (int k = 0; k < K; ++k) st.tos[k] = b[k];
            } else if (op < H_MUL) {
                if (op == H_ADD) {
#pragma unroll
                    for (int k = 0; k < K; ++k) st.tos[k] = st.tos[k] + b[k];
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) st.tos[k] = st.tos[k] - b[k];
                }
            } else if (op < H_BIN_OTHER) {
                if (op == H_MUL) {
#pragma unroll
                    for (int k = 0; k < K; ++k) st.tos[k] = st.tos[k] * b[k];
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) st.tos[k] = b[k] == 0.0f ? __builtin_nanf("") : st.tos[k] / b[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) st.tos[k] = op_binary_other<LEAN>(op, st.tos[k], b[k]);
            }
        } else if (op < H_IF) { // unary
            if (MO) {
                const uint32_t pay = (uint32_t)__builtin_amdgcn_readlane((int)payv, j);
                if (pay != kNoOut) {
#pragma unroll
                    for (int k = 0; k < K; ++k) outs[k][pay] += op_unary<LEAN>(op, st.tos[k]);
                }
                // tos stays: the operand itself is handed to the parent
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) st.tos[k] = op_unary<LEAN>(op, st.tos[k]);
            }
        } else { // ternary IF: cond = top, then = next, else = third
            const int h = st.h - 2;
            st.h = h;
            uint32_t pay = kNoOut;
            if (MO) pay = (uint32_t)__builtin_amdgcn_readlane((int)payv, j);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float b = st.s[k][h + 1], c = st.s[k][h];
                const float r = st.tos[k] > 0.0f ? b : c;
                if (MO) {
                    if (pay != kNoOut) outs[k][pay] += r;
                    st.tos[k] = c;
                } else {
                    st.tos[k] = r;
                }
            }
        }
    }
}

// Result of the structural pre-pass over one tree.
enum TreeClass : int { TREE_OK = 0, TREE_DEEP = 1, TREE_BAD = 2, TREE_HEAVY = 3, TREE_SKIP = 4, TREE_GENERAL = 5 };

// Classify a tree: walk it in chunks of 64 nodes in execution (reverse prefix) order, prefix-sum
// the stack-height deltas, and check 1 <= height everywhere, final height == 1, and the maximum
// height against the register stack.  type/value point at the tree's row; len is already clamped.
// Returns TREE_BAD (malformed), TREE_HEAVY (uses a function this interpreter build lacks:
// lean = 1 -> transcendental/pow, lean = 2 -> anything but leaves and + - * /), TREE_DEEP (operand
// stack > max_height) or TREE_OK.
__device__ inline int classify_tree(const int16_t *__restrict__ type, const float *__restrict__ value, int len,
                                    bool multi, int var_len, int out_len, int max_height, int lean = 0) {
    if (len <= 0) return TREE_BAD;
    const int lane = threadIdx.x & 63;
    int carry = 0, hmax = 0, hmin = 1;
    bool heavy = false;
    for (int base = 0; base < len; base += kWave) {
        const int r = base + lane;
        int delta = 0;
        if (r < len) {
            const int i = len - 1 - r;
            const Decoded d = decode_node(type[i], value[i], multi, var_len, out_len);
            delta = d.delta;
            heavy |= lean == 2 ? d.op > H_DIV : d.heavy;
        }
        const int hh = carry + wave_scan_incl(delta);
        const int hv = r < len ? hh : 1;
        hmax = max(hmax, hv);
        hmin = min(hmin, hv);
        carry = __builtin_amdgcn_readlane(hh, 63);
    }
    hmax = wave_max(hmax);
    hmin = -wave_max(-hmin);
    if (hmin < 1 || carry != 1) return TREE_BAD;
    if (lean != 0 && __any(heavy)) return TREE_HEAVY;
    return hmax > max_height ? TREE_DEEP : TREE_OK;
}

// ---- general (any depth, any var_len) interpreter: operand stack in scratch memory ---------------
// One wave, lanes are input rows (K = 1).  Used only for trees/configurations the register path
// cannot take; correctness first.  `x_row` points at this lane's input row in global memory.
constexpr int kGeneralOuts = 256;

// STRIDE: entry h of the lane's stack is stk[h * STRIDE] -- 1: a private array; 64: a column of an LDS block the wave's lanes share
// (sr_fast_kernel's folded deep path: no private array, so no scratch segment for the launch to set up)
template <bool MO, int STRIDE = 1>
__device__ inline float run_general(const int16_t *__restrict__ type, const float *__restrict__ value, int len,
                                    const float *__restrict__ x_row, int var_len, int out_len, float *outs,
                                    float *stk_ /* [kMaxStack + 2] private, or the lane's column of [len + 2][64] */) {
    struct { float *q; __device__ float &operator[](int h) const { return q[h * STRIDE]; } } stk{stk_};
    int h = 0;
    if (MO) for (int o = 0; o < out_len; ++o) outs[o] = 0.0f;
    for (int i = len - 1; i >= 0; --i) {
        const Decoded d = decode_node(uni((int)type[i]), bits2f(uni(f2bits(value[i]))), MO, var_len, out_len);
        const uint32_t op = d.op;
        if (op < H_ADD) {
            stk[h++] = op == H_CONST ? bits2f(d.pay) : x_row[d.pay];
            continue;
        }
        float r, last;
        if (op < H_UN) {
            const float a = stk[h - 1], b = stk[h - 2];
            h -= 2;
            if (op == H_ADD) r = a + b;
            else if (op == H_SUB) r = a - b;
            else if (op == H_MUL) r = a * b;
            else if (op == H_DIV) r = b == 0.0f ? __builtin_nanf("") : a / b;
            else r = op_binary_other<false>(op, a, b);
            last = b;
        } else if (op < H_IF) {
            const float a = stk[--h];
            r = op_unary<false>(op, a);
            last = a;
        } else {
            const float a = stk[h - 1], b = stk[h - 2], c = stk[h - 3];
            h -= 3;
            r = a > 0.0f ? b : c;
            last = c;
        }
        if (MO) {
            if (d.pay != kNoOut) outs[d.pay] += r;
            r = last;
        }
        stk[h++] = r;
    }
    return h > 0 ? stk[h - 1] : 0.0f;
}

} // namespace evogp
