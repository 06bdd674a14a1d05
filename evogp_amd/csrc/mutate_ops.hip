// mutate_ops.hip — the structural and point mutations of the reference as ONE launch each (round 5).
//
// The reference builds Hoist / Delete from boolean-mask gathers, a (pop, L) array of uniform numbers, an arg-max and tree_crossover
// (src/evogp/algorithm/mutation/hoist.py:43-75, delete.py:44-105, mutation_utils.py:6-48), and the point mutations from six to eight
// (pop, L) random arrays and elementwise programs (single_point.py:43-126, multi_point.py:46-143, single_const.py:39-98,
// multi_const.py:43-95).  evogp_amd/algorithm/mutation.py keeps those programs (`apply`: fed the reference's own draws it reproduces the
// reference bit for bit, tests/test_gpu_mutation_parity.py) -- at 90-120 launches and 1 ms per generation, 5-11 x the fused default step
// (profiles/r05A_n3_generation.log).  Here an operator DRAWS and APPLIES in one kernel: a wave per tree, the random numbers a
// counter-based hash of (seed, call, word, tree or node) (evogp_defs.hpp counter_word, the numbers evogp_amd/parallel.py random_words
// gives: the tests recompute every decision in numpy), the same distributions as the reference's draws:
//   * Delete: mutate with probability `rate` (trees of one node never); the replaced node is uniform among the function nodes whose subtree
//     has at most max_size nodes (the root when there is none: the arg-max of an all-zero row), its replacement child number
//     trunc(1 + u (arity - 1)) -- the reference's randint with the arity as exclusive bound (delete.py:96-101);
//   * Hoist: node p = trunc(u S), inner node trunc(u' size[p]), taken as an ABSOLUTE index like the reference (hoist.py:58-68) or as an offset;
//   * Insert (insert_mutate_kernel): node p = trunc(u S); a fresh tree F of the operator's descriptor -- generated for the mutating rows by
//     evogp_hip_generate_masked_hashed under the SAME words (mask word 4, keys words 7) --; position r = trunc(1 + u' (|F| - 1)) of F takes
//     the subtree at p, and the result takes that subtree's place (insert.py:45-85: tree_crossover, then tree_mutate, each of which leaves
//     its recipient as it is when the row would overflow or a node is out of range);
//   * point mutations: the nodes to redraw -- one per tree, every node of a tree (or node by node) under `intensity`, constants only -- get
//     a payload of their own kind: a function of the same arity from the per-arity roulette (searchsorted left, or scaled to the class
//     total and right with fix_roulette), a variable index, a constant sample; OUT nodes keep or redraw their output index.
// Rows below `skip_rows` (the elites of a generation step) are copied.
#include "evogp_defs.hpp"
#include "launch.hpp"
#include "replace_row.hpp"

#include <cstdint>

namespace evogp {

__device__ inline float word_uniform(unsigned w) { return (float)(w >> 7) * 5.9604644775390625e-08f; }   // 24 bits: [0, 1) like torch.rand

struct StructParams {
    const float *v; const int16_t *t; const int16_t *s;
    float *rv; int16_t *rt; int16_t *rs;
    int *decisions;              // optional [pop][2]: replaced node (-1: copied), donor node
    int pop, gp_len, mode;       // 0 delete, 1 hoist
    int skip_rows, max_size, inner_is_offset;
    float rate;
    unsigned long long base;
    const int *given;            // test entry (evogp_hip_debug_structural_mutate_given): [pop][3] = {mutates, node, child number (delete) / inner
                                 // position (hoist)} in the reference's meaning (delete.py:66-101, hoist.py:53-68) instead of the hashed draws
};

__global__ __launch_bounds__(kRepBlock) void structural_mutate_kernel(StructParams a) {
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(blockIdx.x * (kRepBlock / 64) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (kRepBlock / 64);
    for (int n = wave; n < a.pop; n += nwaves) {
        const size_t off = (size_t)n * a.gp_len;
        const Row L{a.v + off, a.t + off, a.s + off};
        int S = uni((int)L.s[0]);
        S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
        const float u0 = word_uniform(counter_word(a.base, 0u, (unsigned long long)n));
        const unsigned w1 = counter_word(a.base, 1u, (unsigned long long)n), w2 = counter_word(a.base, 2u, (unsigned long long)n);
        bool mutate = n >= a.skip_rows && u0 < a.rate && S >= 1;
        int p = -1, q = 0;
        if (a.given) {   // the reference's own draws (tests): everything behind the draws is the code below
            const int *g = a.given + 3 * (size_t)n;
            mutate = n >= a.skip_rows && uni(g[0]) != 0;
            if (mutate) {
                p = uni(g[1]);
                const int second = uni(g[2]);
                if (a.mode == 0) {
                    const int last = a.gp_len - 1;
                    const int pc = min(max(p, 0), last);
                    const int c1 = min(pc + 1, last), c2 = min(c1 + uni((int)L.s[c1]), last), c3 = min(c2 + uni((int)L.s[c2]), last);
                    q = second == 3 ? c3 : (second == 2 ? c2 : c1);
                } else q = second + (a.inner_is_offset ? p : 0);
            }
        } else if (a.mode == 0) {
            mutate = mutate && S > 1;
            if (mutate) {
                // the k-th function node whose subtree is small enough, k uniform; the root when there is none (delete.py:66-85)
                auto eligible = [&](int i) -> bool {
                    if (i >= S) return false;
                    const int sz = (int)L.s[i];
                    return sz > 1 && (a.max_size <= 0 || sz <= a.max_size);
                };
                int cnt = 0;
                for (int c = 0; c < S; c += 64) cnt += __popcll(__ballot(eligible(c + lane)));
                p = 0;
                if (cnt > 0) {
                    int k = (int)(w1 % (unsigned)cnt);
                    for (int c = 0; c < S; c += 64) {
                        const unsigned long long m = __ballot(eligible(c + lane));
                        const int here = __popcll(m);
                        if (k < here) {   // the k-th set bit of m
                            unsigned long long mm = m;
                            for (int j = 0; j < k; ++j) mm &= mm - 1;
                            p = c + (__ffsll((long long)mm) - 1);
                            break;
                        }
                        k -= here;
                    }
                }
                const int arity = (uni((int)L.t[p]) & T_MASK) - T_UFUNC + 1;
                const int nth = (int)(1.0f + word_uniform(w2) * (float)(arity - 1));   // randint(1, arity): the last child is never drawn
                const int last = a.gp_len - 1;
                const int c1 = min(p + 1, last), c2 = min(c1 + uni((int)L.s[c1]), last), c3 = min(c2 + uni((int)L.s[c2]), last);
                q = nth == 3 ? c3 : (nth == 2 ? c2 : c1);
            }
        } else if (mutate) {
            p = min((int)(word_uniform(w1) * (float)S), S - 1);
            const int sp = uni((int)L.s[p]);
            q = (int)(word_uniform(w2) * (float)sp) + (a.inner_is_offset ? p : 0);
        }
        // tree_crossover of the tree with itself (crossover_kernel): out-of-range nodes and rows that would overflow copy the tree
        bool fallback = !mutate || p < 0 || p >= S || q < 0 || q >= S || q >= a.gp_len;
        int m = 0;
        if (!fallback) {
            m = uni((int)L.s[q]);
            fallback = m < 1 || q + m > a.gp_len || S + (m - uni((int)L.s[p])) > a.gp_len;
        }
        build_row(L, L, S, p, q, m, fallback, a.gp_len, a.rv + off, a.rt + off, a.rs + off);
        if (a.decisions && lane == 0) { a.decisions[2 * (size_t)n] = mutate ? p : -1; a.decisions[2 * (size_t)n + 1] = q; }
    }
}

struct InsertParams {
    const float *v; const int16_t *t; const int16_t *s;       // the forest
    const float *dv; const int16_t *dt; const int16_t *ds;    // the fresh trees, row n for tree n (rows of trees that do not mutate are not read)
    float *rv; int16_t *rt; int16_t *rs;
    int *decisions;              // optional [pop][2]: node of the tree (-1: copied), position inside the fresh tree
    int pop, gp_len, skip_rows;
    unsigned below;              // tree n mutates when word (4, n) < below: the rule of the donor kernel
    unsigned long long base;
    const int *given;            // test entry (evogp_hip_debug_insert_mutate_given): [pop][3] = {mutates, node of the tree, position inside the fresh
                                 // tree} (insert.py:57-79) instead of the hashed draws
};

// A wave per tree; the grafted fresh tree passes through LDS (8 bytes per node and wave).
__global__ __launch_bounds__(kRepBlock) void insert_mutate_kernel(InsertParams a) {
    extern __shared__ float ins_lds[];
    const int lane = threadIdx.x & 63;
    float *gv = ins_lds + (size_t)(threadIdx.x >> 6) * a.gp_len * 2;
    int16_t *gt = reinterpret_cast<int16_t *>(gv + a.gp_len), *gs = gt + a.gp_len;
    const int wave = uni((int)(blockIdx.x * (kRepBlock / 64) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (kRepBlock / 64);
    for (int n = wave; n < a.pop; n += nwaves) {
        const size_t off = (size_t)n * a.gp_len;
        const Row L{a.v + off, a.t + off, a.s + off};
        int S = uni((int)L.s[0]);
        S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
        const bool mutate = n >= a.skip_rows && S >= 1 &&
                            (a.given ? uni(a.given[3 * (size_t)n]) != 0 : counter_word(a.base, 4u, (unsigned long long)n) < a.below);
        int p = -1, r = 0;
        bool fallback = true;
        if (mutate) {
            const Row F{a.dv + off, a.dt + off, a.ds + off};
            int SF = uni((int)F.s[0]);
            SF = SF < 0 ? 0 : (SF > a.gp_len ? a.gp_len : SF);
            if (a.given) {
                p = min(max(uni(a.given[3 * (size_t)n + 1]), 0), S - 1);
                r = uni(a.given[3 * (size_t)n + 2]);
            } else {
                p = min((int)(word_uniform(counter_word(a.base, 1u, (unsigned long long)n)) * (float)S), S - 1);
                r = (int)(1.0f + word_uniform(counter_word(a.base, 2u, (unsigned long long)n)) * (float)(SF - 1));   // randint(1, |F|)
            }
            const int m = uni((int)L.s[p]);          // the subtree that moves
            // F with its subtree at r replaced by the tree's subtree at p (tree_crossover: recipient F); F as it is when that cannot be
            const bool f1 = SF < 1 || r < 1 || r >= SF || m < 1 || p + m > a.gp_len || SF + (m - uni((int)F.s[r < SF ? r : 0])) > a.gp_len;
            build_row(F, L, SF, r, p, m, f1, a.gp_len, gv, gt, gs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const Row G{gv, gt, gs};
            const int SG = uni((int)gs[0]);
            // ... in the place of the subtree at p (tree_mutate); the tree as it is when the row would overflow
            fallback = SG < 1 || S + (SG - m) > a.gp_len;
            build_row(L, G, S, p, 0, SG, fallback, a.gp_len, a.rv + off, a.rt + off, a.rs + off, m);
            __builtin_amdgcn_wave_barrier();         // (the next tree of this wave writes the same LDS rows)
        } else {
            build_row(L, L, S, 0, 0, 0, true, a.gp_len, a.rv + off, a.rt + off, a.rs + off);
        }
        if (a.decisions && lane == 0) { a.decisions[2 * (size_t)n] = mutate && !fallback ? p : -1; a.decisions[2 * (size_t)n + 1] = r; }
    }
}

struct PointParams {
    const float *v; const int16_t *t; const int16_t *s;
    float *rv;                   // new values; types and sizes are the forest's own
    const float *rou_u, *rou_b, *rou_t;   // f32[29] cumulative per-arity roulettes (descriptor.py:113-139)
    const float *consts;
    int pop, gp_len, mode;       // 0 multi-point, 1 single-point, 2 multi-const, 3 single-const
    int skip_rows, per_node, modify_output, fix_roulette;
    int input_len, output_len, n_consts;
    float rate, intensity;
    unsigned long long base;
    // test entry (evogp_hip_debug_point_mutate_given), all [pop][gp_len]: the nodes to redraw and the reference's per-node draws
    // (single_point.py:64-124: the uniform number its roulette search takes for the node's own arity, variable / constant / output indices)
    const unsigned char *g_target;
    const float *g_u;
    const int *g_var, *g_const, *g_out;
};

// searchsorted over a cumulative roulette of kNumFuncs entries: left = entries below x, right = entries not above x
__device__ inline int roulette_left(const float *r, float x) {
    int c = 0;
    for (int i = 0; i < kNumFuncs; ++i) c += r[i] < x ? 1 : 0;
    return c;
}
__device__ inline int roulette_right(const float *r, float x) {
    int c = 0;
    for (int i = 0; i < kNumFuncs; ++i) c += r[i] <= x ? 1 : 0;
    return c;
}

__global__ __launch_bounds__(kRepBlock) void point_mutate_kernel(PointParams a) {
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(blockIdx.x * (kRepBlock / 64) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (kRepBlock / 64);
    for (int n = wave; n < a.pop; n += nwaves) {
        const size_t off = (size_t)n * a.gp_len;
        int S = uni((int)a.s[off]);
        S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
        const float u0 = word_uniform(counter_word(a.base, 0u, (unsigned long long)n));
        const unsigned w1 = counter_word(a.base, 1u, (unsigned long long)n);
        const bool mutate = n >= a.skip_rows && u0 < a.rate;
        int pos = -1;                                   // single-point: the node; single-const: the constant
        bool tree_on = false;                           // multi: the tree's number lies under the intensity
        if (mutate) {
            if (a.mode == 1) pos = min((int)(word_uniform(w1) * (float)S), S - 1);
            else if (a.mode == 3) {
                auto is_const = [&](int i) -> bool { return i < S && (int)a.t[off + i] == T_CONST; };   // (the raw type: single_const.py:55-72)
                int cnt = 0;
                for (int c = 0; c < S; c += 64) cnt += __popcll(__ballot(is_const(c + lane)));
                if (cnt > 0) {
                    int k = (int)(w1 % (unsigned)cnt);
                    for (int c = 0; c < S; c += 64) {
                        const unsigned long long m = __ballot(is_const(c + lane));
                        const int here = __popcll(m);
                        if (k < here) {
                            unsigned long long mm = m;
                            for (int j = 0; j < k; ++j) mm &= mm - 1;
                            pos = c + (__ffsll((long long)mm) - 1);
                            break;
                        }
                        k -= here;
                    }
                }
            } else tree_on = word_uniform(w1) < a.intensity;
        }
        for (int i = lane; i < a.gp_len; i += 64) {
            const float old = a.v[off + i];
            float out = old;
            const int ty = (int)a.t[off + i];
            const unsigned long long node = (unsigned long long)n * (unsigned)a.gp_len + (unsigned)i;
            bool target = mutate && i < S;
            if (a.mode == 1 || a.mode == 3) target = target && i == pos;
            else target = target && (a.per_node ? word_uniform(counter_word(a.base, 12u, node)) < a.intensity : tree_on);
            if (a.mode >= 2) target = target && ty == T_CONST;
            if (a.g_target) target = n >= a.skip_rows && a.g_target[off + i] != 0;
            if (target) {
                const int kind = ty & T_MASK;
                if (a.mode >= 2 || kind == T_CONST) {
                    const int ci = a.g_target ? a.g_const[off + i] : (int)(word_uniform(counter_word(a.base, 10u, node)) * (float)a.n_consts);
                    out = a.consts[min(max(ci, 0), a.n_consts - 1)];
                } else if (kind == T_VAR) {
                    const int vi = a.g_target ? a.g_var[off + i] : (int)(word_uniform(counter_word(a.base, 9u, node)) * (float)a.input_len);
                    out = (float)min(vi, a.input_len - 1);
                } else {
                    const bool is_out = (ty & T_OUT) != 0;
                    const uint32_t bits = f2bits(old);
                    const int old_func = is_out ? (int)(bits & 0xFFFFu) : (int)old;
                    const float *rou = kind >= T_TFUNC ? a.rou_t : (kind == T_BFUNC ? a.rou_b : a.rou_u);   // (single_point.py:86-89: the class index is clamped)
                    const float u = a.g_target ? a.g_u[off + i] : word_uniform(counter_word(a.base, 8u, node));
                    int func;
                    if (!a.fix_roulette) func = roulette_left(rou, u);                                    // may be 29: no function (single_point.py:70-90)
                    else {
                        const float total = rou[kNumFuncs - 1];
                        func = total > 0.0f ? min(roulette_right(rou, u * total), kNumFuncs - 1) : old_func;
                    }
                    if (is_out) {
                        int oi = (int)(bits >> 16);
                        if (a.modify_output)
                            oi = min(a.g_target ? a.g_out[off + i] : (int)(word_uniform(counter_word(a.base, 11u, node)) * (float)a.output_len), a.output_len - 1);
                        out = bits2f((uint32_t)(func + (oi << 16)));
                    } else out = (float)func;
                }
            }
            a.rv[off + i] = out;
        }
    }
}

}  // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_structural_mutate(int pop_size, int gp_len, int mode, float rate, int max_size, int inner_is_offset, int skip_rows,
                                           long long seed, long long call, const float *value, const int16_t *type, const int16_t *size,
                                           float *value_res, int16_t *type_res, int16_t *size_res, int *decisions, evogp_stream_t stream) {
    if (pop_size <= 0 || gp_len <= 0 || gp_len > kMaxStack || mode < 0 || mode > 1 || skip_rows < 0) return EVOGP_E_BADARG;
    if (!value || !type || !size || !value_res || !type_res || !size_res) return EVOGP_E_NULLPTR;
    StructParams a{value, type, size, value_res, type_res, size_res, decisions, pop_size, gp_len, mode, skip_rows, max_size, inner_is_offset, rate,
                   counter_base(seed, call), nullptr};
    long blocks = ((long)pop_size + 3) / 4;
    const long cap = (long)device_info().num_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(structural_mutate_kernel, dim3((unsigned)blocks), dim3(kRepBlock), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// Test entries (include/evogp_hip_debug.h): the same kernels with the draws HANDED IN -- the numbers the reference's Python operators drew,
// recorded by tests/golden/make_mutation_golden.py -- so that everything behind the draws is compared with the reference's own results
// (tests/test_gpu_native_mutation.py).  given: int32 [pop][3] on the device.
extern "C" int evogp_hip_debug_structural_mutate_given(int pop_size, int gp_len, int mode, int inner_is_offset, int skip_rows, const int *given,
                                                       const float *value, const int16_t *type, const int16_t *size, float *value_res,
                                                       int16_t *type_res, int16_t *size_res, evogp_stream_t stream) {
    if (pop_size <= 0 || gp_len <= 0 || gp_len > kMaxStack || mode < 0 || mode > 1 || skip_rows < 0) return EVOGP_E_BADARG;
    if (!given || !value || !type || !size || !value_res || !type_res || !size_res) return EVOGP_E_NULLPTR;
    StructParams a{value, type, size, value_res, type_res, size_res, nullptr, pop_size, gp_len, mode, skip_rows, 0, inner_is_offset, 0.0f, 0ull, given};
    long blocks = ((long)pop_size + 3) / 4;
    const long cap = (long)device_info().num_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(structural_mutate_kernel, dim3((unsigned)blocks), dim3(kRepBlock), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int evogp_hip_debug_insert_mutate_given(int pop_size, int gp_len, int skip_rows, const int *given, const float *value, const int16_t *type,
                                                   const int16_t *size, const float *donor_value, const int16_t *donor_type, const int16_t *donor_size,
                                                   float *value_res, int16_t *type_res, int16_t *size_res, evogp_stream_t stream) {
    if (pop_size <= 0 || gp_len <= 0 || gp_len > kMaxStack || skip_rows < 0) return EVOGP_E_BADARG;
    if (!given || !value || !type || !size || !donor_value || !donor_type || !donor_size || !value_res || !type_res || !size_res) return EVOGP_E_NULLPTR;
    InsertParams a{value, type, size, donor_value, donor_type, donor_size, value_res, type_res, size_res, nullptr, pop_size, gp_len, skip_rows, 0u, 0ull, given};
    long blocks = ((long)pop_size + 3) / 4;
    const long cap = (long)device_info().num_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(insert_mutate_kernel, dim3((unsigned)blocks), dim3(kRepBlock), (size_t)(kRepBlock / 64) * gp_len * 8, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// target: uint8 [pop][gp_len]; u: float [pop][gp_len] (the uniform number of the node's own arity class); var_idx / const_idx / out_idx: int32 [pop][gp_len]
// (out_idx may be null without modify_output).  mode 0 / 1 redraw nodes of every kind, 2 / 3 constants only.
extern "C" int evogp_hip_debug_point_mutate_given(int pop_size, int gp_len, int mode, int modify_output, int fix_roulette, int skip_rows, int input_len,
                                                  int output_len, int n_consts, const unsigned char *target, const float *u, const int *var_idx,
                                                  const int *const_idx, const int *out_idx, const float *value, const int16_t *type, const int16_t *size,
                                                  const float *roulette_ufuncs, const float *roulette_bfuncs, const float *roulette_tfuncs,
                                                  const float *const_samples, float *value_res, evogp_stream_t stream) {
    if (pop_size <= 0 || gp_len <= 0 || gp_len > kMaxStack || mode < 0 || mode > 3 || skip_rows < 0 || input_len <= 0 || output_len <= 0 || n_consts <= 0)
        return EVOGP_E_BADARG;
    if (!target || !const_idx || !value || !type || !size || !value_res || !const_samples) return EVOGP_E_NULLPTR;
    if (mode < 2 && (!u || !var_idx || !roulette_ufuncs || !roulette_bfuncs || !roulette_tfuncs || (modify_output && !out_idx))) return EVOGP_E_NULLPTR;
    PointParams a{value, type, size, value_res, roulette_ufuncs, roulette_bfuncs, roulette_tfuncs, const_samples, pop_size, gp_len, mode, skip_rows,
                  0, modify_output, fix_roulette, input_len, output_len, n_consts, 0.0f, 0.0f, 0ull, target, u, var_idx, const_idx, out_idx};
    long blocks = ((long)pop_size + 3) / 4;
    const long cap = (long)device_info().num_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(point_mutate_kernel, dim3((unsigned)blocks), dim3(kRepBlock), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int evogp_hip_insert_mutate(int pop_size, int gp_len, unsigned mutate_below, int skip_rows, long long seed, long long call,
                                       const float *value, const int16_t *type, const int16_t *size, const float *donor_value,
                                       const int16_t *donor_type, const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res,
                                       int *decisions, evogp_stream_t stream) {
    if (pop_size <= 0 || gp_len <= 0 || gp_len > kMaxStack || skip_rows < 0) return EVOGP_E_BADARG;
    if (!value || !type || !size || !donor_value || !donor_type || !donor_size || !value_res || !type_res || !size_res) return EVOGP_E_NULLPTR;
    InsertParams a{value, type, size, donor_value, donor_type, donor_size, value_res, type_res, size_res, decisions, pop_size, gp_len, skip_rows,
                   mutate_below, counter_base(seed, call), nullptr};
    long blocks = ((long)pop_size + 3) / 4;
    const long cap = (long)device_info().num_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(insert_mutate_kernel, dim3((unsigned)blocks), dim3(kRepBlock), (size_t)(kRepBlock / 64) * gp_len * 8, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int evogp_hip_point_mutate(int pop_size, int gp_len, int mode, float rate, float intensity, int per_node, int modify_output,
                                      int fix_roulette, int skip_rows, int input_len, int output_len, int n_consts, long long seed, long long call,
                                      const float *value, const int16_t *type, const int16_t *size, const float *roulette_ufuncs,
                                      const float *roulette_bfuncs, const float *roulette_tfuncs, const float *const_samples, float *value_res,
                                      evogp_stream_t stream) {
    if (pop_size <= 0 || gp_len <= 0 || gp_len > kMaxStack || mode < 0 || mode > 3 || skip_rows < 0 || input_len <= 0 || output_len <= 0 || n_consts <= 0)
        return EVOGP_E_BADARG;
    if (!value || !type || !size || !value_res || !const_samples) return EVOGP_E_NULLPTR;
    if (mode < 2 && (!roulette_ufuncs || !roulette_bfuncs || !roulette_tfuncs)) return EVOGP_E_NULLPTR;
    PointParams a{value, type, size, value_res, roulette_ufuncs, roulette_bfuncs, roulette_tfuncs, const_samples, pop_size, gp_len, mode, skip_rows,
                  per_node, modify_output, fix_roulette, input_len, output_len, n_consts, rate, intensity, counter_base(seed, call), nullptr, nullptr, nullptr,
                  nullptr, nullptr};
    long blocks = ((long)pop_size + 3) / 4;
    const long cap = (long)device_info().num_cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(point_mutate_kernel, dim3((unsigned)blocks), dim3(kRepBlock), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
