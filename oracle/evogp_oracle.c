/*
 * evogp_oracle.c — plain-C CPU restatement of the reference's hot path.
 * TEST INFRASTRUCTURE ONLY (see evogp_oracle.h).  Parity status: pinned against oracle/_ref
 * and tests/golden/.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/src/evogp/cuda/).  The RNG (Thrust taus88 + uniform_real_distribution<float>)
 * is third-party and un-vendored in the reference (kernel.h:8,20); its published algorithm
 * (L'Ecuyer 1996, three-component Tausworthe; rocThrust 7.2:
 * thrust/random/detail/linear_feedback_shift_engine.inl:47-55, xor_combine_engine.inl:97-105,
 * uniform_real_distribution.inl:62-78) is restated here and pinned by Thrust's documented known
 * answer (10000th draw of the default-seeded engine = 3535848941, thrust/random.h:80-81).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp; no -ffast-math: results are IEEE fp32).
 */
#include "evogp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- encodings: defs.h:5-57 ------------------------------------------------------------- */
enum { MAXSTACK = 1024, MAXDEPTH = 10, NFUNC = 29 };
enum { T_VAR = 0, T_CONST = 1, T_UFUNC = 2, T_BFUNC = 3, T_TFUNC = 4, T_MASK = 0x7F, T_OUT = 0x80 };
enum {
    F_IF = 0, F_ADD, F_SUB, F_MUL, F_DIV, F_LOOSE_DIV, F_POW, F_LOOSE_POW, F_MAX, F_MIN, F_LT, F_GT,
    F_LE, F_GE, F_SIN, F_COS, F_TAN, F_SINH, F_COSH, F_TANH, F_LOG, F_LOOSE_LOG, F_EXP, F_INV,
    F_LOOSE_INV, F_NEG, F_ABS, F_SQRT, F_LOOSE_SQRT
};
static const float DELTA_F = 1e-9f, MAXVAL_F = 1e9f; /* defs.h:7-8 */

/* ---- seed hash: kernel.h:157-172 (FNV-1a-64 over the 12 LE bytes of {n,k1,k2}, low 32 bits) */
uint32_t evogp_oracle_hash(uint32_t n, uint32_t k1, uint32_t k2) {
    const uint32_t words[3] = {n, k1, k2};
    uint64_t h = 14695981039346656037ULL;
    for (int w = 0; w < 3; ++w)
        for (int b = 0; b < 4; ++b) {
            h ^= (uint64_t)((words[w] >> (8 * b)) & 0xFFu);
            h *= 1099511628211ULL;
        }
    return (uint32_t)h;
}

/* ---- taus88: three LFSRs (k,q,s) = (31,13,12),(29,2,4),(28,3,17), all seeded with the raw seed */
typedef struct { uint32_t a, b, c; } taus_t;
static inline void taus_seed(taus_t *g, uint32_t s) { g->a = g->b = g->c = s; }
static inline uint32_t lfsr(uint32_t z, int k, int q, int s) {
    const uint32_t t = ((z << q) ^ z) >> (k - s);
    return ((z & (0xFFFFFFFFu << (32 - k))) << s) ^ t;
}
static inline uint32_t taus_next(taus_t *g) {
    g->a = lfsr(g->a, 31, 13, 12);
    g->b = lfsr(g->b, 29, 2, 4);
    g->c = lfsr(g->c, 28, 3, 17);
    return g->a ^ g->b ^ g->c;
}
/* uniform_real_distribution<float>(0,1): float(u32) / (1.0f + float(0xFFFFFFFF)) == float(u32) * 2^-32;
 * the u32->f32 conversion rounds to nearest, so the result can be exactly 1.0f. */
static inline float taus_uniform(taus_t *g) { return (float)taus_next(g) / 4294967296.0f; }

void evogp_oracle_taus88(uint32_t seed, int count, uint32_t *out, float *fout) {
    taus_t g, h;
    taus_seed(&g, seed);
    taus_seed(&h, seed);
    for (int i = 0; i < count; ++i) {
        if (out) out[i] = taus_next(&g);
        if (fout) fout[i] = taus_uniform(&h);
    }
}

static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ---- generate: generate.cu:33-172 ------------------------------------------------------- */
static void gen_one(unsigned n_global, unsigned gp_len, unsigned var_len, unsigned out_len,
                    unsigned n_const, float out_prob, float const_prob, const unsigned *keys,
                    const float *leaf_probs, const float *roulette, const float *consts,
                    float *o_val, int16_t *o_type, int16_t *o_size) {
    float val[MAXSTACK];
    int16_t typ[MAXSTACK], siz[MAXSTACK];
    int16_t pend_childs[MAXSTACK], pend_depth[MAXSTACK]; /* explicit DFS stack, generate.cu:56-58 */
    int n_nodes = 0, top = 0;
    const int multi = out_len > 1;
    taus_t g;
    taus_seed(&g, evogp_oracle_hash(n_global, keys[0], keys[1])); /* generate.cu:40 */

    pend_childs[0] = 1; pend_depth[0] = 0; top = 1;
    while (top > 0 && n_nodes < MAXSTACK) {
        int childs = pend_childs[--top] - 1;
        const int depth = pend_depth[top];
        int new_childs = 0;
        /* intended semantics of generate.cu:71: depths past the table are leaves */
        const float lp = depth < MAXDEPTH ? leaf_probs[depth] : 1.0f;
        if (taus_uniform(&g) >= lp) { /* function node, generate.cu:71-100 */
            const float r = taus_uniform(&g);
            int k = 0;
            for (int i = NFUNC - 1; i >= 0; --i)
                if (r >= roulette[i]) { k = i + 1; break; }
            const int t = k <= F_IF ? T_TFUNC : (k <= F_GE ? T_BFUNC : T_UFUNC);
            int is_out = 0;
            if (multi && taus_uniform(&g) <= out_prob) { /* generate.cu:86-96 */
                const uint32_t oi = taus_next(&g) % out_len;
                val[n_nodes] = bits_to_float(((oi & 0xFFFFu) << 16) | ((uint32_t)k & 0xFFFFu));
                typ[n_nodes] = (int16_t)(t | T_OUT);
                is_out = 1;
            }
            if (!is_out) { val[n_nodes] = (float)k; typ[n_nodes] = (int16_t)t; }
            new_childs = t - 1; /* generate.cu:102 */
        } else { /* leaf, generate.cu:104-123 */
            if (taus_uniform(&g) <= const_prob) {
                val[n_nodes] = consts[taus_next(&g) % n_const];
                typ[n_nodes] = T_CONST;
            } else {
                val[n_nodes] = (float)(taus_next(&g) % var_len);
                typ[n_nodes] = T_VAR;
            }
        }
        ++n_nodes;
        if (childs > 0) { pend_childs[top] = (int16_t)childs; pend_depth[top] = (int16_t)depth; ++top; }
        if (new_childs > 0) { pend_childs[top] = (int16_t)new_childs; pend_depth[top] = (int16_t)(depth + 1); ++top; }
    }
    /* subtree sizes by a reverse scan with a size stack: generate.cu:130-158 */
    int sstack[MAXSTACK];
    int sp = 0;
    for (int i = n_nodes - 1; i >= 0; --i) {
        const int t = typ[i] & T_MASK;
        int s = 1;
        const int arity = t <= T_CONST ? 0 : t - 1;
        for (int a = 0; a < arity && sp > 0; ++a) s += sstack[--sp];
        sstack[sp++] = s;
        siz[i] = (int16_t)s;
    }
    /* write [0,len), zero the tail (the reference leaves it uninitialised, generate.cu:161-172) */
    const int len = n_nodes > 0 ? siz[0] : 0;
    for (unsigned i = 0; i < gp_len; ++i) {
        const int live = (int)i < len;
        o_val[i] = live ? val[i] : 0.0f;
        o_type[i] = live ? typ[i] : 0;
        o_size[i] = live ? siz[i] : 0;
    }
}

void evogp_oracle_generate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                           unsigned const_samples_len, float out_prob, float const_prob,
                           const unsigned *keys, const float *depth2leaf_probs,
                           const float *roulette_funcs, const float *const_samples,
                           float *value_res, int16_t *type_res, int16_t *size_res,
                           unsigned tree_index_offset) {
#pragma omp parallel for schedule(static)
    for (long n = 0; n < (long)pop_size; ++n)
        gen_one((unsigned)n + tree_index_offset, gp_len, var_len, out_len, const_samples_len, out_prob,
                const_prob, keys, depth2leaf_probs, roulette_funcs, const_samples,
                value_res + (size_t)n * gp_len, type_res + (size_t)n * gp_len, size_res + (size_t)n * gp_len);
}

/* ---- subtree replacement: mutation.cu:5-115 --------------------------------------------- */
static void copy_tree(int gp_len, const float *v, const int16_t *t, const int16_t *s, float *ov,
                      int16_t *ot, int16_t *os) {
    int len = s[0];
    if (len < 0) len = 0;
    if (len > gp_len) len = gp_len;
    for (int i = 0; i < gp_len; ++i) {
        const int live = i < len;
        ov[i] = live ? v[i] : 0.0f; ot[i] = live ? t[i] : 0; os[i] = live ? s[i] : 0;
    }
}

static void replace_subtree(int gp_len, int p, int q, int m, const float *ov, const int16_t *ot,
                            const int16_t *os, const float *nv, const int16_t *nt, const int16_t *ns,
                            float *rv, int16_t *rt, int16_t *rs) {
    const int S = os[0], o = os[p], diff = m - o;
    float bv[MAXSTACK]; int16_t bt[MAXSTACK], bs[MAXSTACK];
    for (int i = 0; i < p; ++i) { bv[i] = ov[i]; bt[i] = ot[i]; bs[i] = os[i]; } /* :30-35 */
    /* root-to-node descent adding diff to every ancestor: mutation.cu:38-88 */
    int cur = 0;
    while (cur < p) {
        bs[cur] = (int16_t)(bs[cur] + diff);
        const int t = bt[cur] & T_MASK;
        ++cur;
        if (cur >= p) break;
        if (t == T_BFUNC) {
            const int right = cur + os[cur];
            if (p >= right) cur = right;
        } else if (t == T_TFUNC) {
            const int mid = cur + os[cur];
            if (p >= mid) {
                const int right = mid + os[mid];
                cur = p < right ? mid : right;
            }
        }
    }
    for (int i = 0; i < m; ++i) { bv[p + i] = nv[q + i]; bt[p + i] = nt[q + i]; bs[p + i] = ns[q + i]; } /* :91-96 */
    for (int i = p + o; i < S; ++i) { bv[i + diff] = ov[i]; bt[i + diff] = ot[i]; bs[i + diff] = os[i]; } /* :99-104 */
    const int len = S + diff;
    for (int i = 0; i < gp_len; ++i) {
        const int live = i < len;
        rv[i] = live ? bv[i] : 0.0f; rt[i] = live ? bt[i] : 0; rs[i] = live ? bs[i] : 0;
    }
}

/* mutation.cu:118-184 */
void evogp_oracle_mutate(int pop_size, int gp_len, const float *value_ori, const int16_t *type_ori,
                         const int16_t *size_ori, const int *mutate_indices, const float *value_new,
                         const int16_t *type_new, const int16_t *size_new, float *value_res,
                         int16_t *type_res, int16_t *size_res) {
#pragma omp parallel for schedule(static)
    for (long n = 0; n < pop_size; ++n) {
        const size_t off = (size_t)n * gp_len;
        const float *ov = value_ori + off; const int16_t *ot = type_ori + off, *os = size_ori + off;
        const float *nv = value_new + off; const int16_t *nt = type_new + off, *ns = size_new + off;
        const int p = mutate_indices[n], S = os[0];
        /* ns[0] outside [1, gp_len] (a malformed donor) is undefined in the reference; defined here as copy-old */
        if (p < 0 || p >= S || ns[0] < 1 || ns[0] > gp_len || S + (ns[0] - os[p]) > gp_len) {
            copy_tree(gp_len, ov, ot, os, value_res + off, type_res + off, size_res + off);
            continue;
        }
        replace_subtree(gp_len, p, 0, ns[0], ov, ot, os, nv, nt, ns, value_res + off, type_res + off,
                        size_res + off);
    }
}

/* mutation.cu:224-309 */
void evogp_oracle_crossover(int pop_size_ori, int pop_size_new, int gp_len, const float *value_ori,
                            const int16_t *type_ori, const int16_t *size_ori, const int *left_idx,
                            const int *right_idx, const int *left_node_idx, const int *right_node_idx,
                            float *value_res, int16_t *type_res, int16_t *size_res) {
#pragma omp parallel for schedule(static)
    for (long n = 0; n < pop_size_new; ++n) {
        const size_t off = (size_t)n * gp_len, lo = (size_t)left_idx[n] * gp_len;
        const float *lv = value_ori + lo; const int16_t *lt = type_ori + lo, *ls = size_ori + lo;
        const int r = right_idx[n], S = ls[0], p = left_node_idx[n], q = right_node_idx[n];
        int fallback = (r < 0 || r >= pop_size_ori);
        const float *rv = 0; const int16_t *rt = 0, *rs = 0;
        if (!fallback) {
            const size_t ro = (size_t)r * gp_len;
            rv = value_ori + ro; rt = type_ori + ro; rs = size_ori + ro;
            /* node indices outside the live trees are undefined in the reference; defined here as copy-left */
            if (p < 0 || p >= S || q < 0 || q >= rs[0] || q >= gp_len) fallback = 1;
            else if (rs[q] < 1 || q + rs[q] > gp_len) fallback = 1;
            else if (S + (rs[q] - ls[p]) > gp_len) fallback = 1; /* :279-289 */
        }
        if (fallback) { copy_tree(gp_len, lv, lt, ls, value_res + off, type_res + off, size_res + off); continue; }
        replace_subtree(gp_len, p, q, rs[q], lv, lt, ls, rv, rt, rs, value_res + off, type_res + off,
                        size_res + off);
    }
}

/* ---- sensitivity probe (TESTS ONLY, not reference semantics) -------------------------------
 * Two correct fp32 math libraries differ by an ulp or two per call, and a tree can amplify that without bound
 * (tan(tan(x)), a / (sin(x) - sin(x))).  With jitter enabled every libm-backed result (sin cos tan sinh cosh tanh log
 * exp pow) is moved by a pseudo-random number of ulps in [-J, +J]; the spread of a tree's fitness over a few seeds is
 * the tolerance a parity test may grant THAT tree (tests/test_gpu_parity.py), instead of a blanket "98 % of the trees". */
static int g_jitter_ulps = 0;
static uint32_t g_jitter_seed = 0;
static __thread uint32_t t_jitter_state = 0;
void evogp_oracle_set_jitter(int ulps, unsigned seed) { g_jitter_ulps = ulps < 0 ? 0 : ulps; g_jitter_seed = seed; }
static inline void jitter_begin(long tree) { t_jitter_state = g_jitter_seed ^ ((uint32_t)tree * 2654435761u); }
static inline float jit(float r) {
    if (g_jitter_ulps == 0 || !isfinite(r) || r == 0.0f) return r;
    t_jitter_state = t_jitter_state * 1664525u + 1013904223u;
    int k = (int)((t_jitter_state >> 8) % (uint32_t)(2 * g_jitter_ulps + 1)) - g_jitter_ulps;
    if (g_jitter_seed == 0xFFFFFFFFu) k = g_jitter_ulps;   /* systematic probes: every result +J ulps / -J ulps (a tree whose */
    if (g_jitter_seed == 0xFFFFFFFEu) k = -g_jitter_ulps;  /* value hangs on ONE library call is otherwise a lottery)         */
    uint32_t b = float_to_bits(r);
    const uint32_t mag = (b & 0x7FFFFFFFu) + (uint32_t)k; /* sign-magnitude: +k ulps away from zero */
    if (mag == 0u || mag >= 0x7F800000u) return r;
    b = (b & 0x80000000u) | mag;
    return bits_to_float(b);
}

/* ---- interpreter: forward.cu:79-244 (node), :246-302 (tree) ------------------------------ */
static inline float apply_unary(unsigned f, float a) {
    switch (f) {
    case F_SIN: return jit(sinf(a));
    case F_COS: return jit(cosf(a));
    case F_TAN: return jit(tanf(a));
    case F_SINH: return jit(sinhf(a));
    case F_COSH: return jit(coshf(a));
    case F_TANH: return jit(tanhf(a));
    case F_LOG: return jit(logf(a));
    case F_LOOSE_LOG: return a == 0.0f ? -MAXVAL_F : jit(logf(fabsf(a)));
    case F_EXP: return jit(expf(a));
    case F_INV: return a == 0.0f ? NAN : 1.0f / a;
    case F_LOOSE_INV: if (fabsf(a) <= DELTA_F) a = copysignf(DELTA_F, a); return 1.0f / a;
    case F_NEG: return -a;
    case F_ABS: return fabsf(a);
    case F_SQRT: return sqrtf(a);
    case F_LOOSE_SQRT: return sqrtf(fabsf(a));
    default: return 0.0f; /* forward.cu:117: unknown ids leave the zero-initialised result */
    }
}
static inline float apply_binary(unsigned f, float a, float b) {
    switch (f) {
    case F_ADD: return a + b;
    case F_SUB: return a - b;
    case F_MUL: return a * b;
    case F_DIV: return b == 0.0f ? NAN : a / b;
    case F_LOOSE_DIV: if (fabsf(b) <= DELTA_F) b = copysignf(DELTA_F, b); return a / b;
    case F_POW: return jit(powf(a, b));
    case F_LOOSE_POW: return (a == 0.0f && b == 0.0f) ? 0.0f : jit(powf(fabsf(a), b));
    case F_MAX: return a >= b ? a : b;
    case F_MIN: return a <= b ? a : b;
    case F_LT: return a < b ? 1.0f : -1.0f;
    case F_GT: return a > b ? 1.0f : -1.0f;
    case F_LE: return a <= b ? 1.0f : -1.0f;
    case F_GE: return a >= b ? 1.0f : -1.0f;
    default: return 0.0f;
    }
}

/* Evaluates one tree on one input row.  Single-output: returns the stack top in outs[0].
 * Multi-output (out_len > 1): outs[0..out_len) are the additive OUT-node accumulators and every
 * function node forwards its LAST popped operand to its parent (forward.cu:237-243).
 * Returns the final stack height (the reference asserts it is 1, forward.cu:298-301). */
static int eval_tree(const float *value, const int16_t *type, int len, const float *vars,
                     unsigned out_len, float *outs, float *stack) {
    const int multi = out_len > 1;
    int top = 0;
    if (multi) for (unsigned o = 0; o < out_len; ++o) outs[o] = 0.0f;
    for (int i = len - 1; i >= 0; --i) { /* reverse scan of the prefix array, forward.cu:281-296 */
        int t = type[i];
        const float v = value[i];
        int is_out = 0;
        if (multi) { is_out = t & T_OUT; t &= T_MASK; } /* forward.cu:91-94: masked only in multi mode */
        if (t == T_CONST) { stack[top++] = v; continue; }
        if (t == T_VAR) { stack[top++] = vars[(int)v]; continue; }
        unsigned f = (unsigned)v, oi = 0;
        if (multi && is_out) { /* forward.cu:106-115 */
            const uint32_t bits = float_to_bits(v);
            f = (unsigned)(int16_t)(bits & 0xFFFFu);
            oi = (unsigned)(int16_t)(bits >> 16);
        }
        float r, last;
        if (t == T_UFUNC) {
            const float a = stack[--top];
            last = a; r = apply_unary(f, a);
        } else if (t == T_BFUNC) {
            const float a = stack[--top], b = stack[--top];
            last = b; r = apply_binary(f, a, b);
        } else { /* everything else is the ternary IF, forward.cu:213-224 */
            const float a = stack[--top], b = stack[--top], c = stack[--top];
            last = c; r = a > 0.0f ? b : c;
        }
        if (multi) {
            if (is_out && oi < out_len) outs[oi] += r;
            r = last;
        }
        stack[top++] = r;
    }
    if (!multi) outs[0] = top > 0 ? stack[top - 1] : 0.0f;
    return top;
}

static inline int live_len(const int16_t *size, unsigned gp_len) {
    int len = size[0];
    if (len < 0) len = 0;
    if ((unsigned)len > gp_len) len = (int)gp_len;
    return len;
}

/* forward.cu:304-351 */
void evogp_oracle_evaluate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                           const float *value, const int16_t *type, const int16_t *size,
                           const float *variables, float *results) {
#pragma omp parallel
    {
        float stack[MAXSTACK + 4], outs[MAXSTACK];
#pragma omp for schedule(static)
        for (long n = 0; n < (long)pop_size; ++n) {
            const size_t off = (size_t)n * gp_len;
            jitter_begin(n);
            eval_tree(value + off, type + off, live_len(size + off, gp_len), variables + (size_t)n * var_len,
                      out_len, outs, stack);
            for (unsigned o = 0; o < out_len; ++o) results[(size_t)n * out_len + o] = outs[o];
        }
    }
}

/* forest.py:143-176 semantics without the replication */
void evogp_oracle_batch_evaluate(unsigned pop_size, unsigned data_points, unsigned gp_len,
                                 unsigned var_len, unsigned out_len, const float *value,
                                 const int16_t *type, const int16_t *size, const float *variables,
                                 float *results) {
#pragma omp parallel
    {
        float stack[MAXSTACK + 4], outs[MAXSTACK];
#pragma omp for schedule(dynamic, 16)
        for (long n = 0; n < (long)pop_size; ++n) {
            const size_t off = (size_t)n * gp_len;
            const int len = live_len(size + off, gp_len);
            jitter_begin(n);
            for (unsigned d = 0; d < data_points; ++d) {
                eval_tree(value + off, type + off, len, variables + (size_t)d * var_len, out_len, outs, stack);
                for (unsigned o = 0; o < out_len; ++o)
                    results[((size_t)n * data_points + d) * out_len + o] = outs[o];
            }
        }
    }
}

/* forward.cu:375-400 (per-datapoint error), :456-471 (pairwise block reduction over 1024 lanes,
 * one partial per block added in block order), :474-479 (divide by D) */
int evogp_oracle_sr_fitness(unsigned pop_size, unsigned data_points, unsigned gp_len,
                            unsigned var_len, unsigned out_len, int use_mse, const float *value,
                            const int16_t *type, const int16_t *size, const float *variables,
                            const float *labels, float *fitnesses, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#else
    (void)threads;
#endif
    enum { BLOCK = 1024 };
#pragma omp parallel num_threads(used)
    {
        float stack[MAXSTACK + 4], outs[MAXSTACK], lane[BLOCK];
#pragma omp for schedule(dynamic, 16)
        for (long n = 0; n < (long)pop_size; ++n) {
            const size_t off = (size_t)n * gp_len;
            const int len = live_len(size + off, gp_len);
            float total = 0.0f;
            jitter_begin(n);
            for (unsigned base = 0; base < data_points; base += BLOCK) {
                for (unsigned l = 0; l < BLOCK; ++l) {
                    const unsigned d = base + l;
                    float err = 0.0f;
                    if (d < data_points) {
                        eval_tree(value + off, type + off, len, variables + (size_t)d * var_len, out_len, outs, stack);
                        for (unsigned o = 0; o < out_len; ++o) {
                            const float diff = labels[(size_t)d * out_len + o] - outs[o];
                            err += use_mse ? diff * diff : fabsf(diff);
                        }
                    }
                    lane[l] = err;
                }
                for (unsigned w = BLOCK / 2; w > 0; w >>= 1)
                    for (unsigned l = 0; l < w; ++l) lane[l] += lane[l + w];
                total += lane[0];
            }
            fitnesses[n] = total / (float)data_points;
        }
    }
    return used;
}

/* size[i] = 1 + sum(children), arity from the masked type: tree.py:361-413 */
int evogp_oracle_validate_tree(int gp_len, const int16_t *type, const int16_t *size) {
    const int len = size[0];
    if (len < 1 || len > gp_len) return -1;
    int sstack[MAXSTACK], sp = 0;
    for (int i = len - 1; i >= 0; --i) {
        const int t = type[i] & T_MASK;
        if (t > T_TFUNC) return i + 1;
        const int arity = t <= T_CONST ? 0 : t - 1;
        if (sp < arity) return i + 1;
        int s = 1;
        for (int a = 0; a < arity; ++a) s += sstack[--sp];
        if (size[i] != s) return i + 1;
        sstack[sp++] = s;
    }
    return sp == 1 ? 0 : 1;
}
