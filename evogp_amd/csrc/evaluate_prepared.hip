// evaluate_prepared.hip — the forward pass of a MULTI-OUTPUT policy population, decoded once and run many times (gfx950).
//
// The reference's rollout problems call Forest.forward once per environment step, 1000 times per generation, on the same
// forest (src/evogp/problem/brax_problem.py:54-93 -> forest.py:112-140 -> treeGPEvalKernel, forward.cu:304-351); every
// call re-reads and re-interprets all three tree arrays.  For multi-output trees the interpretation is almost all waste:
// every function node hands its LAST operand on to its parent and only nodes flagged OUT add their result to an output
// (forward.cu:237-243), so the value a subtree passes upward is the value of its rightmost leaf, an OUT node's operands
// are leaves, and nothing else in the tree is ever observable.  A policy tree is a short list of
//
//        outs[o] += f(leaf, leaf [, leaf])                     in execution (reverse prefix) order
//
// evogp_hip_evaluate_prepare builds that list once per forest: 16-byte records {function, output, operand kinds | three
// operands (variable index or constant)}, record i of tree t at [i][t] so that a wave of 64 trees reads a record with one
// coalesced 1-KiB access.  evogp_hip_evaluate_prepared runs it: one lane per tree, the 64 input rows of the wave staged
// through LDS, accumulators in LDS as [output][lane], results written back coalesced.  Per step and tree that is ~16 B per
// OUT node + the input row + the output row instead of 8 B per NODE three arrays wide — and no decoding, no operand stack.
// Trees the list cannot express (inconsistent subtree sizes, node types outside the five classes, more OUT nodes than the
// workspace has rows) are counted by `prepare`; the run then leaves them to the stack interpreter of evaluate.hip.
#include "interp.hpp"
#include "launch.hpp"

namespace evogp {

constexpr uint32_t kSentinelDeepEvalP = 0x7FC0DEEDu;  // evaluate.hip: "this tree's row is produced by eval_general_kernel"

struct PrepareParams {
    const float *value;
    const int16_t *type;
    const int16_t *size;
    uint4 *rec;      // [maxrec][pop]
    int *count;      // [pop]: records of the tree, -1 malformed (NaN row), -2 left to the stack interpreter
    int *info;       // [0] number of trees with count -2 (zeroed before the launch)
    int pop, gp_len, var_len, out_len, maxrec;
};

__global__ __launch_bounds__(256) void eval_prepare_kernel(PrepareParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int t = wave; t < p.pop; t += nwaves) {
        const size_t row = (size_t)t * p.gp_len;
        int len = uni((int)p.size[row]);
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
        int carry_h = 0, nrec = 0;
        bool bad = len <= 0, fallback = false;
        for (int c = (len + 63) / 64 - 1; c >= 0; --c) {  // last chunk first: execution order
            const int i = c * 64 + lane;
            const bool in = i < len;
            const int ty = in ? (int)p.type[row + i] : T_CONST;
            const float val = in ? p.value[row + i] : 0.0f;
            const int sz = in ? (int)p.size[row + i] : 1;
            const int cls = ty & T_MASK;
            const int arity = cls <= T_CONST ? 0 : (cls <= T_TFUNC ? cls - 1 : 3);  // any other type takes the ternary path (forward.cu:213-224)
            if (__any(in && cls > T_TFUNC)) fallback = true;
            const int delta = in ? 1 - arity : 0;
            const int incl = wave_scan_incl(delta);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (__any(in && carry_h + tot - (incl - delta) < 1)) bad = true;
            carry_h += tot;
            // operands through the subtree sizes (verified: sizes that do not describe the tree send it to the stack interpreter)
            const Decoded d = decode_node(ty, val, true, p.var_len, p.out_len);
            int ci = i + 1, sum = 1;
            bool ok = true;
            uint32_t kinds = 0, pay[3] = {0u, 0u, 0u};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (in && a < arity && ok) {
                    const int sc = ci < len ? (int)p.size[row + ci] : 0;
                    if (sc < 1 || ci + sc > len) ok = false;
                    else {
                        const int leaf = ci + sc - 1;                       // the rightmost leaf of the child subtree: what it passes upward
                        const int lt = (int)p.type[row + leaf] & T_MASK;
                        const float lv = p.value[row + leaf];
                        if (lt > T_CONST) ok = false;
                        else if (lt == T_CONST) { kinds |= 1u << a; pay[a] = f2bits(lv); }
                        else { int v = (int)lv; v = v < 0 ? 0 : (v >= p.var_len ? p.var_len - 1 : v); pay[a] = (uint32_t)v; }
                        sum += sc; ci += sc;
                    }
                }
            }
            if (__any(in && (!ok || sz != sum))) fallback = true;
            const bool adds = in && arity > 0 && d.pay != kNoOut && d.op != H_UN_ZERO && d.op != H_BIN_ZERO;  // unknown ids add 0
            const unsigned long long m = __ballot(adds);
            const int slot = nrec + (lane >= 63 ? 0 : __popcll(m >> (lane + 1)));  // higher node index first
            if (adds && slot < p.maxrec)
                p.rec[(size_t)slot * p.pop + t] = make_uint4(d.op | (d.pay << 8) | (kinds << 16) | ((uint32_t)arity << 20), pay[0], pay[1], pay[2]);
            nrec += __popcll(m);
        }
        if (carry_h != 1) bad = true;
        if (nrec > p.maxrec) fallback = true;
        if (lane == 0) {
            p.count[t] = bad ? -1 : (fallback ? -2 : nrec);
            if (!bad && fallback) atomicAdd(p.info, 1);
        }
    }
}

struct PreparedParams {
    const uint4 *rec;
    const int *count;
    const float *vars;   // [pop][var_len]
    float *results;      // [pop][out_len]
    int pop, var_len, out_len;
};

// (the library functions stay behind a call: inlined, their registers and code would weigh on every caller's arithmetic path)
__device__ __attribute__((noinline)) float prepared_apply_other(uint32_t op, float a, float b) {
    if (op >= H_UN) return op_unary<false>(op, a);
    return op_binary_other<false>(op, a, b);
}
__device__ __attribute__((always_inline)) inline float prepared_apply(uint32_t op, float a, float b, float c) {
    if (op >= H_ADD && op <= H_DIV)
        return op == H_ADD ? a + b : op == H_SUB ? a - b : op == H_MUL ? a * b : (b == 0.0f ? __builtin_nanf("") : a / b);  // forward.cu:177-187
    if (op == H_IF) return a > 0.0f ? b : c;
    return prepared_apply_other(op, a, b);
}

__global__ __launch_bounds__(64) void eval_prepared_kernel(PreparedParams p) {
    extern __shared__ float prep_lds[];
    float *var_s = prep_lds;                    // [var_len][64]
    float *out_s = var_s + p.var_len * 64;      // [out_len][64]
    const int lane = threadIdx.x;
    const int t0 = blockIdx.x * 64;
    const int t = t0 + lane;
    const int trees = p.pop - t0 < 64 ? p.pop - t0 : 64;
    for (int e = lane; e < trees * p.var_len; e += 64) {   // the wave's input rows are one contiguous block
        const int tr = e / p.var_len, v = e - tr * p.var_len;
        var_s[v * 64 + tr] = p.vars[(size_t)t0 * p.var_len + e];
    }
    for (int o = 0; o < p.out_len; ++o) out_s[o * 64 + lane] = 0.0f;
    __syncthreads();
    const int n = t < p.pop ? p.count[t] : 0;
    for (int i = 0; i < n; ++i) {
        const uint4 r = p.rec[(size_t)i * p.pop + t];
        const uint32_t op = r.x & 0xFFu, oi = (r.x >> 8) & 0xFFu, kinds = (r.x >> 16) & 7u, arity = (r.x >> 20) & 3u;
        const float a = (kinds & 1u) ? bits2f(r.y) : var_s[r.y * 64 + lane];
        const float b = arity < 2 ? 0.0f : ((kinds & 2u) ? bits2f(r.z) : var_s[r.z * 64 + lane]);
        const float c = arity < 3 ? 0.0f : ((kinds & 4u) ? bits2f(r.w) : var_s[r.w * 64 + lane]);
        out_s[oi * 64 + lane] += prepared_apply(op, a, b, c);   // forward.cu:239-240, in execution order
    }
    if (n < 0)  // malformed: a NaN row (the reference asserts); left to the stack interpreter: its mark in the first word
        for (int o = 0; o < p.out_len; ++o) out_s[o * 64 + lane] = (n == -2 && o == 0) ? bits2f(kSentinelDeepEvalP) : __builtin_nanf("");
    __syncthreads();
    for (int e = lane; e < trees * p.out_len; e += 64) {
        const int tr = e / p.out_len, o = e - tr * p.out_len;
        p.results[(size_t)t0 * p.out_len + e] = out_s[o * 64 + tr];
    }
}

hipError_t launch_eval_marked_general(unsigned pop, unsigned gp_len, unsigned var_len, unsigned out_len, const float *value,
                                      const int16_t *type, const int16_t *size, const float *vars, float *results, hipStream_t stream);

// ---- the same reading of a multi-output tree WITHOUT a prepared list: tree_evaluate itself (round 4) -------------------------------
// evogp_hip_evaluate on multi-output trees used to run the stack interpreter of evaluate.hip: one lane per tree, bound by the serial
// life of the longest tree of a wave -- 39 us for the 50 000 policy trees of BASELINE configs[4], where the prepared pass takes 12
// but costs 85 us to prepare.  eval_direct_kernel does what eval_prepare_kernel does -- one wave per tree, one node per lane, OUT nodes
// find their leaf operands through the subtree sizes -- but with the tree in REGISTERS (cross-lane permutes instead of scattered
// global gathers), evaluates every OUT node at once (lane = node; the tree's input row sits in the lanes of one register) and adds
// the values to their outputs in execution order (higher node index first, forward.cu:239-240).  Same operations in the same order
// as the stack interpreter and the prepared pass: the same bits.  Trees of more than 64 nodes take the same reading chunk by chunk from
// memory (round 5); trees with node types outside the five classes or with subtree sizes that do not describe them are marked for the
// stack interpreter (launched behind this kernel).
__global__ __launch_bounds__(256) void eval_direct_kernel(PreparedParams q, const float *value, const int16_t *type, const int16_t *size, int gp_len) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    auto bperm = [](int v, int i) -> int { return __builtin_amdgcn_ds_bpermute(i << 2, v); };
    const int width = gp_len < 64 ? gp_len : 64;
    // (the loads of the next tree are issued before this one is evaluated.  Nothing may touch a loaded value before the tree's turn:
    // the length used to be a load of its own, converted at once -- and the wait for that conversion, loads completing in order,
    // was a wait for ALL of the next tree's loads in front of the current tree's work: 26 us at the C5 shape instead of 11.  The
    // length is the subtree size of node 0, lane 0's word.)
    int n_ty = T_CONST, n_sz = 0;
    float n_val = 0.0f, n_var = 0.0f;
    auto fetch = [&](int t) {
        n_ty = T_CONST; n_sz = 0; n_val = 0.0f; n_var = 0.0f;
        if (t < q.pop) {
            const size_t row = (size_t)t * gp_len;
            if (lane < width) { n_ty = (int)type[row + lane]; n_val = value[row + lane]; n_sz = (int)size[row + lane]; }
            if (lane < q.var_len) n_var = q.vars[(size_t)t * q.var_len + lane];
        }
    };
    fetch(wave);
    for (int t = wave; t < q.pop; t += nwaves) {
        int len = __builtin_amdgcn_readfirstlane(n_sz);
        const int rty = n_ty, rsz = n_sz;
        const float rval = n_val, xrow = n_var;
        fetch(t + nwaves);
        float *res = q.results + (size_t)t * q.out_len;
        len = len < 0 ? 0 : (len > gp_len ? gp_len : len);
        if (len > 64) {
            // A tree of more than 64 nodes (rows of up to 1024 exist; example/brax_task.py: max_tree_len 256): the same reading chunk by
            // chunk, last chunk first (execution order), operands gathered from the row in memory the way eval_prepare_kernel does, the
            // per-output sums carried from chunk to chunk in lane o.  (Round 4 marked these trees for the one-tree-per-wave stack
            // interpreter on a grid of 2 x CUs workgroups: an evolved population of long rows ran almost entirely there -- ADVICE r04.)
            const size_t row = (size_t)t * gp_len;
            int carry_h = 0;
            bool lbad = false, lfall = false;
            float mine = 0.0f;
            for (int c = (len + 63) / 64 - 1; c >= 0; --c) {
                const int i = c * 64 + lane;
                const bool in = i < len;
                const int ty = in ? (int)type[row + i] : (int)T_CONST;
                const float val = in ? value[row + i] : 0.0f;
                const int sz = in ? (int)size[row + i] : 1;
                const int cls = ty & T_MASK;
                const int arity = cls <= T_CONST ? 0 : (cls <= T_TFUNC ? cls - 1 : 3);
                if (__any(in && cls > T_TFUNC)) lfall = true;
                const int delta = in ? 1 - arity : 0;
                const int incl = wave_scan_incl(delta);
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                if (__any(in && carry_h + tot - (incl - delta) < 1)) lbad = true;
                carry_h += tot;
                const Decoded d = decode_node(ty, val, true, q.var_len, q.out_len);
                int ci = i + 1, sum = 1;
                bool ok = true;
                float opnd[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int a = 0; a < 3; ++a) {   // (the variable's value comes through a permute: every lane runs it)
                    int sc = 0, lt = T_CONST;
                    float lv = 0.0f;
                    const bool want = in && a < arity && ok;
                    if (want) {
                        sc = ci < len ? (int)size[row + ci] : 0;
                        if (sc < 1 || ci + sc > len) ok = false;
                        else { const int leaf = ci + sc - 1; lt = (int)type[row + leaf] & T_MASK; lv = value[row + leaf]; }
                    }
                    int v = (int)lv;
                    v = v < 0 ? 0 : (v >= q.var_len ? q.var_len - 1 : v);
                    const float xv = bits2f((uint32_t)bperm((int)f2bits(xrow), v));
                    if (want && ok) {
                        if (lt > T_CONST) ok = false;
                        else { opnd[a] = lt == T_CONST ? lv : xv; sum += sc; ci += sc; }
                    }
                }
                if (__any(in && (!ok || sz != sum))) lfall = true;
                const bool adds = in && arity > 0 && ok && d.pay != kNoOut && d.op != H_UN_ZERO && d.op != H_BIN_ZERO;
                float r = 0.0f;
                if (adds) r = prepared_apply(d.op, opnd[0], opnd[1], opnd[2]);
                for (int o = 0; o < q.out_len; ++o) {
                    unsigned long long m = __ballot(adds && d.pay == (uint32_t)o);
                    if (m == 0ull) continue;
                    float acc = bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(mine), o));
                    while (m) {
                        const int hi = 63 - __builtin_clzll(m);
                        m &= ~(1ull << hi);
                        acc += bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(r), hi));
                    }
                    if (lane == o) mine = acc;
                }
            }
            if (carry_h != 1) lbad = true;
            if (lbad || lfall) {
                for (int o = lane; o < q.out_len; o += 64) res[o] = (!lbad && o == 0) ? bits2f(kSentinelDeepEvalP) : __builtin_nanf("");
            } else if (lane < q.out_len) res[lane] = mine;
            continue;
        }
        const bool in = lane < len;
        const int ty = in ? rty : (int)T_CONST, sz = in ? rsz : 1;
        const float val = in ? rval : 0.0f;
        const int cls = ty & T_MASK;
        const int arity = cls <= T_CONST ? 0 : (cls <= T_TFUNC ? cls - 1 : 3);  // any other type takes the ternary path (forward.cu:213-224)
        bool fallback = __any(in && cls > T_TFUNC) != 0;
        const int delta = in ? 1 - arity : 0;
        const int incl = wave_scan_incl(delta);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        const bool bad = len <= 0 || __any(in && tot - (incl - delta) < 1) != 0 || tot != 1;
        const Decoded d = decode_node(ty, val, true, q.var_len, q.out_len);
        // the operands: the rightmost leaf of every child subtree (what the subtree passes upward); kind and size travel in one word
        const int ts = cls | (sz << 8);
        int ci = lane + 1, sum = 1;
        bool ok = true;
        float opnd[3] = {0.0f, 0.0f, 0.0f};
        const int max_arity = __builtin_amdgcn_ballot_w64(in && arity > 2) != 0ull ? 3 : __builtin_amdgcn_ballot_w64(in && arity > 1) != 0ull ? 2 : 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {   // (every lane runs the permutes -- of the operands some node of the tree has)
            if (a >= max_arity) break;
            const int sc = bperm(ts, ci) >> 8;
            const int leaf = ci + sc - 1;
            const int lt = bperm(ts, leaf) & 0xFF;
            const float lv = bits2f((uint32_t)bperm((int)f2bits(val), leaf));
            int v = (int)lv;
            v = v < 0 ? 0 : (v >= q.var_len ? q.var_len - 1 : v);
            const float xv = bits2f((uint32_t)bperm((int)f2bits(xrow), v));
            if (in && a < arity && ok) {
                if (ci >= len || sc < 1 || ci + sc > len || lt > T_CONST) ok = false;
                else { opnd[a] = lt == T_CONST ? lv : xv; sum += sc; ci += sc; }
            }
        }
        if (__any(in && (!ok || sz != sum))) fallback = true;
        if (bad || fallback) {   // malformed: a NaN row (the reference asserts); beyond this kernel: the stack interpreter's mark in the first word
            for (int o = lane; o < q.out_len; o += 64) res[o] = (!bad && o == 0) ? bits2f(kSentinelDeepEvalP) : __builtin_nanf("");
            continue;
        }
        const bool adds = in && arity > 0 && d.pay != kNoOut && d.op != H_UN_ZERO && d.op != H_BIN_ZERO;  // unknown ids add 0
        float r = 0.0f;
        if (adds) r = prepared_apply(d.op, opnd[0], opnd[1], opnd[2]);
        // outs[o] = the sum of its OUT nodes' values in execution order; lane o collects output o
        float mine = 0.0f;
        for (int o = 0; o < q.out_len; ++o) {
            unsigned long long m = __ballot(adds && d.pay == (uint32_t)o);
            float acc = 0.0f;
            while (m) {
                const int hi = 63 - __builtin_clzll(m);
                m &= ~(1ull << hi);
                acc += bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(r), hi));
            }
            if (lane == o) mine = acc;
        }
        if (lane < q.out_len) res[lane] = mine;
    }
}

// (Several trees per pass -- the packed program compiler's scheme, sr_tc.hip -- was built for this kernel in round 4 and taken out
// again: 29.7 us against 27.6 at the C5 shape, 19 against 9 at 12 500 trees.  A pass of this kernel is a chain of dependent
// cross-lane steps, ~3 us whatever it holds; sharing passes saves instructions but makes the serial parts -- the plan of a pass, the
// ordered sums of all its trees' OUT nodes -- longer, and at these sizes the chip is not short of issue slots.  Nodes staged in LDS
// with every load of a batch requested up front made no difference either: it is not memory latency.)
// tree_evaluate for multi-output trees of at most 64 variables and 64 outputs: the direct kernel, then the stack interpreter for what it marked
hipError_t launch_eval_direct(unsigned pop, unsigned gp_len, unsigned var_len, unsigned out_len, const float *value, const int16_t *type,
                              const int16_t *size, const float *vars, float *results, hipStream_t stream) {
    PreparedParams q{nullptr, nullptr, vars, results, (int)pop, (int)var_len, (int)out_len};
    long blocks = ((long)pop + 3) / 4;
    const long cap = (long)device_info().num_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(eval_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, q, value, type, size, (int)gp_len);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_eval_marked_general(pop, gp_len, var_len, out_len, value, type, size, vars, results, stream);
}

} // namespace evogp

using namespace evogp;

static unsigned prepared_rows(unsigned gp_len) { return gp_len < 64u ? gp_len : 64u; }

extern "C" size_t evogp_hip_evaluate_workspace_bytes(unsigned pop_size, unsigned gp_len) {
    return (size_t)prepared_rows(gp_len) * pop_size * sizeof(uint4) + (size_t)pop_size * sizeof(int) + 64;
}

extern "C" int evogp_hip_evaluate_prepare(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                          const float *value, const int16_t *type, const int16_t *size, void *workspace,
                                          size_t workspace_bytes, evogp_stream_t stream_) {
    if (pop_size == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len < 2 || out_len > 32 || var_len > 255)
        return EVOGP_E_BADARG;
    if (!value || !type || !size || !workspace) return EVOGP_E_NULLPTR;
    if (workspace_bytes < evogp_hip_evaluate_workspace_bytes(pop_size, gp_len) || ((uintptr_t)workspace & 15u)) return EVOGP_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned maxrec = prepared_rows(gp_len);
    PrepareParams p{};
    p.value = value; p.type = type; p.size = size;
    p.rec = (uint4 *)workspace;
    p.count = (int *)((char *)workspace + (size_t)maxrec * pop_size * sizeof(uint4));
    p.info = p.count + pop_size;
    p.pop = (int)pop_size; p.gp_len = (int)gp_len; p.var_len = (int)var_len; p.out_len = (int)out_len; p.maxrec = (int)maxrec;
    hipError_t e = zero_words_async(p.info, 16, stream);
    if (e != hipSuccess) return (int)e;
    long blocks = ((long)pop_size + 3) / 4;
    const long cap = (long)device_info().num_cus * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(eval_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

extern "C" int evogp_hip_evaluate_prepared(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                           const float *value, const int16_t *type, const int16_t *size, const void *workspace,
                                           int with_fallback, const float *variables, float *results, evogp_stream_t stream_) {
    if (pop_size == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len < 2 || out_len > 32 || var_len > 255)
        return EVOGP_E_BADARG;
    if (!value || !type || !size || !workspace || !variables || !results) return EVOGP_E_NULLPTR;
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned maxrec = prepared_rows(gp_len);
    PreparedParams p{};
    p.rec = (const uint4 *)workspace;
    p.count = (const int *)((const char *)workspace + (size_t)maxrec * pop_size * sizeof(uint4));
    p.vars = variables; p.results = results;
    p.pop = (int)pop_size; p.var_len = (int)var_len; p.out_len = (int)out_len;
    const size_t lds = (size_t)(var_len + out_len) * 64 * sizeof(float);
    hipLaunchKernelGGL(eval_prepared_kernel, dim3((pop_size + 63) / 64), dim3(64), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !with_fallback) return (int)e;
    return (int)launch_eval_marked_general(pop_size, gp_len, var_len, out_len, value, type, size, variables, results, stream);
}
