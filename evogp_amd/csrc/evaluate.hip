// evaluate.hip — population forward pass with one input row per tree (gfx950).
//
// Replaces  evaluate / treeGPEvalKernel  (src/evogp/cuda/forward.cu:304-371):
//     results[n][:] = tree_n(variables[n][:])
// The reference runs one THREAD per tree (divergent dispatch, 8 KB of local memory per thread,
// row-strided loads).  eval_lane_kernel keeps the lane-per-tree mapping — with one input row per tree it is
// the only one that uses the lanes — but stages nodes, inputs, stacks and accumulators through LDS so that
// every HBM access is coalesced (see the comment at the kernel).
//
// Trees whose operand stack exceeds the LDS stack are marked and redone by a scratch-stack kernel (wave
// per tree) launched behind the fast one; malformed trees yield NaN rows.
#include "interp.hpp"
#include "launch.hpp"
#include <cstdlib>

namespace evogp {

constexpr uint32_t kSentinelDeepEval = 0x7FC0DEEDu;

struct EvalParams {
    const float *value;
    const int16_t *type;
    const int16_t *size;
    const float *vars;  // [pop][var_len]
    float *results;     // [pop][out_len]
    int pop, gp_len, var_len, out_len;
};

// ---- lane-per-tree kernel ---------------------------------------------------------------------------------
// One LANE per tree (64 trees per wave), because this op has ONE input row per tree: there is no second axis to spread
// over the lanes, and a wave per tree (the first version of this file) leaves 63 lanes idle — 123 us for 50 k trees.
// What the reference's thread-per-tree kernel pays for — row-strided loads and 8 KB of local memory per thread — is
// avoided by staging: the wave copies a chunk of 32 nodes of each of its 64 trees from HBM into LDS with coalesced loads
// (two trees per load instruction, nodes in execution = reverse prefix order, already decoded to {handler id,
// payload}), transposed [node][tree] with a row pitch of 65 words so that both the staging writes and the per-lane
// reads are bank-conflict free.  The operand stacks, the input rows and the multi-output accumulators live in LDS as
// [entry][lane].  Lanes then interpret their own tree in lockstep over the execution index; dispatch diverges by node
// class only (leaf / arithmetic / other binary / unary / ternary).
//
// TPW trees per wave: all 64 lanes stage, lanes 0 .. TPW-1 interpret.  The op is bound by the serial chain of its longest
// tree, not by throughput (50 k trees are 782 full waves for 1024 SIMDs), so FEWER trees per wave is faster: the staging
// of a chunk shrinks with TPW (64 lanes load 32 nodes of TPW trees), the longest tree of 16 is shorter than the longest
// of 64, and the chip holds all the extra waves at once.
constexpr int kLaneChunk = 32;
constexpr int kLaneDepth = 24;
constexpr uint32_t kCtlNop = 0xFFu | (2u << 12);  // handler "none", no operands, stack delta 0

template <bool MO, int TPW>
__global__ __launch_bounds__(64) void eval_lane_kernel(EvalParams p) {
    constexpr int kLanePitch = TPW + 1;
    extern __shared__ uint32_t lane_lds[];
    uint32_t *op_s = lane_lds;                                 // [kLaneChunk][kLanePitch]
    uint32_t *pay_s = op_s + kLaneChunk * kLanePitch;          // [kLaneChunk][kLanePitch]
    float *stk = (float *)(pay_s + kLaneChunk * kLanePitch);   // [kLaneDepth + 1][TPW]  (last row: sink for non-pushes)
    float *var_s = stk + (kLaneDepth + 1) * TPW;               // [var_len][TPW]
    float *out_s = var_s + p.var_len * TPW;                    // [out_len + 1][TPW]  (multi-output only; last row: sink)
    const int lane = threadIdx.x;
    const int t0 = blockIdx.x * TPW;
    const int t = t0 + lane;
    const bool active = lane < TPW && t < p.pop;
    int len = 0;
    if (active) {
        len = (int)p.size[(size_t)t * p.gp_len];
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
    }
    // input rows: the 64 rows of this wave are contiguous in memory; copy them coalesced, store transposed
    const int nrows = p.pop - t0 < TPW ? p.pop - t0 : TPW;
    for (int e = lane; e < nrows * p.var_len; e += 64) {
        const int r = e / p.var_len, v = e - r * p.var_len;
        var_s[v * TPW + r] = p.vars[(size_t)t0 * p.var_len + e];
    }
    if (MO && lane < TPW) for (int o = 0; o <= p.out_len; ++o) out_s[o * TPW + lane] = 0.0f;
    const int maxlen = wave_max(len);
    int h = 0;
    float tos = 0.0f;
    int hmin = 0, hmax = 0;  // lowest "height - operands needed" and greatest height seen: validity is judged at the end
    for (int c0 = 0; c0 < maxlen; c0 += kLaneChunk) {
        // ---- stage the next 32 instructions of every tree: all loads first (64 in flight), then decode + store ----
        const int jj = lane & 31;
        const int k = c0 + jj;
        int ty_r[TPW / 2];
        float vl_r[TPW / 2];
#pragma unroll
        for (int it = 0; it < TPW / 2; ++it) {
            const int tl = 2 * it + (lane >> 5);
            const int len_t = __shfl(len, tl);
            ty_r[it] = -1; vl_r[it] = 0.0f;
            if (k < len_t) {
                const size_t at = (size_t)(t0 + tl) * p.gp_len + (size_t)(len_t - 1 - k);
                ty_r[it] = (int)p.type[at]; vl_r[it] = p.value[at];
            }
        }
#pragma unroll
        for (int it = 0; it < TPW / 2; ++it) {
            const int tl = 2 * it + (lane >> 5);
            uint32_t op = kCtlNop, pay = 0;  // past the end of the tree: nothing happens
            if (ty_r[it] != -1) {
                const Decoded dn = decode_node(ty_r[it], vl_r[it], MO, p.var_len, p.out_len);
                // control word: handler id | operands needed << 8 | (stack delta + 2) << 12
                const uint32_t need = dn.delta > 0 ? 0u : (uint32_t)(1 - dn.delta);
                op = dn.op | (need << 8) | ((uint32_t)(dn.delta + 2) << 12);
                pay = dn.pay;
            }
            op_s[jj * kLanePitch + tl] = op;
            pay_s[jj * kLanePitch + tl] = pay;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- interpret: elements 0 .. h-2 of the operand stack live in LDS, the top element in a register; the next
        // instruction is fetched before the current one executes ----
        const int n = maxlen - c0 < kLaneChunk ? maxlen - c0 : kLaneChunk;
        const int ln = lane < TPW ? lane : 0;  // lanes beyond the wave's trees idle (all-NOP) but must touch valid LDS
        const bool mine = lane < TPW;  // the other lanes run NOPs (their LDS traffic lands in the sink rows)
        uint32_t nctl = mine ? op_s[ln] : kCtlNop, npay = pay_s[ln];
        // ONE straight-line step for every node class.  The lanes of a wave sit on different classes; a branch per class
        // costs an LDS round trip and a page of mask bookkeeping each, and a lone wave issues one instruction per ~4
        // clocks, so the step's INSTRUCTION COUNT is the op's running time (first version: ~300 instructions, 1.4 k
        // clocks per step).  Here every lane forms one operand address (its variable, or the element under the top of
        // its stack), one push address and one accumulator address — classes that do not push / accumulate aim at a
        // sink row — and the classes are resolved with selects.  Validity (operand underflow, stack overflow) is
        // tracked as a running minimum / maximum and judged once at the end; addresses are clamped meanwhile.
        // Only divisions, the non-arithmetic functions and IF take a wave-uniform branch, when some lane needs them.
        float *const lds_f = stk;  // stk | var_s | out_s are contiguous
        constexpr int var_base = (kLaneDepth + 1) * TPW;
        const int out_base = var_base + p.var_len * TPW;
        for (int j = 0; j < n; ++j) {
            const uint32_t ctl = nctl, pay = npay;
            if (j + 1 < n) { nctl = mine ? op_s[(j + 1) * kLanePitch + ln] : kCtlNop; npay = pay_s[(j + 1) * kLanePitch + ln]; }
            const uint32_t op = ctl & 0xFFu;
            const int need = (int)((ctl >> 8) & 3u), delta = (int)((ctl >> 12) & 7u) - 2;
            hmin = min(hmin, h - need);
            const bool isleaf = op < H_ADD, isvar = op == H_VAR;
            int hw = h - 1; hw = hw < 0 ? 0 : (hw > kLaneDepth - 1 ? kLaneDepth - 1 : hw);
            int hr = h - 2; hr = hr < 0 ? 0 : (hr > kLaneDepth - 1 ? kLaneDepth - 1 : hr);
            lds_f[(isleaf ? hw : kLaneDepth) * TPW + ln] = tos;  // a leaf pushes the old top
            const float lda = lds_f[isvar ? var_base + (int)pay * TPW + ln : hr * TPW + ln];
            const bool hasout = MO && need != 0 && pay != kNoOut;
            const int io = out_base + (hasout ? (int)pay : p.out_len) * TPW + ln;
            float ldo = 0.0f, ldc = 0.0f;
            if (MO) ldo = lds_f[io];
            const bool anyter = __any(need == 3);
            if (anyter) { int hc = h - 3; hc = hc < 0 ? 0 : (hc > kLaneDepth - 1 ? kLaneDepth - 1 : hc); ldc = lds_f[hc * TPW + ln]; }
            const float x = tos, y = lda;
            float fr = x + y;
            fr = op == H_SUB ? x - y : fr;
            fr = op == H_MUL ? x * y : fr;
            if (__any(op == H_DIV)) { const float q = y == 0.0f ? __builtin_nanf("") : x / y; fr = op == H_DIV ? q : fr; }
            if (__any(need == 2 && op > H_DIV)) { if (need == 2 && op > H_DIV) fr = op_binary_other<false>(op, x, y); }
            if (__any(need == 1)) { if (need == 1) fr = op_unary<false>(op, x); }
            if (anyter) fr = need == 3 ? (x > 0.0f ? lda : ldc) : fr;
            const float leafval = isvar ? lda : bits2f(pay);
            float r;
            if (MO) {
                // a function node adds its value to its output and hands its LAST popped operand to its parent
                // (forward.cu:237-243): b for binary, the operand itself for unary, c for IF
                lds_f[io] = ldo + fr;
                r = need == 2 ? lda : tos;
                r = need == 3 ? ldc : r;
            } else {
                r = need == 0 ? tos : fr;
            }
            tos = isleaf ? leafval : r;
            h += delta;
            hmax = max(hmax, h);
        }
        __builtin_amdgcn_wave_barrier();  // the next chunk overwrites the staging area
    }
    if (active) {
        // one store path for all outcomes: a tree too deep for the LDS stack leaves the sentinel (redone by
        // eval_general_kernel), a malformed one NaN (forward.cu:298-301 asserts)
        float *res = p.results + (size_t)t * p.out_len;
        const bool deep = hmax > kLaneDepth + 1;
        const bool bad = len <= 0 || hmin < 0 || h != 1;
        const float nan = __builtin_nanf("");
        if (!MO) {
            res[0] = deep ? bits2f(kSentinelDeepEval) : (bad ? nan : tos);
        } else {
            for (int o = 0; o < p.out_len; ++o) {
                const float val = out_s[o * TPW + lane];
                res[o] = (deep && o == 0) ? bits2f(kSentinelDeepEval) : (bad ? nan : val);
            }
        }
    }
}

template <bool MO>
__global__ __launch_bounds__(64) void eval_general_kernel(EvalParams p, int only_marked) {
    const int lane = threadIdx.x & 63;
    float stk[kMaxStack + 2];
    float outs[MO ? kGeneralOuts : 1];
    auto process = [&](int t) {
        float *res = p.results + (size_t)t * p.out_len;
        const size_t row = (size_t)t * p.gp_len;
        const float *tv = p.value + row;
        const int16_t *tt = p.type + row;
        int len = uni((int)p.size[row]);
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
        const int cls = uni(classify_tree(tt, tv, len, MO, p.var_len, p.out_len, kMaxStack));
        if (cls != TREE_OK) {
            for (int o = lane; o < p.out_len; o += kWave) res[o] = __builtin_nanf("");
            return;
        }
        const float r = run_general<MO>(tt, tv, len, p.vars + (size_t)t * p.var_len, p.var_len, p.out_len, outs, stk);
        if (lane == 0) {
            if (!MO) res[0] = r;
            else for (int o = 0; o < p.out_len; ++o) res[o] = outs[o];
        }
    };
    if (!only_marked) {
        for (int t = blockIdx.x; t < p.pop; t += gridDim.x) process(t);
        return;
    }
    // behind the lane kernel: the wave looks at 64 result words at a time (one tree per lane) and redoes the marked ones
    for (int base = blockIdx.x * kWave; base < p.pop; base += gridDim.x * kWave) {
        const int t = base + lane;
        const bool marked = t < p.pop && f2bits(p.results[(size_t)t * p.out_len]) == kSentinelDeepEval;
        unsigned long long mask = __ballot(marked);
        while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            process(base + b);
        }
    }
}

template <bool MO>
static hipError_t launch_eval_general(const EvalParams &p, int only_marked, hipStream_t stream) {
    const DeviceInfo &dev = device_info();
    long blocks = (long)dev.num_cus * (only_marked ? 2 : 16);  // behind the lane kernel only the (rare) deep trees are left
    if (blocks > p.pop) blocks = p.pop;
    hipLaunchKernelGGL(eval_general_kernel<MO>, dim3((unsigned)blocks), dim3(64), 0, stream, p, only_marked);
    return hipGetLastError();
}

template <bool MO, int TPW>
static hipError_t launch_eval_lane_tpw(const EvalParams &p, hipStream_t stream) {
    const size_t lds = (size_t)(2 * kLaneChunk * (TPW + 1) + (kLaneDepth + 1 + p.var_len + (MO ? p.out_len + 1 : 0)) * TPW) * 4;
    const unsigned blocks = (unsigned)((p.pop + TPW - 1) / TPW);
    hipLaunchKernelGGL((eval_lane_kernel<MO, TPW>), dim3(blocks), dim3(64), lds, stream, p);
    return hipGetLastError();
}

template <bool MO>
static hipError_t launch_eval_lane(const EvalParams &p, hipStream_t stream) {
    // trees per wave: 16 until the population exceeds what the chip holds at once (~8 k waves), then 32 / 64
    static const int forced = [] { const char *e = getenv("EVOGP_EVAL_TPW"); return e ? atoi(e) : 0; }();
    int tpw = p.pop <= 16 * 8192 ? 16 : (p.pop <= 32 * 8192 ? 32 : 64);
    if (forced == 16 || forced == 32 || forced == 64) tpw = forced;
    hipError_t e = tpw == 16 ? launch_eval_lane_tpw<MO, 16>(p, stream)
                 : tpw == 32 ? launch_eval_lane_tpw<MO, 32>(p, stream) : launch_eval_lane_tpw<MO, 64>(p, stream);
    if (e != hipSuccess) return e;
    return launch_eval_general<MO>(p, 1, stream);
}

// the trees a previous kernel marked (kSentinelDeepEval in the first result word) through the stack interpreter
// (evaluate_prepared.hip leaves what its operation lists cannot express to it)
hipError_t launch_eval_marked_general(unsigned pop, unsigned gp_len, unsigned var_len, unsigned out_len, const float *value,
                                      const int16_t *type, const int16_t *size, const float *vars, float *results, hipStream_t stream) {
    EvalParams p{value, type, size, vars, results, (int)pop, (int)gp_len, (int)var_len, (int)out_len};
    return out_len > 1 ? launch_eval_general<true>(p, 1, stream) : launch_eval_general<false>(p, 1, stream);
}

hipError_t launch_eval_direct(unsigned pop, unsigned gp_len, unsigned var_len, unsigned out_len, const float *value, const int16_t *type,
                              const int16_t *size, const float *vars, float *results, hipStream_t stream);

} // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_evaluate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                  const float *value, const int16_t *type, const int16_t *size,
                                  const float *variables, float *results, evogp_stream_t stream_) {
    // argument contract of torch_wrapper.cu:205-208
    if (pop_size == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len == 0) return EVOGP_E_BADARG;
    if (!value || !type || !size || !variables || !results) return EVOGP_E_NULLPTR;
    if (out_len > (unsigned)kGeneralOuts) return EVOGP_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    EvalParams p{value, type, size, variables, results, (int)pop_size, (int)gp_len, (int)var_len, (int)out_len};
    const bool mo = out_len > 1;
    // multi-output trees: every OUT node at once, one wave per tree (evaluate_prepared.hip eval_direct_kernel; EVOGP_EVAL_DIRECT=0: the
    // lane-per-tree stack interpreter below)
    static const bool direct = [] { const char *e = getenv("EVOGP_EVAL_DIRECT"); return !e || atoi(e) != 0; }();
    if (mo && direct && var_len <= 64 && out_len <= 64)
        return (int)launch_eval_direct(pop_size, gp_len, var_len, out_len, value, type, size, variables, results, stream);
    if (var_len > 64 || out_len > 32)  // the transposed input rows / accumulators would not fit the wave's LDS budget
        return (int)(mo ? launch_eval_general<true>(p, 0, stream) : launch_eval_general<false>(p, 0, stream));
    return (int)(mo ? launch_eval_lane<true>(p, stream) : launch_eval_lane<false>(p, stream));
}
