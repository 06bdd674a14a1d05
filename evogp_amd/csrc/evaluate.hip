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
constexpr int kLaneChunk = 32;
constexpr int kLanePitch = 65;
constexpr int kLaneDepth = 24;

template <bool MO>
__global__ __launch_bounds__(64) void eval_lane_kernel(EvalParams p) {
    extern __shared__ uint32_t lane_lds[];
    uint32_t *op_s = lane_lds;                                 // [kLaneChunk][kLanePitch]
    uint32_t *pay_s = op_s + kLaneChunk * kLanePitch;          // [kLaneChunk][kLanePitch]
    float *stk = (float *)(pay_s + kLaneChunk * kLanePitch);   // [kLaneDepth][64]
    float *var_s = stk + kLaneDepth * 64;                      // [var_len][64]
    float *out_s = var_s + p.var_len * 64;                     // [out_len][64]  (multi-output only)
    const int lane = threadIdx.x;
    const int t0 = blockIdx.x * 64;
    const int t = t0 + lane;
    const bool active = t < p.pop;
    int len = 0;
    if (active) {
        len = (int)p.size[(size_t)t * p.gp_len];
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
    }
    // input rows: the 64 rows of this wave are contiguous in memory; copy them coalesced, store transposed
    const int nrows = p.pop - t0 < 64 ? p.pop - t0 : 64;
    for (int e = lane; e < nrows * p.var_len; e += 64) {
        const int r = e / p.var_len, v = e - r * p.var_len;
        var_s[v * 64 + r] = p.vars[(size_t)t0 * p.var_len + e];
    }
    if (MO) for (int o = 0; o < p.out_len; ++o) out_s[o * 64 + lane] = 0.0f;
    const int maxlen = wave_max(len);
    int h = 0;
    float tos = 0.0f;
    bool bad = len <= 0, deep = false;
    for (int c0 = 0; c0 < maxlen; c0 += kLaneChunk) {
        // ---- stage the next 32 instructions of every tree: all loads first (64 in flight), then decode + store ----
        const int jj = lane & 31;
        const int k = c0 + jj;
        int ty_r[32];
        float vl_r[32];
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int tl = 2 * it + (lane >> 5);
            const int len_t = __shfl(len, tl);
            ty_r[it] = -1; vl_r[it] = 0.0f;
            if (k < len_t) {
                const size_t at = (size_t)(t0 + tl) * p.gp_len + (size_t)(len_t - 1 - k);
                ty_r[it] = (int)p.type[at]; vl_r[it] = p.value[at];
            }
        }
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int tl = 2 * it + (lane >> 5);
            uint32_t op = 0xFFFFFFFFu, pay = 0;
            if (ty_r[it] != -1) {
                const Decoded dn = decode_node(ty_r[it], vl_r[it], MO, p.var_len, p.out_len);
                op = dn.op; pay = dn.pay;
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
        uint32_t nop = op_s[lane], npay = pay_s[lane];
        for (int j = 0; j < n; ++j) {
            const uint32_t op = nop, pay = npay;
            if (j + 1 < n) { nop = op_s[(j + 1) * kLanePitch + lane]; npay = pay_s[(j + 1) * kLanePitch + lane]; }
            if (c0 + j >= len || bad || deep) continue;
            if (op < H_ADD) {  // leaf: push
                if (h > kLaneDepth) { deep = true; continue; }
                if (h >= 1) stk[(h - 1) * 64 + lane] = tos;
                tos = op == H_CONST ? bits2f(pay) : var_s[pay * 64 + lane];
                ++h;
            } else if (op < H_UN) {  // binary: a = top (left operand), b = next (right operand)
                if (h < 2) { bad = true; continue; }
                const float a = tos, b = stk[(h - 2) * 64 + lane];
                float r;
                if (op <= H_DIV) r = op == H_ADD ? a + b : op == H_SUB ? a - b : op == H_MUL ? a * b : (b == 0.0f ? __builtin_nanf("") : a / b);
                else r = op_binary_other<false>(op, a, b);
                --h;
                if (MO) {
                    if (pay != kNoOut) out_s[pay * 64 + lane] += r;
                    r = b;  // a function node hands its LAST popped operand to its parent (forward.cu:237-243)
                }
                tos = r;
            } else if (op < H_IF) {  // unary
                if (h < 1) { bad = true; continue; }
                const float r = op_unary<false>(op, tos);
                if (MO) { if (pay != kNoOut) out_s[pay * 64 + lane] += r; }
                else tos = r;
            } else {  // ternary IF: cond = top, then = next, else = third
                if (h < 3) { bad = true; continue; }
                const float b = stk[(h - 2) * 64 + lane], c = stk[(h - 3) * 64 + lane];
                float r = tos > 0.0f ? b : c;
                h -= 2;
                if (MO) { if (pay != kNoOut) out_s[pay * 64 + lane] += r; r = c; }
                tos = r;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next chunk overwrites the staging area
    }
    if (!active) return;
    float *res = p.results + (size_t)t * p.out_len;
    if (deep) { res[0] = bits2f(kSentinelDeepEval); return; }        // redone by eval_general_kernel
    if (bad || h != 1) { for (int o = 0; o < p.out_len; ++o) res[o] = __builtin_nanf(""); return; }  // forward.cu:298-301 asserts
    if (!MO) res[0] = tos;
    else for (int o = 0; o < p.out_len; ++o) res[o] = out_s[o * 64 + lane];
}

template <bool MO>
__global__ __launch_bounds__(64) void eval_general_kernel(EvalParams p, int only_marked) {
    const int lane = threadIdx.x & 63;
    float stk[kMaxStack + 2];
    float outs[MO ? kGeneralOuts : 1];
    for (int t = blockIdx.x; t < p.pop; t += gridDim.x) {
        float *res = p.results + (size_t)t * p.out_len;
        if (only_marked && uni(f2bits(res[0])) != kSentinelDeepEval) continue;
        const size_t row = (size_t)t * p.gp_len;
        const float *tv = p.value + row;
        const int16_t *tt = p.type + row;
        int len = uni((int)p.size[row]);
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
        const int cls = uni(classify_tree(tt, tv, len, MO, p.var_len, p.out_len, kMaxStack));
        if (cls != TREE_OK) {
            for (int o = lane; o < p.out_len; o += kWave) res[o] = __builtin_nanf("");
            continue;
        }
        const float r = run_general<MO>(tt, tv, len, p.vars + (size_t)t * p.var_len, p.var_len, p.out_len, outs, stk);
        if (lane == 0) {
            if (!MO) res[0] = r;
            else for (int o = 0; o < p.out_len; ++o) res[o] = outs[o];
        }
    }
}

template <bool MO>
static hipError_t launch_eval_general(const EvalParams &p, int only_marked, hipStream_t stream) {
    const DeviceInfo &dev = device_info();
    long blocks = (long)dev.num_cus * 16;
    if (blocks > p.pop) blocks = p.pop;
    hipLaunchKernelGGL(eval_general_kernel<MO>, dim3((unsigned)blocks), dim3(64), 0, stream, p, only_marked);
    return hipGetLastError();
}

template <bool MO>
static hipError_t launch_eval_lane(const EvalParams &p, hipStream_t stream) {
    const size_t lds = (size_t)(2 * kLaneChunk * kLanePitch + kLaneDepth * 64 + p.var_len * 64 + (MO ? p.out_len * 64 : 0)) * 4;
    const unsigned blocks = (unsigned)((p.pop + 63) / 64);
    hipLaunchKernelGGL((eval_lane_kernel<MO>), dim3(blocks), dim3(64), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_eval_general<MO>(p, 1, stream);
}

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
    if (var_len > 64 || out_len > 32)  // the transposed input rows / accumulators would not fit the wave's LDS budget
        return (int)(mo ? launch_eval_general<true>(p, 0, stream) : launch_eval_general<false>(p, 0, stream));
    return (int)(mo ? launch_eval_lane<true>(p, stream) : launch_eval_lane<false>(p, stream));
}
