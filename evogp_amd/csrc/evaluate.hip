// evaluate.hip — population forward pass with one input row per tree (gfx950).
//
// Replaces  evaluate / treeGPEvalKernel  (src/evogp/cuda/forward.cu:304-371):
//     results[n][:] = tree_n(variables[n][:])
// The reference runs one THREAD per tree (divergent dispatch, 8 KB of local memory per thread,
// row-strided loads).  Here one WAVE interprets one tree with the wave-uniform register-stack
// interpreter of interp.hpp: the tree row is loaded coalesced (one node per lane), decode is scalar,
// there is no divergence and no memory traffic in the inner loop.  All lanes evaluate the same input
// row (the op is latency/launch bound at its real sizes — 50 k trees x 17 inputs per call in the
// policy-rollout config — so idle lanes cost nothing measurable).
//
// Trees whose operand stack exceeds the register stack are marked and redone by a scratch-stack
// kernel launched behind the fast one; malformed trees yield NaN rows.
#include "interp.hpp"
#include "launch.hpp"

namespace evogp {

constexpr uint32_t kSentinelDeepEval = 0x7FC0DEEDu;
constexpr int kEvalDepth = 32;

struct EvalParams {
    const float *value;
    const int16_t *type;
    const int16_t *size;
    const float *vars;  // [pop][var_len]
    float *results;     // [pop][out_len]
    int pop, gp_len, var_len, out_len;
};

template <int VL, bool MO>
__global__ __launch_bounds__(256) void eval_fast_kernel(EvalParams p) {
    using VARS = typename VecOf<VL>::type;
    const int lane = threadIdx.x & 63;
    const int wave0 = uni((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * 4;
    for (int t = wave0; t < p.pop; t += nwaves) {
        const size_t row = (size_t)t * p.gp_len;
        const float *tv = p.value + row;
        const int16_t *tt = p.type + row;
        float *res = p.results + (size_t)t * p.out_len;
        int len = uni((int)p.size[row]);
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
        const int cls = uni(classify_tree(tt, tv, len, MO, p.var_len, p.out_len, kEvalDepth));
        if (cls != TREE_OK) {
            if (cls == TREE_DEEP) { if (lane == 0) res[0] = bits2f(kSentinelDeepEval); }
            else for (int o = lane; o < p.out_len; o += kWave) res[o] = __builtin_nanf("");
            continue;
        }
        VARS vars[1];
        const float *xr = p.vars + (size_t)t * p.var_len;
#pragma unroll
        for (int v = 0; v < VL; ++v) vars[0][v] = v < p.var_len ? xr[v] : 0.0f;
        v16f outs[1];
        if (MO) {
#pragma unroll
            for (int o = 0; o < kMaxOutRegs; ++o) outs[0][o] = 0.0f;
        }
        RegStack<1, kEvalDepth> st;
        st.h = 0;
        st.tos[0] = 0.0f;
        for (int base = 0; base < len; base += kWave) {
            const int r = base + lane;
            uint32_t opv = 0, payv = 0;
            if (r < len) {
                const int i = len - 1 - r;
                const Decoded dn = decode_node(tt[i], tv[i], MO, p.var_len, p.out_len);
                opv = dn.op; payv = dn.pay;
            }
            const int n = len - base < kWave ? len - base : kWave;
            run_chunk<MO, false, 1, kEvalDepth, VL>(opv, payv, n, st, vars, outs);
        }
        if (!MO) {
            if (lane == 0) res[0] = st.tos[0];
        } else if (lane == 0) {
#pragma unroll
            for (int o = 0; o < kMaxOutRegs; ++o)
                if (o < p.out_len) res[o] = outs[0][o];
        }
    }
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

template <int VL, bool MO>
static hipError_t launch_eval_fast(const EvalParams &p, hipStream_t stream) {
    const DeviceInfo &dev = device_info();
    long blocks = (long)dev.num_cus * 8; // 8 x 4 waves = a full CU
    const long need = ((long)p.pop + 3) / 4;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL((eval_fast_kernel<VL, MO>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
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
    if (var_len > 32 || out_len > (unsigned)kMaxOutRegs)
        return (int)(mo ? launch_eval_general<true>(p, 0, stream) : launch_eval_general<false>(p, 0, stream));
    if (var_len <= 16) return (int)(mo ? launch_eval_fast<16, true>(p, stream) : launch_eval_fast<16, false>(p, stream));
    return (int)(mo ? launch_eval_fast<32, true>(p, stream) : launch_eval_fast<32, false>(p, stream));
}
