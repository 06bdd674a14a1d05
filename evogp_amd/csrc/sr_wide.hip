// sr_wide.hip — batch evaluation for WIDE inputs and the fused classification epilogue (gfx950).
//
// results[t][d][:] = tree_t(X[d][:])  for every tree t and every row d of a shared dataset, like
// evogp_hip_batch_evaluate (src/evogp/tree/forest.py:143-176), for the shapes the register kernels of
// sr_fitness.hip cannot keep resident: more than 32 variables, or a dataset too large for one workgroup (the classifier
// config: 1797 rows x 64 variables = 460 KB).  The dataset is cut into GROUPS of W tiles of 64 rows; a workgroup owns
// one group for its whole life: wave w stages its tile's rows into LDS as [variable][lane] (any number of variables up
// to the LDS size) and then walks the workgroup's share of the trees with the wave-uniform register-stack interpreter
// of interp.hpp (variables are read from LDS instead of a register tuple).  Waves never synchronise: there is nothing
// to reduce across rows.  Workgroup (g, i) takes trees i, i + n, i + 2n, ... on group g.
//
// MODE 0 stores the outputs; MODE 1 is the epilogue of the Classification problem (src/evogp/problem/classification.py
// :62-75): per (tree, row) the arg-max of the outputs is compared with the row's label and only the COUNT of matches
// leaves the chip (one integer atomic per tree and wave) — the (pop, D, classes) tensor (14.4 GB at pop 200 k) is never
// written.  The arg-max follows torch.argmax(clip(softmax(x))): a NaN or an infinite maximum makes the soft-max row
// all-NaN, whose arg-max is index 0; otherwise the first maximum wins.
//
// Trees whose operand stack exceeds the register stack are marked (MODE 0: sentinel in results[t][0][0], redone by
// sr_general_kernel; MODE 1: counted by a scratch-stack fallback inside this kernel's slow path below).
#include "interp.hpp"
#include "launch.hpp"
#include "sr_params.hpp"

namespace evogp {

struct WideParams {
    const float *value;
    const int16_t *type;
    const int16_t *size;
    const float *X;        // [D][var_len]
    float *results;        // MODE 0: [pop][D][out_len]
    const int *labels;     // MODE 1: [D]
    unsigned *counts;      // MODE 1: [pop], zeroed before the launch
    unsigned *marks;       // pending-marks flags (MODE 0)
    int pop, D, gp_len, var_len, out_len;
    int ngroups, workers;  // grid = ngroups * workers workgroups
};

constexpr int kWideDepth = 32;

template <bool MO, int MODE>
__global__ __launch_bounds__(512) void sr_wide_kernel(WideParams p) {
    extern __shared__ float wide_lds[];  // [wave][variable][lane]
    const int lane = threadIdx.x & 63;
    const int w = uni((int)(threadIdx.x >> 6));
    const int W = blockDim.x >> 6;
    const int group = blockIdx.x % p.ngroups, worker = blockIdx.x / p.ngroups;
    const int d = (group * W + w) * 64 + lane;          // this lane's row
    const int dc = d < p.D ? d : p.D - 1;
    const bool tile_live = (group * W + w) * 64 < p.D;  // a wave whose tile lies past the dataset has nothing to do
    float *mine = wide_lds + (size_t)w * p.var_len * 64;
    for (int v = 0; v < p.var_len; ++v) mine[v * 64 + lane] = p.X[(size_t)dc * p.var_len + v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!tile_live) return;
    const LdsVars vars{mine + lane};
    const int label = MODE == 1 ? p.labels[dc] : 0;

    for (int t = worker; t < p.pop; t += p.workers) {
        const size_t row = (size_t)t * p.gp_len;
        const float *tv = p.value + row;
        const int16_t *tt = p.type + row;
        int len = uni((int)p.size[row]);
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
        const int cls = uni(classify_tree(tt, tv, len, MO, p.var_len, p.out_len, kWideDepth));
        float *res = MODE == 0 ? p.results + ((size_t)t * p.D + d) * p.out_len : nullptr;
        if (cls != TREE_OK) {
            if (MODE == 0) {
                if (cls == TREE_DEEP) {  // redone by sr_general_kernel (every group writes the same mark)
                    if (lane == 0) { p.results[(size_t)t * p.D * p.out_len] = bits2f(kSentinelDeep); if (p.marks) p.marks[1] = 1u; }
                } else if (d < p.D) {
                    for (int o = 0; o < p.out_len; ++o) res[o] = __builtin_nanf("");
                }
                continue;
            }
            if (cls != TREE_DEEP) {
                // malformed tree: all outputs NaN -> arg-max 0
                const unsigned long long hit = __ballot(d < p.D && label == 0);
                if (lane == 0 && hit) atomicAdd(p.counts + t, (unsigned)__popcll(hit));
                continue;
            }
        }
        v16f outs[1];
        float top;
        if (MODE == 0 || cls == TREE_OK) {
            if (MO) {
#pragma unroll
                for (int o = 0; o < kMaxOutRegs; ++o) outs[0][o] = 0.0f;
            }
            RegStack<1, kWideDepth> st;
            st.h = 0;
            st.tos[0] = 0.0f;
            for (int base = 0; base < len; base += kWave) {
                const int r = base + lane;
                uint32_t opv = 0, payv = 0;
                if (r < len) {
                    const Decoded dn = decode_node(tt[len - 1 - r], tv[len - 1 - r], MO, p.var_len, p.out_len);
                    opv = dn.op; payv = dn.pay;
                }
                const int n = len - base < kWave ? len - base : kWave;
                run_chunk<MO, false, 1, kWideDepth>(opv, payv, n, st, vars, outs);
            }
            top = st.tos[0];
        } else {
            // MODE 1, deep tree: scratch-stack interpreter on this lane's row (rare)
            float stk[kMaxStack + 2];
            float o16[MO ? kMaxOutRegs : 1];
            top = run_general<MO>(tt, tv, len, p.X + (size_t)dc * p.var_len, p.var_len, p.out_len, o16, stk);
            if (MO) {
#pragma unroll
                for (int o = 0; o < kMaxOutRegs; ++o) outs[0][o] = o < p.out_len ? o16[o] : 0.0f;
            }
        }
        if (MODE == 0) {
            if (d < p.D) {
                if (!MO) res[0] = top;
                else {
#pragma unroll
                    for (int o = 0; o < kMaxOutRegs; ++o)
                        if (o < p.out_len) res[o] = outs[0][o];
                }
            }
        } else {
            // arg-max as torch.argmax(clip(softmax(x))) sees it
            int best = 0;
            float m = outs[0][0];
            bool poisoned = m != m;
#pragma unroll
            for (int o = 1; o < kMaxOutRegs; ++o) {
                if (o < p.out_len) {
                    const float x = outs[0][o];
                    poisoned |= x != x;
                    if (x > m) { m = x; best = o; }
                }
            }
            if (poisoned || __builtin_isinf(m)) best = 0;
            const unsigned long long hit = __ballot(d < p.D && best == label);
            if (lane == 0 && hit) atomicAdd(p.counts + t, (unsigned)__popcll(hit));
        }
    }
}

template <bool MO, int MODE>
static hipError_t launch_wide(WideParams p, hipStream_t stream) {
    const DeviceInfo &dev = device_info();
    // waves per workgroup: as many tiles as the LDS of a CU holds (<= 8), never more than the dataset has
    const int tiles = (p.D + 63) / 64;
    const size_t per_wave = (size_t)p.var_len * 64 * 4;
    int W = (int)((dev.lds_per_cu - 2048) / per_wave);
    W = W > 8 ? 8 : W;
    W = W > tiles ? tiles : W;
    if (W < 1) return hipErrorInvalidValue;
    p.ngroups = (tiles + W - 1) / W;
    int workers = dev.num_cus / p.ngroups;
    workers = workers < 1 ? 1 : workers;
    // small inputs leave LDS for more than one workgroup per CU
    const int per_cu = (int)((dev.lds_per_cu - 2048) / (per_wave * W));
    workers *= per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    if (workers > p.pop) workers = p.pop;
    p.workers = workers;
    const size_t lds = per_wave * W;
    auto kern = sr_wide_kernel<MO, MODE>;
    static bool attr_done = false;
    if (!attr_done || lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { (void)hipGetLastError(); if (lds > 64 * 1024) return e; }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.ngroups * p.workers)), dim3(W * 64), lds, stream, p);
    return hipGetLastError();
}

// called by run_population<STORE> (sr_fitness.hip) for inputs the register kernels do not take
hipError_t launch_wide_store(const SrParams &s, hipStream_t stream) {
    WideParams p{};
    p.value = s.value; p.type = s.type; p.size = s.size; p.X = s.X; p.results = s.results; p.marks = s.marks;
    p.pop = s.pop; p.D = s.D; p.gp_len = s.gp_len; p.var_len = s.var_len; p.out_len = s.out_len;
    return s.out_len > 1 ? launch_wide<true, 0>(p, stream) : launch_wide<false, 0>(p, stream);
}

} // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_batch_argmax_count(unsigned pop_size, unsigned data_points, unsigned gp_len, unsigned var_len,
                                            unsigned out_len, const float *value, const int16_t *type, const int16_t *size,
                                            const float *variables, const int *labels, unsigned *counts,
                                            evogp_stream_t stream_) {
    if (pop_size == 0 || data_points == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len < 2)
        return EVOGP_E_BADARG;
    if (!value || !type || !size || !variables || !labels || !counts) return EVOGP_E_NULLPTR;
    if (out_len > (unsigned)kMaxOutRegs || (size_t)var_len * 256 > 150 * 1024) return EVOGP_E_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)pop_size * sizeof(unsigned), stream);
    if (e != hipSuccess) return (int)e;
    WideParams p{};
    p.value = value; p.type = type; p.size = size; p.X = variables; p.labels = labels; p.counts = counts;
    p.pop = (int)pop_size; p.D = (int)data_points; p.gp_len = (int)gp_len; p.var_len = (int)var_len; p.out_len = (int)out_len;
    return (int)launch_wide<true, 1>(p, stream);
}
