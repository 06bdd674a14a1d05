// sr_fitness.hip — fused symbolic-regression fitness for a whole population (gfx950).
//
// Replaces the reference's  SR_fitness  dispatcher and its four kernel strategies
// (src/evogp/cuda/forward.cu:402-479, 481-549, 551-692, 694-825, 827-856) with ONE fused kernel:
//
//   fitness[t] = (1/D) * sum_d sum_o err(labels[d][o] - tree_t(X[d])_o)        err = square | abs
//
// Work decomposition ("tile-resident" persistent workgroups):
//   * the D datapoints are cut into tiles of 64*K rows; wave w of a workgroup OWNS tile w for its
//     whole life: the lane's K input rows and labels are loaded from HBM once into VGPRs and never
//     touched again (the dataset is read once per workgroup, not once per tree);
//   * the workgroup pulls batches of B consecutive trees from a global atomic counter (dynamic
//     load balance: tree lengths vary 1..gp_len).  For each batch the waves first split the B
//     trees between them to classify them (valid? operand stack <= DEPTH?), then EVERY wave
//     interprets all B trees on its own tile: it loads the tree coalesced (one node per lane),
//     pre-decodes it into two VGPRs and runs the wave-uniform register-stack interpreter
//     (interp.hpp).  All waves of a workgroup execute the same instruction stream on different
//     rows, so they reach the batch barrier together;
//   * per tree each wave reduces its 64*K errors with a fixed butterfly and parks the partial sum
//     in LDS; after the batch barrier one thread per tree adds the partials in tile order and
//     writes the mean — no memset of the output, no float atomics, no second "average" kernel,
//     bit-reproducible from run to run.
//
// HBM traffic per launch = the live prefixes of value/type (6 B per node), size[t][0], 4 B of
// fitness per tree, the dataset once per workgroup: the algorithmic bytes of SURVEY.md §8d (trees
// are re-read once per tile from L2).  The kernel is bound by instruction issue (scalar decode +
// VALU), not by HBM; see DESIGN.md.
//
// Trees the register path cannot take (operand stack deeper than DEPTH) are marked with a sentinel
// NaN and evaluated by sr_general_kernel — a wave-per-tree interpreter with its stack in scratch
// memory — launched right behind the fast kernel on the same stream; it only touches marked trees.
// Configurations the register path cannot take at all (more than 32 variables, more than 16
// outputs) run entirely on the general kernel.  Malformed trees (stack underflow, final height
// != 1 — the reference asserts, forward.cu:298-301) get a NaN fitness.
#include "interp.hpp"
#include <cstdio>
#include "launch.hpp"
#include <mutex>
#include <vector>
#include "sr_params.hpp"

namespace evogp {

static unsigned long long *g_stats = nullptr; // set by evogp_hip_debug_set_stats

// Per-stage timing of tree_SR_fitness calls (evogp_hip_debug_profile): four events per call on the launch stream --
// before the call, between the program compiler and the interpreter, behind the interpreter, behind the follow-up kernels.
struct ProfCall {
    hipEvent_t ev[4];
    bool mid;  // ev[1] was recorded (the threaded-code path took the call)
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static bool g_prof_skip_followups = false;   // evogp_hip_debug_profile(2): the call stops behind the threaded code (marked trees keep their sentinel words)
static std::vector<ProfCall> g_prof;

// flag word `i` of the call's scratch block, over the chunks the threaded code cut the population into
__device__ inline bool marks_pending(const SrParams &p, int i) {
    unsigned any = 0u;
    const int n = p.mark_chunks > 1 ? p.mark_chunks : 1;
    for (int c = 0; c < n; ++c) any |= p.marks[(size_t)c * kCallScratchChunkWords + i];
    return uni((int)any) != 0;
}

__device__ inline float err_term(float diff, int use_mse) { return use_mse ? diff * diff : fabsf(diff); }

// The fitness of ONE single-output tree with the operand stack in scratch memory (sr_general_kernel's per-tree work): one wave, lanes are
// datapoints.  sr_fast_kernel's build behind the threaded code runs it itself for the rare tree whose stack its registers do not hold
// (only_marked == 5, round 5: the scratch-stack kernel is then not launched at all -- two launches that find nothing were 10 us of every call).
constexpr int kFoldMaxLen = 126;   // (gp_len + 2) * 256 B of LDS for the folded deep path: 32 KB
template <int STRIDE>
__device__ __attribute__((always_inline)) inline void general_tree_fitness(const SrParams &p, int t, float *stk) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)t * p.gp_len;
    const float *tv = p.value + row;
    const int16_t *tt = p.type + row;
    int len = uni((int)p.size[row]);
    len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
    const int cls = uni(classify_tree(tt, tv, len, false, p.var_len, p.out_len, kMaxStack));
    if (cls != TREE_OK) {
        if (lane == 0) p.fitness[t] = __builtin_nanf("");
        return;
    }
    float acc = 0.0f, outs[1];
    for (int base = 0; base < p.D; base += kWave) {
        const int d = base + lane;
        const int dc = d < p.D ? d : p.D - 1;
        const float res = run_general<false, STRIDE>(tt, tv, len, p.X + (size_t)dc * p.var_len, p.var_len, p.out_len, outs, stk);
        const float e = err_term(p.y[dc] - res, p.use_mse);
        acc += d < p.D ? e : 0.0f;
    }
    const float total = wave_sum(acc);
    if (lane == 0) p.fitness[t] = total / (float)p.D;
}

// K rows per lane, DEPTH-entry register stack, VL variable registers, MO = multi-output,
// MAXW = waves per workgroup the launch bound allows.
// The LEAN build is held to 128 VGPRs (4 resident waves per SIMD): the interpreter is latency bound,
// throughput scales with K x resident waves.
#ifndef EVOGP_LEAN_WAVES
#define EVOGP_LEAN_WAVES 4
#endif
template <int K, int DEPTH, int VL, bool MO, int MAXW, bool STORE, bool LEAN>
__global__ __launch_bounds__(MAXW * 64, (LEAN && K == 4 && VL <= 16) ? EVOGP_LEAN_WAVES : 1) void sr_fast_kernel(SrParams p) {
    __shared__ float part[2][kMaxBatch][kMaxWaves]; // per-tree partial sums, double-buffered by batch parity
    __shared__ int cls_s[2][kMaxBatch];
    __shared__ int next_s[2];

    if (p.only_marked && p.marks && !marks_pending(p, 0) && !(p.only_marked == 5 && marks_pending(p, 4))) return;  // nothing was marked for this build
    // Behind another kernel the batch size follows the share of marked trees the marking kernel sampled: few marked
    // trees -> full 64-tree batches (the work is one mark word per lane), many -> the launcher's load-balancing size.
    int batch = p.batch;
    if (p.only_marked && p.marks && p.mark_sample > 0 && uni((int)p.marks[2]) * 8 < p.mark_sample) batch = kMaxBatch;
    using VARS = typename VecOf<VL>::type;
    constexpr int TILE = kWave * K;
    const int lane = threadIdx.x & 63;
    const int w = uni((int)(threadIdx.x >> 6));
    const int W = blockDim.x >> 6;
    // a wave owns tile w; when there are more tiles than waves it also takes w+W, w+2W, ...
    // (rows are then re-loaded per tree — the slow corner, only for D > 64*K*MAXW)
    const bool single = p.ntiles <= W;

    VARS vars[K];
    float yv[K];
    int d[K], dc[K];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            d[k] = tile * TILE + k * kWave + lane;
            dc[k] = d[k] < p.D ? d[k] : p.D - 1;
            const float *xr = p.X + (size_t)dc[k] * p.var_len;
#pragma unroll
            for (int v = 0; v < VL; ++v) vars[k][v] = v < p.var_len ? xr[v] : 0.0f;
            yv[k] = (MO || STORE) ? 0.0f : p.y[dc[k]];
        }
    };
    if (single) load_tile(w < p.ntiles ? w : 0);

    // Two ways to get a batch.  A full pass pulls batches of consecutive trees from the global counter.  A pass behind
    // another kernel (only_marked) looks for the few trees that kernel marked: a global batch counter would cost one
    // same-address atomic per 64 UNMARKED trees (~11 ns each, serialised over the chip: 0.2 ms per million trees) --
    // instead every workgroup owns a contiguous chunk of the population, reads its mark words coalesced, collects the
    // hits in an LDS queue and works the queue off in batches.  No global atomics, no counter.
    const bool marked = p.only_marked != 0;
    constexpr int SCAN = 4;  // mark words per thread and scan step
    __shared__ int q_s[kMaxBatch + SCAN * MAXW * 64];
    __shared__ int tid_s[2][kMaxBatch];
    __shared__ int qn_s;
    int scan = 0, scan_end = 0;
    if (marked) {
        const int chunk = ((p.pop + (int)gridDim.x - 1) / (int)gridDim.x + 63) & ~63;
        scan = (int)blockIdx.x * chunk < p.pop ? (int)blockIdx.x * chunk : p.pop;
        scan_end = scan + chunk < p.pop ? scan + chunk : p.pop;
        if (threadIdx.x == 0) qn_s = 0;
    } else if (threadIdx.x == 0) next_s[0] = (int)atomicAdd(p.counter, (unsigned)batch);
    __syncthreads();
    int par = 0;
    for (;;) {
        int t0 = 0, nb = 0;
        if (!marked) {
            t0 = uni(next_s[par]);
            if (t0 >= p.pop) break;
            nb = p.pop - t0 < batch ? p.pop - t0 : batch;
            if (threadIdx.x == 0) next_s[par ^ 1] = (int)atomicAdd(p.counter, (unsigned)batch); // prefetch
        } else {
            int qn;
            for (;;) {
                qn = uni(qn_s);
                __syncthreads();  // everybody has read the count before anybody adds to it
                if (qn >= batch || scan >= scan_end) break;
#pragma unroll
                for (int j = 0; j < SCAN; ++j) {
                    const int t = scan + j * (int)blockDim.x + (int)threadIdx.x;
                    if (t < scan_end) {
                        const float *mark = STORE ? p.results + (size_t)t * p.D * p.out_len : p.fitness + t;
                        const uint32_t mw = f2bits(*mark);   // (only_marked 5: the ONLY follow-up behind the threaded code -- a tree left for a general compiler that was not launched is this kernel's too)
                        if (mw == kSentinelHeavy || (p.only_marked == 5 && mw == kSentinelGeneral)) q_s[atomicAdd(&qn_s, 1)] = t;
                    }
                }
                scan += SCAN * (int)blockDim.x;
                __syncthreads();
            }
            if (qn == 0) break;  // chunk exhausted, queue empty
            nb = qn < batch ? qn : batch;
            if ((int)threadIdx.x < nb) tid_s[par][threadIdx.x] = q_s[qn - nb + (int)threadIdx.x];  // the tail of the queue
            if (threadIdx.x == 0) qn_s = qn - nb;
            __syncthreads();
        }
        auto tree_of = [&](int b) -> int { return marked ? tid_s[par][b] : t0 + b; };

        // ---- phase 1: classify the batch, trees split between the waves ----
        for (int b = w; b < nb; b += W) {
            const size_t row = (size_t)uni(tree_of(b)) * p.gp_len;
            int len = uni((int)p.size[row]);
            len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
            const int c = classify_tree(p.type + row, p.value + row, len, MO, p.var_len, p.out_len, DEPTH, LEAN ? 1 : 0);
            if (lane == 0) cls_s[par][b] = c;
        }
        __syncthreads();

        // ---- phase 2: every wave interprets every tree of the batch on its own rows ----
        for (int b = 0; b < nb; ++b) {
            const int cls_b = uni(cls_s[par][b]);
            const int tb = uni(tree_of(b));
            if (cls_b != TREE_OK) {
                if (STORE && cls_b == TREE_BAD) { // malformed tree: the whole result row is NaN
                    for (int tile = w; tile < p.ntiles; tile += W)
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const int dd = tile * TILE + k * kWave + lane;
                            if (dd < p.D)
                                for (int o = 0; o < p.out_len; ++o)
                                    p.results[((size_t)tb * p.D + dd) * p.out_len + o] = __builtin_nanf("");
                        }
                }
                continue;
            }
            const size_t row = (size_t)tb * p.gp_len;
            const float *tv = p.value + row;
            const int16_t *tt = p.type + row;
            int len = uni((int)p.size[row]);
            len = len > p.gp_len ? p.gp_len : len;

            float acc = 0.0f;
            for (int tile = w; tile < p.ntiles; tile += W) {
                if (!single) load_tile(tile);
                v16f outs[K];
                if (MO) {
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int o = 0; o < kMaxOutRegs; ++o) outs[k][o] = 0.0f;
                }
                RegStack<K, DEPTH> st;
                st.h = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) st.tos[k] = 0.0f;
                for (int base = 0; base < len; base += kWave) {
                    const int r = base + lane;
                    uint32_t opv = 0, payv = 0;
                    if (r < len) {
                        const int i = len - 1 - r;
                        const Decoded dn = decode_node(tt[i], tv[i], MO, p.var_len, p.out_len);
                        opv = dn.op; payv = dn.pay;
                    }
                    const int n = len - base < kWave ? len - base : kWave;
                    run_chunk<MO, LEAN, K, DEPTH>(opv, payv, n, st, RegVars<VL, K>{vars}, outs);
                }
                if (STORE) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (d[k] < p.D) {
                            float *rr = p.results + ((size_t)tb * p.D + d[k]) * p.out_len;
                            if (!MO) rr[0] = st.tos[k];
                            else {
#pragma unroll
                                for (int o = 0; o < kMaxOutRegs; ++o)
                                    if (o < p.out_len) rr[o] = outs[k][o];
                            }
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float e;
                    if (!MO) {
                        e = err_term(yv[k] - st.tos[k], p.use_mse);
                    } else {
                        e = 0.0f;
                        const float *yr = p.y + (size_t)dc[k] * p.out_len;
#pragma unroll
                        for (int o = 0; o < kMaxOutRegs; ++o)
                            if (o < p.out_len) e += err_term(yr[o] - outs[k][o], p.use_mse);
                    }
                    acc += d[k] < p.D ? e : 0.0f;
                }
            }
            if (!STORE) {
                const float total = wave_sum(acc);
                if (lane == 0) part[par][b][w] = total;
            }
        }
        __syncthreads();

        // ---- phase 3: one thread per tree adds the tile partials in tile order ----
        if (p.marks && threadIdx.x < 64) {  // batches are at most 64 trees: wave 0 sees every class of the batch
            const int c = (int)threadIdx.x < nb ? cls_s[par][threadIdx.x] : TREE_OK;
            if (__any(c == TREE_HEAVY) && threadIdx.x == 0) p.marks[0] = 1u;
            if (__any(c == TREE_DEEP) && threadIdx.x == 0 && p.only_marked != 5) p.marks[1] = 1u;
        }
        if (STORE) {
            if ((int)threadIdx.x < nb) {
                const int c = cls_s[par][threadIdx.x];
                if (c == TREE_DEEP || c == TREE_HEAVY)
                    p.results[(size_t)tree_of(threadIdx.x) * p.D * p.out_len] = bits2f(c == TREE_DEEP ? kSentinelDeep : kSentinelHeavy);
            }
        } else if ((int)threadIdx.x < nb) {
            const int b = threadIdx.x;
            const int c = cls_s[par][b];
            float f;
            if (c == TREE_OK) {
                const int nw = p.ntiles < W ? p.ntiles : W;
                float s = 0.0f;
                for (int i = 0; i < nw; ++i) s += part[par][b][i];
                f = s / (float)p.D;
            } else {
                f = c == TREE_DEEP ? bits2f(kSentinelDeep) : c == TREE_HEAVY ? bits2f(kSentinelHeavy) : __builtin_nanf("");
            }
            p.fitness[tree_of(b)] = f;
        }
        if (!STORE && !MO && !LEAN && p.only_marked == 5) {   // a tree too deep for the register stack: the scratch stack, here and now (wave 0)
            bool deep_any = false;
            for (int b = 0; b < nb; ++b) deep_any |= uni(cls_s[par][b]) == TREE_DEEP;
            if (deep_any) {
                __syncthreads();   // (phase 3's sentinel words are written before the values replace them)
                if (w == 0) {   // (the stack: the lane's column of [gp_len + 2][64] floats of dynamic LDS, launch_fast -- a private array is a scratch segment, ~3 us of every launch)
                    extern __shared__ float deep_stack_lds[];
                    for (int b = 0; b < nb; ++b)
                        if (uni(cls_s[par][b]) == TREE_DEEP) general_tree_fitness<kWave>(p, uni(tree_of(b)), deep_stack_lds + lane);
                }
            }
        }
        par ^= 1;
    }
}

// General path: one wave per tree, lanes are datapoints, stack/outputs in scratch memory, dataset
// read row-major from global memory.  only_marked != 0: evaluate only trees whose first output
// word holds the sentinel written by the fast kernel.
template <bool MO, bool STORE>
__global__ __launch_bounds__(64) void sr_general_kernel(SrParams p, int only_marked) {
    const int lane = threadIdx.x & 63;
    // only_marked 1: behind the register kernels, the trees they marked too deep; 2: behind the threaded code alone (more
    // variables than the register kernels take), every tree it left marked
    // 4: behind the threaded code and the FULL register build (which has taken the heavy marks) when the caller's function mask kept
    // the general compiler from being launched: too deep, or carrying that compiler's sentinel after all (a mask that promised too much)
    if (only_marked == 4 && p.marks) {
        if (!marks_pending(p, 4) && uni((int)p.marks[1]) == 0) return;
    }
    if (only_marked == 1 && p.marks && uni((int)p.marks[1]) == 0) return;  // no tree needs the general path
    if (only_marked == 2 && p.marks && !marks_pending(p, 0) && uni((int)p.marks[1]) == 0) return;
    float stk[kMaxStack + 2];
    float outs[MO ? kGeneralOuts : 1];
    auto process = [&](int t) {
        const size_t row = (size_t)t * p.gp_len;
        const float *tv = p.value + row;
        const int16_t *tt = p.type + row;
        int len = uni((int)p.size[row]);
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
        const int cls = uni(classify_tree(tt, tv, len, MO, p.var_len, p.out_len, kMaxStack));
        if (cls != TREE_OK && !STORE) {
            if (lane == 0) p.fitness[t] = __builtin_nanf("");
            return;
        }
        float acc = 0.0f;
        for (int base = 0; base < p.D; base += kWave) {
            const int d = base + lane;
            const int dc = d < p.D ? d : p.D - 1;
            float res = __builtin_nanf("");
            if (cls == TREE_OK)
                res = run_general<MO>(tt, tv, len, p.X + (size_t)dc * p.var_len, p.var_len, p.out_len, outs, stk);
            if (STORE) {
                if (d < p.D) {
                    float *rr = p.results + ((size_t)t * p.D + d) * p.out_len;
                    if (!MO) rr[0] = res;
                    else for (int o = 0; o < p.out_len; ++o) rr[o] = cls == TREE_OK ? outs[o] : __builtin_nanf("");
                }
                continue;
            }
            float e;
            if (!MO) {
                e = err_term(p.y[dc] - res, p.use_mse);
            } else {
                e = 0.0f;
                for (int o = 0; o < p.out_len; ++o) e += err_term(p.y[(size_t)dc * p.out_len + o] - outs[o], p.use_mse);
            }
            acc += d < p.D ? e : 0.0f;
        }
        if (!STORE) {
            const float total = wave_sum(acc);
            if (lane == 0) p.fitness[t] = total / (float)p.D;
        }
    };
    if (!only_marked) {
        for (int t = blockIdx.x; t < p.pop; t += gridDim.x) process(t);
        return;
    }
    // behind the register kernels: every workgroup (one wave) owns a contiguous chunk of the population and reads its
    // mark words 64 at a time -- walking the population one dependent load per tree costs ~1 us per UNMARKED tree
    const int chunk = ((p.pop + (int)gridDim.x - 1) / (int)gridDim.x + 63) & ~63;
    const int c0 = (int)blockIdx.x * chunk < p.pop ? (int)blockIdx.x * chunk : p.pop;
    const int c1 = c0 + chunk < p.pop ? c0 + chunk : p.pop;
    for (int base = c0; base < c1; base += kWave) {
        const int t = base + lane;
        bool hit = false;
        if (t < c1) {
            const float *mark = STORE ? p.results + (size_t)t * p.D * p.out_len : p.fitness + t;
            const uint32_t w = f2bits(*mark);
            hit = w == kSentinelDeep || (only_marked >= 2 && w == kSentinelHeavy) || (only_marked == 4 && w == kSentinelGeneral);
        }
        unsigned long long m = __ballot(hit);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            process(base + b);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------
template <int K, int DEPTH, int VL, bool MO, int MAXW, bool STORE, bool LEAN>
static hipError_t launch_fast(SrParams p, int only_marked, hipStream_t stream, unsigned *zeroed_counter = nullptr) {
    auto kern = sr_fast_kernel<K, DEPTH, VL, MO, MAXW, STORE, LEAN>;
    const DeviceInfo &dev = device_info();
    p.ntiles = (p.D + 64 * K - 1) / (64 * K);
    p.only_marked = only_marked;
    const int W = p.ntiles < MAXW ? p.ntiles : MAXW;
    static int per_cu_cache[kMaxWaves + 1] = {0}; // occupancy per block size of THIS instantiation
    int per_cu = per_cu_cache[W];
    if (per_cu == 0) {
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, W * 64, 0);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        per_cu_cache[W] = per_cu;
    }
    long blocks = (long)dev.num_cus * per_cu;
    // batch size: ~16 batches per workgroup keeps the tail short and the atomics rare
    long batch = p.pop / (blocks * 16);
    batch = batch < 4 ? 4 : (batch > kMaxBatch ? kMaxBatch : batch);
    if (const char *env = getenv("EVOGP_SR_BATCH")) {
        const int b = atoi(env);
        batch = b < 1 ? 1 : (b > kMaxBatch ? kMaxBatch : b);
    }
    p.batch = (int)batch;
    const long need = (p.pop + batch - 1) / batch;
    if (blocks > need) blocks = need;
    hipError_t e;
    p.counter = zeroed_counter ? zeroed_counter : acquire_counter(stream, &e);
    if (!p.counter) return e;
    // only_marked == 5: wave 0 evaluates a tree too deep for the register stack on a stack in LDS (run_population folds only rows of at most kFoldMaxLen nodes)
    const size_t lds = only_marked == 5 ? (size_t)(p.gp_len + 2) * kWave * sizeof(float) : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(W * 64), lds, stream, p);
    return hipGetLastError();
}

// LEAN pass over every tree, then the FULL build over the trees the lean pass marked heavy.
template <int K, int DEPTH, int VL, bool MO, int MAXW, bool STORE>
static hipError_t launch_pair(const SrParams &p, hipStream_t stream) {
    hipError_t e = launch_fast<K, DEPTH, VL, MO, MAXW, STORE, true>(p, 0, stream);
    if (e != hipSuccess) return e;
    return launch_fast<K, DEPTH, VL, MO, MAXW, STORE, false>(p, 1, stream);
}

template <bool STORE>
static hipError_t launch_general(const SrParams &p, int only_marked, hipStream_t stream) {
    const DeviceInfo &dev = device_info();
    long blocks = (long)dev.num_cus * (only_marked ? 4 : 16);  // behind a fast kernel only the marked trees are left
    if (blocks > p.pop) blocks = p.pop;
    if (p.out_len > 1) hipLaunchKernelGGL((sr_general_kernel<true, STORE>), dim3((unsigned)blocks), dim3(64), 0, stream, p, only_marked);
    else hipLaunchKernelGGL((sr_general_kernel<false, STORE>), dim3((unsigned)blocks), dim3(64), 0, stream, p, only_marked);
    return hipGetLastError();
}

// Configurations (chosen from the sweep in profiles/: throughput ~ K x resident waves):
//   single output, D >= 256 : K = 4 rows per lane, 16-entry stack, variable registers sized to var_len
//   multi output,  D >= 128 : K = 2 (16 output accumulators per row live in registers as well)
//   small datasets          : K = 1, 32-entry stack
static std::mutex g_chain_mu;  // one for both instantiations of run_population: they share the scratch blocks

template <bool STORE>
static int run_population(const SrParams &p_in, hipStream_t stream) {
    SrParams p = p_in;
    p.hint_general = 1;   // the general compiler is launched unless the function mask below says that nothing can be left for it
    // The scratch block handed out below was zeroed by the previous call's first kernel ON THE SAME STREAM; two host
    // threads feeding one stream must therefore not interleave their acquire + launch sequences.
    std::lock_guard<std::mutex> chain_lock(g_chain_mu);
    {   // pending-marks flags of this call: follow-up kernels leave at once when nothing was marked for them
        hipError_t me;
        p.marks = acquire_call_scratch(stream, &p.zero_next, &me);
        if (!p.marks) return (int)me;
    }
    static const bool forced = getenv("EVOGP_SR_FORCE_GENERAL") != nullptr;  // A/B switches are read once per process
    if (STORE && !forced && p.out_len <= kMaxOutRegs && ((size_t)p.var_len + (p.out_len > 1 ? 8 * (size_t)p.out_len : 0)) * 256 <= 150 * 1024 &&
        (p.var_len > 32 || p.D > 1024)) {
        // more variables than a register tuple holds, or more rows than one workgroup keeps resident: tile-group kernel
        hipError_t we = launch_wide_store(p, stream);
        if (we != hipSuccess) return (int)we;
        return (int)launch_general<STORE>(p, 1, stream);
    }
    const bool fast_ok = p.var_len <= 32 && p.out_len <= kMaxOutRegs && !forced;
    // EVOGP_SR_ASM: 0 = C++ interpreter only, 3 = threaded code (default)
    static const int asm_depth = env_int("EVOGP_SR_ASM", EVOGP_SR_DEFAULT_ASM);
    hipError_t e;
    if (!fast_ok) {
        // more variables (or outputs) than the register kernels hold: the threaded code keeps its dataset in LDS and does not
        // mind; what it leaves marked goes to the scratch-stack kernel
        if (!STORE && !forced && asm_depth == 3) {
            bool done = false;
            if ((e = launch_threaded_code(p, stream, &done, &p.mark_sample, &p.mark_chunks)) != hipSuccess) return (int)e;
            if (done) return (int)launch_general<STORE>(p, 2, stream);
        }
        return (int)launch_general<STORE>(p, 0, stream);
    }
    const bool mo = p.out_len > 1;
    bool tc_done = false, folded = false;
    ProfCall prof{};
    bool profiling = false;
    if (!STORE) {
        std::lock_guard<std::mutex> pl(g_prof_mu);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (g_prof_on && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            profiling = true;
            for (auto &ev : prof.ev) if (hipEventCreate(&ev) != hipSuccess) profiling = false;
            if (profiling) { (void)hipEventRecord(prof.ev[0], stream); p.prof_mid = prof.ev[1]; }
        }
    }
    auto prof_done = [&](bool mid) {
        if (!profiling) return;
        (void)hipEventRecord(prof.ev[3], stream);
        prof.mid = mid;
        std::lock_guard<std::mutex> pl(g_prof_mu);
        g_prof.push_back(prof);
    };
    // A call WITHOUT a function mask -- torch.ops.evogp_cuda.tree_SR_fitness, the operator the reference's own Python calls
    // (tree/forest.py:340-351, torch_wrapper.cu:235-284) -- is given the mask that the last completed call on a forest of this shape
    // OBSERVED on the device (the packed compiler notes which of its lines it needed, the interpreter's launch publishes that to a word
    // of host memory: sr_tc.hip tc_learned_class): nothing but + - * / -> the arithmetic line and ONE array of records (256 instead of
    // 768 MB at 1 M trees), the kernels a caller's mask chooses.  The first call on a shape has no observation: it looks (one pass over
    // the nodes, waited for -- once per shape; inside a stream capture it takes round 5's generic line and three arrays instead).  Only
    // speed follows the observation: it is of an EARLIER forest, and a tree the chosen line cannot take goes the way it goes under a
    // mask that promised too much (the register kernels: the same value up to the order of the sum over the rows) -- for one call, whose
    // own observation corrects the next.  EVOGP_TC_LEARN=0: such calls always take the generic line and three arrays.
    if (!STORE && asm_depth == 3 && !mo && !p.classify && p.func_mask == 0u && p.gp_len <= 64 && p.marks) {
        static const int env_learn = env_int("EVOGP_TC_LEARN", 1);
        constexpr unsigned kArith = (1u << F_ADD) | (1u << F_SUB) | (1u << F_MUL) | (1u << F_DIV);
        constexpr unsigned kUnaryOwn = (1u << F_SIN) | (1u << F_COS) | (1u << F_TAN) | (1u << F_LOG) | (1u << F_LOOSE_LOG) | (1u << F_EXP) | (1u << F_INV) |
                                       (1u << F_NEG) | (1u << F_ABS) | (1u << F_SQRT) | (1u << F_LOOSE_SQRT);
        if (env_learn) {
            int cls = tc_learned_class(p.pop, p.gp_len, stream, &p.feedback, &p.feedback_expected);
            if (p.feedback && cls < 0) {
                cls = tc_detect_class(p, p.marks + 5, stream);
                if (cls >= 0) p.feedback_expected = tc_store_class(p.feedback, cls);
            }
            if (p.feedback && cls == 0) p.func_mask = kArith;
            else if (p.feedback && cls == 1) p.func_mask = kArith | kUnaryOwn;
        }
    }
    if (!STORE && asm_depth == 3 && !mo && !profiling && p.func_mask != 0u && p.gp_len <= 64) {
        // The caller knows the forest's function set.  Nothing but + - * / and the unary functions with handlers of their own:
        // no tree can be left for the general compiler (rows of at most 64 nodes), so that launch is not made.  The last follow-up
        // kernel (mode 4: behind the FULL build) also takes a tree that carries the general compiler's sentinel after all, so a mask
        // that promises too much costs speed, never a result.  Decided by the mask alone: the same forest always takes the same
        // kernels.  5 launches -> 4 on the headline's function set.
        constexpr unsigned kOwnHandlers = (1u << F_ADD) | (1u << F_SUB) | (1u << F_MUL) | (1u << F_DIV) | (1u << F_SIN) | (1u << F_COS) | (1u << F_TAN) |
                                          (1u << F_LOG) | (1u << F_LOOSE_LOG) | (1u << F_EXP) | (1u << F_INV) | (1u << F_NEG) | (1u << F_ABS) |
                                          (1u << F_SQRT) | (1u << F_LOOSE_SQRT);
        constexpr unsigned kBailOut = (1u << F_SIN) | (1u << F_COS) | (1u << F_TAN);
        // (EVOGP_TC_FUNC_MASK=0 ignores the mask.)  The promise used to fail about once per million trees: tree_generate's roulette
        // scan returns function id 29 -- no function, a unary node worth 0 (forward.cu:117) -- whenever its uniform draw is exactly
        // 1.0, which the reference's float(u32) * 2^-32 reaches once in ~3e7 draws (generate.cu:77-84); the headline's forest holds
        // one such tree (index 549654), only the general compiler took it, and without that launch the scratch-stack kernel spent
        // 65 us on this ONE tree (1.54 instead of 1.44 ms per call, profiles/r03r_*, r03s_*).  The one-chunk compiler now takes
        // such nodes itself (kUnaryZero, sr_tc.hip), which also spares the unmasked call the general compiler's scan of a million
        // mark words: 1.443 -> 1.420 ms without the mask, 1.409 with it; 125 k trees 0.2115 / 0.2095 (profiles/r03u_shard_model_mask.log).
        static const int env_mask = env_int("EVOGP_TC_FUNC_MASK", 1);
        if (env_mask && (p.func_mask & ~kOwnHandlers) == 0u) p.hint_general = 0;
        // (Also leaving out the FULL register build where no run-time bail-out can occur was measured and taken back: the few
        // trees whose operand stack is too deep for the threaded code -- there are always some in a million -- then go to the
        // scratch-stack kernel, 1.54 instead of 1.44 ms at 1 M trees, profiles/r03r_shard_model.log.)
        (void)kBailOut;
    }
    if (!STORE && asm_depth == 3) {
        // threaded-code path (sr_tc.hip); trees it cannot take come back marked for the FULL register build
        p.stats = g_stats;
        e = launch_threaded_code(p, stream, &tc_done, &p.mark_sample, &p.mark_chunks);
        p.stats = nullptr;
        if (e != hipSuccess) return (int)e;
    }
    if (profiling) (void)hipEventRecord(prof.ev[2], stream);
    if (profiling && tc_done && g_prof_skip_followups) {   // diagnostics: which trees does the threaded code leave to the register kernels?
        prof_done(tc_done);
        return 0;
    }
    if (tc_done && mo) {  // multi-output trees the threaded code left marked: FULL register build, then the general kernel
        if (p.D >= 128) e = p.var_len <= 16 ? launch_fast<2, 16, 16, true, 8, STORE, false>(p, 1, stream, p.marks + 3) : launch_fast<2, 16, 32, true, 8, STORE, false>(p, 1, stream, p.marks + 3);
        else e = p.var_len <= 16 ? launch_fast<1, 32, 16, true, 16, STORE, false>(p, 1, stream, p.marks + 3) : launch_fast<1, 32, 32, true, 16, STORE, false>(p, 1, stream, p.marks + 3);
    } else if (tc_done) {
        // (one follow-up launch: the FULL register build takes what the threaded code marked and runs the scratch stack itself for the rare
        // tree that is too deep for its registers; EVOGP_SR_FOLD=0: the scratch-stack kernel as a launch of its own, as in round 4)
        static const int env_fold = env_int("EVOGP_SR_FOLD", 1);
        folded = env_fold != 0 && !STORE && p.gp_len <= kFoldMaxLen;   // (longer rows: the scratch-stack kernel as a launch of its own)
        const int om = folded ? 5 : 1;
        if (p.var_len <= 10) e = launch_fast<4, 16, 10, false, 4, STORE, false>(p, om, stream, p.marks + 3);
        else if (p.var_len <= 12) e = launch_fast<4, 16, 12, false, 4, STORE, false>(p, om, stream, p.marks + 3);
        else if (p.var_len <= 16) e = launch_fast<4, 16, 16, false, 4, STORE, false>(p, om, stream, p.marks + 3);
        else e = launch_fast<4, 16, 32, false, 4, STORE, false>(p, om, stream, p.marks + 3);
    } else if (!mo && p.D >= 256) {
        if (p.var_len <= 9) e = launch_pair<4, 16, 9, false, 4, STORE>(p, stream);
        else if (p.var_len <= 10) e = launch_pair<4, 16, 10, false, 4, STORE>(p, stream);
        else if (p.var_len <= 12) e = launch_pair<4, 16, 12, false, 4, STORE>(p, stream);
        else if (p.var_len <= 16) e = launch_pair<4, 16, 16, false, 4, STORE>(p, stream);
        else e = launch_pair<4, 16, 32, false, 4, STORE>(p, stream);
    } else if (mo && p.D >= 128) {
        e = p.var_len <= 16 ? launch_pair<2, 16, 16, true, 8, STORE>(p, stream) : launch_pair<2, 16, 32, true, 8, STORE>(p, stream);
    } else if (!mo) {
        e = p.var_len <= 16 ? launch_pair<1, 32, 16, false, 16, STORE>(p, stream) : launch_pair<1, 32, 32, false, 16, STORE>(p, stream);
    } else {
        e = p.var_len <= 16 ? launch_pair<1, 32, 16, true, 16, STORE>(p, stream) : launch_pair<1, 32, 32, true, 16, STORE>(p, stream);
    }
    if (e != hipSuccess) return (int)e;
    if (!folded) e = launch_general<STORE>(p, tc_done && !mo && !p.hint_general ? 4 : 1, stream);
    static const bool dbg_marks = getenv("EVOGP_DEBUG_MARKS") != nullptr;   // diagnostics: the call's flag words (synchronises)
    if (dbg_marks && p.marks) {
        unsigned h[8] = {};
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, p.marks, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[evogp] sr_fitness pop %d: marks %u %u %u %u %u %u %u %u, general compiler %d, chunks %d\n", p.pop, h[0], h[1], h[2], h[3], h[4],
                h[5], h[6], h[7], p.hint_general, p.mark_chunks);
    }
    prof_done(tc_done);
    return (int)e;
}

} // namespace evogp

namespace evogp {

hipError_t run_argmax_count_threaded(const SrParams &p_in, const int *labels, unsigned *counts, unsigned *wide_marks, hipStream_t stream, bool *handled) {
    *handled = false;
    static const int asm_depth = env_int("EVOGP_SR_ASM", EVOGP_SR_DEFAULT_ASM);
    if (asm_depth != 3) return hipSuccess;
    SrParams p = p_in;
    p.hint_general = 1;
    std::lock_guard<std::mutex> chain_lock(g_chain_mu);   // (the call-scratch chain of run_population)
    hipError_t e;
    p.marks = acquire_call_scratch(stream, &p.zero_next, &e);
    if (!p.marks) return e;
    if ((e = launch_threaded_code(p, stream, handled, &p.mark_sample, &p.mark_chunks)) != hipSuccess) return e;
    if (!*handled) {  // not eligible: the scratch block stays clean for the next call, nothing to undo
        return hipSuccess;
    }
    return launch_tc_count(counts, p.pop, labels, p.D, wide_marks, stream);
}

} // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_sr_fitness(unsigned pop_size, unsigned data_points, unsigned gp_len, unsigned var_len,
                                    unsigned out_len, int use_mse, const float *value, const int16_t *type,
                                    const int16_t *size, const float *variables, const float *labels,
                                    float *fitnesses, unsigned kernel_type, evogp_stream_t stream_) {
    return evogp_hip_sr_fitness_hinted(pop_size, data_points, gp_len, var_len, out_len, use_mse, value, type, size, variables, labels,
                                       fitnesses, kernel_type, 0u, stream_);
}

extern "C" int evogp_hip_sr_fitness_hinted(unsigned pop_size, unsigned data_points, unsigned gp_len, unsigned var_len,
                                           unsigned out_len, int use_mse, const float *value, const int16_t *type,
                                           const int16_t *size, const float *variables, const float *labels,
                                           float *fitnesses, unsigned kernel_type, unsigned function_mask, evogp_stream_t stream_) {
    // argument contract of torch_wrapper.cu:250-254
    if (pop_size == 0 || data_points == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len == 0)
        return EVOGP_E_BADARG;
    if (kernel_type > 4) return EVOGP_E_BADARG;
    if (!value || !type || !size || !variables || !labels || !fitnesses) return EVOGP_E_NULLPTR;
    if (out_len > (unsigned)kGeneralOuts) return EVOGP_E_UNSUPPORTED;
    SrParams p{};
    p.value = value; p.type = type; p.size = size; p.X = variables; p.y = labels; p.fitness = fitnesses;
    p.pop = (int)pop_size; p.D = (int)data_points; p.gp_len = (int)gp_len; p.var_len = (int)var_len;
    p.out_len = (int)out_len; p.use_mse = use_mse ? 1 : 0;
    p.func_mask = function_mask;
    return run_population<false>(p, (hipStream_t)stream_);
}

extern "C" int evogp_hip_batch_evaluate(unsigned pop_size, unsigned data_points, unsigned gp_len, unsigned var_len,
                                        unsigned out_len, const float *value, const int16_t *type, const int16_t *size,
                                        const float *variables, float *results, evogp_stream_t stream_) {
    if (pop_size == 0 || data_points == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len == 0)
        return EVOGP_E_BADARG;
    if (!value || !type || !size || !variables || !results) return EVOGP_E_NULLPTR;
    if (out_len > (unsigned)kGeneralOuts) return EVOGP_E_UNSUPPORTED;
    SrParams p{};
    p.value = value; p.type = type; p.size = size; p.X = variables; p.y = nullptr; p.results = results;
    p.pop = (int)pop_size; p.D = (int)data_points; p.gp_len = (int)gp_len; p.var_len = (int)var_len;
    p.out_len = (int)out_len; p.use_mse = 1;
    return run_population<true>(p, (hipStream_t)stream_);
}

// Profiling hook for bench scripts (not part of the reference boundary): point the threaded-code kernel at a
// device buffer of 8 x u64 cycle counters {asm, batch loop, barrier wait, trees, nodes, kernel, waves, -};
// nullptr switches the accounting off (the default).
extern "C" int evogp_hip_debug_set_stats(unsigned long long *device_counters) {
    g_stats = device_counters;
    return 0;
}

extern "C" int evogp_hip_debug_profile(int enable) {
    std::lock_guard<std::mutex> pl(g_prof_mu);
    for (auto &c : g_prof) for (auto &ev : c.ev) (void)hipEventDestroy(ev);
    g_prof.clear();
    g_prof_on = enable != 0;
    g_prof_skip_followups = enable == 2;
    return 0;
}

extern "C" int evogp_hip_debug_profile_read(float *stage_ms, int *calls) {
    if (!stage_ms || !calls) return EVOGP_E_NULLPTR;
    std::lock_guard<std::mutex> pl(g_prof_mu);
    double sum[3] = {0, 0, 0};
    int n = 0;
    for (auto &c : g_prof) {
        hipError_t e = hipEventSynchronize(c.ev[3]);
        if (e != hipSuccess) return (int)e;
        if (!c.mid) continue;  // not a threaded-code call: no compiler / interpreter split
        for (int i = 0; i < 3; ++i) {
            float ms = 0;
            if ((e = hipEventElapsedTime(&ms, c.ev[i], c.ev[i + 1])) != hipSuccess) return (int)e;
            sum[i] += ms;
        }
        ++n;
    }
    for (int i = 0; i < 3; ++i) stage_ms[i] = n ? (float)(sum[i] / n) : 0.0f;
    *calls = n;
    return 0;
}
