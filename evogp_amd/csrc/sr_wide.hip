// sr_wide.hip — batch evaluation for WIDE inputs and the fused classification epilogue (gfx950).
//
// results[t][d][:] = tree_t(X[d][:])  for every tree t and every row d of a shared dataset, like
// evogp_hip_batch_evaluate (src/evogp/tree/forest.py:143-176), for the shapes the register kernels of
// sr_fitness.hip cannot keep resident: more than 32 variables, or a dataset too large for one workgroup (the classifier
// config: 1797 rows x 64 variables = 460 KB).  The dataset is cut into GROUPS of R = 64 K rows; a workgroup owns one
// group for its whole life: it stages the group's rows ONCE into LDS as [variable][row] (any number of variables up to
// the LDS size) and its eight waves then walk DIFFERENT trees over the SAME rows — lane l holds rows K l .. K l + K - 1 —
// with the wave-uniform register-stack interpreter of interp.hpp (variables are read from LDS instead of a register
// tuple).  Sharing the rows between the waves is what makes K > 1 affordable (one copy of 64 variables x 256 rows is
// 64 KB) and K rows per dispatch is what amortises the interpreter: the first version (a private tile per wave, K = 1,
// two waves per SIMD) ran at 1.4 node-rows per clock and CU.  Waves never synchronise after the staging.
// Workgroup (g, i) of n takes trees 8 i + w, 8 (i + n) + w, ... on group g.
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
#include <cstdio>
#include <vector>
#include "launch.hpp"
#include "sr_params.hpp"
#include <cstdlib>

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

constexpr int kWideWaves = 8;
struct __attribute__((packed, aligned(4))) Unaligned4 { float x, y, z, w; };  // 16-byte store at 4-byte alignment

// MODE 1 follow-up: trees marked deep -- and trees with a row whose arg-max hangs on the soft-max's rounding (interp.hpp) -- are
// counted with the scratch-stack interpreter and torch's own arithmetic for the arg-max.  A wave takes ONE block of 64 rows of a
// marked tree (blockIdx.y = row block): the first version gave a wave the whole tree, and a handful of marked trees (44 of
// 200 000 on the digits data) then cost the fitness pass the serial walk of one wave through all 29 row blocks, 0.6-1.4 ms.
// Three launches, all leaving at once when nothing is marked: clear (a marked tree's count word becomes the bare mark -- the
// tile-group kernel's other row groups may have added their hits to it), count (atomic adds under the mark), unmark.
__global__ void wide_deep_clear_kernel(WideParams p, int unmark) {
    if (p.marks && p.marks[1] == 0u) return;
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= p.pop) return;
    const unsigned c = p.counts[t];
    if (c & kDeepCountBit) p.counts[t] = unmark ? (c & ~kDeepCountBit) : kDeepCountBit;
}

__global__ __launch_bounds__(64) void wide_deep_count_kernel(WideParams p) {
    if (p.marks && p.marks[1] == 0u) return;
    const int lane = threadIdx.x;
    float stk[kMaxStack + 2];
    float o16[kMaxOutRegs];
    const int nrb = (p.D + 63) / 64;
    for (int rb = (int)blockIdx.y; rb < nrb; rb += (int)gridDim.y) {   // (the grid's y extent is capped: datasets beyond 4 M rows loop)
    const int d = rb * 64 + lane, dc = d < p.D ? d : p.D - 1;
    for (int t0 = blockIdx.x * 64; t0 < p.pop; t0 += gridDim.x * 64) {
        // 64 count words at a time: which of these trees are marked
        const int tl = t0 + lane;
        unsigned long long marked = __ballot(tl < p.pop && (p.counts[tl] & kDeepCountBit) != 0u);
        while (marked) {
            const int t = t0 + __builtin_ctzll(marked);
            marked &= marked - 1;
            const size_t row = (size_t)t * p.gp_len;
            int len = uni((int)p.size[row]);
            len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
            (void)run_general<true>(p.type + row, p.value + row, len, p.X + (size_t)dc * p.var_len, p.var_len, p.out_len, o16, stk);
            const int best = argmax_as_torch(o16, p.out_len);
            const unsigned hits = (unsigned)__popcll(__ballot(d < p.D && best == p.labels[dc]));
            if (lane == 0 && hits) atomicAdd(p.counts + t, hits);
        }
    }
    }
}

// the three launches behind a fused count (threaded code: the count words of marked trees are bare marks already)
static hipError_t launch_deep_recount(const WideParams &p, bool words_are_bare_marks, hipStream_t stream) {
    const unsigned tb = (unsigned)((p.pop + 255) / 256);
    if (!words_are_bare_marks) hipLaunchKernelGGL(wide_deep_clear_kernel, dim3(tb), dim3(256), 0, stream, p, 0);
    long blocks = ((long)p.pop + 63) / 64;
    if (blocks > 512) blocks = 512;
    const unsigned row_blocks = (unsigned)((p.D + 63) / 64);
    hipLaunchKernelGGL(wide_deep_count_kernel, dim3((unsigned)blocks, row_blocks > 65535u ? 65535u : row_blocks), dim3(64), 0, stream, p);
    hipLaunchKernelGGL(wide_deep_clear_kernel, dim3(tb), dim3(256), 0, stream, p, 1);
    return hipGetLastError();
}

template <bool MO, int MODE, int K, int DEPTH>
__global__ __launch_bounds__(kWideWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void sr_wide_kernel(WideParams p) {
    extern __shared__ float wide_lds[];  // [variable][R]
    constexpr int R = K * 64;
    const int lane = threadIdx.x & 63;
    const int w = uni((int)(threadIdx.x >> 6));
    const int group = blockIdx.x % p.ngroups, worker = blockIdx.x / p.ngroups;
    const int row0 = group * R;
    for (int e = threadIdx.x; e < R * p.var_len; e += blockDim.x) {
        const int dl = e / p.var_len, v = e - dl * p.var_len;
        const int d = row0 + dl < p.D ? row0 + dl : p.D - 1;
        wide_lds[v * R + dl] = p.X[(size_t)d * p.var_len + v];
    }
    __syncthreads();
    const LdsVars vars{wide_lds + lane * K, R};
    float *out_s = wide_lds + (size_t)p.var_len * R;  // MODE 0, multi-output: [wave][row][output] transposition blocks
    const int d0 = row0 + lane * K;  // this lane's first row
    int label[K];
#pragma unroll
    for (int k = 0; k < K; ++k) label[k] = MODE == 1 ? p.labels[d0 + k < p.D ? d0 + k : p.D - 1] : 0;

    for (int t = worker * kWideWaves + w; t < p.pop; t += p.workers * kWideWaves) {
        const size_t row = (size_t)t * p.gp_len;
        const float *tv = p.value + row;
        const int16_t *tt = p.type + row;
        int len = uni((int)p.size[row]);
        len = len < 0 ? 0 : (len > p.gp_len ? p.gp_len : len);
        const int cls = uni(classify_tree(tt, tv, len, MO, p.var_len, p.out_len, DEPTH));
        if (cls != TREE_OK) {
            if (MODE == 0) {
                if (cls == TREE_DEEP) {  // redone by sr_general_kernel (every group writes the same mark)
                    if (lane == 0) { p.results[(size_t)t * p.D * p.out_len] = bits2f(kSentinelDeep); if (p.marks) p.marks[1] = 1u; }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        if (d0 + k < p.D)
                            for (int o = 0; o < p.out_len; ++o) p.results[((size_t)t * p.D + d0 + k) * p.out_len + o] = __builtin_nanf("");
                }
                continue;
            }
            if (cls == TREE_DEEP) {  // recounted by wide_deep_count_kernel
                if (lane == 0) { atomicOr(p.counts + t, kDeepCountBit); if (p.marks) p.marks[1] = 1u; }
                continue;
            }
            // malformed tree: all outputs NaN -> arg-max 0
            unsigned hits = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) hits += (unsigned)__popcll(__ballot(d0 + k < p.D && label[k] == 0));
            if (lane == 0 && hits) atomicAdd(p.counts + t, hits);
            continue;
        }
        v16f outs[K];
        float top[K];
        {
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
                    const Decoded dn = decode_node(tt[len - 1 - r], tv[len - 1 - r], MO, p.var_len, p.out_len);
                    opv = dn.op; payv = dn.pay;
                }
                const int n = len - base < kWave ? len - base : kWave;
                run_chunk<MO, false, K, DEPTH>(opv, payv, n, st, vars, outs);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) top[k] = st.tos[k];
        }
        if (MODE == 0 && !MO) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (d0 + k < p.D) p.results[(size_t)t * p.D + d0 + k] = top[k];
        } else if (MODE == 0) {
            // The wave's rows x outputs are ONE contiguous block of the result tensor, but a lane holds out_len values
            // of each of its K rows: transpose through LDS and store 16 bytes per lane instead of out_len x K scalars at
            // a stride of out_len floats.
            float *mine = out_s + (size_t)w * R * p.out_len;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int o = 0; o < kMaxOutRegs; ++o)
                    if (o < p.out_len) mine[(lane * K + k) * p.out_len + o] = outs[k][o];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int rows_here = p.D - row0 < R ? p.D - row0 : R;
            const int nvalid = rows_here * p.out_len;
            float *gbase = p.results + ((size_t)t * p.D + row0) * p.out_len;
            for (int e = lane * 4; e < nvalid; e += 256) {
                if (e + 4 <= nvalid) {
                    Unaligned4 q;
                    q.x = mine[e]; q.y = mine[e + 1]; q.z = mine[e + 2]; q.w = mine[e + 3];
                    *reinterpret_cast<Unaligned4 *>(gbase + e) = q;
                } else {
                    for (int i = e; i < nvalid; ++i) gbase[i] = mine[i];
                }
            }
            __builtin_amdgcn_wave_barrier();  // the block is rewritten for the wave's next tree
        } else {
            // arg-max as torch.argmax(clip(softmax(x))) sees it: the raw arg-max, except in rows where an output in front of the
            // first maximum is within the soft-max's rounding of it -- a tree with such a row is recounted with torch's arithmetic
            unsigned hits = 0;
            bool any_amb = false;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                int best = 0;
                float m = outs[k][0];
                bool poisoned = m != m;
#pragma unroll
                for (int o = 1; o < kMaxOutRegs; ++o) {
                    if (o < p.out_len) {
                        const float x = outs[k][o];
                        poisoned |= x != x;
                        if (x > m) { m = x; best = o; }
                    }
                }
                const bool forced = poisoned || __builtin_isinf(m);
                const float thr = m - kSoftmaxTieMargin;
                bool amb = false;
#pragma unroll
                for (int o = 0; o < kMaxOutRegs; ++o)
                    if (o < p.out_len) amb |= o < best && outs[k][o] >= thr;
                any_amb |= amb && !forced && d0 + k < p.D;
                if (forced) best = 0;
                hits += (unsigned)__popcll(__ballot(d0 + k < p.D && best == label[k]));
            }
            if (__ballot(any_amb) != 0ull) {   // (wave-uniform: the recount kernel overwrites whatever the groups add)
                if (lane == 0) { atomicOr(p.counts + t, kDeepCountBit); if (p.marks) p.marks[1] = 1u; }
            } else if (lane == 0 && hits) atomicAdd(p.counts + t, hits);
        }
    }
}

template <bool MO, int MODE, int K, int DEPTH>
static hipError_t launch_wide_k(WideParams p, hipStream_t stream) {
    const DeviceInfo &dev = device_info();
    constexpr int R = K * 64;
    const size_t lds = ((size_t)p.var_len * R + (MO && MODE == 0 ? (size_t)kWideWaves * R * p.out_len : 0)) * 4;
    p.ngroups = (p.D + R - 1) / R;
    int per_cu = (int)((dev.lds_per_cu - 2048) / lds);
    per_cu = per_cu > 2 ? 2 : (per_cu < 1 ? 1 : per_cu);  // 2 x 8 waves = four per SIMD
    int workers = dev.num_cus * per_cu / p.ngroups;
    workers = workers < 1 ? 1 : workers;
    const int max_workers = (p.pop + kWideWaves - 1) / kWideWaves;
    if (workers > max_workers) workers = max_workers;
    p.workers = workers;
    auto kern = sr_wide_kernel<MO, MODE, K, DEPTH>;
    static bool attr_done = false;
    if (!attr_done || lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { (void)hipGetLastError(); if (lds > 64 * 1024) return e; }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.ngroups * p.workers)), dim3(kWideWaves * 64), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || MODE != 1) return e;
    return launch_deep_recount(p, false, stream);
}

// rows per lane: as many as the variables leave LDS for (two workgroups per CU when possible); EVOGP_WIDE_K overrides
static int wide_rows_per_lane(const WideParams &p, int kmax) {
    const DeviceInfo &dev = device_info();
    int k = kmax;
    auto bytes = [&](int kk) { return ((size_t)p.var_len + (p.results && p.out_len > 1 ? (size_t)kWideWaves * p.out_len : 0)) * kk * 64 * 4; };
    while (k > 1 && bytes(k) * 2 > dev.lds_per_cu - 2048) k >>= 1;
    if (p.D <= 64 * (k >> 1) && k > 1) k >>= 1;  // a short dataset does not fill the rows
    if (const char *e = getenv("EVOGP_WIDE_K")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) k = v < kmax ? v : kmax; }
    return k;
}

template <bool MO, int MODE>
static hipError_t launch_wide(WideParams p, hipStream_t stream) {
    if (((size_t)p.var_len + (MO && MODE == 0 ? (size_t)kWideWaves * p.out_len : 0)) * 64 * 4 > device_info().lds_per_cu - 2048)
        return hipErrorInvalidValue;
    if (MO) {
        switch (wide_rows_per_lane(p, 2)) {
            case 2: return launch_wide_k<true, MODE, 2, 16>(p, stream);
            default: return launch_wide_k<true, MODE, 1, 32>(p, stream);
        }
    }
    if (MODE == 0) {
        switch (wide_rows_per_lane(p, 4)) {
            case 4: return launch_wide_k<false, 0, 4, 16>(p, stream);
            case 2: return launch_wide_k<false, 0, 2, 32>(p, stream);
            default: return launch_wide_k<false, 0, 1, 32>(p, stream);
        }
    }
    return hipErrorInvalidValue;  // the arg-max epilogue needs several outputs
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
    hipError_t e;
    WideParams p{};
    p.marks = acquire_counter(stream, &e);  // [1] != 0: some tree was too deep for the register stack
    if (!p.marks) return (int)e;
    p.value = value; p.type = type; p.size = size; p.X = variables; p.labels = labels; p.counts = counts;
    p.pop = (int)pop_size; p.D = (int)data_points; p.gp_len = (int)gp_len; p.var_len = (int)var_len; p.out_len = (int)out_len;
    // first choice: compiled programs on the threaded-code interpreter with the classifier's END handler (3-4 x the tile-group
    // kernel at the UCI classifier shape); what it leaves marked is recounted by wide_deep_count_kernel
    static const bool tc_ok = [] { const char *v = getenv("EVOGP_TC_CLASSIFY"); return !(v && v[0] == '0'); }();
    if (tc_ok) {
        SrParams s{};
        s.value = value; s.type = type; s.size = size; s.X = variables; s.y = (const float *)labels; s.fitness = (float *)counts;
        s.pop = p.pop; s.D = p.D; s.gp_len = p.gp_len; s.var_len = p.var_len; s.out_len = p.out_len; s.use_mse = 0; s.classify = 1;
        bool handled = false;
        if ((e = run_argmax_count_threaded(s, labels, counts, p.marks, stream, &handled)) != hipSuccess) return (int)e;
        if (handled) {
            static const bool dbg = getenv("EVOGP_DEBUG_CLS") != nullptr;   // how many trees go to the recount kernel (diagnostics)
            if (dbg) {
                std::vector<unsigned> h(pop_size);
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(h.data(), counts, (size_t)pop_size * sizeof(unsigned), hipMemcpyDeviceToHost);
                size_t marked = 0;
                for (unsigned c : h) marked += (c & kDeepCountBit) ? 1 : 0;
                fprintf(stderr, "[evogp] batch_argmax_count: %zu of %u trees marked for the recount kernel\n", marked, pop_size);
            }
            return (int)launch_deep_recount(p, true, stream);
        }
    }
    if ((e = zero_words_async(counts, (size_t)pop_size, stream)) != hipSuccess) return (int)e;
    return (int)launch_wide<true, 1>(p, stream);
}
