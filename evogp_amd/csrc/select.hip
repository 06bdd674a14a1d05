// select.hip — the selection step of a generation as ONE launch: which trees survive, which are elites (gfx950).
//
// DefaultSelection (src/evogp/algorithm/selection/default.py:21-39) sorts the whole fitness vector and keeps the best
// n_elite / n_surv indices.  Nothing downstream needs the ORDER inside those sets: elites are copied, parents are drawn
// uniformly from the survivors (crossover/default.py:30-45).  torch.sort of 100 k floats is 61-68 us in ten launches
// (219 us at 1 M; torch.kthvalue, the textbook alternative, 0.39 / 3.8 ms), the largest item of a generation outside the
// fitness call.  Here: an exact three-pass radix select (11 + 11 + 10 bits of an order-preserving key) for BOTH ranks at
// once, then a compaction in index order -- one cooperative kernel, phases separated by a grid barrier:
//
//     order[0 .. n_elite)        the n_elite best trees, ascending tree index
//     order[n_elite .. n_surv)   the other survivors, ascending tree index
//
// Ties at a threshold are resolved towards the lower index, so the two SETS are exactly those a stable descending sort
// yields, and the result is deterministic.  NaN counts as the worst fitness.  The grid is at most 64 workgroups of 1024 threads,
// so all workgroups are resident and the barrier (an atomic counter in the caller's zeroed workspace) cannot deadlock.
#include "evogp_defs.hpp"
#include "launch.hpp"
#include <cstdlib>

namespace evogp {

constexpr int kSelThreads = 1024;
constexpr int kSelBins = 2048;            // 11-bit digits (the last pass uses 1024 of them)
constexpr int kSelMaxBlocks = 64;    // few, large workgroups: a barrier's cost grows with the number of arrivals on its counter
// workspace words: [0] barrier, [16 .. 16 + 6 * kSelBins) histograms {pass 0, pass 1 elite, pass 1 keep, pass 2 elite, pass 2 keep},
// then 4 counts per workgroup
constexpr int kSelHist = 16;
constexpr int kSelCounts = kSelHist + 5 * kSelBins;
constexpr int kSelWords = kSelCounts + 4 * kSelMaxBlocks;

// larger fitness <=> larger key; NaN -> 0 (worse than -inf); -0 and +0 share a key, so ties between them go by index as in a
// stable sort of the float values (evogp_amd/parallel.py select_order computes the same key in torch where there is no GPU)
__device__ inline uint32_t select_key(float f) {
    uint32_t u = f2bits(f);
    if (f != f) return 0u;
    if (f == 0.0f) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// words other workgroups wrote with atomics: read at the device's coherence point, not through this CU's vector cache
__device__ inline unsigned ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline void grid_barrier(unsigned *bar, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(bar, 1u);
        while (ld(bar) < target) __builtin_amdgcn_s_sleep(1);   // (polling with a read-modify-write made 256 pollers queue behind each other)
        __threadfence();
    }
    __syncthreads();
}

struct SelectParams {
    const float *fitness;
    int *order;
    unsigned *ws;
    unsigned *zero_next;   // optional: the workspace of the NEXT call on this stream, zeroed here (no memset launch per call)
    int n, n_elite, n_keep;
};

// Inclusive scan over the workgroup's 1024 threads: the DPP scan inside each wave, the 16 wave totals through LDS -- two
// workgroup barriers where a Hillis-Steele scan in LDS takes twenty (the kernel was bound by its ~250 barriers, not by the data).
// `tot_s`: 16 words.  Every thread of the workgroup must call it.
__device__ inline unsigned block_scan_incl(unsigned v, unsigned *tot_s, unsigned *total) {
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
    const unsigned incl = (unsigned)wave_scan_incl((int)v);
    if (lane == 63) tot_s[w] = incl;
    __syncthreads();
    unsigned before = 0u, all = 0u;
#pragma unroll
    for (int q = 0; q < kSelThreads / 64; ++q) {
        const unsigned tq = tot_s[q];
        all += tq;
        before += q < w ? tq : 0u;
    }
    __syncthreads();
    *total = all;
    return incl + before;
}

// bin of `hist` (bins counted from the TOP) in which the rank-th largest element falls, and the rank inside that bin
__device__ inline void find_bin(const unsigned *hist, int bins, unsigned rank, unsigned *scan_s, int *bin_out, unsigned *rank_out) {
    // every thread sums a contiguous run of bins from the top
    const int per = bins / kSelThreads;                     // 2 or 1
    const int top = bins - 1 - (int)threadIdx.x * per;      // this thread's highest bin
    // (plain vector loads: these words were only ever touched by atomics, which live in L2, and the barrier in front of this call
    // ends with a fence -- no stale copy can sit in this CU's cache; coherent scalar loads cost ~0.7 us EACH, 34 us per call)
    unsigned hv[2] = {0u, 0u};                               // bins top - per + 1 .. top (per is 2 or 1)
    for (int j = 0; j < per; ++j) hv[j] = hist[top - per + 1 + j];
    unsigned mine = 0;
    for (int j = 0; j < per; ++j) mine += hv[j];
    // inclusive scan of the partial sums (top bins first); the thread whose run contains the rank walks its bins
    unsigned all;
    const unsigned incl = block_scan_incl(mine, scan_s, &all);
    __shared__ int s_bin;
    __shared__ unsigned s_rank;
    unsigned before = incl - mine;
    if (before < rank && rank <= incl) {   // exactly one thread (ranks beyond the total: none, the caller never asks)
        int bb = top;
        for (int j = 0; j < per - 1; ++j) {
            const unsigned h = hv[per - 1 - j];              // bin `top - j`
            if (before + h >= rank) break;
            before += h; --bb;
        }
        s_bin = bb;
        s_rank = rank - before;                              // 1-based rank inside the bin
    }
    __syncthreads();
    *bin_out = s_bin;
    *rank_out = s_rank;
    __syncthreads();
}

__global__ __launch_bounds__(kSelThreads) void select_kernel(SelectParams p) {
    __shared__ unsigned hist_s[2 * kSelBins];
    __shared__ unsigned scan_s[4 * (kSelThreads / 64)];   // wave totals of the workgroup scans (four at a time in the compaction)
    unsigned *bar = p.ws;
    unsigned *g0 = p.ws + kSelHist, *g1e = g0 + kSelBins, *g1k = g1e + kSelBins, *g2e = g1k + kSelBins, *g2k = g2e + kSelBins;
    unsigned *counts = p.ws + kSelCounts;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (p.zero_next && b == nb - 1)
        for (int i = tid; i < kSelWords; i += kSelThreads) p.zero_next[i] = 0u;
    const int chunk = ((p.n + nb - 1) / nb + kSelThreads - 1) / kSelThreads * kSelThreads;   // contiguous slice per workgroup
    const int lo = b * chunk < p.n ? b * chunk : p.n, hi = lo + chunk < p.n ? lo + chunk : p.n;
    const unsigned ke = (unsigned)p.n_elite, kk = (unsigned)p.n_keep;

    auto clear = [&](int words) { for (int i = tid; i < words; i += kSelThreads) hist_s[i] = 0u; __syncthreads(); };
    auto flush = [&](unsigned *g, const unsigned *s, int bins) {
        for (int i = tid; i < bins; i += kSelThreads) if (s[i]) atomicAdd(g + i, s[i]);
    };

    // pass 0: top 11 bits
    clear(kSelBins);
    for (int i = lo + tid; i < hi; i += kSelThreads) atomicAdd(&hist_s[select_key(p.fitness[i]) >> 21], 1u);
    __syncthreads();
    flush(g0, hist_s, kSelBins);
    grid_barrier(bar, (unsigned)nb);
    int be0 = 0, bk0 = 0;
    unsigned re = 0, rk = 0;
    if (ke) find_bin(g0, kSelBins, ke, scan_s, &be0, &re);
    find_bin(g0, kSelBins, kk, scan_s, &bk0, &rk);

    // pass 1: the next 11 bits of the elements in the two threshold bins
    clear(2 * kSelBins);
    for (int i = lo + tid; i < hi; i += kSelThreads) {
        const uint32_t k = select_key(p.fitness[i]);
        const int d0 = (int)(k >> 21), d1 = (int)((k >> 10) & 2047u);
        if (ke && d0 == be0) atomicAdd(&hist_s[d1], 1u);
        if (d0 == bk0) atomicAdd(&hist_s[kSelBins + d1], 1u);
    }
    __syncthreads();
    if (ke) flush(g1e, hist_s, kSelBins);
    flush(g1k, hist_s + kSelBins, kSelBins);
    grid_barrier(bar, 2u * (unsigned)nb);
    int be1 = 0, bk1 = 0;
    if (ke) find_bin(g1e, kSelBins, re, scan_s, &be1, &re);
    find_bin(g1k, kSelBins, rk, scan_s, &bk1, &rk);

    // pass 2: the last 10 bits
    clear(2 * kSelBins);
    const uint32_t pe = ((uint32_t)be0 << 11) | (uint32_t)be1, pk = ((uint32_t)bk0 << 11) | (uint32_t)bk1;   // 22-bit prefixes
    for (int i = lo + tid; i < hi; i += kSelThreads) {
        const uint32_t k = select_key(p.fitness[i]);
        if (ke && (k >> 10) == pe) atomicAdd(&hist_s[k & 1023u], 1u);
        if ((k >> 10) == pk) atomicAdd(&hist_s[kSelBins + (k & 1023u)], 1u);
    }
    __syncthreads();
    if (ke) flush(g2e, hist_s, 1024);
    flush(g2k, hist_s + kSelBins, 1024);
    grid_barrier(bar, 3u * (unsigned)nb);
    int be2 = 0, bk2 = 0;
    if (ke) find_bin(g2e, 1024, re, scan_s, &be2, &re);
    find_bin(g2k, 1024, rk, scan_s, &bk2, &rk);
    const uint32_t te = ke ? ((pe << 10) | (uint32_t)be2) : 0xFFFFFFFFu, tk = (pk << 10) | (uint32_t)bk2;  // threshold keys
    // re / rk: how many of the elements EQUAL to the threshold belong to the set (those of lowest index)

    // compaction, in index order.  Per workgroup: elements above each threshold and elements equal to it.
    unsigned c[4] = {0u, 0u, 0u, 0u};   // > te, == te, > tk, == tk
    for (int i = lo + tid; i < hi; i += kSelThreads) {
        const uint32_t k = select_key(p.fitness[i]);
        c[0] += ke && k > te; c[1] += ke && k == te; c[2] += k > tk; c[3] += k == tk;
    }
    {
        const int lane = tid & 63, w = tid >> 6;
        for (int j = 0; j < 4; ++j) {
            const unsigned ws = (unsigned)__builtin_amdgcn_readlane(wave_scan_incl((int)c[j]), 63);
            if (lane == 0) scan_s[j * (kSelThreads / 64) + w] = ws;
        }
        __syncthreads();
        if (tid < 4) {
            unsigned v = 0;
            for (int q = 0; q < kSelThreads / 64; ++q) v += scan_s[tid * (kSelThreads / 64) + q];
            counts[4 * b + tid] = v;
        }
        __syncthreads();
    }
    grid_barrier(bar, 4u * (unsigned)nb);
    // what the workgroups in front of this one hold
    unsigned pre[4] = {0u, 0u, 0u, 0u};
    {
        static_assert(kSelMaxBlocks <= 64, "one wave sums the counts of the workgroups in front");
        if (tid < 64) {
            for (int j = 0; j < 4; ++j) {
                const unsigned v = tid < b ? ld(counts + 4 * tid + j) : 0u;
                const unsigned ws = (unsigned)__builtin_amdgcn_readlane(wave_scan_incl((int)v), 63);
                if (tid == 0) scan_s[j] = ws;
            }
        }
        __syncthreads();
        for (int j = 0; j < 4; ++j) pre[j] = scan_s[j];
        __syncthreads();
    }
    // walk the slice one workgroup-width at a time; ranks inside a step by a scan over the workgroup
    unsigned run[4] = {pre[0], pre[1], pre[2], pre[3]};   // elements of each kind in front of the current step
    for (int base = lo; base < hi; base += kSelThreads) {
        const int i = base + tid;
        const uint32_t k = i < hi ? select_key(p.fitness[i]) : 0u;
        const bool in = i < hi;
        const unsigned f[4] = {(unsigned)(in && ke && k > te), (unsigned)(in && ke && k == te), (unsigned)(in && k > tk), (unsigned)(in && k == tk)};
        // exclusive scans over the workgroup for all four flags: two 16-bit fields per word (a field counts to 1024 at most)
        unsigned all_lo, all_hi;
        const unsigned incl_lo = block_scan_incl(f[0] | (f[1] << 16), scan_s, &all_lo);
        const unsigned incl_hi = block_scan_incl(f[2] | (f[3] << 16), scan_s, &all_hi);
        const unsigned ex[4] = {(incl_lo & 0xFFFFu) - f[0], (incl_lo >> 16) - f[1], (incl_hi & 0xFFFFu) - f[2], (incl_hi >> 16) - f[3]};
        const unsigned tot[4] = {all_lo & 0xFFFFu, all_lo >> 16, all_hi & 0xFFFFu, all_hi >> 16};
        if (in) {
            const unsigned above_e = run[0] + ex[0], tie_e = run[1] + ex[1], above_k = run[2] + ex[2], tie_k = run[3] + ex[3];
            const bool elite = ke && (k > te || (k == te && tie_e < re));
            const bool kept = k > tk || (k == tk && tie_k < rk);
            if (elite) {
                // elites in front of this one: those above the threshold, plus the ties taken so far
                const unsigned pos = above_e + (tie_e < re ? tie_e : re);
                p.order[pos] = i;
            } else if (kept) {
                // survivors in front of this one that are not elites
                const unsigned kept_before = above_k + (tie_k < rk ? tie_k : rk);
                const unsigned elite_before = above_e + (tie_e < re ? tie_e : re);
                p.order[ke + kept_before - elite_before] = i;
            }
        }
        for (int j = 0; j < 4; ++j) run[j] += tot[j];
    }
}

// ---- tournaments (selection/tournament.py:59-133 with its defaults: contenders drawn with replacement, the best one wins) --------
// winners[i] = the best of t_size contenders of tournament i; contender k of tournament i is tree
//     word(seed, generation, 16 + k, i) % n        (word: the counter-based hash of breed.hip / parallel.random_words)
// so every rank of a sharded run -- and the torch formulation for tensors that are not on a GPU -- names the same contenders.
// "Best" is the order of select_key (NaN worst, -0 = +0); of equal contenders the first drawn wins, as torch.argmax does.
// One lane per tournament: t_size dependent-free gathers from a vector that lives in L2 (4 MB at 1 M trees).
__global__ __launch_bounds__(256) void tournament_kernel(const float *fitness, unsigned n, unsigned n_tournaments, unsigned t_size,
                                                         unsigned long long base, int *winners) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tournaments) return;
    uint32_t best_key = 0u;
    unsigned best = 0u;
    for (unsigned k = 0; k < t_size; ++k) {
        const unsigned c = counter_word(base, 16u + k, (unsigned long long)i) % n;
        const uint32_t key = select_key(fitness[c]);
        if (k == 0u || key > best_key) { best_key = key; best = c; }
    }
    winners[i] = (int)best;
}

} // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_tournament_select(unsigned n, unsigned n_tournaments, unsigned t_size, long long seed, long long generation,
                                           const float *fitness, int *winners, evogp_stream_t stream_) {
    if (n == 0 || n_tournaments == 0 || t_size == 0 || t_size > (1u << 20)) return EVOGP_E_BADARG;
    if (!fitness || !winners) return EVOGP_E_NULLPTR;
    const unsigned long long base = counter_base(seed, generation);
    hipLaunchKernelGGL(tournament_kernel, dim3((n_tournaments + 255) / 256), dim3(256), 0, (hipStream_t)stream_, fitness, n, n_tournaments, t_size,
                       base, winners);
    return (int)hipGetLastError();
}

// scores[i] = NaN -> -inf, else -errors[i] (negate != 0) or errors[i]: what a generation does with the fitness pass's errors before it
// selects -- SymbolicRegression.evaluate's sign (problem/symbolic_regression.py:82-96) and StandardPipeline.step's NaN scrub
// (pipeline/standard.py:41-43) -- as ONE launch instead of torch's four (neg, isnan, full_like, where: 15-20 us of a 190 us generation
// at 100 k trees).
namespace evogp {
__global__ __launch_bounds__(256) void fitness_scores_kernel(const float *in, float *out, unsigned n, int negate) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    out[i] = x != x ? -__builtin_inff() : (negate ? -x : x);
}
}  // namespace evogp

extern "C" int evogp_hip_fitness_scores(unsigned n, int negate, const float *errors, float *scores, evogp_stream_t stream_) {
    if (n == 0) return EVOGP_E_BADARG;
    if (!errors || !scores) return EVOGP_E_NULLPTR;
    hipLaunchKernelGGL(evogp::fitness_scores_kernel, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream_, errors, scores, n, negate);
    return (int)hipGetLastError();
}

extern "C" size_t evogp_hip_select_workspace_bytes(void) { return (size_t)kSelWords * sizeof(unsigned); }

extern "C" int evogp_hip_select(unsigned n, unsigned n_elite, unsigned n_keep, const float *fitness, int *order, void *zeroed_workspace,
                                evogp_stream_t stream_) {
    return evogp_hip_select_alternating(n, n_elite, n_keep, fitness, order, zeroed_workspace, nullptr, stream_);
}

extern "C" int evogp_hip_select_alternating(unsigned n, unsigned n_elite, unsigned n_keep, const float *fitness, int *order,
                                            void *zeroed_workspace, void *next_workspace, evogp_stream_t stream_) {
    if (n == 0 || n_keep == 0 || n_keep > n || n_elite > n_keep) return EVOGP_E_BADARG;
    if (!fitness || !order || !zeroed_workspace) return EVOGP_E_NULLPTR;
    if (next_workspace == zeroed_workspace) return EVOGP_E_BADARG;
    SelectParams p{fitness, order, (unsigned *)zeroed_workspace, (unsigned *)next_workspace, (int)n, (int)n_elite, (int)n_keep};
    // The grid barrier needs every workgroup resident at once: never launch more than the occupancy calculator says fit
    // (register growth, partitioned modes).  EVOGP_SELECT_COOP=1 additionally asks the runtime for a cooperative launch, which
    // fails instead of hanging when the grid would not be co-resident (a CU mask the occupancy query does not see) -- opt-in,
    // because hipLaunchCooperativeKernel costs 22 us per call on this stack (100 k values: 54 instead of 32 us; 1 M: 86 instead of
    // 62; profiles/r03b_10_select_time.log), two thirds of the kernel itself.
    static const int per_cu = [] {
        int b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, select_kernel, kSelThreads, 0) != hipSuccess) b = 0;
        return b;
    }();
    static const bool coop = [] { const char *e = getenv("EVOGP_SELECT_COOP"); return e && e[0] == '1'; }();
    if (per_cu < 1) return EVOGP_E_UNSUPPORTED;
    int blocks = device_info().num_cus;
    if (blocks > kSelMaxBlocks) blocks = kSelMaxBlocks;
    const long fit = (long)per_cu * device_info().num_cus;
    if (blocks > fit) blocks = (int)fit;
    const int need = ((int)n + kSelThreads - 1) / kSelThreads;
    if (blocks > need) blocks = need;
    if (coop) {
        void *args[] = {(void *)&p};
        const hipError_t e = hipLaunchCooperativeKernel((const void *)select_kernel, dim3((unsigned)blocks), dim3(kSelThreads), args, 0,
                                                        (hipStream_t)stream_);
        if (e == hipSuccess) return EVOGP_OK;
        (void)hipGetLastError();   // not supported on this device / in this context: the clamped plain launch below
    }
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)blocks), dim3(kSelThreads), 0, (hipStream_t)stream_, p);
    return (int)hipGetLastError();
}
