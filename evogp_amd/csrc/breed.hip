// breed.hip — the default generation step as ONE pass over the next population (gfx950).
//
// Fuses what  GeneticProgramming.step  does with the default operators
// (src/evogp/algorithm/genetic_programming.py:105-124, selection/default.py:42-71, crossover/default.py:16-66,
// mutation/default.py:32-75) after the fitness sort:
//
//     next[0 .. e)        = forest[order[0 .. e)]                                           elites
//     child_i             = crossover(forest[order[r0 % s]], p = r2 % size_left,
//                                     forest[order[r1 % s]], q = r3 % size_right)           i in [0, pop - e)
//     next[e + i]         = r4 < mutate_below ? mutate(child_i, (r5 % 1024) % size(child_i), donor_i) : child_i
//
// `order` is the descending fitness order, s the number of survivors, r0..r5 six raw 31-bit random words per offspring
// (drawn by ONE torch.randint), donor_i the tree that tree_generate produces for tree index i (evogp_hip_generate_masked
// writes only the rows whose r4 word says "mutate").  The reference composes this from a gather of the survivors, four
// index tensors, tree_crossover, a CPU Bernoulli mask, boolean-mask gathers and scatters, tree_generate, tree_mutate and
// three concatenations — about 90 kernel launches and two host syncs per generation in PyTorch.  The subtree surgery is
// the row builder of replace.hip (same fallback rules: mutation.cu:150-160,170-180,256-266,279-289); a mutated child
// is staged in LDS between the two replacements, so every output row is written exactly once.
//
// HBM traffic per offspring: the live prefixes of both parents' rows, the donor (20 % of the
// rows), one full output row.
#include "evogp_defs.hpp"
#include "launch.hpp"
#include "breed_group.hpp"
#include <cstdint>
#include <cstdlib>

namespace evogp {


// A workgroup (4 waves) takes 64 consecutive output rows in two phases.  DECIDE: lane l of wave 0 owns row n0 + l and
// chases the dependent loads of its decisions (random words -> order[] -> tree sizes -> subtree sizes -> donor size):
// 64 chains in flight instead of one per wave.  BUILD: every wave builds every fourth row with all arguments ready in
// LDS — two memory round trips per row (loads, stores) instead of six.
__global__ __launch_bounds__(kRepBlock) void breed_kernel(BreedParams a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char breed_lds[];
    const int w = uni((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // this wave's staging row: value f32[gp_len] | type i16[gp_len] | size i16[gp_len]
    unsigned char *mine = breed_lds + (size_t)w * a.gp_len * 8;
    float *cv = (float *)mine;
    int16_t *ct = (int16_t *)(mine + (size_t)a.gp_len * 4);
    int16_t *cs = ct + a.gp_len;
    __shared__ int dec_s[10][64];  // decisions of the chunk's 64 rows
    const int nchunks = (a.row_count + 63) >> 6;
    const int row_end = a.row_begin + a.row_count;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int n0 = a.row_begin + (c << 6);
        // ---- DECIDE (wave 0) ----
        const int n = n0 + lane;
        int li = 0, ri = 0, S = 0, p = 0, q = 0, m = 0, o = 0, dm = 0;
        unsigned r5 = 0;
        bool fallback = true, mutating = false;
        if (w == 0 && n < row_end) {
            if (n < a.n_elite) {
                li = a.order[n];
                li = li < 0 ? 0 : (li >= a.table_rows ? a.table_rows - 1 : li);
                ri = li;
                S = (int)a.s[(size_t)li * a.gp_len];
                S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
            } else {
                const int i = n - a.n_elite;
                unsigned r0, r1, r2, r3, r4;
                if (a.hashed) {
                    r0 = counter_word(a.hash_base, 0u, (unsigned long long)i); r1 = counter_word(a.hash_base, 1u, (unsigned long long)i);
                    r2 = counter_word(a.hash_base, 2u, (unsigned long long)i); r3 = counter_word(a.hash_base, 3u, (unsigned long long)i);
                    r4 = counter_word(a.hash_base, 4u, (unsigned long long)i); r5 = counter_word(a.hash_base, 5u, (unsigned long long)i);
                } else {
                    r0 = (unsigned)a.rnd[i]; r1 = (unsigned)a.rnd[a.n_new + i]; r2 = (unsigned)a.rnd[2 * a.n_new + i];
                    r3 = (unsigned)a.rnd[3 * a.n_new + i]; r4 = (unsigned)a.rnd[4 * a.n_new + i];
                    r5 = (unsigned)a.rnd[5 * a.n_new + i];
                }
                li = a.parents[r0 % (unsigned)a.n_surv];
                ri = a.parents[r1 % (unsigned)a.n_surv];
                li = li < 0 ? 0 : (li >= a.table_rows ? a.table_rows - 1 : li);
                ri = ri < 0 ? 0 : (ri >= a.table_rows ? a.table_rows - 1 : ri);
                const int16_t *ls = a.s + (size_t)li * a.gp_len, *rs = a.s + (size_t)ri * a.gp_len;
                S = (int)ls[0];
                int RS = (int)rs[0];
                S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
                RS = RS < 0 ? 0 : (RS > a.gp_len ? a.gp_len : RS);
                // positions: u % tree_size (crossover/default.py:49-58); an empty tree falls back to a copy
                p = S > 0 ? (int)(r2 % (unsigned)S) : 0;
                q = RS > 0 ? (int)(r3 % (unsigned)RS) : 0;
                fallback = S <= 0 || RS <= 0;
                if (!fallback) {
                    m = (int)rs[q];
                    o = (int)ls[p];
                    fallback = m < 1 || q + m > a.gp_len || S + (m - o) > a.gp_len;  // mutation.cu:279-289
                }
                mutating = r4 < a.mutate_below;
                if (mutating) dm = (int)a.ds[(size_t)(n - a.row_begin) * a.gp_len];
            }
        }
        if (w == 0) {
            dec_s[0][lane] = li; dec_s[1][lane] = ri; dec_s[2][lane] = S; dec_s[3][lane] = p; dec_s[4][lane] = q;
            dec_s[5][lane] = m; dec_s[6][lane] = o; dec_s[7][lane] = dm; dec_s[8][lane] = (int)r5;
            dec_s[9][lane] = (fallback ? 1 : 0) | (mutating ? 2 : 0);
        }
        __syncthreads();
        // ---- BUILD (every wave takes every fourth row of the chunk) ----
        const int rows = row_end - n0 < 64 ? row_end - n0 : 64;
        for (int l = w; l < rows; l += kRepBlock / 64) {
            const int nn = n0 + l;
            const size_t off = (size_t)(nn - a.row_begin) * a.gp_len;
            const int li_ = uni(dec_s[0][l]), ri_ = uni(dec_s[1][l]), S_ = uni(dec_s[2][l]);
            const size_t lo = (size_t)li_ * a.gp_len, ro = (size_t)ri_ * a.gp_len;
            const Row L{a.v + lo, a.t + lo, a.s + lo}, R{a.v + ro, a.t + ro, a.s + ro};
            const int flags = uni(dec_s[9][l]);
            const bool fb = (flags & 1) != 0, mu = (flags & 2) != 0;
            const int p_ = uni(dec_s[3][l]), q_ = uni(dec_s[4][l]), m_ = uni(dec_s[5][l]), o_ = uni(dec_s[6][l]);
            int pm = -1;
            if (!mu) {
                build_row(L, R, S_, p_, q_, m_, fb, a.gp_len, a.ov + off, a.ot + off, a.os + off, o_);
            } else {
                build_row(L, R, S_, p_, q_, m_, fb, a.gp_len, cv, ct, cs, o_);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const size_t doff = off;
                const Row C{cv, ct, cs}, D{a.dv + doff, a.dt + doff, a.ds + doff};
                int CS = uni((int)C.s[0]);
                CS = CS < 0 ? 0 : (CS > a.gp_len ? a.gp_len : CS);
                const unsigned r5_ = (unsigned)uni(dec_s[8][l]);
                pm = CS > 0 ? (int)((r5_ % (unsigned)kMaxStack) % (unsigned)CS) : 0;  // mutation/default.py:59-66
                const int dm_ = uni(dec_s[7][l]);
                bool mfall = CS <= 0 || dm_ < 1 || dm_ > a.gp_len;                   // mutation.cu:150-160 (+ donor sanity)
                if (!mfall) mfall = CS + (dm_ - uni((int)C.s[pm])) > a.gp_len;       // :170-180
                build_row(C, D, CS, pm, 0, dm_, mfall, a.gp_len, a.ov + off, a.ot + off, a.os + off);
                __builtin_amdgcn_wave_barrier();  // the staging row is rewritten by this wave's next mutating offspring
            }
            if (a.decisions && lane == 0 && nn >= a.n_elite) {
                int *d = a.decisions + (size_t)(nn - a.row_begin) * 6;
                d[0] = li_; d[1] = ri_; d[2] = p_; d[3] = q_; d[4] = mu ? 1 : 0; d[5] = pm;
            }
        }
        __syncthreads();  // the next chunk's decisions overwrite dec_s
    }
}

__global__ __launch_bounds__(kRepBlock) void breed_group_kernel(BreedParams a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char breed_lds[];
    __shared__ int dec_s[kBreedUnit][10][64];
    NoBreedHook hook;
    breed_group_body(a, breed_lds, dec_s, hook);
}

} // namespace evogp

using namespace evogp;

static int breed_impl(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv, const float *value, const int16_t *type,
                      const int16_t *size, const int *elite_rows, const int *parent_rows, const int *rnd, int hashed,
                      unsigned long long hash_base, unsigned mutate_below, const float *donor_value, const int16_t *donor_type,
                      const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res, int *decisions, int row_begin,
                      int row_count, evogp_stream_t stream_);

extern "C" int evogp_hip_breed_default(int pop_size, int gp_len, int n_elite, int n_surv, const float *value,
                                       const int16_t *type, const int16_t *size, const int *order, const int *rnd,
                                       unsigned mutate_below, const float *donor_value, const int16_t *donor_type,
                                       const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res,
                                       int *decisions, evogp_stream_t stream_) {
    return evogp_hip_breed_default_rows(pop_size, gp_len, n_elite, n_surv, value, type, size, order, rnd, mutate_below,
                                        donor_value ? donor_value - (size_t)n_elite * gp_len : nullptr,
                                        donor_type ? donor_type - (size_t)n_elite * gp_len : nullptr,
                                        donor_size ? donor_size - (size_t)n_elite * gp_len : nullptr, value_res, type_res,
                                        size_res, decisions ? decisions - (size_t)n_elite * 6 : nullptr, 0, pop_size, stream_);
}

extern "C" int evogp_hip_breed_default_rows(int pop_size, int gp_len, int n_elite, int n_surv, const float *value,
                                            const int16_t *type, const int16_t *size, const int *order, const int *rnd,
                                            unsigned mutate_below, const float *donor_value, const int16_t *donor_type,
                                            const int16_t *donor_size, float *value_res, int16_t *type_res,
                                            int16_t *size_res, int *decisions, int row_begin, int row_count,
                                            evogp_stream_t stream_) {
    return evogp_hip_breed_default_table(pop_size, pop_size, gp_len, n_elite, n_surv, value, type, size, order, rnd, mutate_below,
                                         donor_value, donor_type, donor_size, value_res, type_res, size_res, decisions, row_begin,
                                         row_count, stream_);
}

extern "C" int evogp_hip_breed_default_table(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv,
                                             const float *value, const int16_t *type, const int16_t *size, const int *order,
                                             const int *rnd, unsigned mutate_below, const float *donor_value,
                                             const int16_t *donor_type, const int16_t *donor_size, float *value_res,
                                             int16_t *type_res, int16_t *size_res, int *decisions, int row_begin, int row_count,
                                             evogp_stream_t stream_) {
    // elites and parents are both prefixes of one ranking
    if (n_surv > pop_size) return EVOGP_E_BADARG;
    return evogp_hip_breed_lists(pop_size, table_rows, gp_len, n_elite, n_surv, value, type, size, order, order, rnd, mutate_below,
                                 donor_value, donor_type, donor_size, value_res, type_res, size_res, decisions, row_begin, row_count,
                                 stream_);
}

extern "C" int evogp_hip_breed_lists(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv, const float *value,
                                     const int16_t *type, const int16_t *size, const int *elite_rows, const int *parent_rows,
                                     const int *rnd, unsigned mutate_below, const float *donor_value, const int16_t *donor_type,
                                     const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res,
                                     int *decisions, int row_begin, int row_count, evogp_stream_t stream_) {
    return breed_impl(pop_size, table_rows, gp_len, n_elite, n_surv, value, type, size, elite_rows, parent_rows, rnd, 0, 0ull, mutate_below,
                      donor_value, donor_type, donor_size, value_res, type_res, size_res, decisions, row_begin, row_count, stream_);
}


extern "C" int evogp_hip_breed_lists_hashed(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv, const float *value,
                                            const int16_t *type, const int16_t *size, const int *elite_rows, const int *parent_rows,
                                            long long seed, long long generation, unsigned mutate_below, const float *donor_value,
                                            const int16_t *donor_type, const int16_t *donor_size, float *value_res, int16_t *type_res,
                                            int16_t *size_res, int *decisions, int row_begin, int row_count, evogp_stream_t stream_) {
    return breed_impl(pop_size, table_rows, gp_len, n_elite, n_surv, value, type, size, elite_rows, parent_rows, nullptr, 1,
                      counter_base(seed, generation), mutate_below, donor_value, donor_type, donor_size, value_res, type_res, size_res, decisions,
                      row_begin, row_count, stream_);
}

static int breed_impl(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv, const float *value, const int16_t *type,
                      const int16_t *size, const int *elite_rows, const int *parent_rows, const int *rnd, int hashed,
                      unsigned long long hash_base, unsigned mutate_below, const float *donor_value, const int16_t *donor_type,
                      const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res, int *decisions, int row_begin,
                      int row_count, evogp_stream_t stream_) {
    const int *order = elite_rows;
    // n_surv may exceed pop_size: a selection that draws with replacement may name more parents than there are trees
    if (pop_size <= 0 || table_rows <= 0 || gp_len <= 0 || gp_len > kMaxStack || n_elite < 0 || n_elite > pop_size || n_surv <= 0)
        return EVOGP_E_BADARG;
    if (!value || !type || !size || !parent_rows || (n_elite > 0 && !order) || !value_res || !type_res || !size_res) return EVOGP_E_NULLPTR;
    const int n_new = pop_size - n_elite;
    if (n_new > 0 && !rnd && !hashed) return EVOGP_E_NULLPTR;
    if (mutate_below != 0 && n_new > 0 && (!donor_value || !donor_type || !donor_size)) return EVOGP_E_NULLPTR;
    if (row_begin < 0 || row_count <= 0 || row_begin + row_count > pop_size) return EVOGP_E_BADARG;
    BreedParams a{value, type, size, order, parent_rows, rnd, donor_value, donor_type, donor_size, value_res, type_res, size_res,
                  decisions, pop_size, gp_len, n_elite, n_surv, n_new, table_rows, mutate_below, row_begin, row_count, hashed, hash_base, 1};
    const DeviceInfo &dev = device_info();
    long blocks = ((long)row_count + 63) / 64;  // one workgroup per 64 rows
    const long cap = (long)dev.num_cus * 8 * 4;
    if (blocks > cap) blocks = cap;
    static const bool groups_on = [] { const char *e = getenv("EVOGP_REPLACE_GROUPS"); return !(e && e[0] == '0'); }();
    // EVOGP_BREED_UNIT=4: every wave decides a chunk and the workgroup builds the four (breed_group.hpp).  Measured SLOWER: 1 M rows
    // 379 against 343 us, 500 k 203 against 152 (profiles/r03k_breed_unit.log) -- a chunk takes ~45 us under load, not the ~20 us
    // of its idle round trips: the pass is bound by the memory system's rate of scattered 128- / 256-byte row reads (2.6 TB/s),
    // not by the decision chain, and fewer, longer work units only balance worse.  Kept as a switch, default one chunk at a time.
    static const int env_unit = [] { const char *e = getenv("EVOGP_BREED_UNIT"); return e ? atoi(e) : 0; }();
    a.chunks_per_unit = env_unit == 4 ? 4 : 1;
    if (a.chunks_per_unit > 1) {
        const long units = (((long)row_count + 63) / 64 + kBreedUnit - 1) / kBreedUnit;
        if (blocks > units) blocks = units;
    }
    const size_t lds_groups = (size_t)(kRepBlock / kGroupLanes) * gp_len * 8;
    if (groups_on && gp_len % 4 == 0 && lds_groups <= 48 * 1024 && (uintptr_t)value_res % 16 == 0 && (uintptr_t)type_res % 8 == 0 &&
        (uintptr_t)size_res % 8 == 0) {
        hipLaunchKernelGGL(breed_group_kernel, dim3((unsigned)blocks), dim3(kRepBlock), lds_groups, (hipStream_t)stream_, a);
        return (int)hipGetLastError();
    }
    const size_t lds = (size_t)(kRepBlock / 64) * gp_len * 8;
    hipLaunchKernelGGL(breed_kernel, dim3((unsigned)blocks), dim3(kRepBlock), lds, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

// ---- counter-based random words -----------------------------------------------------------------------------------------------
// Word k of offspring i of generation g is a hash of (seed, g, k, i): the splitmix64 finaliser, the same arithmetic as
// evogp_amd/parallel.py random_words (which serves the CPU paths and the tests).  A rank of a sharded run fills exactly the
// columns of its own offspring; every rank computes the same word for the same offspring whatever the world size.
namespace evogp {
__global__ void random_words_kernel(unsigned long long base, int rows, long long n_cols, long long lo, long long hi, int *out) {
    const long long n = hi - lo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n * rows; e += (long long)gridDim.x * blockDim.x) {
        const long long k = e / n, i = lo + (e - k * n);
        out[k * n_cols + i] = (int)counter_word(base, (unsigned)k, (unsigned long long)i);
    }
}
}  // namespace evogp

extern "C" int evogp_hip_random_words(long long seed, long long generation, int rows, long long n_cols, long long lo, long long hi,
                                      int *out, evogp_stream_t stream) {
    if (rows <= 0 || n_cols <= 0 || lo < 0 || hi > n_cols || lo > hi) return EVOGP_E_BADARG;
    if (!out) return EVOGP_E_NULLPTR;
    if (hi == lo) return EVOGP_OK;
    const unsigned long long base = evogp::counter_base(seed, generation);
    const long long n = (hi - lo) * rows;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(evogp::random_words_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, base, rows, n_cols, lo, hi, out);
    return (int)hipGetLastError();
}
