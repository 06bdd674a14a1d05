// breed_group.hpp — the four-rows-per-wave breeding pass (breed.hip) as a body with a hook behind every chunk of 64 rows.
// (Rounds 3-4 instantiated it a second time with a hook that compiled the rows just written into the program records of the
// next fitness call; measured, nothing gained, removed in round 5: docs/DESIGN_history_r01_r03.md section 3.5a.)
#pragma once
#include "replace_row.hpp"
#include <hip/hip_runtime.h>

namespace evogp {

struct BreedParams {
    const float *v; const int16_t *t; const int16_t *s;    // current generation [pop][gp_len]
    const int *order;                                        // [n_elite] rows copied unchanged (the elites)
    const int *parents;                                      // [n_surv] rows the parents are drawn from (repeats allowed)
    const int *rnd;                                          // [6][n_new] raw words in [0, 2^31 - 1)
    const float *dv; const int16_t *dt; const int16_t *ds;  // donors [n_new][gp_len] (rows of mutating offspring only)
    float *ov; int16_t *ot; int16_t *os;                    // next generation [pop][gp_len]
    int *decisions;                                          // optional [n_new][6]: left, right, p, q, mutated, mutate position
    int pop, gp_len, n_elite, n_surv, n_new;
    int table_rows;  // rows of v/t/s: the whole population, or only the trees `order` can name (a sharded run's survivor table)
    unsigned mutate_below;
    int row_begin, row_count;  // rows [row_begin, row_begin + row_count) of the next generation are built; output and donor
                               // arrays hold exactly these rows (donor row k belongs to next-generation row row_begin + k)
    int hashed;                // != 0: no rnd array: word k of offspring i is counter_word(hash_base, k, i) (evogp_defs.hpp)
    unsigned long long hash_base;
    int chunks_per_unit;       // 1: a workgroup decides and builds one chunk of 64 rows at a time; 4: every wave decides a chunk,
                               // then the workgroup builds the four (large launches: the decision chains run four abreast)
};

constexpr int kBreedUnit = kRepBlock / 64;   // chunks a workgroup can decide at once: one per wave

struct NoBreedHook {
    __device__ inline void chunk_done(const BreedParams &, int, int) const {}
};

// ---- four rows per wave -------------------------------------------------------------------------------------------------
// Same two phases, but BUILD works in groups of 16 lanes (replace_row.hpp): the workgroup's 16 groups build 16 rows at a
// time, so a chunk of 64 rows is four steps of dependent memory round trips instead of sixteen.  A mutated child is staged
// in the group's own LDS row (8 bytes per node) between the two replacements.
// DECIDE is a chain of four dependent loads during which three of the four waves wait -- a third of a chunk's life.  In
// launches with many chunks per workgroup (chunks_per_unit = 4) every wave decides a chunk of its own, so four chains run
// abreast, and the workgroup then builds the four chunks; small launches keep one chunk per workgroup at a time (there are
// not enough chunks to fill the chip otherwise).  `all_dec`: kBreedUnit tables of [10][64] decisions.
template <class Hook>
__device__ inline void breed_group_body(const BreedParams &a, unsigned char *breed_lds, int (*all_dec)[10][64], Hook &hook) {
    constexpr int kGroups = kRepBlock / kGroupLanes;
    const int w = uni((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x / kGroupLanes;           // group in the workgroup
    const int gl = threadIdx.x & (kGroupLanes - 1);
    unsigned char *mine = breed_lds + (size_t)g * a.gp_len * 8;
    float *cv = (float *)mine;
    int16_t *ct = (int16_t *)(mine + (size_t)a.gp_len * 4);
    int16_t *cs = ct + a.gp_len;
    const int nchunks = (a.row_count + 63) >> 6;
    const int row_end = a.row_begin + a.row_count;
    const int cpu = uni(a.chunks_per_unit > 1 ? kBreedUnit : 1);
    const int nunits = (nchunks + cpu - 1) / cpu;
    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
      {
        // ---- DECIDE: wave w decides chunk u * cpu + w (one chunk per unit: wave 0 alone), as in breed_kernel ----
        const int c = u * cpu + w;
        const bool decider = w < cpu && c < nchunks;
        int (*dec_s)[64] = all_dec[w < cpu ? w : 0];
        const int n0 = a.row_begin + (c << 6);
        const int n = n0 + lane;
        int li = 0, ri = 0, S = 0, p = 0, q = 0, m = 0, o = 0, dm = 0;
        unsigned r5 = 0;
        bool fallback = true, mutating = false;
        if (decider && n < row_end) {
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
        if (decider) {
            dec_s[0][lane] = li; dec_s[1][lane] = ri; dec_s[2][lane] = S; dec_s[3][lane] = p; dec_s[4][lane] = q;
            dec_s[5][lane] = m; dec_s[6][lane] = o; dec_s[7][lane] = dm; dec_s[8][lane] = (int)r5;
            dec_s[9][lane] = (fallback ? 1 : 0) | (mutating ? 2 : 0);
        }
      }
        __syncthreads();
      for (int cc = 0; cc < cpu; ++cc) {
        const int c = u * cpu + cc;
        if (c >= nchunks) break;
        int (*dec_s)[64] = all_dec[cc];
        const int n0 = a.row_begin + (c << 6);
        // ---- BUILD: 16 rows per step ----
        const int rows = row_end - n0 < 64 ? row_end - n0 : 64;
        for (int l0 = 0; l0 < rows; l0 += kGroups) {
            const int l = l0 + g;
            const bool active = l < rows;
            const int lc = active ? l : 0;
            const int nn = n0 + lc;
            const size_t off = (size_t)(nn - a.row_begin) * a.gp_len;
            const int li_ = dec_s[0][lc], ri_ = dec_s[1][lc], S_ = dec_s[2][lc], p_ = dec_s[3][lc], q_ = dec_s[4][lc],
                      m_ = dec_s[5][lc], o_ = dec_s[6][lc], flags = dec_s[9][lc];
            const size_t lo = (size_t)li_ * a.gp_len, ro = (size_t)ri_ * a.gp_len;
            const bool fb = (flags & 1) != 0, mu = active && (flags & 2) != 0;
            // the child of the crossover goes straight to its row, or to the group's LDS row when it mutates next
            float *tv = mu ? cv : a.ov + off;
            int16_t *tt = mu ? ct : a.ot + off, *ts = mu ? cs : a.os + off;
            build_row_group(a.v + lo, a.t + lo, a.s + lo, a.v + ro, a.t + ro, a.s + ro, S_, p_, q_, m_, o_, fb, active,
                            a.gp_len, tv, tt, ts);
            int pm = -1;
            if (__any(mu)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (mu) {
                    int CS = (int)cs[0];
                    CS = CS < 0 ? 0 : (CS > a.gp_len ? a.gp_len : CS);
                    const unsigned r5_ = (unsigned)dec_s[8][lc];
                    pm = CS > 0 ? (int)((r5_ % (unsigned)kMaxStack) % (unsigned)CS) : 0;  // mutation/default.py:59-66
                    const int dm_ = dec_s[7][lc];
                    bool mfall = CS <= 0 || dm_ < 1 || dm_ > a.gp_len;                   // mutation.cu:150-160 (+ donor sanity)
                    int co = 0;
                    if (!mfall) {
                        co = (int)cs[pm];
                        mfall = CS + (dm_ - co) > a.gp_len;                              // :170-180
                    }
                    build_row_group(cv, ct, cs, a.dv + off, a.dt + off, a.ds + off, CS, pm, 0, dm_, co, mfall, true, a.gp_len,
                                    a.ov + off, a.ot + off, a.os + off);
                }
                __builtin_amdgcn_wave_barrier();  // the staging rows are rewritten in the next step
            }
            if (a.decisions && active && gl == 0 && nn >= a.n_elite) {
                int *d = a.decisions + (size_t)(nn - a.row_begin) * 6;
                d[0] = li_; d[1] = ri_; d[2] = p_; d[3] = q_; d[4] = mu ? 1 : 0; d[5] = pm;
            }
        }
      }
        __syncthreads();  // the next unit's decisions overwrite the tables; every row of the unit's chunks is written
      for (int cc = 0; cc < cpu; ++cc) {
        const int c = u * cpu + cc;
        if (c >= nchunks) break;
        const int n0 = a.row_begin + (c << 6);
        hook.chunk_done(a, n0, row_end - n0 < 64 ? row_end - n0 : 64);
      }
    }
}


} // namespace evogp
