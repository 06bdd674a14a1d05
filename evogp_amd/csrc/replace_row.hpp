// replace_row.hpp — the subtree-replacement row builder shared by replace.hip (tree_mutate / tree_crossover) and
// breed.hip (the fused default generation step).  See replace.hip for the derivation.
#pragma once
#include "evogp_defs.hpp"

namespace evogp {

constexpr int kRepBlock = 256; // 4 waves, one output tree per wave per iteration

struct Row {
    const float *v;
    const int16_t *t;
    const int16_t *s;
};

// Build one output row.  fallback => copy the left tree.  All arguments are wave-uniform.
// `o_known` >= 0: size of the replaced subtree (L.s[p]) when the caller has it already, else it is read here.
__device__ inline void build_row(const Row &L, const Row &R, int S, int p, int q, int m, bool fallback, int gp_len,
                                 float *ov, int16_t *ot, int16_t *os, int o_known = -1) {
    const int lane = threadIdx.x & 63;
    int o = 0, diff = 0;
    if (fallback) { p = S; m = 0; q = 0; } // "everything is the untouched prefix"
    else { o = o_known >= 0 ? o_known : uni((int)L.s[p]); diff = m - o; }
    const int len = S + diff;
    for (int j = lane; j < gp_len; j += kWave) {
        float v = 0.0f;
        int t = 0, s = 0;
        if (j < p) {
            v = L.v[j]; t = L.t[j]; s = L.s[j];
            if (j + s > p) s += diff; // ancestor of the replaced node (mutation.cu:38-88)
        } else if (j < p + m) {
            const int k = q + (j - p);
            v = R.v[k]; t = R.t[k]; s = R.s[k];
        } else if (j < len) {
            const int k = j - diff;
            v = L.v[k]; t = L.t[k]; s = L.s[k];
        }
        ov[j] = v; ot[j] = (int16_t)t; os[j] = (int16_t)s;
    }
}

// ---- four rows per wave ------------------------------------------------------------------------------------------------
// A wave that builds ONE row per iteration spends its life waiting: the parents' indices, their sizes, the subtree sizes
// and only then the row itself are dependent memory round trips, and the chip holds 8 k waves for 100 k rows.  Here a
// group of 16 lanes owns a row and every lane four consecutive output positions: the four rows of a wave chase their
// chains together, the gathers of a lane (4 positions x value/type/size) are all in flight before the first is used, and
// the row leaves as 16-byte (values) and 8-byte (types, sizes) stores.  All arguments are per-lane values that agree
// within a group.  Needs gp_len % 4 == 0 (row bases are then 16- / 8-byte aligned).
constexpr int kGroupLanes = 16;
constexpr int kGroupSpan = kGroupLanes * 4;  // output positions a group covers per step

__device__ inline void build_row_group(const float *Lv, const int16_t *Lt, const int16_t *Ls, const float *Rv,
                                       const int16_t *Rt, const int16_t *Rs, int S, int p, int q, int m, int o,
                                       bool fallback, bool active, int gp_len, float *ov, int16_t *ot, int16_t *os) {
    const int gl = threadIdx.x & (kGroupLanes - 1);
    int diff = 0;
    if (fallback) { p = S; m = 0; q = 0; }  // "everything is the untouched prefix"
    else diff = m - o;
    const int len = S + diff;
    for (int j0 = gl * 4; j0 < gp_len; j0 += kGroupSpan) {
        float v[4];
        int t[4], sz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // all twelve gathers first
            const int j = j0 + k;
            const bool pre = j < p, don = !pre && j < p + m;
            int idx = pre ? j : (don ? q + (j - p) : j - diff);
            idx = j < len ? idx : 0;
            idx = idx < 0 ? 0 : (idx >= gp_len ? gp_len - 1 : idx);
            v[k] = (don ? Rv : Lv)[idx];
            t[k] = (don ? Rt : Lt)[idx];
            sz[k] = (don ? Rs : Ls)[idx];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k;
            if (j < p && j + sz[k] > p) sz[k] += diff;  // ancestor of the replaced node (mutation.cu:38-88)
            if (j >= len) { v[k] = 0.0f; t[k] = 0; sz[k] = 0; }
        }
        if (active) {
            *reinterpret_cast<float4 *>(ov + j0) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<uint2 *>(ot + j0) = make_uint2(((uint32_t)t[0] & 0xFFFFu) | ((uint32_t)t[1] << 16),
                                                               ((uint32_t)t[2] & 0xFFFFu) | ((uint32_t)t[3] << 16));
            *reinterpret_cast<uint2 *>(os + j0) = make_uint2(((uint32_t)sz[0] & 0xFFFFu) | ((uint32_t)sz[1] << 16),
                                                               ((uint32_t)sz[2] & 0xFFFFu) | ((uint32_t)sz[3] << 16));
        }
    }
}

} // namespace evogp
