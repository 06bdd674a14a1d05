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

} // namespace evogp
