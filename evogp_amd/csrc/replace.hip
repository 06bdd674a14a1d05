// replace.hip — subtree replacement: the shared primitive of mutation and crossover (gfx950).
//
// Replaces  _gpTreeReplace / treeGPMutationKernel / treeGPCrossoverKernel
// (src/evogp/cuda/mutation.cu:5-115, 118-184, 224-309).  The reference runs one THREAD per output
// tree: three serial element-wise copy loops through an 8 KB local-memory staging buffer, a serial
// root-to-node walk to find the ancestors, and row-strided (uncoalesced) loads and stores.
//
// Here one WAVE builds one output tree and every lane owns one output POSITION j, so every load and
// store is a contiguous run across the lanes:
//
//     j <  p            out[j] = left[j], size += diff if j is an ancestor of p
//     p <= j < p + m    out[j] = donor[q + j - p]                      (value, type AND size)
//     p + m <= j < len  out[j] = left[j - diff]
//     len <= j          out[j] = 0                                     (reference: uninitialised)
//
// with o = size_left[p], diff = m - o, len = S + diff.  "j is an ancestor of p" needs no tree walk:
// in prefix order it is simply  j < p < j + size_left[j]  — one compare per lane, evaluated in
// parallel (fuzz-checked against the reference's walk through oracle/_ref).
//
// HBM traffic per output tree: the live prefix of the left row, the donor subtree, and one full
// output row (8 B per node each) — the algorithmic bytes of SURVEY.md §8d.
#include "evogp_defs.hpp"
#include "launch.hpp"
#include "replace_row.hpp"
#include <cstdint>
#include <cstdlib>

namespace evogp {

struct MutateParams {
    const float *ov; const int16_t *ot; const int16_t *os; // old forest
    const int *idx;
    const float *nv; const int16_t *nt; const int16_t *ns; // new (donor) forest, whole trees
    float *rv; int16_t *rt; int16_t *rs;
    int pop, gp_len;
};

__global__ __launch_bounds__(kRepBlock) void mutate_kernel(MutateParams a) {
    const int wave = uni((int)(blockIdx.x * (kRepBlock / 64) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (kRepBlock / 64);
    for (int n = wave; n < a.pop; n += nwaves) {
        const size_t off = (size_t)n * a.gp_len;
        const Row L{a.ov + off, a.ot + off, a.os + off}, R{a.nv + off, a.nt + off, a.ns + off};
        int S = uni((int)L.s[0]);
        S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
        const int p = uni(a.idx[n]);
        const int m = uni((int)R.s[0]);
        bool fallback = p < 0 || p >= S || m < 1 || m > a.gp_len; // mutation.cu:150-160 (+ donor sanity)
        if (!fallback) fallback = S + (m - uni((int)L.s[p])) > a.gp_len; // :170-180
        build_row(L, R, S, p, 0, m, fallback, a.gp_len, a.rv + off, a.rt + off, a.rs + off);
    }
}

struct CrossParams {
    const float *v; const int16_t *t; const int16_t *s; // survivor forest
    const int *left_idx, *right_idx, *left_node, *right_node;
    float *rv; int16_t *rt; int16_t *rs;
    int pop_ori, pop_new, gp_len;
};

__global__ __launch_bounds__(kRepBlock) void crossover_kernel(CrossParams a) {
    const int wave = uni((int)(blockIdx.x * (kRepBlock / 64) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (kRepBlock / 64);
    for (int n = wave; n < a.pop_new; n += nwaves) {
        int li = uni(a.left_idx[n]);
        li = li < 0 ? 0 : (li >= a.pop_ori ? a.pop_ori - 1 : li); // the reference does not check (mutation.cu:246-248)
        const int ri = uni(a.right_idx[n]);
        const size_t lo = (size_t)li * a.gp_len, off = (size_t)n * a.gp_len;
        const Row L{a.v + lo, a.t + lo, a.s + lo};
        int S = uni((int)L.s[0]);
        S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
        const int p = uni(a.left_node[n]), q = uni(a.right_node[n]);
        bool fallback = ri < 0 || ri >= a.pop_ori; // mutation.cu:256-266
        Row R = L;
        int m = 0;
        if (!fallback) {
            const size_t ro = (size_t)ri * a.gp_len;
            R = Row{a.v + ro, a.t + ro, a.s + ro};
            const int RS = uni((int)R.s[0]);
            // node indices outside the live trees are undefined in the reference; here: copy left
            fallback = p < 0 || p >= S || q < 0 || q >= RS || q >= a.gp_len;
            if (!fallback) {
                m = uni((int)R.s[q]);
                fallback = m < 1 || q + m > a.gp_len || S + (m - uni((int)L.s[p])) > a.gp_len; // :279-289
            }
        }
        build_row(L, R, S, p, q, m, fallback, a.gp_len, a.rv + off, a.rt + off, a.rs + off);
    }
}

// ---- group variants: four output trees per wave (replace_row.hpp) ----------------------------------------------------
__global__ __launch_bounds__(kRepBlock) void mutate_group_kernel(MutateParams a) {
    constexpr int kGroups = kRepBlock / kGroupLanes;
    const int g = threadIdx.x / kGroupLanes;
    for (int base = blockIdx.x * kGroups; base < a.pop; base += gridDim.x * kGroups) {
        const int n = base + g;
        const bool active = n < a.pop;
        const size_t off = (size_t)(active ? n : 0) * a.gp_len;
        const float *Lv = a.ov + off; const int16_t *Lt = a.ot + off, *Ls = a.os + off;
        const float *Rv = a.nv + off; const int16_t *Rt = a.nt + off, *Rs = a.ns + off;
        int S = (int)Ls[0];
        S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
        const int p = a.idx[active ? n : 0];
        const int m = (int)Rs[0];
        bool fallback = p < 0 || p >= S || m < 1 || m > a.gp_len;  // mutation.cu:150-160 (+ donor sanity)
        int o = 0;
        if (!fallback) {
            o = (int)Ls[p];
            fallback = S + (m - o) > a.gp_len;  // :170-180
        }
        build_row_group(Lv, Lt, Ls, Rv, Rt, Rs, S, p, 0, m, o, fallback, active, a.gp_len, a.rv + off, a.rt + off, a.rs + off);
    }
}

__global__ __launch_bounds__(kRepBlock) void crossover_group_kernel(CrossParams a) {
    constexpr int kGroups = kRepBlock / kGroupLanes;
    const int g = threadIdx.x / kGroupLanes;
    for (int base = blockIdx.x * kGroups; base < a.pop_new; base += gridDim.x * kGroups) {
        const int n = base + g;
        const bool active = n < a.pop_new;
        const int nn = active ? n : 0;
        int li = a.left_idx[nn];
        li = li < 0 ? 0 : (li >= a.pop_ori ? a.pop_ori - 1 : li);  // the reference does not check (mutation.cu:246-248)
        const int ri = a.right_idx[nn];
        const int p = a.left_node[nn], q = a.right_node[nn];
        bool fallback = ri < 0 || ri >= a.pop_ori;  // mutation.cu:256-266
        const size_t lo = (size_t)li * a.gp_len, ro = (size_t)(fallback ? li : ri) * a.gp_len, off = (size_t)nn * a.gp_len;
        const float *Lv = a.v + lo; const int16_t *Lt = a.t + lo, *Ls = a.s + lo;
        const float *Rv = a.v + ro; const int16_t *Rt = a.t + ro, *Rs = a.s + ro;
        int S = (int)Ls[0];
        S = S < 0 ? 0 : (S > a.gp_len ? a.gp_len : S);
        const int RS = (int)Rs[0];
        int m = 0, o = 0;
        // node indices outside the live trees are undefined in the reference; here: copy left
        fallback = fallback || p < 0 || p >= S || q < 0 || q >= RS || q >= a.gp_len;
        if (!fallback) {
            m = (int)Rs[q];
            o = (int)Ls[p];
            fallback = m < 1 || q + m > a.gp_len || S + (m - o) > a.gp_len;  // :279-289
        }
        build_row_group(Lv, Lt, Ls, Rv, Rt, Rs, S, p, q, m, o, fallback, active, a.gp_len, a.rv + off, a.rt + off, a.rs + off);
    }
}

// the group kernels store 16 / 8 bytes at a time: rows must start on such boundaries
static bool group_ok(int gp_len, const void *rv, const void *rt, const void *rs) {
    static const bool enabled = [] { const char *e = getenv("EVOGP_REPLACE_GROUPS"); return !(e && e[0] == '0'); }();
    return enabled && gp_len % 4 == 0 && (uintptr_t)rv % 16 == 0 && (uintptr_t)rt % 8 == 0 && (uintptr_t)rs % 8 == 0;
}

static unsigned grid_for_groups(long trees) {
    const DeviceInfo &dev = device_info();
    const long per_block = kRepBlock / kGroupLanes;
    long blocks = (trees + per_block - 1) / per_block;
    const long cap = (long)dev.num_cus * 8 * 4;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

static unsigned grid_for(long trees) {
    const DeviceInfo &dev = device_info();
    long blocks = (trees + (kRepBlock / 64) - 1) / (kRepBlock / 64);
    const long cap = (long)dev.num_cus * 8 * 4; // persistent beyond 4 rounds of full occupancy
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

} // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_mutate(int pop_size, int gp_len, const float *value_ori, const int16_t *type_ori,
                                const int16_t *size_ori, const int *mutate_indices, const float *value_new,
                                const int16_t *type_new, const int16_t *size_new, float *value_res,
                                int16_t *type_res, int16_t *size_res, evogp_stream_t stream_) {
    if (pop_size <= 0 || gp_len <= 0 || gp_len > kMaxStack) return EVOGP_E_BADARG; // torch_wrapper.cu:103-104
    if (!value_ori || !type_ori || !size_ori || !mutate_indices || !value_new || !type_new || !size_new || !value_res ||
        !type_res || !size_res)
        return EVOGP_E_NULLPTR;
    MutateParams a{value_ori, type_ori, size_ori, mutate_indices, value_new, type_new, size_new,
                   value_res, type_res, size_res, pop_size, gp_len};
    if (group_ok(gp_len, value_res, type_res, size_res)) hipLaunchKernelGGL(mutate_group_kernel, dim3(grid_for_groups(pop_size)), dim3(kRepBlock), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(mutate_kernel, dim3(grid_for(pop_size)), dim3(kRepBlock), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

extern "C" int evogp_hip_crossover(int pop_size_ori, int pop_size_new, int gp_len, const float *value_ori,
                                   const int16_t *type_ori, const int16_t *size_ori, const int *left_idx,
                                   const int *right_idx, const int *left_node_idx, const int *right_node_idx,
                                   float *value_res, int16_t *type_res, int16_t *size_res, evogp_stream_t stream_) {
    if (pop_size_ori <= 0 || pop_size_new <= 0 || gp_len <= 0 || gp_len > kMaxStack) return EVOGP_E_BADARG; // :154-156
    if (!value_ori || !type_ori || !size_ori || !left_idx || !right_idx || !left_node_idx || !right_node_idx ||
        !value_res || !type_res || !size_res)
        return EVOGP_E_NULLPTR;
    CrossParams a{value_ori, type_ori, size_ori, left_idx, right_idx, left_node_idx, right_node_idx,
                  value_res, type_res, size_res, pop_size_ori, pop_size_new, gp_len};
    if (group_ok(gp_len, value_res, type_res, size_res)) hipLaunchKernelGGL(crossover_group_kernel, dim3(grid_for_groups(pop_size_new)), dim3(kRepBlock), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(crossover_kernel, dim3(grid_for(pop_size_new)), dim3(kRepBlock), 0, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}
