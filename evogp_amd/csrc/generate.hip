// generate.hip — random tree generation (gfx950).
//
// Replaces  generate / treeGPGenerate  (src/evogp/cuda/generate.cu:16-233).  The draw order of the
// per-tree taus88 stream is part of the contract (trees must be bit-identical for identical keys),
// so the depth-first expansion itself is inherently serial per tree and stays one LANE per tree.
// What is different from the reference:
//
//   * no 12 KB per-thread local-memory arrays: the pending-children stack holds at most one frame
//     per depth (depths strictly increase from bottom to top and depth <= 10), and subtree sizes are
//     produced ON THE FLY instead of by a second reverse pass with a size stack — in prefix order a
//     new node at depth d closes every still-open node of depth >= d, whose size is then
//     (index of the new node) - (its own index).  Both per-lane structures are 11 entries and live
//     in LDS ([entry][lane], conflict-free);
//   * nodes are written straight to their final place (4+2 B per node, size 2 B when the node
//     closes); every byte of the output rows is written exactly once: the tail [len, gp_len) is
//     zero-filled by the whole wave with coalesced stores (the reference leaves it uninitialised,
//     torch_wrapper.cu:64-66) so results are deterministic and comparable across runs and shards;
//   * `tree_index_offset` is added to the tree index before it is hashed into the seed
//     (generate.cu:35,40 hash the local index), so a population sharded over several GPUs is
//     bit-identical to the single-GPU one.
#include "evogp_defs.hpp"
#include "launch.hpp"

namespace evogp {

constexpr int kGenBlock = 256;
constexpr int kLevels = kMaxFullDepth + 2; // depths 0..10 plus one spare

struct GenParams {
    unsigned pop, gp_len, var_len, out_len, n_const;
    float out_prob, const_prob;
    const unsigned *keys;
    const float *leaf_probs; // [10]
    const float *roulette;   // [29]
    const float *consts;     // [n_const]
    float *value;
    int16_t *type;
    int16_t *size;
    unsigned index_offset;
    const int *active_word;    // optional: tree n is generated only when (unsigned)active_word[n] < active_below
    unsigned active_below;
};

template <bool MO>
__global__ __launch_bounds__(kGenBlock) void generate_kernel(GenParams p) {
    __shared__ uint32_t frame_s[kLevels][kGenBlock]; // pending children: childs | depth << 16
    __shared__ uint32_t open_s[kLevels][kGenBlock];  // index of the open node at each depth
    __shared__ float leaf_s[kMaxFullDepth + 1];
    const int tid = threadIdx.x;
    const unsigned n = blockIdx.x * kGenBlock + tid;
    const bool active = n < p.pop && (p.active_word == nullptr || (unsigned)p.active_word[n] < p.active_below);
    if (tid < kMaxFullDepth) leaf_s[tid] = p.leaf_probs[tid];
    if (tid == kMaxFullDepth) leaf_s[tid] = 1.0f; // depths past the table are leaves
    __syncthreads();

    const size_t row = (size_t)n * p.gp_len;
    unsigned cnt = 0;
    if (active) {
        Taus88 rng(seed_hash(n + p.index_offset, p.keys[0], p.keys[1]));
        int top = 1, deepest_open = -1;
        frame_s[0][tid] = 1u; // {childs = 1, depth = 0}
        while (top > 0 && cnt < (unsigned)kMaxStack) {
            const uint32_t fr = frame_s[--top][tid];
            const int childs = (int)(fr & 0xFFFFu) - 1;
            const int depth = (int)(fr >> 16);
            const int dl = depth < kMaxFullDepth ? depth : kMaxFullDepth;
            float v;
            int t, new_childs = 0;
            if (rng.uniform() >= leaf_s[dl]) { // function node (generate.cu:71-100)
                const float r = rng.uniform();
                int k = 0; // largest i with r >= roulette[i], plus one (reverse scan with break, :77-84)
#pragma unroll
                for (int i = 0; i < kNumFuncs; ++i) k = r >= p.roulette[i] ? i + 1 : k;
                t = k <= F_IF ? T_TFUNC : (k <= F_GE ? T_BFUNC : T_UFUNC);
                v = (float)k;
                if (MO) {
                    if (rng.uniform() <= p.out_prob) { // output node: {int16 function, int16 out index} (:86-96)
                        const uint32_t oi = rng.next() % p.out_len;
                        v = bits2f(((oi & 0xFFFFu) << 16) | ((uint32_t)k & 0xFFFFu));
                        new_childs = t - 1;
                        t |= T_OUT;
                    }
                }
                new_childs = (t & T_MASK) - 1;
            } else { // leaf (:104-123)
                if (rng.uniform() <= p.const_prob) {
                    v = p.consts[rng.next() % p.n_const];
                    t = T_CONST;
                } else {
                    v = (float)(rng.next() % p.var_len);
                    t = T_VAR;
                }
            }
            // close every open node at depth >= this depth: its subtree ended at this index
            for (int dd = depth; dd <= deepest_open; ++dd) {
                const unsigned start = open_s[dd][tid];
                if (start < p.gp_len) p.size[row + start] = (int16_t)(cnt - start);
            }
            open_s[depth][tid] = cnt;
            deepest_open = depth;
            if (cnt < p.gp_len) {
                p.value[row + cnt] = v;
                p.type[row + cnt] = (int16_t)t;
            }
            ++cnt;
            if (childs > 0) frame_s[top++][tid] = (uint32_t)childs | ((uint32_t)depth << 16);
            if (new_childs > 0) frame_s[top++][tid] = (uint32_t)new_childs | ((uint32_t)(depth + 1) << 16);
        }
        for (int dd = 0; dd <= deepest_open; ++dd) {
            const unsigned start = open_s[dd][tid];
            if (start < p.gp_len) p.size[row + start] = (int16_t)(cnt - start);
        }
    }

    // ---- zero the tails: the wave walks its 64 rows, lanes cover [len, gp_len) coalesced ----
    const int lane = tid & 63;
    const unsigned wave_first = n - lane;
    for (int l = 0; l < kWave; ++l) {
        const unsigned tn = wave_first + l;
        if (tn >= p.pop) break;
        const unsigned len = (unsigned)__shfl(active ? (int)cnt : (int)p.gp_len, l, 64);  // rows that were skipped stay untouched
        const size_t r0 = (size_t)tn * p.gp_len;
        for (unsigned i = len + lane; i < p.gp_len; i += kWave) {
            p.value[r0 + i] = 0.0f;
            p.type[r0 + i] = 0;
            p.size[r0 + i] = 0;
        }
    }
}

} // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_generate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                  unsigned const_samples_len, float out_prob, float const_prob, const unsigned *keys,
                                  const float *depth2leaf_probs, const float *roulette_funcs, const float *const_samples,
                                  float *value_res, int16_t *type_res, int16_t *size_res, unsigned tree_index_offset,
                                  evogp_stream_t stream_) {
    // argument contract of torch_wrapper.cu:48-54
    if (pop_size == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len == 0 || const_samples_len == 0)
        return EVOGP_E_BADARG;
    if (!(out_prob >= 0.0f && out_prob <= 1.0f) || !(const_prob >= 0.0f && const_prob <= 1.0f)) return EVOGP_E_BADARG;
    if (!keys || !depth2leaf_probs || !roulette_funcs || !const_samples || !value_res || !type_res || !size_res)
        return EVOGP_E_NULLPTR;
    return evogp_hip_generate_masked(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys,
                                     depth2leaf_probs, roulette_funcs, const_samples, value_res, type_res, size_res,
                                     tree_index_offset, nullptr, 0u, stream_);
}

extern "C" int evogp_hip_generate_masked(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                         unsigned const_samples_len, float out_prob, float const_prob, const unsigned *keys,
                                         const float *depth2leaf_probs, const float *roulette_funcs,
                                         const float *const_samples, float *value_res, int16_t *type_res, int16_t *size_res,
                                         unsigned tree_index_offset, const int *active_word, unsigned active_below,
                                         evogp_stream_t stream_) {
    if (pop_size == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len == 0 || const_samples_len == 0)
        return EVOGP_E_BADARG;
    if (!(out_prob >= 0.0f && out_prob <= 1.0f) || !(const_prob >= 0.0f && const_prob <= 1.0f)) return EVOGP_E_BADARG;
    if (!keys || !depth2leaf_probs || !roulette_funcs || !const_samples || !value_res || !type_res || !size_res)
        return EVOGP_E_NULLPTR;
    GenParams p{pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys, depth2leaf_probs,
                roulette_funcs, const_samples, value_res, type_res, size_res, tree_index_offset, active_word, active_below};
    const unsigned blocks = (pop_size + kGenBlock - 1) / kGenBlock;
    hipStream_t stream = (hipStream_t)stream_;
    if (out_len > 1) hipLaunchKernelGGL(generate_kernel<true>, dim3(blocks), dim3(kGenBlock), 0, stream, p);
    else hipLaunchKernelGGL(generate_kernel<false>, dim3(blocks), dim3(kGenBlock), 0, stream, p);
    return (int)hipGetLastError();
}
