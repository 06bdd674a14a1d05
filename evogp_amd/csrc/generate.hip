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
#include <cstdlib>

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
    int hashed;                // != 0: no keys / active_word arrays -- the two keys are word (7, 0 / 1) % 10^6 and tree n is generated
    unsigned long long hash_base;  // when word (4, n + index_offset) < active_below, of the counter-based words of (seed, generation)
    unsigned hkey0, hkey1;     // hashed: the two keys, worked out by the host (generate_impl)
    uint32_t m_const, m_var, m_out;   // fast_mod_magic of n_const / var_len / out_len (a 64-bit division each: the host's)
};

__device__ inline bool gen_active(const GenParams &p, unsigned n) {
    if (p.hashed) return counter_word(p.hash_base, 4u, (unsigned long long)n + p.index_offset) < p.active_below;
    return p.active_word == nullptr || (unsigned)p.active_word[n] < p.active_below;
}
__device__ inline uint32_t gen_seed(const GenParams &p, unsigned n) {
    if (p.hashed) return seed_hash(n + p.index_offset, p.hkey0, p.hkey1);
    return seed_hash(n + p.index_offset, p.keys[0], p.keys[1]);
}

template <bool MO>
__global__ __launch_bounds__(kGenBlock) void generate_kernel(GenParams p) {
    __shared__ uint32_t frame_s[kLevels][kGenBlock]; // pending children: childs | depth << 16
    __shared__ uint32_t open_s[kLevels][kGenBlock];  // index of the open node at each depth
    __shared__ float leaf_s[kMaxFullDepth + 1];
    const int tid = threadIdx.x;
    const unsigned n = blockIdx.x * kGenBlock + tid;
    const bool active = n < p.pop && gen_active(p, n);
    if (tid < kMaxFullDepth) leaf_s[tid] = p.leaf_probs[tid];
    if (tid == kMaxFullDepth) leaf_s[tid] = 1.0f; // depths past the table are leaves
    __syncthreads();

    const size_t row = (size_t)n * p.gp_len;
    unsigned cnt = 0;
    if (active) {
        Taus88 rng(gen_seed(p, n));
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


// ---- staged variant (gp_len <= kStagedMaxLen): one WAVE per workgroup, nothing but LDS inside the serial loop ----------
//
// The lane-per-tree loop above is latency-bound: every node costs scattered 2- and 4-byte global stores (64 cache lines
// per instruction) and, for constant leaves, a dependent global load.  Here the serial loop touches only LDS and the rows
// leave the chip in coalesced pieces:
//   * every live lane emits exactly ONE node per iteration, so iteration `it` fills column `it` of all 64 rows (finished
//     lanes emit zeros — the zero tail comes for free).  Values and types go through a ring of kGenChunk columns that the
//     whole wave flushes every kGenChunk iterations as 64-byte (values) / 32-byte (types) row segments;
//   * subtree sizes close late (the root closes last), so they are staged for the whole row and flushed at the end;
//   * the constant table, the leaf probabilities and the pending/open frames live in LDS; the 29-entry roulette table is
//     reduced once per wave to its non-dominated thresholds (entry i can only be the "largest i with r >= roulette[i]"
//     if every later entry is strictly larger) — with an ordinary cumulative table that is one entry per function in use.
//   * masked launches (the mutation donors of a generation: one row in five is generated) first GATHER the rows to generate:
//     a workgroup looks at 64 * G consecutive rows (G = 0.8 / the expected share of live rows) and works through the list of
//     live ones 64 at a time, so the serial loop runs with ~80 % of its lanes live instead of 20 % (1 M rows: 152 -> 82 us, DESIGN
//     section 3.3).  Which lane generates a tree does not matter: the draw order belongs to the tree (its index seeds the stream).
__device__ inline uint32_t fast_mod(uint32_t x, uint32_t d, uint32_t m) {   // x % d with m = floor(2^32 / d) (d == 1: 2^32 - 1): the quotient estimate is at most one short
    const uint32_t r = x - __umulhi(x, m) * d;
    return r >= d ? r - d : r;
}
__host__ __device__ inline uint32_t fast_mod_magic(unsigned d) { return d <= 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / d); }

constexpr int kGenChunk = 16;
constexpr int kGenMaxGather = 16;     // G: at most 1024 rows per workgroup
constexpr int kGenPitch = 68;         // == 4 (mod 64): the 16x4 flush pattern and the per-lane pattern are both conflict-free
constexpr int kGenConstLds = 256;
constexpr int kGenThr = 8;
constexpr int kStagedMaxLen = 256;

// LDS of a wave (round 6: byte-sized where a byte will do -- node types in the ring, subtree sizes, frames -- and constants by their number:
// 12.6 instead of 19.8 KB at rows of 64 nodes, twelve instead of eight waves per CU; with the kernel held to 168 VGPRs that is three
// waves per SIMD instead of two.  A subtree size beyond 255 does not fit its byte: the (rare) tree of more than 255 nodes has its sizes
// written again, straight to memory, by a second walk over the same draws: generate_staged_kernel `replay`)
static size_t staged_consts_in_lds(unsigned n_const) { return n_const <= (unsigned)kGenConstLds ? (size_t)((n_const + 3u) & ~3u) : 0; }
static size_t staged_lds_bytes(unsigned gp_len, unsigned gather, unsigned n_const) {
    return (size_t)kGenChunk * kGenPitch * 4 + staged_consts_in_lds(n_const) * 4 + 64 * 4 + (size_t)gather * kWave * 4 + (size_t)kLevels * kWave * 2 +
           (size_t)kGenChunk * kGenPitch + (size_t)gp_len * kGenPitch + (size_t)kLevels * kWave + 16;
}

template <bool MO>
__global__ __launch_bounds__(kWave) void generate_staged_kernel(GenParams p, unsigned gather, unsigned consts_lds) {
    extern __shared__ uint32_t gen_lds[];
    float *ring_v = reinterpret_cast<float *>(gen_lds);                                // [kGenChunk][kGenPitch]
    float *const_s = ring_v + kGenChunk * kGenPitch;                                   // [consts_lds]: the constants, when there are at most kGenConstLds
    float *misc_s = const_s + consts_lds;                                              // leaf probs [11], roulette [29]
    unsigned *rows_s = reinterpret_cast<unsigned *>(misc_s + 64);                      // [64 * gather]: the rows to generate
    uint16_t *open_s = reinterpret_cast<uint16_t *>(rows_s + (size_t)gather * kWave);  // [kLevels][64]: index of the open function at a depth
    uint8_t *ring_t = reinterpret_cast<uint8_t *>(open_s + kLevels * kWave);           // [kGenChunk][kGenPitch]: node types (0 .. 0x84)
    uint8_t *size_s = ring_t + kGenChunk * kGenPitch;                                  // [gp_len][kGenPitch]: subtree sizes up to 255 (else: replay)
    uint8_t *frame_s = size_s + (size_t)p.gp_len * kGenPitch;                          // [kLevels][64]: childs | depth << 4 (depth <= 11)

    const int lane = threadIdx.x;
    // ---- the rows of this workgroup's 64 * gather that are to be generated, in index order ----
    unsigned n_rows = 0;
    // (four groups of 64 rows at a time: their mask words are loaded together -- one memory latency per four groups -- without the sixteen
    // registers and the sixteen unrolled hashes that held the kernel at two waves per SIMD)
    for (unsigned g0 = 0; g0 < gather; g0 += 4u) {
        unsigned word[4];
#pragma unroll
        for (unsigned g = 0; g < 4u; ++g) {
            const unsigned idx = (blockIdx.x * gather + g0 + g) * kWave + lane;
            word[g] = 0u;
            if (g0 + g < gather && idx < p.pop && p.active_word != nullptr && !p.hashed) word[g] = (unsigned)p.active_word[idx];
        }
#pragma unroll
        for (unsigned g = 0; g < 4u; ++g) {
            if (g0 + g >= gather) break;
            const unsigned idx = (blockIdx.x * gather + g0 + g) * kWave + lane;
            const bool a = idx < p.pop && (p.hashed ? gen_active(p, idx) : (p.active_word == nullptr || word[g] < p.active_below));
            const unsigned long long m = __ballot(a);
            if (a) rows_s[n_rows + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = idx;
            n_rows += (unsigned)__popcll(m);
        }
    }
    if (n_rows == 0u) return;  // nothing to generate in these rows: they stay untouched

    // ---- tables ----
    if (lane < kMaxFullDepth) misc_s[lane] = p.leaf_probs[lane];
    if (lane == kMaxFullDepth) misc_s[lane] = 1.0f;  // depths past the table are leaves
    const float my_roul = lane < kNumFuncs ? p.roulette[lane] : 0.0f;
    if (lane < kNumFuncs) misc_s[16 + lane] = my_roul;
    const bool consts_in_lds = consts_lds != 0u;
    if (consts_in_lds)
        for (unsigned i = lane; i < p.n_const; i += kWave) const_s[i] = p.consts[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // entry i survives unless a later entry qualifies whenever it does (r_j <= r_i for some j > i): the minimum over the later lanes,
    // five shuffles (NaN entries qualify for nothing and fminf passes them over; the lanes behind the table hold +inf)
    float later = lane < kNumFuncs ? my_roul : __builtin_inff();
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) later = fminf(later, __shfl_down(later, d, 64));
    later = __shfl_down(later, 1, 64);
    const bool surv = lane < kNumFuncs && my_roul == my_roul && !(lane + 1 < kWave && later <= my_roul);
    unsigned long long smask = __ballot(surv);
    const int n_thr = __popcll(smask);
    float thr[kGenThr];
    int kv[kGenThr];
#pragma unroll
    for (int m = 0; m < kGenThr; ++m) {
        if (smask) {
            const int i = __builtin_ctzll(smask);
            smask &= smask - 1;
            thr[m] = bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(my_roul), i));
            kv[m] = i + 1;
        } else {
            thr[m] = __builtin_inff();  // never reached: draws are <= 1
            kv[m] = 0;
        }
    }

    const uint32_t m_const = p.m_const, m_var = p.m_var, m_out = p.m_out;

    for (unsigned b0 = 0; b0 < n_rows; b0 += kWave) {   // 64 rows of the list at a time
    const bool active = b0 + lane < n_rows;
    const unsigned n = active ? rows_s[b0 + lane] : 0u;
    const unsigned long long amask = __ballot(active);
    // sizes_of: the rows whose subtree sizes go out with the chunk (the chunks behind a tree's last node leave once, values, types and
    // sizes together; the chunks flushed inside the loop get their sizes when the trees are complete)
    auto flush_chunk = [&](unsigned chunk, unsigned filled, unsigned long long sizes_of) {
        // lanes: 16 columns x 4 rows per instruction
        const unsigned j = lane & 15, rs = lane >> 4;
        const unsigned node = chunk * kGenChunk + j;
        const bool in_row = node < p.gp_len;
        const unsigned steps = (unsigned)(64 - __builtin_clzll(amask | 1ull) + 3) / 4;   // (a masked launch's list is short: its live rows are the first ones)
#pragma unroll 4
        for (unsigned r4 = 0; r4 < steps; ++r4) {
            const unsigned row = r4 * 4 + rs;
            float v = 0.0f;
            unsigned t = 0;
            if (j < filled) {
                v = ring_v[j * kGenPitch + row];
                t = ring_t[j * kGenPitch + row];
            }
            if (in_row && ((amask >> row) & 1ull)) {
                const size_t at = (size_t)rows_s[b0 + row] * p.gp_len + node;
                p.value[at] = v;
                p.type[at] = (int16_t)t;
                if ((sizes_of >> row) & 1ull) p.size[at] = (int16_t)size_s[node * kGenPitch + row];
            }
        }
    };

    // Round 6, per node: the frame on top of the pending-children stack stays in registers (LDS holds the frames below it: a push when a
    // function has siblings left, a pop when a subtree is complete -- every second node instead of every node); a leaf's size is 1 when it
    // is emitted, so only FUNCTION nodes are "open" and a new node closes the functions at its depth and below (one every second node
    // instead of at least one per node, and the longest such loop in the wave is what every lane waits for); the second draw is common to
    // both kinds of node and the third to both kinds of leaf; x % n is a multiplication by floor(2^32 / n) and one correction.  The draw
    // order is the reference's (generate.cu:55-172): the same trees, bit for bit.
    {   // sizes start at zero: the flush below then reads a row's column whatever the row's length (no cross-lane fetch of the length per step)
        uint32_t *z = reinterpret_cast<uint32_t *>(size_s);
        const unsigned words = p.gp_len * (unsigned)kGenPitch / 4u;
        for (unsigned i = lane; i < words; i += kWave) z[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    Taus88 rng(gen_seed(p, n));
    int tos_childs = 1, tos_depth = 0, sp = 0, deepest_open = -1;
    bool running = active;
    unsigned cnt = 0, it = 0;
    for (;; ++it) {
        const bool live = running && it < (unsigned)kMaxStack;
        if (!__any(live)) break;
        float v = 0.0f;
        int t = 0;
        if (live) {
            const int childs = tos_childs - 1, depth = tos_depth;
            const int dl = depth < kMaxFullDepth ? depth : kMaxFullDepth;
            const float u1 = rng.uniform(), u2 = rng.uniform();
            const bool is_func = u1 >= misc_s[dl];
            for (int dd = depth; dd <= deepest_open; ++dd) {   // the functions at this depth and below ended with the node before this one
                const unsigned start = open_s[dd * kWave + lane];
                if (start < p.gp_len) size_s[start * kGenPitch + lane] = (uint8_t)min(it - start, 255u);
            }
            if (is_func) {  // function node (generate.cu:71-100)
                int k = 0;  // largest i with r >= roulette[i], plus one (:77-84)
                if (n_thr <= kGenThr) {
#pragma unroll
                    for (int m = 0; m < kGenThr; ++m) k = u2 >= thr[m] ? kv[m] : k;
                } else {
#pragma unroll
                    for (int i = 0; i < kNumFuncs; ++i) k = u2 >= misc_s[16 + i] ? i + 1 : k;
                }
                t = k <= F_IF ? T_TFUNC : (k <= F_GE ? T_BFUNC : T_UFUNC);
                v = (float)k;
                const int arity = t - 1;
                if (MO) {
                    if (rng.uniform() <= p.out_prob) {  // output node (:86-96)
                        const uint32_t oi = fast_mod(rng.next(), p.out_len, m_out);
                        v = bits2f(((oi & 0xFFFFu) << 16) | ((uint32_t)k & 0xFFFFu));
                        t |= T_OUT;
                    }
                }
                open_s[depth * kWave + lane] = (uint16_t)it;
                deepest_open = depth;
                if (childs > 0) frame_s[(sp++) * kWave + lane] = (uint8_t)((unsigned)childs | ((unsigned)depth << 4));
                tos_childs = arity; tos_depth = depth + 1;
            } else {  // leaf (:104-123)
                const uint32_t r3 = rng.next();
                if (u2 <= p.const_prob) {
                    const uint32_t ci = fast_mod(r3, p.n_const, m_const);
                    if (consts_in_lds) v = const_s[ci];  // wave-uniform choice: a ds_read, not a flat load
                    else v = p.consts[ci];
                    t = T_CONST;
                } else {
                    v = (float)fast_mod(r3, p.var_len, m_var);
                    t = T_VAR;
                }
                if (it < p.gp_len) size_s[it * kGenPitch + lane] = 1;
                deepest_open = min(deepest_open, depth - 1);
                if (childs > 0) tos_childs = childs;
                else if (sp > 0) {
                    const unsigned fr = frame_s[(--sp) * kWave + lane];
                    tos_childs = (int)(fr & 0xFu); tos_depth = (int)(fr >> 4);
                } else running = false;
            }
            cnt = it + 1;
        }
        const unsigned slot = it & (kGenChunk - 1);
        ring_v[slot * kGenPitch + lane] = v;
        ring_t[slot * kGenPitch + lane] = (uint8_t)t;
        if (slot == kGenChunk - 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            flush_chunk(it / kGenChunk, kGenChunk, 0ull);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (active)
        for (int dd = 0; dd <= deepest_open; ++dd) {
            const unsigned start = open_s[dd * kWave + lane];
            if (start < p.gp_len) size_s[start * kGenPitch + lane] = (uint8_t)min(cnt - start, 255u);
        }
    const unsigned long long bigmask = __ballot(active && cnt > 255u);   // trees whose sizes do not fit a byte: written again below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- subtree sizes of the chunks that left inside the loop ----
    const unsigned nchunks = (p.gp_len + kGenChunk - 1) / kGenChunk;
    const unsigned first_tail = min(it / kGenChunk, nchunks);
    {
        const unsigned j = lane & 15, rs = lane >> 4;
        for (unsigned ch = 0; ch < first_tail; ++ch) {
            const unsigned node = ch * kGenChunk + j;
            const unsigned steps = (unsigned)(64 - __builtin_clzll(amask | 1ull) + 3) / 4;
#pragma unroll 4
            for (unsigned r4 = 0; r4 < steps; ++r4) {
                const unsigned row = r4 * 4 + rs;
                if (node < p.gp_len && (((amask & ~bigmask) >> row) & 1ull))
                    p.size[(size_t)rows_s[b0 + row] * p.gp_len + node] = (int16_t)size_s[node * kGenPitch + row];
            }
        }
    }
    // ---- the partially filled chunk, then zeros up to gp_len: values, types and sizes in one walk ----
    for (unsigned c = first_tail; c < nchunks; ++c) flush_chunk(c, c == first_tail ? (it & (kGenChunk - 1)) : 0u, amask & ~bigmask);

    // ---- a tree of more than 255 nodes (deep descriptors on rows it overflows): the same draws once more, the sizes of the nodes its row
    // holds written straight to memory as they close (generate_kernel's way: scattered stores, rare)
    if (bigmask != 0ull) {
        const bool mine = ((bigmask >> lane) & 1ull) != 0ull;
        const size_t row = (size_t)n * p.gp_len;
        Taus88 r2(gen_seed(p, n));
        int c2 = 1, d2 = 0, sp2 = 0, deep2 = -1;
        bool run2 = mine;
        unsigned cnt2 = 0;
        for (unsigned i2 = 0;; ++i2) {
            const bool live = run2 && i2 < (unsigned)kMaxStack;
            if (!__any(live)) break;
            if (live) {
                const int childs = c2 - 1, depth = d2;
                const int dl = depth < kMaxFullDepth ? depth : kMaxFullDepth;
                const float u1 = r2.uniform(), u2 = r2.uniform();
                for (int dd = depth; dd <= deep2; ++dd) {
                    const unsigned start = open_s[dd * kWave + lane];
                    if (start < p.gp_len) p.size[row + start] = (int16_t)(i2 - start);
                }
                if (u1 >= misc_s[dl]) {
                    int k = 0;
                    if (n_thr <= kGenThr) {
#pragma unroll
                        for (int m = 0; m < kGenThr; ++m) k = u2 >= thr[m] ? kv[m] : k;
                    } else {
#pragma unroll
                        for (int i = 0; i < kNumFuncs; ++i) k = u2 >= misc_s[16 + i] ? i + 1 : k;
                    }
                    const int arity = (k <= F_IF ? T_TFUNC : (k <= F_GE ? T_BFUNC : T_UFUNC)) - 1;
                    if (MO) { if (r2.uniform() <= p.out_prob) (void)r2.next(); }
                    open_s[depth * kWave + lane] = (uint16_t)i2;
                    deep2 = depth;
                    if (childs > 0) frame_s[(sp2++) * kWave + lane] = (uint8_t)((unsigned)childs | ((unsigned)depth << 4));
                    c2 = arity; d2 = depth + 1;
                } else {
                    (void)r2.next();
                    if (i2 < p.gp_len) p.size[row + i2] = 1;
                    deep2 = min(deep2, depth - 1);
                    if (childs > 0) c2 = childs;
                    else if (sp2 > 0) { const unsigned fr = frame_s[(--sp2) * kWave + lane]; c2 = (int)(fr & 0xFu); d2 = (int)(fr >> 4); }
                    else run2 = false;
                }
                cnt2 = i2 + 1;
            }
        }
        if (mine)
            for (int dd = 0; dd <= deep2; ++dd) {
                const unsigned start = open_s[dd * kWave + lane];
                if (start < p.gp_len) p.size[row + start] = (int16_t)(cnt2 - start);
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the rings and frames are reused by the next 64 rows
    __builtin_amdgcn_wave_barrier();
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

static int generate_impl(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len, unsigned const_samples_len, float out_prob,
                         float const_prob, const unsigned *keys, const float *depth2leaf_probs, const float *roulette_funcs,
                         const float *const_samples, float *value_res, int16_t *type_res, int16_t *size_res, unsigned tree_index_offset,
                         const int *active_word, unsigned active_below, int hashed, unsigned long long hash_base, evogp_stream_t stream_);

extern "C" int evogp_hip_generate_masked(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                         unsigned const_samples_len, float out_prob, float const_prob, const unsigned *keys,
                                         const float *depth2leaf_probs, const float *roulette_funcs,
                                         const float *const_samples, float *value_res, int16_t *type_res, int16_t *size_res,
                                         unsigned tree_index_offset, const int *active_word, unsigned active_below,
                                         evogp_stream_t stream_) {
    if (!keys) return EVOGP_E_NULLPTR;
    return generate_impl(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys, depth2leaf_probs, roulette_funcs,
                         const_samples, value_res, type_res, size_res, tree_index_offset, active_word, active_below, 0, 0ull, stream_);
}

extern "C" int evogp_hip_generate_masked_hashed(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                                unsigned const_samples_len, float out_prob, float const_prob,
                                                const float *depth2leaf_probs, const float *roulette_funcs, const float *const_samples,
                                                float *value_res, int16_t *type_res, int16_t *size_res, unsigned tree_index_offset,
                                                long long seed, long long generation, unsigned active_below, evogp_stream_t stream_) {
    return generate_impl(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, nullptr, depth2leaf_probs, roulette_funcs,
                         const_samples, value_res, type_res, size_res, tree_index_offset, nullptr, active_below, 1, counter_base(seed, generation),
                         stream_);
}

static int generate_impl(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len, unsigned const_samples_len, float out_prob,
                         float const_prob, const unsigned *keys, const float *depth2leaf_probs, const float *roulette_funcs,
                         const float *const_samples, float *value_res, int16_t *type_res, int16_t *size_res, unsigned tree_index_offset,
                         const int *active_word, unsigned active_below, int hashed, unsigned long long hash_base, evogp_stream_t stream_) {
    if (pop_size == 0 || gp_len == 0 || gp_len > (unsigned)kMaxStack || var_len == 0 || out_len == 0 || const_samples_len == 0)
        return EVOGP_E_BADARG;
    if (!(out_prob >= 0.0f && out_prob <= 1.0f) || !(const_prob >= 0.0f && const_prob <= 1.0f)) return EVOGP_E_BADARG;
    if ((!keys && !hashed) || !depth2leaf_probs || !roulette_funcs || !const_samples || !value_res || !type_res || !size_res)
        return EVOGP_E_NULLPTR;
    GenParams p{pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys, depth2leaf_probs,
                roulette_funcs, const_samples, value_res, type_res, size_res, tree_index_offset, active_word, active_below, hashed, hash_base,
                hashed ? counter_word(hash_base, 7u, 0ull) % 1000000u : 0u, hashed ? counter_word(hash_base, 7u, 1ull) % 1000000u : 0u,
                fast_mod_magic(const_samples_len), fast_mod_magic(var_len), fast_mod_magic(out_len)};
    hipStream_t stream = (hipStream_t)stream_;
    static const bool staged_ok = [] { const char *e = getenv("EVOGP_GEN_STAGED"); return !(e && e[0] == '0'); }();
    if (staged_ok && gp_len <= (unsigned)kStagedMaxLen) {
        // masked launches gather their live rows: 64 * G rows per workgroup hold ~51 live ones on average (the spread of a
        // binomial keeps all but ~2 % of the workgroups at one pass of the serial loop); EVOGP_GEN_GATHER=1 switches it off
        static const int env_gather = [] { const char *e = getenv("EVOGP_GEN_GATHER"); return e ? atoi(e) : 0; }();
        unsigned gather = 1;
        if (active_word || hashed) {
            const double share = (double)active_below / 2147483648.0;
            gather = share > 0.0 ? (unsigned)(0.8 / share) : (unsigned)kGenMaxGather;
            // ... but only as far as the gathered grid still fills the chip: a launch whose workgroups are all resident at once
            // (pop 100 k: 1563 of them) lasts as long as one workgroup, and a gathered workgroup lasts longer (22.7 -> 37.5 us)
            const unsigned resident = (unsigned)device_info().num_cus * 12u;
            const unsigned rounds = ((pop_size + kWave - 1) / kWave) / resident;
            if (gather > rounds) gather = rounds;
            if (env_gather > 0) gather = (unsigned)env_gather;
            gather = gather < 1u ? 1u : (gather > (unsigned)kGenMaxGather ? (unsigned)kGenMaxGather : gather);
        }
        const unsigned wgs = (pop_size + kWave * gather - 1) / (kWave * gather);
        const size_t lds = staged_lds_bytes(gp_len, gather, const_samples_len);
        const unsigned consts_lds = (unsigned)staged_consts_in_lds(const_samples_len);
        if (out_len > 1) hipLaunchKernelGGL(generate_staged_kernel<true>, dim3(wgs), dim3(kWave), lds, stream, p, gather, consts_lds);
        else hipLaunchKernelGGL(generate_staged_kernel<false>, dim3(wgs), dim3(kWave), lds, stream, p, gather, consts_lds);
        return (int)hipGetLastError();
    }
    const unsigned blocks = (pop_size + kGenBlock - 1) / kGenBlock;
    if (out_len > 1) hipLaunchKernelGGL(generate_kernel<true>, dim3(blocks), dim3(kGenBlock), 0, stream, p);
    else hipLaunchKernelGGL(generate_kernel<false>, dim3(blocks), dim3(kGenBlock), 0, stream, p);
    return (int)hipGetLastError();
}
