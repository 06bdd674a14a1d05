/*
 * evogp_hip.h — C ABI of the MI355X (gfx950) tree-evaluation / genetic-operation engine.
 *
 * This is the drop-in boundary.  Each entry point replaces one of the five host launchers
 * the reference declares in  src/evogp/cuda/kernel.h:23-97  and calls from
 * src/evogp/cuda/torch_wrapper.cu  (generate :68, mutate :121, crossover :173,
 * evaluate :220, SR_fitness :267).  Argument order, meaning and units are the
 * reference's; the differences are deliberate and minimal:
 *
 *   - every function takes the HIP stream to enqueue on as its last argument
 *     (the reference uses the legacy default stream, SURVEY.md §8b "Threading / streams");
 *   - every function returns an int: 0 on success, a hipError_t value when the launch
 *     failed, or a negative EVOGP_E_* code for an argument the kernels cannot honour
 *     (the reference's launchers return void and never check, torch_wrapper.cu:84,135,188,231,282);
 *   - evogp_hip_generate has one extra argument, tree_index_offset, added to the
 *     tree index before it is hashed into the per-tree RNG seed (reference: generate.cu:35,40
 *     uses the local thread index).  0 reproduces the reference; a rank offset makes a
 *     population sharded over several GPUs bit-identical to the single-GPU one.
 *
 * All pointers are DEVICE pointers (HBM), row-major, contiguous, exactly the layouts of
 * the reference's Forest tensors (src/evogp/tree/forest.py:13-40):
 *     value  float32 [pop][gp_len]   node payload (var index / constant / function id / OUT bits)
 *     type   int16   [pop][gp_len]   NodeType, bit 7 = OUT_NODE
 *     size   int16   [pop][gp_len]   subtree size, size[t][0] = live length of tree t
 * Output rows are written on [0, len) and zero-filled on [len, gp_len) (the reference leaves
 * the tail uninitialised: torch_wrapper.cu:64-66, generate.cu:167-172).
 *
 * No torch types, no C++ types: this header is plain C and is what a cgo / JNI / ctypes /
 * libtorch binding includes (INTEGRATION.md shows the libtorch and ctypes stubs).
 */
#ifndef EVOGP_HIP_H
#define EVOGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* encodings shared with the reference (src/evogp/cuda/defs.h:5-57) */
#define EVOGP_MAX_STACK 1024      /* defs.h:5  — upper bound on gp_len                     */
#define EVOGP_MAX_FULL_DEPTH 10   /* defs.h:5  — length of depth2leaf_probs                */
#define EVOGP_NUM_FUNCS 29        /* defs.h:56 — Function::END, length of roulette_funcs   */

/* negative return codes (argument errors detected on the host before any launch) */
#define EVOGP_E_BADARG (-1)       /* size/shape argument out of the range the reference's wrapper accepts */
#define EVOGP_E_NULLPTR (-2)      /* a required pointer is NULL */
#define EVOGP_E_UNSUPPORTED (-3)  /* var_len/out_len larger than the interpreter's staging area */

typedef void *evogp_stream_t; /* a hipStream_t; NULL = the null stream */

/* kernel.h:23-38 `generate`.  keys: u32[2]; depth2leaf_probs: f32[10]; roulette_funcs: f32[29]
 * (cumulative); const_samples: f32[const_samples_len].  Outputs: [pop_size][gp_len]. */
int evogp_hip_generate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                       unsigned const_samples_len, float out_prob, float const_prob,
                       const unsigned *keys, const float *depth2leaf_probs,
                       const float *roulette_funcs, const float *const_samples,
                       float *value_res, int16_t *type_res, int16_t *size_res,
                       unsigned tree_index_offset, evogp_stream_t stream);

/* kernel.h:40-53 `mutate`.  out[n] = old[n] with the subtree at mutate_indices[n] replaced by the
 * whole tree new[n]; old[n] is copied when the index is outside [0, len(old[n])) or the result
 * would exceed gp_len (mutation.cu:150-160,170-180). */
int evogp_hip_mutate(int pop_size, int gp_len,
                     const float *value_ori, const int16_t *type_ori, const int16_t *size_ori,
                     const int *mutate_indices,
                     const float *value_new, const int16_t *type_new, const int16_t *size_new,
                     float *value_res, int16_t *type_res, int16_t *size_res,
                     evogp_stream_t stream);

/* kernel.h:55-69 `crossover`.  out[n] = ori[left_idx[n]] with the subtree at left_node_idx[n]
 * replaced by subtree right_node_idx[n] of ori[right_idx[n]]; the left tree is copied when
 * right_idx[n] is outside [0, pop_size_ori) or the result would exceed gp_len
 * (mutation.cu:256-266,279-289).  Node indices outside the live tree (undefined behaviour in the
 * reference) also produce a copy of the left tree. */
int evogp_hip_crossover(int pop_size_ori, int pop_size_new, int gp_len,
                        const float *value_ori, const int16_t *type_ori, const int16_t *size_ori,
                        const int *left_idx, const int *right_idx,
                        const int *left_node_idx, const int *right_node_idx,
                        float *value_res, int16_t *type_res, int16_t *size_res,
                        evogp_stream_t stream);

/* kernel.h:71-81 `evaluate`.  results[n][:] = tree_n(variables[n][:]);
 * variables: f32[pop][var_len], results: f32[pop][out_len]. */
int evogp_hip_evaluate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                       const float *value, const int16_t *type, const int16_t *size,
                       const float *variables, float *results, evogp_stream_t stream);

/* kernel.h:83-97 `SR_fitness`.  fitnesses[t] = (1/D) * sum_d sum_o err(labels[d][o] - tree_t(X[d])_o),
 * err = square (use_mse != 0) or abs.  variables: f32[D][var_len], labels: f32[D][out_len].
 * kernel_type 0..4 is accepted for compatibility (forward.cu:827-856); every value runs the same
 * fused kernel. */
int evogp_hip_sr_fitness(unsigned pop_size, unsigned data_points, unsigned gp_len,
                         unsigned var_len, unsigned out_len, int use_mse,
                         const float *value, const int16_t *type, const int16_t *size,
                         const float *variables, const float *labels, float *fitnesses,
                         unsigned kernel_type, evogp_stream_t stream);

/* evogp_hip_generate restricted to the trees n with (unsigned)active_word[n] < active_below (active_word == NULL: all of
 * them).  Rows of the other trees are not touched.  Used by the fused generation step: donors are only produced for
 * the offspring that will mutate. */
int evogp_hip_generate_masked(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                              unsigned const_samples_len, float out_prob, float const_prob,
                              const unsigned *keys, const float *depth2leaf_probs,
                              const float *roulette_funcs, const float *const_samples,
                              float *value_res, int16_t *type_res, int16_t *size_res,
                              unsigned tree_index_offset, const int *active_word, unsigned active_below,
                              evogp_stream_t stream);

/* The default generation step in one pass (replaces the PyTorch composition of
 * src/evogp/algorithm/genetic_programming.py:105-124 with selection/default.py, crossover/default.py and
 * mutation/default.py):
 *     res[0 .. n_elite)      = forest[order[0 .. n_elite)]
 *     child_i                = crossover(forest[order[r0 % n_surv]] at r2 % size, forest[order[r1 % n_surv]] at r3 % size)
 *     res[n_elite + i]       = r4 < mutate_below ? mutate(child_i, (r5 % 1024) % size(child_i), donor_i) : child_i
 * order: i32[n_surv] (descending fitness order), rnd: i32[6][pop_size - n_elite] raw words in [0, 2^31 - 1),
 * donors: [pop_size - n_elite][gp_len] (only the rows with r4 < mutate_below are read), decisions: optional
 * i32[pop_size - n_elite][6] = {left, right, p, q, mutated, mutate position} for tests, or NULL.
 * Fallback rules of the subtree replacement: mutation.cu:150-160,170-180,256-266,279-289. */
int evogp_hip_breed_default(int pop_size, int gp_len, int n_elite, int n_surv,
                            const float *value, const int16_t *type, const int16_t *size,
                            const int *order, const int *rnd, unsigned mutate_below,
                            const float *donor_value, const int16_t *donor_type, const int16_t *donor_size,
                            float *value_res, int16_t *type_res, int16_t *size_res,
                            int *decisions, evogp_stream_t stream);

/* evogp_hip_breed_default restricted to the rows [row_begin, row_begin + row_count) of the next generation (a rank's
 * slice of a sharded population).  value_res/type_res/size_res, the donor arrays and `decisions` hold
 * exactly row_count rows: row k of each belongs to next-generation row row_begin + k (donor rows of elite rows and of
 * offspring that do not mutate are never read). */
int evogp_hip_breed_default_rows(int pop_size, int gp_len, int n_elite, int n_surv,
                                 const float *value, const int16_t *type, const int16_t *size,
                                 const int *order, const int *rnd, unsigned mutate_below,
                                 const float *donor_value, const int16_t *donor_type, const int16_t *donor_size,
                                 float *value_res, int16_t *type_res, int16_t *size_res,
                                 int *decisions, int row_begin, int row_count, evogp_stream_t stream);

/* The same pass when value / type / size hold only the trees that `order` can name — `table_rows` rows, e.g. the
 * survivor table a sharded run gathers (evogp_amd/parallel.py) — instead of the whole population of `pop_size` trees;
 * `order` then holds table rows.  evogp_hip_breed_default_rows is this call with table_rows = pop_size. */
int evogp_hip_breed_default_table(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv, const float *value,
                                  const int16_t *type, const int16_t *size, const int *order, const int *rnd,
                                  unsigned mutate_below, const float *donor_value, const int16_t *donor_type,
                                  const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res,
                                  int *decisions, int row_begin, int row_count, evogp_stream_t stream);

/* The same pass for ANY selection operator (src/evogp/algorithm/genetic_programming.py:110-122 with e.g.
 * selection/tournament.py:59-133): the elites and the parents are two separate lists of table rows,
 *     res[0 .. n_elite)  = table[elite_rows[0 .. n_elite)]
 *     child_i            = crossover(table[parent_rows[r0 % n_surv]] at r2 % size, table[parent_rows[r1 % n_surv]] at r3 % size)
 * parent_rows: i32[n_surv], repeats allowed (a tournament winner that won k times is k entries, exactly the
 * `forest[survivor_indices]` gather of crossover/default.py:37; n_surv is not bounded by pop_size); elite_rows: i32[n_elite]
 * (may be NULL when n_elite == 0).  evogp_hip_breed_default_table is this call with both lists = order. */
int evogp_hip_breed_lists(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv, const float *value,
                          const int16_t *type, const int16_t *size, const int *elite_rows, const int *parent_rows,
                          const int *rnd, unsigned mutate_below, const float *donor_value, const int16_t *donor_type,
                          const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res,
                          int *decisions, int row_begin, int row_count, evogp_stream_t stream);

/* evogp_hip_sr_fitness for a caller that knows which functions can occur in the forest: bit f of function_mask = function id f
 * (defs.h:10-57) may occur, 0 = unknown.  A fitness call is up to five launches of which three usually find nothing to do; a
 * forest of + - * / and the unary functions of at most 64 nodes per tree cannot leave a tree for the general compiler, so that
 * launch need not be made (5 -> 4 launches: 5-11 us per call).  Whatever a tree carries that the mask did not promise is still
 * evaluated correctly by the last follow-up kernel: a wrong mask costs speed, never a result.  (tree_generate emits the
 * invalid function id 29 when its uniform draw is exactly 1.0 -- generate.cu:77-84, one tree in the 1 M-tree headline forest; the
 * compiler takes such nodes itself, so they do not break the promise.)  EVOGP_TC_FUNC_MASK=0 makes the engine ignore the mask.  evogp_amd.tree.Forest derives the mask from the GenerateDescriptors its trees came from. */
int evogp_hip_sr_fitness_hinted(unsigned pop_size, unsigned data_points, unsigned gp_len, unsigned var_len, unsigned out_len,
                                int use_mse, const float *value, const int16_t *type, const int16_t *size,
                                const float *variables, const float *labels, float *fitnesses, unsigned kernel_type,
                                unsigned function_mask, evogp_stream_t stream);
/* (Rounds 3-4 also shipped two opt-in experiments -- program records compiled ahead by the breeding pass under a stamp protocol
 * (evogp_hip_breed_lists_compiled, evogp_hip_sr_fitness_stamped, evogp_hip_set_breed_compile) and history-driven launch skipping
 * (EVOGP_TC_HINTS) -- that gained nothing measurable and, switched on globally, failed tests.  Round 5 removed both: ABI version 4.) */

/* The same two passes with their random words computed in the kernels (no counterpart in the reference): word k of offspring i is
 * hash(seed, generation, k, i) -- exactly the numbers evogp_hip_random_words writes -- so no array of words is drawn, written and read
 * (one launch and 24 B per offspring less per generation), and the two generation keys are words (7, 0) and (7, 1) modulo 10^6.
 * evogp_hip_generate_masked_hashed generates the trees n whose word (4, n + tree_index_offset) is below active_below;
 * evogp_hip_breed_lists_hashed is evogp_hip_breed_lists without `rnd`.  Results equal those of the array forms fed with
 * evogp_hip_random_words(seed, generation, ...) bit for bit (tests/test_gpu_breed.py). */
int evogp_hip_generate_masked_hashed(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                     unsigned const_samples_len, float out_prob, float const_prob, const float *depth2leaf_probs,
                                     const float *roulette_funcs, const float *const_samples, float *value_res, int16_t *type_res,
                                     int16_t *size_res, unsigned tree_index_offset, long long seed, long long generation,
                                     unsigned active_below, evogp_stream_t stream);
int evogp_hip_breed_lists_hashed(int pop_size, int table_rows, int gp_len, int n_elite, int n_surv, const float *value,
                                 const int16_t *type, const int16_t *size, const int *elite_rows, const int *parent_rows,
                                 long long seed, long long generation, unsigned mutate_below, const float *donor_value,
                                 const int16_t *donor_type, const int16_t *donor_size, float *value_res, int16_t *type_res,
                                 int16_t *size_res, int *decisions, int row_begin, int row_count, evogp_stream_t stream);

/* The reference's structural and point mutations, drawn and applied in one launch each (no counterpart in the
 * reference's ABI, whose operators are torch programs around tree_crossover: src/evogp/algorithm/mutation/hoist.py:43-75,
 * delete.py:44-105, single_point.py:43-126, multi_point.py:46-143, single_const.py:39-98, multi_const.py:43-95).  The random numbers are the
 * counter-based words of evogp_hip_random_words under (seed, call): word 0 of tree n decides whether it mutates (u < rate), words 1 and 2
 * the nodes; words 8-12 of node n * gp_len + i its new payload.  Trees below skip_rows (the elites of a generation step) are copied.
 *   evogp_hip_structural_mutate  mode 0 = DeleteMutation (a function node whose subtree has at most max_size nodes -- 0: any -- is replaced by
 *       its child number trunc(1 + u (arity - 1)), the reference's draw), 1 = HoistMutation (node trunc(u S) is replaced by node
 *       trunc(u' size) taken as an absolute index, or as an offset into the subtree with inner_is_offset); the fall-back rules of
 *       tree_crossover apply (mutation.cu:256-266, 279-289).  decisions: optional i32[pop][2] = {replaced node or -1, donor node}.
 *   evogp_hip_point_mutate       mode 0 = MultiPointMutation, 1 = SinglePointMutation, 2 = MultiConstMutation, 3 = SingleConstMutation;
 *       only the value array changes.  roulette_*funcs: the descriptor's cumulative per-arity tables, f32[29] each. */
int evogp_hip_structural_mutate(int pop_size, int gp_len, int mode, float rate, int max_size, int inner_is_offset, int skip_rows,
                                long long seed, long long call, const float *value, const int16_t *type, const int16_t *size,
                                float *value_res, int16_t *type_res, int16_t *size_res, int *decisions, evogp_stream_t stream);
/* InsertMutation (mutation/insert.py:45-85): tree n mutates when counter word (4, n) of (seed, call) lies below mutate_below (of 2^31) -- the
 * rule under which evogp_hip_generate_masked_hashed(seed, call, mutate_below) generated the fresh trees handed in as donor_* (row n for tree n).
 * decisions (optional, [pop][2]): node of the tree that was replaced (-1: the tree was copied), position inside the fresh tree. */
int evogp_hip_insert_mutate(int pop_size, int gp_len, unsigned mutate_below, int skip_rows, long long seed, long long call,
                            const float *value, const int16_t *type, const int16_t *size, const float *donor_value,
                            const int16_t *donor_type, const int16_t *donor_size, float *value_res, int16_t *type_res, int16_t *size_res,
                            int *decisions, evogp_stream_t stream);
int evogp_hip_point_mutate(int pop_size, int gp_len, int mode, float rate, float intensity, int per_node, int modify_output,
                           int fix_roulette, int skip_rows, int input_len, int output_len, int n_consts, long long seed, long long call,
                           const float *value, const int16_t *type, const int16_t *size, const float *roulette_ufuncs,
                           const float *roulette_bfuncs, const float *roulette_tfuncs, const float *const_samples, float *value_res,
                           evogp_stream_t stream);

/* Counter-based random words for the breeding pass of a sharded run (no counterpart in the reference, which draws with
 * torch's generator): out[k][i] for k < rows, i in [lo, hi) = hash(seed, generation, k, i) mapped to [0, 2^31 - 1), the value
 * evogp_amd/parallel.py random_words computes on any device; out: i32[rows][n_cols], only columns [lo, hi) are written.  Every
 * rank fills the columns of its own offspring; equal arguments give equal words on every rank and for every world size. */
int evogp_hip_random_words(long long seed, long long generation, int rows, long long n_cols, long long lo, long long hi,
                           int *out, evogp_stream_t stream);

/* scores[i] = -inf where errors[i] is NaN, else -errors[i] (negate != 0) or errors[i] (no counterpart in the reference's ABI: its
 * SymbolicRegression.evaluate negates in torch, problem/symbolic_regression.py:82-96, and its pipeline scrubs NaN with a boolean-mask
 * assignment, pipeline/standard.py:41-43 -- four small launches and a host sync per generation).  errors and scores may be the same array. */
int evogp_hip_fitness_scores(unsigned n, int negate, const float *errors, float *scores, evogp_stream_t stream);

/* Selection in one launch (no counterpart in the reference, whose DefaultSelection sorts the whole fitness vector,
 * src/evogp/algorithm/selection/default.py:21-39): order[0 .. n_elite) = the n_elite trees of highest fitness, order[n_elite ..
 * n_keep) = the next n_keep - n_elite, each group in ascending tree index.  Ties at a threshold go to the lower index, so both
 * SETS are those of a stable descending sort; NaN is the worst fitness.  n_elite <= n_keep <= n.  `zeroed_workspace`:
 * evogp_hip_select_workspace_bytes() bytes, all zero (the kernel's grid barrier and histograms live there). */
size_t evogp_hip_select_workspace_bytes(void);
int evogp_hip_select(unsigned n, unsigned n_elite, unsigned n_keep, const float *fitness, int *order, void *zeroed_workspace,
                     evogp_stream_t stream);
/* The same launch for a caller that alternates between two workspaces on one stream: `next_workspace` (the other one, used by the
 * previous call on that stream) is zeroed by this launch, so the steady state needs no memset per call (3-4 us). */
int evogp_hip_select_alternating(unsigned n, unsigned n_elite, unsigned n_keep, const float *fitness, int *order,
                                 void *zeroed_workspace, void *next_workspace, evogp_stream_t stream);

/* Tournament selection in one launch (no counterpart in the reference's ABI; src/evogp/algorithm/selection/tournament.py:59-133 with its
 * default arguments, which is also the setting of example/uci_sr.py:73-75: contenders drawn WITH replacement, the best contender
 * wins): winners[i] = the tree of highest fitness among t_size contenders, contender k of tournament i being tree
 * hash(seed, generation, 16 + k, i) % n -- the counter-based words of evogp_hip_random_words, so every rank of a sharded run names
 * the same contenders (evogp_amd/parallel.py random_words gives the same numbers on any device).  NaN is the worst fitness; of
 * equal contenders the first drawn wins (torch.argmax).  winners: i32[n_tournaments]. */
int evogp_hip_tournament_select(unsigned n, unsigned n_tournaments, unsigned t_size, long long seed, long long generation,
                                const float *fitness, int *winners, evogp_stream_t stream);

/* Non-replicating batch evaluation (replaces the repeat_interleave + tree_evaluate
 * composition of src/evogp/tree/forest.py:143-176): results[t][d][:] = tree_t(variables[d][:]),
 * variables: f32[D][var_len], results: f32[pop][D][out_len]. */
int evogp_hip_batch_evaluate(unsigned pop_size, unsigned data_points, unsigned gp_len,
                             unsigned var_len, unsigned out_len,
                             const float *value, const int16_t *type, const int16_t *size,
                             const float *variables, float *results, evogp_stream_t stream);

/* Fused epilogue of the Classification problem (src/evogp/problem/classification.py:62-75) on top of the batch
 * evaluation: counts[t] = number of rows d with argmax_o tree_t(variables[d])_o == labels[d], where the arg-max is the
 * one torch.argmax(clip(softmax(outputs))) returns (first maximum; index 0 when an output is NaN or the maximum is
 * infinite).  out_len in [2, 16]; labels: i32[data_points]; counts: u32[pop_size] (zeroed here).  The
 * (pop, data_points, out_len) output tensor is never materialised. */
int evogp_hip_batch_argmax_count(unsigned pop_size, unsigned data_points, unsigned gp_len,
                                 unsigned var_len, unsigned out_len,
                                 const float *value, const int16_t *type, const int16_t *size,
                                 const float *variables, const int *labels, unsigned *counts,
                                 evogp_stream_t stream);

/* Prepared forward pass of a MULTI-OUTPUT population (no counterpart in the reference's ABI).  The
 * reference's rollout problems call `evaluate` once per environment step on the same forest, 1000 times per generation
 * (src/evogp/problem/brax_problem.py:54-93).  In multi-output mode only nodes flagged OUT are observable and their operands
 * are leaves (forward.cu:237-243: every function hands its last operand on), so a tree reduces to a short list of
 * "outs[o] += f(leaf, leaf)".  _prepare builds the lists once per forest into `workspace` (device memory,
 * >= evogp_hip_evaluate_workspace_bytes, 16-byte aligned; the last 64 bytes hold an int: the number of trees the lists cannot
 * express); _prepared evaluates results[n][:] = tree_n(variables[n][:]) from them — same values as evogp_hip_evaluate, bit for
 * bit (same operations in the same order) — and, when with_fallback != 0, runs the stack interpreter on the trees counted
 * above (pass 0 only if that count, read back after _prepare, is 0).  out_len in [2, 32], var_len <= 255. */
size_t evogp_hip_evaluate_workspace_bytes(unsigned pop_size, unsigned gp_len);
int evogp_hip_evaluate_prepare(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                               const float *value, const int16_t *type, const int16_t *size,
                               void *workspace, size_t workspace_bytes, evogp_stream_t stream);
int evogp_hip_evaluate_prepared(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                                const float *value, const int16_t *type, const int16_t *size,
                                const void *workspace, int with_fallback,
                                const float *variables, float *results, evogp_stream_t stream);

/* Engine-owned device memory (no counterpart in the reference, whose kernels keep a private copy of the tree per thread,
 * forward.cu:284-287).  evogp_hip_sr_fitness compiles every tree into PROGRAM RECORDS that its interpreter kernel reads: one
 * buffer per device, allocated with hipMalloc on first use, grown on demand (never while a HIP graph is being captured),
 * reused by every later eager call on that device and invisible to the caller's allocator; calls recorded into HIP graphs share a
 * second buffer of the same law that eager calls never touch (at most two population-sized buffers per device: round 5).  Size law:
 *     bytes = ceil(pop_size * 256 / 4096) * 4096 * max(2, ceil((gp_len + 2) / 31))    (+ 1/8 slack when it grows)
 * i.e. 768 MB for 1 M trees of gp_len 64 (three arrays of records; two up to gp_len 60), 8.7 GB for 1 M trees of gp_len 1024.
 * A single-output forest of gp_len <= 64 whose function mask (evogp_hip_sr_fitness_hinted) holds no unary function has programs of at
 * most 32 words: ONE array, 256 MB at 1 M trees (round 4).  Round 6: evogp_hip_sr_fitness -- the call WITHOUT a mask, the reference's
 * own operator (torch_wrapper.cu:235-284) -- is given the mask the engine observes on the device: the first call on a population
 * shape (pop_size, gp_len) looks at its forest (one pass over the nodes, waited for once per shape; not inside a stream capture,
 * where the call takes the law above), every later call reads what the last completed call on that shape observed.  The observation
 * decides which kernels run and how many arrays are held, never a result: a tree the chosen compiler cannot take is evaluated by the
 * register kernels (the value up to the order of the sum over the rows), and the call's own observation corrects the next call.
 * EVOGP_TC_LEARN=0 switches the observations off (unmasked calls then always hold three arrays).
 *   evogp_hip_set_program_buffer_limit  caps the buffer (default 16 GiB): a call that would need more runs on the register
 *                                       interpreters instead (same results, 3-6x slower); 0 disables the compiled path.
 *   evogp_hip_program_buffer_bytes      bytes currently held on the current device (the eager buffer + the graphs').
 *   evogp_hip_release_workspaces        waits for the current device and frees the buffer (and the rings below); the next
 *                                       fitness call allocates again.
 * Round 4: a call whose trees cannot need the general program compiler (single output, gp_len <= 64, a dataset that fits LDS, and
 * a function mask that says so: evogp_hip_sr_fitness_hinted) runs as ONE kernel whose waves compile the batch of eight trees they
 * are about to interpret into a RING of records of their own -- no population-sized buffer at all.  A ring is
 *     bytes = compute units * 16 waves * 2 arrays * 16 records * 256     (32 MiB on a 256-CU device, whatever the population)
 * and lives in L2; every stream that makes such calls gets one (plus one spare per device, which the first call on a CAPTURING
 * stream takes: nothing is allocated inside a capture), they are never freed or moved before evogp_hip_release_workspaces, so a
 * HIP graph that holds such a call stays valid while later calls grow or shrink the population.
 *   evogp_hip_record_ring_bytes         bytes of rings currently held on the current device. */
int evogp_hip_set_program_buffer_limit(unsigned long long bytes);
/* Where that memory comes from (round 5): by default hipMalloc / hipFree; a caller with an allocator of its own hands in a pair of functions
 * (alloc returns NULL on failure) and the buffers become visible in its statistics -- the libtorch binding installs torch's caching
 * allocator when it is loaded, so torch.cuda.memory_allocated() counts them.  The engine synchronises the device before it first touches a
 * block from a caller's pool.  Only while the engine holds no memory (evogp_hip_release_workspaces first), else EVOGP_E_BADARG; NULL, NULL
 * restores hipMalloc / hipFree. */
typedef void *(*evogp_alloc_fn)(size_t bytes);
typedef void (*evogp_free_fn)(void *ptr);
int evogp_hip_set_allocator(evogp_alloc_fn alloc, evogp_free_fn free_fn);
unsigned long long evogp_hip_program_buffer_bytes(void);
unsigned long long evogp_hip_record_ring_bytes(void);
int evogp_hip_release_workspaces(void);

/* Timers, per-stage profiling, compiler / twin selection for A/B runs, handler histograms and the test entries of the native mutation
 * kernels are not part of this boundary: include/evogp_hip_debug.h. */

/* Human-readable text for a return code of any function above. */
const char *evogp_hip_error_string(int code);

/* Division in the threaded-code SR-fitness path (no counterpart in the reference's ABI: the reference fixes its division
 * at BUILD time -- setup.py:55 compiles the kernels with -use_fast_math, i.e. CUDA's approximate 2-ulp division).
 *   EVOGP_DIV_SHORT (default)  the IEEE sequence with all of its range and special-case handling (v_div_scale,
 *                   v_div_fmas, v_div_fixup) but ONE residual correction instead of a refined reciprocal plus two: 9
 *                   operations per quotient instead of 13.  Faithfully rounded; it is the correctly rounded quotient
 *                   except for about 1 operand pair in 4e9 (1 of 2^32 random mantissa pairs, scripts/ubench/
 *                   div_faithful.hip), where it is the neighbouring float.  Fitness vectors of 200 k trees x 1024 rows
 *                   were bit-identical to the IEEE mode (scripts/div_modes.py).  A block of 64 lanes x K rows whose operands
 *                   all lie in [2^-46, 2^46] -- where the range scaling does nothing -- runs the same four operations
 *                   without it (same quotients, bit for bit; docs/DESIGN_history_r01_r03.md section 3.1d).
 *                   Round 5: under SHORT and FAST a launch whose whole dataset lies in [2^-46, 2^46] (NaN allowed) keeps the correctly
 *                   rounded reciprocals of the variables' columns in LDS, and a division BY A VARIABLE reads its reciprocal instead of
 *                   computing it with v_rcp_f32 (1 ulp); the residual correction against the divisor stays, so the quotient is
 *                   faithfully rounded as before but need not be the same float in the rare pairs where SHORT and IEEE differ
 *                   (EVOGP_TC_RECIP=0 switches it off).
 *   EVOGP_DIV_IEEE  every quotient is the correctly rounded IEEE-754 quotient, as the CPU oracle computes it (+45 % time).
 *   EVOGP_DIV_FAST  as SHORT, but a block with an operand outside [2^-46, 2^46] takes rows without range scaling:
 *                   |b| > 2^126 gives 0 and |a/b| >= 2^128 gives NaN instead of inf (-2 % time against SHORT).
 * A NaN fitness is always the canonical quiet NaN 0x7FC00000 on this path.
 * Affects only tree_SR_fitness with one output on the + - * / function set (the threaded-code path); every other
 * kernel (evaluate, batch_evaluate, the register interpreters) divides with the IEEE sequence.
 * Environment: EVOGP_SR_DIV=ieee|short|fast selects the mode of a process that never calls the setter. */
#define EVOGP_OK 0
#define EVOGP_DIV_IEEE 0
#define EVOGP_DIV_FAST 1
#define EVOGP_DIV_SHORT 2
int evogp_hip_set_sr_division(int mode);
int evogp_hip_get_sr_division(void);

/* ABI version of this header (5): bumped when a signature changes or an entry point is added (5: the debug hooks moved to evogp_hip_debug.h). */
int evogp_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EVOGP_HIP_H */
