/* evogp_hip_debug.h -- measurement and test hooks of libevogp_hip.so.  NOT part of the drop-in boundary (include/evogp_hip.h is what a
 * maintainer of the reference binds): bench.py, scripts/ and tests/ use these to time kernels on their stream, to
 * pin a code path for an A/B run, to read the compiled programs back, and to feed the native mutation kernels the reference's own
 * draws.  Nothing here changes a result; nothing in evogp_amd/ calls any of it. */
#ifndef EVOGP_HIP_DEBUG_H
#define EVOGP_HIP_DEBUG_H
#include "evogp_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Average duration in milliseconds of the most recent `evogp_hip_*` launch sequence that was
 * bracketed by evogp_hip_timer_begin/_end on `stream` (hipEvent pair recorded on that stream).
 * Used by bench.py to time the kernel on the stream it is launched on. */
int evogp_hip_timer_begin(evogp_stream_t stream);
int evogp_hip_timer_end(evogp_stream_t stream, float *elapsed_ms);

/* Profiling hook (no counterpart in the reference): when device_counters != NULL the threaded-code fitness
 * kernel adds per-wave shader-clock cycle counts to device_counters[0..7] = {interpreter core, batch loop,
 * barrier wait, trees, nodes, whole kernel, waves, unused}; NULL (the default) disables the accounting. */
int evogp_hip_debug_set_stats(unsigned long long *device_counters);

/* Per-stage timing of evogp_hip_sr_fitness (profiling, no counterpart in the reference).  While enabled, every call records
 * HIP events on its launch stream: before the call, between the program compiler and the interpreter kernel, behind the
 * interpreter, behind the follow-up kernels.  _read waits for the recorded calls and returns the average duration in ms
 * of {compiler, interpreter, follow-ups} over the calls the threaded-code path took; enable(…) also clears the record.
 * enable == 2: as 1, and the calls STOP behind the threaded code -- trees it leaves to the register kernels keep their sentinel
 * words (0x7FC0FEED, 0x7FC0BEEF, 0x7FC0DEED) in the fitness vector, so a script can count them (bench.py). */
int evogp_hip_debug_profile(int enable);
int evogp_hip_debug_profile_read(float *stage_ms /* [3] */, int *calls);

/* Which program compiler evogp_hip_sr_fitness uses for single-output trees of at most 64 nodes (tests and A/B measurements; no
 * counterpart in the reference): -1 = the packed compiler with the batch size chosen by the population (DEFAULT), 0 = the older one-tree-per-pass
 * compiler, 8 / 16 / 32 / 64 = the packed compiler with that many trees per wave.  The fitness words do not depend on the choice
 * (tests/test_gpu_tc_wide.py compares them bit for bit).  The environment variable EVOGP_TC_PACKED sets the same before the first call. */
int evogp_hip_debug_compile_batch(int trees);
/* Which compiler takes single-output trees of MORE than 64 nodes over + - * / and the unary functions with handlers of their own (tests
 * and A/B measurements): 1 = the straight-line staged compiler of round 5 (DEFAULT; csrc/sr_tc.hip compile_long_arith), 0 = the general
 * compiler's staged passes, as every other long tree; -1 = back to the default / the environment (EVOGP_TC_LONG_FAST).  Same fitness words. */
int evogp_hip_debug_long_compiler(int fast);
/* Whether a program word whose successor has no variable operand names its handler's twin that does not prefetch:
 * -1 = by the launch's trees per CU (DEFAULT: from 900 on; the environment variable EVOGP_TC_TWINS = 0 / 2 sets never / always before the
 * first call), 0 = never, 1 = always.  The fitness words do not depend on the choice (tests/test_gpu_tc_wide.py compares them bit for bit). */
int evogp_hip_debug_twins(int mode);

/* Handler histogram of the program records the most recent evogp_hip_sr_fitness call on the current device compiled:
 * device_hist[flavour * N + id] = number of program words with that handler among the first `pop` trees, N =
 * evogp_hip_debug_tc_nhandlers(), hist_len >= 2 N.  Handler ids and their instruction counts: evogp_amd/lib/tc_handlers.json
 * (written by csrc/gen/gen_tc_asm.py).  bench.py derives the VALU-issue roofline of the interpreter from it. */
int evogp_hip_debug_tc_histogram(unsigned pop, unsigned long long *device_hist, int hist_len, evogp_stream_t stream);
int evogp_hip_debug_tc_nhandlers(void);
/* The program of one tree as that call compiled it (diagnostics; waits for the device): up to max_words pairs {word 0, word 1} in execution
 * order, NEXT words followed; returns the number of words (END / SKIP included), -1 on error. */
int evogp_hip_debug_tc_program(unsigned tree, unsigned *host_words, int max_words);

/* Calls without a function mask (evogp_hip_sr_fitness) choose their program compiler by what the last completed call on a forest of the
 * same shape observed (csrc/sr_tc.hip tc_learned_class).  This forgets every observation of the current device:
 * the next such call looks at its own forest first, like a process's first call.  Tests call it in front of every unhinted call so
 * that a test's kernels do not depend on the tests that ran before it.  (evogp_hip_release_workspaces forgets as well.) */
int evogp_hip_debug_forget_function_classes(void);

/* The native mutation kernels (csrc/mutate_ops.hip) with the draws HANDED IN instead of hashed -- the numbers the reference's Python
 * operators drew, recorded by tests/golden/make_mutation_golden.py --, so that everything behind the draws is compared with the
 * reference's own results (tests/test_gpu_native_mutation.py).  All pointers are device pointers.
 *   given       int32 [pop][3]: {mutates, node, child number 1..3 (mode 0, delete.py:96-101) / inner position (mode 1, hoist.py:61-68)};
 *               for Insert {mutates, node of the tree (insert.py:57-62), position inside the fresh tree (:74-79)}, donor_* row n = the fresh
 *               tree of tree n;
 *   target      uint8 [pop][gp_len]: the nodes to redraw; u: float32, the uniform number the roulette search of the node's OWN arity class
 *               takes (single_point.py:70-90); var_idx / const_idx / out_idx: int32 (out_idx may be NULL without modify_output);
 *               mode 0 / 1: nodes of every kind, 2 / 3: constants only (u, var_idx and the roulettes may be NULL). */
int evogp_hip_debug_structural_mutate_given(int pop_size, int gp_len, int mode, int inner_is_offset, int skip_rows, const int *given,
                                            const float *value, const int16_t *type, const int16_t *size, float *value_res,
                                            int16_t *type_res, int16_t *size_res, evogp_stream_t stream);
int evogp_hip_debug_insert_mutate_given(int pop_size, int gp_len, int skip_rows, const int *given, const float *value, const int16_t *type,
                                        const int16_t *size, const float *donor_value, const int16_t *donor_type, const int16_t *donor_size,
                                        float *value_res, int16_t *type_res, int16_t *size_res, evogp_stream_t stream);
int evogp_hip_debug_point_mutate_given(int pop_size, int gp_len, int mode, int modify_output, int fix_roulette, int skip_rows, int input_len,
                                       int output_len, int n_consts, const unsigned char *target, const float *u, const int *var_idx,
                                       const int *const_idx, const int *out_idx, const float *value, const int16_t *type, const int16_t *size,
                                       const float *roulette_ufuncs, const float *roulette_bfuncs, const float *roulette_tfuncs,
                                       const float *const_samples, float *value_res, evogp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EVOGP_HIP_DEBUG_H */
