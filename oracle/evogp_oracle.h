/*
 * evogp_oracle.h — CPU restatement of the reference algorithm.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product (evogp_amd) never
 * imports, links or calls anything in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_vs_ref.py,
 * tests/test_golden.py) against
 *   (i)  the reference's own device code compiled for the host (oracle/_ref, built by
 *        oracle/build_ref.py from the sources where they lie under /root/reference), and
 *   (ii) the golden vectors committed under tests/golden/ that were generated with (i)
 *        (SURVEY.md Appendix B fixtures + RNG known answers of Thrust's taus88).
 *
 * All pointers are HOST pointers; layouts are those of include/evogp_hip.h.
 */
#ifndef EVOGP_ORACLE_H
#define EVOGP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

uint32_t evogp_oracle_hash(uint32_t n, uint32_t k1, uint32_t k2);
/* fills out[0..count) with raw taus88 draws and fout[0..count) with the uniform floats of a
 * second engine seeded identically (either pointer may be NULL) */
void evogp_oracle_taus88(uint32_t seed, int count, uint32_t *out, float *fout);

void evogp_oracle_generate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                           unsigned const_samples_len, float out_prob, float const_prob,
                           const unsigned *keys, const float *depth2leaf_probs,
                           const float *roulette_funcs, const float *const_samples,
                           float *value_res, int16_t *type_res, int16_t *size_res,
                           unsigned tree_index_offset);

void evogp_oracle_mutate(int pop_size, int gp_len,
                         const float *value_ori, const int16_t *type_ori, const int16_t *size_ori,
                         const int *mutate_indices,
                         const float *value_new, const int16_t *type_new, const int16_t *size_new,
                         float *value_res, int16_t *type_res, int16_t *size_res);

void evogp_oracle_crossover(int pop_size_ori, int pop_size_new, int gp_len,
                            const float *value_ori, const int16_t *type_ori, const int16_t *size_ori,
                            const int *left_idx, const int *right_idx,
                            const int *left_node_idx, const int *right_node_idx,
                            float *value_res, int16_t *type_res, int16_t *size_res);

void evogp_oracle_evaluate(unsigned pop_size, unsigned gp_len, unsigned var_len, unsigned out_len,
                           const float *value, const int16_t *type, const int16_t *size,
                           const float *variables, float *results);

/* threads <= 0: use every core OpenMP sees.  Returns the number of threads used. */
int evogp_oracle_sr_fitness(unsigned pop_size, unsigned data_points, unsigned gp_len,
                            unsigned var_len, unsigned out_len, int use_mse,
                            const float *value, const int16_t *type, const int16_t *size,
                            const float *variables, const float *labels, float *fitnesses,
                            int threads);

void evogp_oracle_batch_evaluate(unsigned pop_size, unsigned data_points, unsigned gp_len,
                                 unsigned var_len, unsigned out_len,
                                 const float *value, const int16_t *type, const int16_t *size,
                                 const float *variables, float *results);

/* structural validator: returns 0 when row is a well-formed prefix tree of length size[0]
 * (size[i] = 1 + sum of children sizes, arities by type), otherwise the 1-based index of the
 * first offending node (or -1 for a bad length).  Follows the invariant of
 * src/evogp/tree/tree.py:361-413. */
int evogp_oracle_validate_tree(int gp_len, const int16_t *type, const int16_t *size);

/* Sensitivity probe for parity tests (NOT reference semantics; 0 = off, the default): every libm-backed function result of the
 * evaluation entry points is moved by a pseudo-random number of ulps in [-ulps, +ulps] (seeded per tree from `seed`). */
void evogp_oracle_set_jitter(int ulps, unsigned seed);

#ifdef __cplusplus
}
#endif
#endif
