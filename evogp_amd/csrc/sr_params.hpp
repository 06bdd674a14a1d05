// sr_params.hpp — argument block and marks shared by the SR-fitness kernels (sr_fitness.hip, sr_tc.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evogp {

constexpr uint32_t kSentinelDeep = 0x7FC0DEEDu;  // quiet-NaN payload: "evaluate me in the general kernel"
constexpr uint32_t kSentinelHeavy = 0x7FC0FEEDu; // quiet-NaN payload: "evaluate me in the FULL register kernel"
constexpr uint32_t kSentinelGeneral = 0x7FC0BEEFu; // quiet-NaN payload: "compile me with the general program compiler" (sr_tc.hip)
constexpr int kMaxBatch = 64;                    // trees per batch (LDS partial-sum slots)
constexpr int kMaxWaves = 16;

struct SrParams {
    const float *value;
    const int16_t *type;
    const int16_t *size;
    const float *X;  // [D][var_len]
    const float *y;  // [D][out_len]
    float *fitness;  // [pop]            (fitness mode)
    float *results;  // [pop][D][out_len] (store mode: batch evaluation, no reduction)
    unsigned *counter; // batch counter (zeroed before the launch)
    int pop, D, gp_len, var_len, out_len;
    int use_mse;
    int classify;    // threaded code only: y holds ONE column of int32 class labels; fitness[t] receives the number of rows whose
                     // arg-max output is their label (as a float; tc_count_kernel turns the words into the caller's counts)
    int batch;       // trees per batch, <= kMaxBatch
    int ntiles;      // ceil(D / (64*K))
    int only_marked; // != 0: only trees whose output word holds kSentinelHeavy are evaluated
    int mark_sample; // trees the marking kernel SAMPLED for marks[2] (0: no estimate)
    unsigned long long *stats; // optional cycle counters (profiling builds of the bench only), else nullptr
    unsigned *zero_next; // four words the first kernel of the chain zeroes for the next call on this stream (or nullptr)
    int mark_chunks;     // the threaded code ran the population in this many chunks: chunk c's flags are marks[c * kCallScratchChunkWords + i]
    hipEvent_t prof_mid; // profiling (evogp_hip_debug_profile): recorded between the compiler and the interpreter launch, or nullptr
    int hint_general;         // != 0: launch the general compiler behind the packed one.  0 only where the caller's function mask (below) says
                              // that no tree can be left for it; a tree that carries its sentinel after all is evaluated by the last follow-up kernel
    unsigned func_mask;       // bit f: function id f (defs.h:10-57) may occur in the forest; 0 = unknown.  A caller that knows the
                              // forest's function set (evogp_amd.tree.Forest tracks it from the descriptors its trees came from) lets
                              // the call skip launches that such a forest cannot need.  A call that comes without one (the reference's operator)
                              // is given the mask the last completed call on a forest of its shape observed (run_population, tc_learned_class)
    unsigned *feedback;       // a call without a function mask: where its interpreter launch publishes the function class the program compiler
                              // found (tc_learned_class), or nullptr
    unsigned feedback_expected;   // ... the word the host read there before the call (the launch writes only what differs)
    unsigned *marks; // [0] != 0: some tree carries kSentinelHeavy, [1] != 0: some tree carries kSentinelDeep,
                     // [2]: how many of the mark_sample sampled trees were marked heavy (may be nullptr)
};

// Threaded-code path (sr_tc.hip): compile the population into fused programs and interpret them with the
// assembly core.  Returns hipSuccess and sets *handled when it took the launch (trees it could not take are
// marked kSentinelHeavy / NaN in p.fitness for the follow-up kernels); *handled == false means "not eligible".
hipError_t launch_threaded_code(const SrParams &p, hipStream_t stream, bool *handled, int *mark_sample, int *mark_chunks);
int tc_learned_class(int pop, int gp_len, hipStream_t stream, unsigned **publish, unsigned *word);
int tc_detect_class(const SrParams &p, unsigned *flags, hipStream_t stream);
unsigned tc_store_class(unsigned *publish, int cls);

// Classification epilogue on the threaded code (sr_fitness.hip: it shares the call-scratch chain with the fitness calls):
// counts[t] = rows whose arg-max output equals labels[row]; trees the path cannot take come back with kDeepCountBit set and
// wide_marks[1] raised for sr_wide.hip's recount kernel.  *handled == false: not eligible, nothing was launched.
constexpr unsigned kDeepCountBit = 0x80000000u;  // set in counts[t] for a tree that wide_deep_count_kernel still has to count
hipError_t run_argmax_count_threaded(const SrParams &p, const int *labels, unsigned *counts, unsigned *wide_marks, hipStream_t stream, bool *handled);
hipError_t launch_tc_count(unsigned *counts, int pop, const int *labels, int D, unsigned *wide_marks, hipStream_t stream);

// Tile-group kernel for shapes the register kernels cannot keep resident (sr_wide.hip): STORE mode of batch_evaluate.
hipError_t launch_wide_store(const SrParams &p, hipStream_t stream);

} // namespace evogp
