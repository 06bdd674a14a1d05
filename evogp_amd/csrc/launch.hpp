// launch.hpp — host-side helpers shared by the C-ABI launchers (device properties, work counters).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/evogp_hip.h"
#include "../../include/evogp_hip_debug.h"   // (definitions are checked against their declarations)

#ifndef EVOGP_SR_DEFAULT_K
#define EVOGP_SR_DEFAULT_K 4
#endif
#ifndef EVOGP_SR_DEFAULT_ASM
#define EVOGP_SR_DEFAULT_ASM 3
#endif
#ifndef EVOGP_SR_DEFAULT_DEPTH
#define EVOGP_SR_DEFAULT_DEPTH 16
#endif

namespace evogp {

struct DeviceInfo {
    int device = -1;
    int num_cus = 256;
    int max_waves_per_cu = 32;
    size_t lds_per_cu = 160 * 1024;
};

// Largest power of two <= the number of XCDs that actually run workgroups (<= 8): probed once per device, outside stream
// captures (a call on a capturing stream gets 1 until some other call has probed).
int xcc_regions(hipStream_t stream);

// Properties of the CURRENT device (cached per device id).
const DeviceInfo &device_info();

// Four zero-initialised 32-bit words in device memory (a work counter, or the pending-marks flags) for one launch on `stream`
// (dynamic batch distribution inside persistent kernels).  Counters come from a small per-device
// ring; the slot is cleared with hipMemsetAsync on `stream` before it is handed out, so reuse is
// ordered by the stream.  Returns nullptr and sets *err on failure.
unsigned *acquire_counter(hipStream_t stream, hipError_t *err);

// Zero `nwords` 32-bit words on `stream`.  Inside a stream capture a small kernel does it: a memset node recorded from hipMemsetAsync
// was found to write other values than zero when its graph is replayed (ROCm 7.2, gfx950: the call-scratch block of a captured
// tree_SR_fitness came back holding what looks like another kernel's argument block; scripts/dbg/graph_ring4.py), which left the work
// counters of every replay but the first dirty.
hipError_t zero_words_async(void *ptr, size_t nwords, hipStream_t stream);

// integer value of an environment switch, or `def` when it is not set
int env_int(const char *name, int def);

// 0 = IEEE division in every kernel (default), 1 = the threaded-code fitness path uses the fast division
// (evogp_hip_set_sr_division / EVOGP_SR_DIV=fast)
int sr_division_mode();

// A zeroed scratch block (kCallScratchWords words) for one SR-fitness call on `stream`, without a memset in the steady state: *zero_for_next is the block
// the NEXT call on this stream will get; a kernel of this call zeroes it and the caller then reports that with
// call_scratch_next_is_clean (otherwise the next acquire memsets it).
// Layout of a block: words 0..3 = pending-marks flags and work counters (sr_params.hpp), then one work counter per XCD of
// the threaded-code kernel's dynamic tail, each on its own 128-byte line (word 32 * (1 + xcd)).
constexpr int kCallScratchXcds = 8;
constexpr int kCallScratchChunkWords = 32 * (1 + kCallScratchXcds);   // one flag line + one counter line per XCD
constexpr int kCallScratchChunks = 8;                                  // population chunks of one call (sr_tc.hip), each with its own lines
constexpr int kCallScratchWords = kCallScratchChunkWords * kCallScratchChunks;
unsigned *acquire_call_scratch(hipStream_t stream, unsigned **zero_for_next, hipError_t *err);
void call_scratch_next_is_clean(hipStream_t stream);

} // namespace evogp
