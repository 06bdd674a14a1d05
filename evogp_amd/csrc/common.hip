// common.hip — device-property cache, work-counter ring, event timer and error strings of the C ABI.
#include "launch.hpp"

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

namespace evogp {

static std::mutex g_mu;
static DeviceInfo g_info[64];

// Which XCC ids do workgroups of this device see?  The threaded-code interpreter cuts its dynamic tail into one region per
// XCD (HW_REG_XCC_ID); a region nobody runs on would never be worked off, so the count is measured, not assumed: a
// partitioned or differently configured device reports fewer ids.
__global__ void xcc_probe_kernel(unsigned *seen) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
    if (threadIdx.x == 0) atomicOr(seen, 1u << (id & 31u));
}

// Runs on a private non-blocking stream (never the null stream: no device-wide synchronisation with the caller's streams).
// Returns 0 when the probe itself failed (the caller then works with one region and tries again later).
static int probe_xcc_regions() {
    unsigned *d = nullptr, h = 0;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int regions = 0;
    if (hipMalloc((void **)&d, sizeof(unsigned)) == hipSuccess) {
        if (hipMemsetAsync(d, 0, sizeof(unsigned), s) == hipSuccess) {
            hipLaunchKernelGGL(xcc_probe_kernel, dim3(4096), dim3(64), 0, s, d);
            if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, s) == hipSuccess &&
                hipStreamSynchronize(s) == hipSuccess) {
                int n = 0;
                while (n < 8 && ((h >> n) & 1u)) ++n;  // ids 0 .. n-1 all present
                regions = 1;
                while (regions * 2 <= n) regions *= 2;
            }
        }
        (void)hipFree(d);
    }
    (void)hipGetLastError();
    (void)hipStreamDestroy(s);
    return regions;
}

const DeviceInfo &device_info() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceInfo &d = g_info[dev];
    if (d.device != dev) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            d.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            d.max_waves_per_cu = prop.maxThreadsPerMultiProcessor > 0 ? prop.maxThreadsPerMultiProcessor / 64 : 32;
            d.lds_per_cu = 160 * 1024; // gfx950: 160 KiB per CU (prop.sharedMemPerBlock reports the 64 KiB default cap)
        }
        d.device = dev;
    }
    return d;
}

// Number of dynamic regions (XCDs that run workgroups) of the current device: probed once, on the first call whose stream
// is not being captured into a graph (the probe allocates, launches and waits: none of that is legal during a capture;
// such a call works with one region and leaves the probe to a later one).
static int g_xcc[64];  // 0 = not probed yet
int xcc_regions(hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (g_xcc[dev] > 0) return g_xcc[dev];
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 1; }
    const int r = probe_xcc_regions();  // outside the lock: it waits for the device
    if (r <= 0) return 1;
    std::lock_guard<std::mutex> lock(g_mu);
    g_xcc[dev] = r;
    return r;
}

// Zeroed four-word counter slots.  Every stream has a ring of its own: a slot that comes round again is reused on the stream
// that used it last, so stream order alone puts the memset behind the slot's previous kernel -- no device-wide wait, nothing
// held across a wait, legal inside a stream capture (the ring of a capturing stream must exist already: it is allocated by
// the first call outside a capture).
constexpr int kCounterRing = 256;
struct CounterRing {
    hipStream_t stream = nullptr;
    unsigned *base = nullptr;
    unsigned next = 0;
};
static std::vector<CounterRing> g_ring[64];

__global__ void zero_words_kernel(unsigned *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

hipError_t zero_words_async(void *ptr, size_t nwords, hipStream_t stream) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (!capturing) return hipMemsetAsync(ptr, 0, nwords * sizeof(unsigned), stream);
    size_t blocks = (nwords + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned *)ptr, nwords);
    return hipGetLastError();
}

unsigned *acquire_counter(hipStream_t stream, hipError_t *err) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    unsigned *slot = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        CounterRing *r = nullptr;
        for (auto &c : g_ring[dev]) if (c.stream == stream) r = &c;
        if (!r) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
                (void)hipGetLastError();
                *err = hipErrorStreamCaptureUnsupported;   // call once on this stream before capturing
                return nullptr;
            }
            unsigned *ptr = nullptr;
            const hipError_t e = hipMalloc((void **)&ptr, kCounterRing * 4 * sizeof(unsigned));
            if (e != hipSuccess) { *err = e; return nullptr; }
            g_ring[dev].push_back(CounterRing{stream, ptr, 0});
            r = &g_ring[dev].back();
        }
        slot = r->base + 4 * (r->next++ % kCounterRing);
    }
    const hipError_t e = zero_words_async(slot, 4, stream);
    if (e != hipSuccess) { *err = e; return nullptr; }
    *err = hipSuccess;
    return slot;
}

// Per-stream scratch blocks for the SR-fitness call chain: two blocks of four words that alternate between calls.  The
// block a call uses was zeroed by the PREVIOUS call's first kernel on the same stream (stream order makes that safe), so
// the steady state needs no hipMemsetAsync in front of every call.
struct StreamScratch {
    unsigned *blocks = nullptr;  // 2 x kCallScratchWords
    int cur = 0;
    bool next_clean = false;     // the other block was zeroed by the last call's kernel
};
static std::mutex g_scratch_mu;
static std::vector<std::pair<hipStream_t, StreamScratch>> g_scratch[64];

unsigned *acquire_call_scratch(hipStream_t stream, unsigned **zero_for_next, hipError_t *err) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    StreamScratch *s = nullptr;
    for (auto &kv : g_scratch[dev]) if (kv.first == stream) s = &kv.second;
    if (!s) {
        g_scratch[dev].emplace_back(stream, StreamScratch{});
        s = &g_scratch[dev].back().second;
        hipError_t e = hipMalloc((void **)&s->blocks, 2 * kCallScratchWords * sizeof(unsigned));
        if (e != hipSuccess) { g_scratch[dev].pop_back(); *err = e; return nullptr; }
    }
    const int use = s->cur ^ 1;  // alternate
    // A call recorded into a HIP graph is replayed without this host code: it must carry its own memset and must not rely
    // on (or promise) zeroing across calls.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (capturing) {
        hipError_t e = zero_words_async(s->blocks + kCallScratchWords * use, kCallScratchWords, stream);
        if (e != hipSuccess) { *err = e; return nullptr; }
        s->cur = use;
        s->next_clean = false;
        *zero_for_next = nullptr;
        *err = hipSuccess;
        return s->blocks + kCallScratchWords * use;
    }
    if (!s->next_clean) {
        hipError_t e = zero_words_async(s->blocks + kCallScratchWords * use, kCallScratchWords, stream);
        if (e != hipSuccess) { *err = e; return nullptr; }
    }
    s->cur = use;
    s->next_clean = false;
    *zero_for_next = s->blocks + kCallScratchWords * (use ^ 1);
    *err = hipSuccess;
    return s->blocks + kCallScratchWords * use;
}

void call_scratch_next_is_clean(hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto &kv : g_scratch[dev]) if (kv.first == stream) kv.second.next_clean = true;
}

// Calibration reads for the FETCH_SIZE counter (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access
// pattern"): every lane reads W bytes per load instruction, consecutive lanes consecutive addresses -- W = 2 and 4 are the program
// compiler's node loads (type / size, value), 16 the wide streaming read the guide calibrated.  scripts/pmc_calibrate.py runs them
// over a buffer far larger than the Infinity Cache under rocprofv3 --pmc FETCH_SIZE.
template <int W>
__global__ __launch_bounds__(256) void calib_read_kernel(const unsigned char *buf, size_t bytes, unsigned *sink) {
    const size_t n = bytes / W, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (W == 2) acc += ((const unsigned short *)buf)[i];
        else if (W == 4) acc += ((const unsigned *)buf)[i];
        else { const uint4 v = ((const uint4 *)buf)[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;   // (keeps the loads)
}

}  // namespace evogp

extern "C" int evogp_hip_debug_calibrate_read(const void *buf, unsigned long long bytes, int width, unsigned *sink, void *stream) {
    using namespace evogp;
    const dim3 grid(256 * 16), block(256);
    if (width == 2) hipLaunchKernelGGL(calib_read_kernel<2>, grid, block, 0, (hipStream_t)stream, (const unsigned char *)buf, (size_t)bytes, sink);
    else if (width == 4) hipLaunchKernelGGL(calib_read_kernel<4>, grid, block, 0, (hipStream_t)stream, (const unsigned char *)buf, (size_t)bytes, sink);
    else if (width == 16) hipLaunchKernelGGL(calib_read_kernel<16>, grid, block, 0, (hipStream_t)stream, (const unsigned char *)buf, (size_t)bytes, sink);
    else return EVOGP_E_BADARG;
    return (int)hipGetLastError();
}

// Debugging aid (not declared in the header): both call-scratch blocks of `stream` and the index of the current one, copied to the
// host after a device synchronisation.  host_words: 2 * 2304 words.
extern "C" int evogp_hip_debug_call_scratch(void *stream, unsigned *host_words, int *current) {
    using namespace evogp;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto &kv : g_scratch[dev & 63])
        if (kv.first == (hipStream_t)stream) {
            (void)hipDeviceSynchronize();
            *current = kv.second.cur;
            return (int)hipMemcpy(host_words, kv.second.blocks, 2 * kCallScratchWords * sizeof(unsigned), hipMemcpyDeviceToHost);
        }
    return -1;
}

namespace evogp {

struct TimerSlot {
    hipEvent_t begin = nullptr, end = nullptr;
};
static TimerSlot g_timer[64];

} // namespace evogp

using namespace evogp;

extern "C" int evogp_hip_timer_begin(evogp_stream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    TimerSlot &t = g_timer[dev & 63];
    hipError_t e;
    if (!t.begin) {
        if ((e = hipEventCreate(&t.begin)) != hipSuccess) return (int)e;
        if ((e = hipEventCreate(&t.end)) != hipSuccess) return (int)e;
    }
    return (int)hipEventRecord(t.begin, (hipStream_t)stream);
}

extern "C" int evogp_hip_timer_end(evogp_stream_t stream, float *elapsed_ms) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    TimerSlot &t = g_timer[dev & 63];
    if (!t.begin || !elapsed_ms) return EVOGP_E_NULLPTR;
    hipError_t e;
    if ((e = hipEventRecord(t.end, (hipStream_t)stream)) != hipSuccess) return (int)e;
    if ((e = hipEventSynchronize(t.end)) != hipSuccess) return (int)e;
    return (int)hipEventElapsedTime(elapsed_ms, t.begin, t.end);
}

extern "C" const char *evogp_hip_error_string(int code) {
    switch (code) {
    case 0: return "success";
    case EVOGP_E_BADARG: return "evogp: size argument out of range (pop/gp_len/var_len/out_len/data_points must be > 0, gp_len <= 1024, probabilities in [0,1], kernel_type in 0..4)";
    case EVOGP_E_NULLPTR: return "evogp: a required pointer is NULL";
    case EVOGP_E_UNSUPPORTED: return "evogp: out_len larger than the interpreter's output staging area (256)";
    default: return hipGetErrorString((hipError_t)code);
    }
}

namespace evogp {
int env_int(const char *name, int def) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : def;
}
}  // namespace evogp

// ---- division mode of the SR-fitness fast path ---------------------------------------------------------------------------
static std::atomic<int> g_sr_division{-1};
namespace evogp {
int sr_division_mode() {
    int m = g_sr_division.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *e = getenv("EVOGP_SR_DIV");
        m = EVOGP_DIV_SHORT;
        if (e && (e[0] == 'i' || e[0] == 'I' || e[0] == '0')) m = EVOGP_DIV_IEEE;
        if (e && (e[0] == 'f' || e[0] == 'F' || e[0] == '1')) m = EVOGP_DIV_FAST;
        g_sr_division.store(m, std::memory_order_relaxed);
    }
    return m;
}
}  // namespace evogp

extern "C" int evogp_hip_set_sr_division(int mode) {
    if (mode != EVOGP_DIV_IEEE && mode != EVOGP_DIV_FAST && mode != EVOGP_DIV_SHORT) return EVOGP_E_BADARG;
    g_sr_division.store(mode, std::memory_order_relaxed);
    return EVOGP_OK;
}

extern "C" int evogp_hip_get_sr_division(void) { return evogp::sr_division_mode(); }

extern "C" int evogp_hip_abi_version(void) { return 5; }
