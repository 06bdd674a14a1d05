// common.hip — device-property cache, work-counter ring, event timer and error strings of the C ABI.
#include "launch.hpp"

#include <atomic>
#include <mutex>

namespace evogp {

static std::mutex g_mu;
static DeviceInfo g_info[64];

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

constexpr int kCounterRing = 4096;
struct CounterRing {
    unsigned *base = nullptr;
    std::atomic<unsigned> next{0};
};
static CounterRing g_ring[64];

unsigned *acquire_counter(hipStream_t stream, hipError_t *err) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    CounterRing &r = g_ring[dev];
    if (!r.base) {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!r.base) {
            unsigned *ptr = nullptr;
            hipError_t e = hipMalloc((void **)&ptr, kCounterRing * 4 * sizeof(unsigned));
            if (e != hipSuccess) { *err = e; return nullptr; }
            r.base = ptr;
        }
    }
    unsigned *slot = r.base + 4 * (r.next.fetch_add(1) % kCounterRing);
    hipError_t e = hipMemsetAsync(slot, 0, 4 * sizeof(unsigned), stream);
    if (e != hipSuccess) { *err = e; return nullptr; }
    *err = hipSuccess;
    return slot;
}

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

extern "C" int evogp_hip_abi_version(void) { return 1; }
