// Latency of scalar loads of a 256-byte record (gfx950): 1..4 x s_load_dwordx16, cold (first touch after a flush) vs
// L2-warm (the same wave pulled the lines in with one vector load) vs repeated (scalar-cache hit).
// One wave per block, 1024 blocks; each wave owns its own record; result = average ticks per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NLOADS, int MODE>  // MODE 0 cold, 1 warmed by a vector load, 2 second scalar read
__global__ void k(const uint4 *buf, unsigned long long *out, size_t stride16) {
    const uint4 *rec = buf + (size_t)blockIdx.x * stride16;
    unsigned long long t0, t1;
    float sink = 0;
    if (MODE == 1) { uint4 v = rec[threadIdx.x & 15]; sink += v.x; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); for (int i = 0; i < 2000; ++i) asm volatile("s_nop 7"); }
    if (MODE == 2) { asm volatile("s_load_dwordx16 s[36:51], %0, 0x0\n\ts_load_dwordx16 s[52:67], %0, 0x40\n\ts_load_dwordx16 s[68:83], %0, 0x80\n\ts_load_dwordx16 s[84:99], %0, 0xc0\n\ts_waitcnt lgkmcnt(0)" :: "s"(rec) : "memory", "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99"); }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (NLOADS >= 1) asm volatile("s_load_dwordx16 s[36:51], %0, 0x0" :: "s"(rec) : "memory", "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51");
    if (NLOADS >= 2) asm volatile("s_load_dwordx16 s[52:67], %0, 0x40" :: "s"(rec) : "memory", "s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67");
    if (NLOADS >= 3) asm volatile("s_load_dwordx16 s[68:83], %0, 0x80" :: "s"(rec) : "memory", "s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83");
    if (NLOADS >= 4) asm volatile("s_load_dwordx16 s[84:99], %0, 0xc0" :: "s"(rec) : "memory", "s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (unsigned long long)(sink == 12345.f);
}
static void flush(char *big, size_t n) { (void)hipMemset(big, 1, n); (void)hipDeviceSynchronize(); }
template <int N, int M> static void run(const char *what, const uint4 *buf, unsigned long long *dout, char *big, size_t bign, int waves_per_cu) {
    const int blocks = 256 * waves_per_cu;
    flush(big, bign);
    hipLaunchKernelGGL((k<N, M>), dim3(blocks), dim3(64), 0, 0, buf, dout, (size_t)48);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), dout, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-8s %d x s_load_dwordx16, %2d waves/CU: %8.0f ticks\n", what, N, waves_per_cu, s / blocks);
}
int main() {
    uint4 *buf; unsigned long long *dout; char *big; const size_t bign = 1ull << 30;
    (void)hipMalloc(&buf, 768ull * 8192); (void)hipMalloc(&dout, 8 * 8192); (void)hipMalloc(&big, bign);
    (void)hipMemset(buf, 0, 768ull * 8192);
    for (int w : {1, 16}) {
        run<1, 0>("cold", buf, dout, big, bign, w); run<2, 0>("cold", buf, dout, big, bign, w); run<4, 0>("cold", buf, dout, big, bign, w);
        run<1, 1>("warmed", buf, dout, big, bign, w); run<4, 1>("warmed", buf, dout, big, bign, w);
        run<1, 2>("repeat", buf, dout, big, bign, w); run<4, 2>("repeat", buf, dout, big, bign, w);
    }
    return 0;
}
