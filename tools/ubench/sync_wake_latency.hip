// How long after a kernel's end does the host know?  hipStreamSynchronize (the driver's wait) against the host polling a word in
// pinned host memory that the kernel's last thread writes.  Same kernel (spins ~300 us on s_memrealtime), 500 repetitions each,
// host wall time from launch to return; the difference of the medians is the wake-up cost a synchronous entry point pays per call.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/sync_wake tools/ubench/sync_wake_latency.hip && /tmp/sync_wake
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin_kernel(unsigned long long ticks, volatile unsigned int *flag, unsigned int seq, double *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (sink) sink[0] = (double)seq;  // (a device-side result, as the fit's scalars are)
    if (flag) {
        __threadfence_system();
        *flag = seq;
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    unsigned int *flag;
    hipHostMalloc((void **)&flag, 64, hipHostMallocDefault);
    *flag = 0;
    double *sink, *h_sink;
    hipMalloc((void **)&sink, 4096);
    hipHostMalloc((void **)&h_sink, 4096, hipHostMallocDefault);
    const unsigned long long ticks = 30000;  // 300 us at 100 MHz
    const int reps = 500;
    std::vector<double> a, b, c;
    unsigned int seq = 0;
    for (int r = 0; r < reps + 20; r++) {
        double t0 = now_us();
        spin_kernel<<<1, 64, 0, st>>>(ticks, nullptr, ++seq, sink);
        hipStreamSynchronize(st);
        double t1 = now_us();
        if (r >= 20) a.push_back(t1 - t0);
    }
    for (int r = 0; r < reps + 20; r++) {  // as the library ends a call today: D2H copy of the scalars, then the wait
        double t0 = now_us();
        spin_kernel<<<1, 64, 0, st>>>(ticks, nullptr, ++seq, sink);
        hipMemcpyAsync(h_sink, sink, 1024, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        double t1 = now_us();
        if (r >= 20) c.push_back(t1 - t0);
    }
    for (int r = 0; r < reps + 20; r++) {
        double t0 = now_us();
        spin_kernel<<<1, 64, 0, st>>>(ticks, flag, ++seq, sink);
        while (*(volatile unsigned int *)flag != seq) { }
        double t1 = now_us();
        if (r >= 20) b.push_back(t1 - t0);
    }
    hipStreamSynchronize(st);
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto p90 = [](std::vector<double> &v) { return v[v.size() * 9 / 10]; };
    printf("kernel spins 300 us; host wall time launch -> return, median (p90) of %d:\n", reps);
    printf("  hipStreamSynchronize                     %.1f us (%.1f)\n", med(a), p90(a));
    printf("  hipMemcpyAsync D2H 1 KB + synchronize    %.1f us (%.1f)\n", med(c), p90(c));
    printf("  host polls a pinned word the kernel sets %.1f us (%.1f)\n", med(b), p90(b));
    return 0;
}
