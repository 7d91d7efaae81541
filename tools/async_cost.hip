// Which HIP operations wake the runtime's background thread?  Process CPU vs calling-thread CPU per operation.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/async_cost tools/async_cost.hip && /tmp/async_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <cstdint>
#include <functional>
static double now(clockid_t c) { timespec ts; clock_gettime(c, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
__global__ void k_work(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
__global__ void k_flag(volatile uint32_t *flag, uint32_t v) { if (threadIdx.x == 0) { __threadfence_system(); *flag = v; } }
__global__ void k_copy_out(const uint32_t *src, uint32_t *dst, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }
static void wait_stream(hipStream_t s) {
    while (hipStreamQuery(s) == hipErrorNotReady) { timespec ts{0, 20000}; nanosleep(&ts, nullptr); }
}
int main() {
    hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int n = 1 << 20;
    float *d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
    uint32_t *dsmall; hipMalloc(&dsmall, 4096);
    uint32_t *hpin; hipHostMalloc(&hpin, 1 << 20, hipHostMallocDefault);
    uint32_t *hmap; hipHostMalloc(&hmap, 1 << 20, hipHostMallocMapped);
    uint32_t *hmap_dev; hipHostGetDevicePointer((void **)&hmap_dev, hmap, 0);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    uint32_t seq = 0;
    auto wait_flag = [&](uint32_t v) { while (*(volatile uint32_t *)hmap != v) { timespec ts{0, 20000}; nanosleep(&ts, nullptr); } };
    struct V { const char *name; std::function<void()> body; };
    V vs[] = {
        {"10 kernels + flag kernel, poll MEMORY only", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); ++seq; hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, hmap_dev, seq); wait_flag(seq); }},
        {"10 kernels + flag kernel, poll memory, then 1 query", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); ++seq; hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, hmap_dev, seq); wait_flag(seq); wait_stream(s); }},
        {"graph of 10 kernels + flag kernel, poll memory", [&] { hipGraphLaunch(ge, s); ++seq; hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, hmap_dev, seq); wait_flag(seq); }},
        {"10 kernels + wait", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); wait_stream(s); }},
        {"10 kernels + hipStreamSynchronize", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); hipStreamSynchronize(s); }},
        {"10 kernels + 1 d2h 256B pinned + wait", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); hipMemcpyAsync(hpin, dsmall, 256, hipMemcpyDeviceToHost, s); wait_stream(s); }},
        {"10 kernels + 4 d2h 256B pinned + wait", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); for (int q = 0; q < 4; ++q) hipMemcpyAsync(hpin + 64 * q, dsmall, 256, hipMemcpyDeviceToHost, s); wait_stream(s); }},
        {"10 kernels + 1 d2h 256KB pinned + wait", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); hipMemcpyAsync(hpin, d, 262144, hipMemcpyDeviceToHost, s); wait_stream(s); }},
        {"10 kernels + copy kernel to mapped host + wait", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); hipLaunchKernelGGL(k_copy_out, dim3(1), dim3(64), 0, s, dsmall, hmap_dev, 64); wait_stream(s); }},
        {"10 kernels + 1 h2d 256B pinned + wait", [&] { hipMemcpyAsync(dsmall, hpin, 256, hipMemcpyHostToDevice, s); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); wait_stream(s); }},
        {"10 kernels + memsetAsync 4KB + wait", [&] { hipMemsetAsync(dsmall, 0, 4096, s); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); wait_stream(s); }},
        {"graph of 10 kernels + wait", [&] { hipGraphLaunch(ge, s); wait_stream(s); }},
        {"10 kernels + event record/wait on 2nd stream + wait", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); hipEventRecord(ev, s); hipStreamWaitEvent(s2, ev, 0); hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, d, 64); wait_stream(s); wait_stream(s2); }},
        {"10 kernels + d2d 4KB + wait", [&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n); hipMemcpyAsync(dsmall, d, 4096, hipMemcpyDeviceToDevice, s); wait_stream(s); }},
        {"24MB h2d pinned + wait", [&] { static float *big = nullptr, *dbig = nullptr; if (!big) { hipHostMalloc(&big, 24 << 20, hipHostMallocDefault); hipMalloc(&dbig, 24 << 20); } hipMemcpyAsync(dbig, big, 24 << 20, hipMemcpyHostToDevice, s); wait_stream(s); }},
    };
    for (auto &v : vs) {
        for (int i = 0; i < 50; ++i) v.body();
        const int N = 1000;
        const double w0 = now(CLOCK_MONOTONIC), p0 = now(CLOCK_PROCESS_CPUTIME_ID), t0 = now(CLOCK_THREAD_CPUTIME_ID);
        for (int i = 0; i < N; ++i) v.body();
        const double w1 = now(CLOCK_MONOTONIC), p1 = now(CLOCK_PROCESS_CPUTIME_ID), t1 = now(CLOCK_THREAD_CPUTIME_ID);
        printf("%-52s wall %7.1f us  thread cpu %6.1f us  other threads %6.1f us\n", v.name, 1e6 * (w1 - w0) / N, 1e6 * (t1 - t0) / N,
               1e6 * ((p1 - p0) - (t1 - t0)) / N);
    }
    return 0;
}
