// tools/launch_rate.hip -- how many dependent kernel launches per second does one gfx950 take, in total, when T
// host threads each drive their own stream?  Separates "the device retires commands at a fixed rate" from "the
// kernels themselves keep the CUs busy" for the in-flight registration mode of bench.py.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_rate.hip -o build/launch_rate -lpthread && build/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_tiny(unsigned *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
// ~`iters` dependent FMAs per lane: a kernel with a known duration that occupies `blocks` workgroups
__global__ void k_spin(float *p, int iters) {
    float a = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.f) p[0] = a;
}

static double run(int T, int N, int blocks, int iters, bool graph) {
    std::vector<std::thread> th;
    std::vector<hipStream_t> st(T);
    std::vector<void *> buf(T);
    std::vector<hipGraphExec_t> ex(T);
    for (int t = 0; t < T; ++t) {
        if (getenv("LR_CUMASK")) {   // a CU-masked stream (all CUs enabled) owns a hardware queue of its own
            uint32_t mask[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
            if (hipExtStreamCreateWithCUMask(&st[t], 8, mask) != hipSuccess) { printf("cumask stream failed\n"); exit(1); }
        } else
            hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking);
        hipMalloc(&buf[t], 4096 * 4);
        hipMemset(buf[t], 0, 4096 * 4);
        if (graph) {
            hipGraph_t g;
            hipStreamBeginCapture(st[t], hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < 100; ++i) {
                if (iters) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[t], (float *)buf[t], iters);
                else hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(64), 0, st[t], (unsigned *)buf[t]);
            }
            hipStreamEndCapture(st[t], &g);
            hipGraphInstantiate(&ex[t], g, nullptr, nullptr, 0);
        }
    }
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            if (graph) {
                for (int i = 0; i < N / 100; ++i) hipGraphLaunch(ex[t], st[t]);
            } else {
                for (int i = 0; i < N; ++i) {
                    if (iters) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[t], (float *)buf[t], iters);
                    else hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(64), 0, st[t], (unsigned *)buf[t]);
                }
            }
            hipStreamSynchronize(st[t]);
        });
    for (auto &x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int t = 0; t < T; ++t) { hipStreamDestroy(st[t]); hipFree(buf[t]); }
    return s;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int N = 5000;
    printf("%-8s %-6s %-8s %-7s %12s %12s\n", "mode", "thr", "blocks", "iters", "launch/s", "us/launch/str");
    const bool quick = getenv("LR_QUICK") != nullptr;
    for (int graph = 0; graph < (quick ? 1 : 2); ++graph)
        for (int iters : {0, 2000})
            for (int blocks : {1, 256})
                if (!quick || (iters == 2000 && blocks == 1))
                for (int T : {1, 2, 4, 8, 16}) {
                    const double s = run(T, N, blocks, iters, graph);
                    printf("%-8s %-6d %-8d %-7d %12.0f %12.2f\n", graph ? "graph" : "direct", T, blocks, iters, T * (double)N / s, s / N * 1e6);
                }
    return 0;
}
