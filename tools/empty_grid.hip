// tools/empty_grid.hip -- what does a launch cost whose workgroups have nothing to do?  The extraction's scan kernels are launched
// over the tiles of all sixteen clouds of a group (~15 600 workgroups of 256 lanes) from a captured graph; behind the first
// iteration most of those workgroups read two words of the loop state and return.  B workgroups x (two dependent loads, return),
// 100 launches per graph, on T streams at once.
//   hipcc --offload-arch=gfx950 -O3 tools/empty_grid.hip -o build_tools/empty_grid -lpthread && build_tools/empty_grid
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

struct St { unsigned nc, pad[63]; };
struct Tab { St *st; unsigned pad[30]; };
__global__ __launch_bounds__(256) void k_empty(const Tab *tab, unsigned *out) {
    const Tab &t = tab[blockIdx.x & 15];
    if (t.st->nc == 0) return;
    out[blockIdx.x * 256 + threadIdx.x] = 1;
}

static double run(int T, int blocks) {
    std::vector<hipStream_t> st(T);
    std::vector<hipGraphExec_t> ex(T);
    for (int t = 0; t < T; ++t) {
        hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking);
        St *s; Tab *tab; unsigned *out;
        hipMalloc(&s, 16 * sizeof(St)); hipMemset(s, 0, 16 * sizeof(St));
        hipMalloc(&out, 4);
        Tab h[16];
        for (int i = 0; i < 16; ++i) h[i].st = s + i;
        hipMalloc(&tab, sizeof(h)); hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
        hipGraph_t g;
        hipStreamBeginCapture(st[t], hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st[t], tab, out);
        hipStreamEndCapture(st[t], &g);
        hipGraphInstantiate(&ex[t], g, nullptr, nullptr, 0);
    }
    hipDeviceSynchronize();
    for (int t = 0; t < T; ++t) hipGraphLaunch(ex[t], st[t]);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { for (int i = 0; i < 20; ++i) hipGraphLaunch(ex[t], st[t]); hipStreamSynchronize(st[t]); });
    for (auto &x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return s / 2000 * 1e6;   // us per launch of one stream's chain
}

int main() {
    for (int T : {1, 4})
        for (int b : {16, 128, 1024, 2048, 4096, 16384}) printf("streams %d workgroups %6d: %7.2f us per launch\n", T, b, run(T, b));
    return 0;
}
