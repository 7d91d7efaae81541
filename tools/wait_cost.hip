// tools/wait_cost.hip -- what does one host wait for a short kernel cost in CPU time and in wake-up latency, per
// waiting method?  (bench.py keeps 8 registrations in flight per GPU with ~90 host waits each; the GPU boxes give
// a container 16 CPUs, so spinning waits cap the number of registrations in flight.)
//   hipcc --offload-arch=gfx950 -O3 tools/wait_cost.hip -o build/wait_cost && build/wait_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <thread>

__global__ void k_spin(float *p, int iters) {
    float a = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.f) p[0] = a;
}

static double cpu_now(clockid_t c) { timespec t; clock_gettime(c, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float *buf;
    hipMalloc(&buf, 4096);
    hipMemset(buf, 0, 4096);
    hipEvent_t ev_spin, ev_block;
    hipEventCreateWithFlags(&ev_spin, hipEventDisableTiming);
    hipEventCreateWithFlags(&ev_block, hipEventDisableTiming | hipEventBlockingSync);
    const int N = 300;
    printf("%-22s %8s %12s %14s %14s\n", "method", "iters", "wall us/rep", "thread cpu us", "process cpu us");
    for (int iters : {1000, 6000, 60000}) {
        for (int m = 0; m < 5; ++m) {
            const char *name[] = {"hipStreamSynchronize", "event spin", "event blocking", "query + yield", "query + nanosleep 20us"};
            for (int w = 0; w < 10; ++w) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, buf, iters); hipStreamSynchronize(st); }
            const double c0 = cpu_now(CLOCK_THREAD_CPUTIME_ID), p0 = cpu_now(CLOCK_PROCESS_CPUTIME_ID);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, buf, iters);
                switch (m) {
                    case 0: hipStreamSynchronize(st); break;
                    case 1: hipEventRecord(ev_spin, st); hipEventSynchronize(ev_spin); break;
                    case 2: hipEventRecord(ev_block, st); hipEventSynchronize(ev_block); break;
                    case 3: while (hipStreamQuery(st) == hipErrorNotReady) std::this_thread::yield(); break;
                    case 4: {
                        timespec ts{0, 20000};
                        while (hipStreamQuery(st) == hipErrorNotReady) nanosleep(&ts, nullptr);
                        break;
                    }
                }
            }
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%-22s %8d %12.1f %14.1f %14.1f\n", name[m], iters, wall / N * 1e6, (cpu_now(CLOCK_THREAD_CPUTIME_ID) - c0) / N * 1e6,
                   (cpu_now(CLOCK_PROCESS_CPUTIME_ID) - p0) / N * 1e6);
        }
    }
    return 0;
}
