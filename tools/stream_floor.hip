// tools/stream_floor.hip -- how fast can gfx950 stream the 28 B/point of one K1 pass at the bench size?
// Plain read kernels over 7 arrays of n floats (x y z nx ny nz + shapeIndex) with the same tiling as
// k_score_mark_batch, to separate "HBM roofline" from "what a 28 MB launch can reach at all".
//   hipcc --offload-arch=gfx950 -O3 tools/stream_floor.hip -o /tmp/stream_floor && /tmp/stream_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PPT>
__global__ __launch_bounds__(256) void k_read(const float *__restrict__ a, size_t pitch, uint32_t n, float *__restrict__ out) {
    const uint32_t base = (blockIdx.x * 256 + threadIdx.x) * PPT;
    float acc = 0.f;
    if (base + PPT <= n) {
#pragma unroll
        for (int arr = 0; arr < 7; ++arr)
#pragma unroll
            for (int k = 0; k < PPT; k += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(a + arr * pitch + base + k);
                acc += v.x + v.y + v.z + v.w;
            }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    for (uint32_t n : {1000000u, 4000000u, 16000000u, 64000000u}) {
        const size_t pitch = ((size_t)n + 3) & ~(size_t)3;
        float *d, *o;
        hipMalloc(&d, 7 * pitch * 4);
        hipMalloc(&o, 64);
        hipMemset(d, 0, 7 * pitch * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int ppt : {4, 8, 16}) {
            const uint32_t nb = (n + 256 * ppt - 1) / (256 * ppt);
            auto launch = [&]() {
                if (ppt == 4) hipLaunchKernelGGL(k_read<4>, dim3(nb), dim3(256), 0, 0, d, pitch, n, o);
                else if (ppt == 8) hipLaunchKernelGGL(k_read<8>, dim3(nb), dim3(256), 0, 0, d, pitch, n, o);
                else hipLaunchKernelGGL(k_read<16>, dim3(nb), dim3(256), 0, 0, d, pitch, n, o);
            };
            for (int i = 0; i < 20; ++i) launch();
            hipDeviceSynchronize();
            const int iters = 200;
            hipEventRecord(e0, 0);
            for (int i = 0; i < iters; ++i) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / iters, gb = 28.0 * n / 1e9;
            printf("n %9u  points/lane %2d  blocks %7u  %8.2f us/launch  %7.1f GB/s\n", n, ppt, nb, us, gb / (us * 1e-6));
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
