// tools/pmc_calibrate.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for access patterns of KNOWN byte counts
// (/opt/skills/guides/MI355X_MICROARCH.md calibrates x2 for wide coalesced streaming reads only).
//   k_cal_stream16   every lane one float4 (16 B), coalesced:                       n * 16 B read
//   k_cal_stream4    every lane one float (4 B), coalesced:                         n * 4 B read
//   k_cal_gather12   lane i reads x, y, z (12 B) of row idx[i] of an N x 6 float array (24 B rows, random rows, each once):
//                    12 B useful per lane; the rows' 64 B sectors: 1 or 2 per row
//   k_cal_gather4    lane i reads one float at a random position (each once)
//   k_cal_write16    every lane stores one float4, coalesced:                       n * 16 B written
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/pmc_calibrate.sh); prints the expected byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_cal_stream16(const float4 *__restrict__ a, size_t n, float *sink) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1234.5f) *sink = acc;
}
__global__ void k_cal_stream4(const float *__restrict__ a, size_t n, float *sink) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 1234.5f) *sink = acc;
}
__global__ void k_cal_gather12(const float *__restrict__ rows, const uint32_t *__restrict__ idx, size_t m, float *sink) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        const float *p = rows + 6 * (size_t)idx[i];
        acc += p[0] + p[1] + p[2];
    }
    if (acc == 1234.5f) *sink = acc;
}
__global__ void k_cal_gather4(const float *__restrict__ a, const uint32_t *__restrict__ idx, size_t m, float *sink) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) acc += a[idx[i]];
    if (acc == 1234.5f) *sink = acc;
}
__global__ void k_cal_write16(float4 *__restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const size_t bytes = 1ull << 30;                 // 1 GiB: four times the Infinity Cache
    const size_t n16 = bytes / 16, n4 = bytes / 4, rows = bytes / 24, m = rows / 4;   // gathers touch every fourth row / 64th float once
    float *buf, *sink;
    uint32_t *idx;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&idx, m * 4));
    CK(hipMemset(buf, 0, bytes));
    std::vector<uint32_t> h(m);
    std::mt19937 rng(7);
    for (size_t i = 0; i < m; ++i) h[i] = (uint32_t)(4 * i);
    std::shuffle(h.begin(), h.end(), rng);
    CK(hipMemcpy(idx, h.data(), m * 4, hipMemcpyHostToDevice));
    size_t sectors = 0;                              // 64 B sectors the x, y, z of the gathered rows lie in
    for (size_t i = 0; i < m; ++i) { const size_t b0 = (size_t)h[i] * 24, b1 = b0 + 11; sectors += (b1 / 64 != b0 / 64) ? 2 : 1; }
    const dim3 g(4096), b(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_cal_stream16, g, b, 0, 0, (const float4 *)buf, n16, sink);
        hipLaunchKernelGGL(k_cal_stream4, g, b, 0, 0, buf, n4, sink);
        hipLaunchKernelGGL(k_cal_gather12, g, b, 0, 0, buf, idx, m, sink);
        for (size_t i = 0; i < m; ++i) h[i] = (uint32_t)((64 * i) % n4);
        hipLaunchKernelGGL(k_cal_gather4, g, b, 0, 0, buf, idx, m, sink);
        hipLaunchKernelGGL(k_cal_write16, g, b, 0, 0, (float4 *)buf, n16);
        CK(hipDeviceSynchronize());
    }
    printf("{\"k_cal_stream16_read_bytes\": %zu, \"k_cal_stream4_read_bytes\": %zu, \"k_cal_gather12_useful_bytes\": %zu, "
           "\"k_cal_gather12_index_bytes\": %zu, \"k_cal_gather12_sector64_bytes\": %zu, \"k_cal_gather12_line128_bytes_upper\": %zu, "
           "\"k_cal_gather4_useful_bytes\": %zu, \"k_cal_gather4_index_bytes\": %zu, \"k_cal_gather4_sector64_bytes\": %zu, "
           "\"k_cal_write16_written_bytes\": %zu}\n",
           bytes, bytes, m * 12, m * 4, sectors * 64, m * 128, m * 4, m * 4, m * 64, bytes);
    return 0;
}
