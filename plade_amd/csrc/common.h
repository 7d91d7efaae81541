// plade_amd/csrc/common.h -- shared host/device helpers of libplade_hip.so (gfx950 only).
//
// fp32 expressions follow the evaluation order of the reference's Eigen 3.4 / PCL /
// Schnabel code so that integer decisions taken on the GPU agree with the CPU path
// bit for bit (the library is compiled with -ffp-contract=off; + - * / sqrt are
// correctly rounded on CDNA4).  Conventions:
//   fixed-size Eigen 3-vector reductions:  x0 + (x1 + x2)
//   dynamic-size Eigen reductions and Schnabel's Vec3f::dot:  (x0 + x1) + x2
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cfloat>
#include <string>
#include <vector>
#include <map>
#include <chrono>
#include <atomic>

#define HD __host__ __device__ __forceinline__

namespace plade {

struct f3 {
    float x, y, z;
    HD f3() : x(0.f), y(0.f), z(0.f) {}
    HD f3(float a, float b, float c) : x(a), y(b), z(c) {}
};
HD f3 operator+(f3 a, f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
HD f3 operator-(f3 a, f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
HD f3 operator-(f3 a) { return f3(-a.x, -a.y, -a.z); }
HD f3 operator*(float s, f3 a) { return f3(s * a.x, s * a.y, s * a.z); }
HD f3 operator/(f3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
HD float dot_e(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }     // Eigen fixed-size
HD float dot_s(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }     // sequential
HD float sqn_e(f3 a) { return dot_e(a, a); }
HD float norm_e(f3 a) { return sqrtf(sqn_e(a)); }
HD f3 cross(f3 a, f3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HD f3 normalized_e(f3 a) {  // Eigen MatrixBase::normalize()
    float z = sqn_e(a);
    if (z > 0.f) { float s = sqrtf(z); return a / s; }
    return a;
}

struct m3 { float m[3][3]; };
HD f3 mul_e(const m3 &R, f3 v) {
    return f3(R.m[0][0] * v.x + (R.m[0][1] * v.y + R.m[0][2] * v.z),
              R.m[1][0] * v.x + (R.m[1][1] * v.y + R.m[1][2] * v.z),
              R.m[2][0] * v.x + (R.m[2][1] * v.y + R.m[2][2] * v.z));
}
// pcl::transformPointCloud dense branch (pcl-1.8.1/common/include/pcl/common/impl/transforms.hpp:69-71)
HD f3 pcl_xform(const float *T, f3 p) {
    return f3(T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3], T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7],
              T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11]);
}
// FLANN L2_Simple<float> (flann/algorithms/dist.h:74-98)
HD float flann_d2(f3 q, f3 p) {
    float ax = q.x - p.x, ay = q.y - p.y, az = q.z - p.z;
    float r = ax * ax;
    r += ay * ay;
    r += az * az;
    return r;
}
// pcl::KdTreeFLANN::radiusSearch squared radius (kdtree_flann.hpp:193)
inline float pcl_r2(double radius) { return static_cast<float>(radius * radius); }

// ---------------------------------------------------------------------------
struct Err {
    int code;
    std::string msg;
};

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            throw plade::Err{-2, std::string(#expr) + ": " + hipGetErrorString(_e)};           \
        }                                                                                      \
    } while (0)

#define PLADE_REQUIRE(cond, code, text)                        \
    do {                                                       \
        if (!(cond)) throw plade::Err{(code), std::string(text)}; \
    } while (0)

// allocation statistics of this process (PLADE_DEBUG_ALLOC=1 prints them when a context is destroyed)
struct AllocStats { std::atomic<uint64_t> calls{0}, bytes{0}, nanos{0}; };
inline AllocStats &alloc_stats() { static AllocStats s; return s; }

// Device memory of the work areas.  A registration context owns ~700 buffers, most of them a few KB to a few hundred KB, and
// every one used to be a hipMalloc of its own (1 400 calls and 150 ms for the CLI's 64-pair list, 90 us each): blocks of up to
// 1 MB now come out of 32 MB slabs (api.hip: bump allocation, 256-byte granules, one lock; a slab whose blocks have all been
// returned is recycled behind a device synchronisation, as hipFree would have waited); larger blocks are hipMalloc / hipFree.
void *dev_alloc(size_t bytes);
void dev_free(void *p);

// While the pairs of a group run in lock step (launch.h) what a pair launches is queued, not issued: an allocation that is
// replaced by a larger one meanwhile may still be named by queued launches, so it is kept until the pair's next wait has
// returned (hipFree on a stream-ordered path waits for the device; here the work has not even been queued).
inline thread_local std::vector<void *> *tl_deferred_free = nullptr;

// grow-only device buffer
template <class T>
struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
    ~DBuf() { if (p) dev_free(p); }
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    T *ensure(size_t n) {
        if (n > cap) {
            if (p) { if (tl_deferred_free) tl_deferred_free->push_back(p); else dev_free(p); }
            p = nullptr;
            size_t want = n + n / 4 + 64;
            p = static_cast<T *>(dev_alloc(want * sizeof(T)));
            cap = want;
        }
        return p;
    }
    operator T *() const { return p; }
    // exchange the allocations of two buffers (hand a result over without a device copy)
    void swap(DBuf &o) { T *tp = p; p = o.p; o.p = tp; size_t tc = cap; cap = o.cap; o.cap = tc; }
};

// pinned host buffer
template <class T>
struct HBuf {
    T *p = nullptr;
    size_t cap = 0;
    ~HBuf() { if (p) (void)hipHostFree(p); }
    HBuf() = default;
    HBuf(const HBuf &) = delete;
    HBuf &operator=(const HBuf &) = delete;
    T *ensure(size_t n, unsigned flags = hipHostMallocDefault) {
        if (n > cap) {
            if (p) HIP_TRY(hipHostFree(p));
            p = nullptr;
            size_t want = n + n / 4 + 64;
            HIP_TRY(hipHostMalloc((void **)&p, want * sizeof(T), flags));
            cap = want;
        }
        return p;
    }
    T &operator[](size_t i) { return p[i]; }
    operator T *() const { return p; }
};

#ifdef __HIPCC__
// order-preserving float <-> int mapping for atomicMin/atomicMax on floats
__device__ __forceinline__ int ordered_int(float f) { int v = __float_as_int(f); return v >= 0 ? v : v ^ 0x7fffffff; }
__device__ __forceinline__ float ordered_float(int v) { return __int_as_float(v >= 0 ? v : v ^ 0x7fffffff); }
// Commit per-lane minima/maxima (K of each) to global ordered-int slots out[0..K) (min) and
// out[K..2K) (max): wave shuffle reduce -> LDS -> one lane per slot, and the atomic is only issued
// when it would change the stored value (after the first few blocks almost never).
template <int K>
__device__ __forceinline__ void block_minmax_commit(float (&mn)[K], float (&mx)[K], int *out, float (*s_lds)[8]) {
    for (int k = 0; k < K; ++k)
        for (int d = 32; d >= 1; d >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], d, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) for (int k = 0; k < K; ++k) { s_lds[k][wave] = mn[k]; s_lds[K + k][wave] = mx[k]; }
    __syncthreads();
    if ((int)threadIdx.x < 2 * K) {
        const int k = threadIdx.x;
        float v = s_lds[k][0];
        for (int w = 1; w < nw; ++w) v = k < K ? fminf(v, s_lds[k][w]) : fmaxf(v, s_lds[k][w]);
        const int iv = ordered_int(v);
        if (k < K) { if (iv < out[k]) atomicMin(&out[k], iv); }
        else { if (iv > out[k]) atomicMax(&out[k], iv); }
    }
}
#endif

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

using Clock = std::chrono::steady_clock;
inline double secs_since(Clock::time_point t0) {
    return std::chrono::duration<double>(Clock::now() - t0).count();
}

}  // namespace plade
