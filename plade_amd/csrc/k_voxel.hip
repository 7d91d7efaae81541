// plade_amd/csrc/k_voxel.hip -- K9: voxel-grid downsampling and average point spacing.
//
// (1) pcl::VoxelGrid<PointT>::applyFilter as reached through DownSamplePointCloud
//     (code/PLADE/util.h:161-184; pcl-1.8.1/filters/include/pcl/filters/impl/voxel_grid.hpp:214-262,
//     310-345, 416-426; centroid = fp32 sum / fp32 count, accumulators.hpp:65-84).  Output order =
//     ascending voxel index i + j*dx + k*dx*dy, i.e. lexicographic in (k, j, i).  PCL orders the
//     points of one voxel with an UNSTABLE std::sort, so its fp32 centroid sum order is
//     implementation-defined; this kernel sums in ascending input position (the order a stable sort
//     gives) -- the oracle's `sort_mode = 1`; DESIGN.md quantifies the <= few-ulp difference.
//     Several point groups (per-plane clouds, plade.cpp:93-105) are voxelised in one pass by putting
//     the group id in the top bits of the sort key.
// (2) average_spacing (code/PLADE/util.cpp:1619-1648): k = 6 nearest neighbours (FLANN fp32
//     distances) of <= 10000 strided sample points, (sum_{i=1..k-1} sqrt(d_i)) / k averaged in
//     double.  Exact brute-force kNN on the GPU: the k smallest fp32 distances are the same numbers
//     whatever search structure finds them.
#include "voxel.h"
#include "prims.h"

namespace plade {

// ------------------------------------------------------------------------------------------------
__global__ void k_voxel_keys(const float *__restrict__ xyz, uint32_t stride, const uint32_t *__restrict__ item_point,
                             const uint32_t *__restrict__ item_group, uint32_t n_items, float inv, int lminx, int lminy,
                             int lminz, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t p = item_point ? item_point[i] : i;
    const float x = xyz[(size_t)p * stride], y = xyz[(size_t)p * stride + 1], z = xyz[(size_t)p * stride + 2];
    // voxel_grid.hpp:330-332: floor(x * inverse_leaf_size) (fp32), then the integer offset
    const uint64_t lx = (uint64_t)((int)floorf(x * inv) - lminx);
    const uint64_t ly = (uint64_t)((int)floorf(y * inv) - lminy);
    const uint64_t lz = (uint64_t)((int)floorf(z * inv) - lminz);
    const uint64_t g = item_group ? item_group[i] : 0u;
    keys[i] = (g << 54) | (lz << 36) | (ly << 18) | lx;
    vals[i] = i;
}

__global__ void k_head_flags(const uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_heads(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ seg, uint32_t n,
                        uint32_t *__restrict__ heads) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) heads[seg[i]] = i;
}

__global__ void k_voxel_centroids(const float *__restrict__ xyz, uint32_t stride,
                                  const uint32_t *__restrict__ item_point, const uint64_t *__restrict__ keys,
                                  const uint32_t *__restrict__ vals, const uint32_t *__restrict__ heads,
                                  uint32_t n_seg, uint32_t n_items, float *__restrict__ out_xyz,
                                  uint32_t *__restrict__ out_group) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t b = heads[s], e = (s + 1 < n_seg) ? heads[s + 1] : n_items;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (uint32_t j = b; j < e; ++j) {
        const uint32_t it = vals[j];
        const uint32_t p = item_point ? item_point[it] : it;
        ax += xyz[(size_t)p * stride];
        ay += xyz[(size_t)p * stride + 1];
        az += xyz[(size_t)p * stride + 2];
    }
    const float cnt = (float)(e - b);
    out_xyz[3 * (size_t)s] = ax / cnt;
    out_xyz[3 * (size_t)s + 1] = ay / cnt;
    out_xyz[3 * (size_t)s + 2] = az / cnt;
    if (out_group) out_group[s] = (uint32_t)(keys[b] >> 54);
}

__global__ void k_group_offsets(const uint32_t *__restrict__ seg_group, uint32_t n_seg, uint32_t n_groups,
                                uint32_t *__restrict__ offsets /* n_groups + 1, pre-filled with n_seg */) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t g = seg_group[s];
    if (s == 0 || seg_group[s - 1] != g) atomicMin(&offsets[g], s);
}

__global__ void k_fix_offsets(uint32_t *__restrict__ offsets, uint32_t n_groups, uint32_t n_seg) {
    // empty groups inherit the next group's start (single thread; n_groups is tiny)
    if (blockIdx.x || threadIdx.x) return;
    offsets[n_groups] = n_seg;
    for (int g = (int)n_groups - 1; g >= 0; --g)
        if (offsets[g] > offsets[g + 1]) offsets[g] = offsets[g + 1];
}

uint32_t VoxelWork::run(plade_ctx *ctx, const float *d_xyz, uint32_t stride, const uint32_t *d_item_point,
                        const uint32_t *d_item_group, uint32_t n_items, uint32_t n_groups, float leaf,
                        const float bbox_min[3], const float bbox_max[3]) {
    n_out = 0;
    PLADE_REQUIRE(leaf > 0.f, PLADE_EINVAL, "voxel: leaf must be positive");
    PLADE_REQUIRE(n_groups >= 1 && n_groups <= 1024, PLADE_ELIMIT, "voxel: at most 1024 point groups");
    if (n_items == 0) return 0;
    const float inv = 1.f / leaf;
    const int lmin[3] = {(int)floorf(bbox_min[0] * inv), (int)floorf(bbox_min[1] * inv), (int)floorf(bbox_min[2] * inv)};
    const int lmax[3] = {(int)floorf(bbox_max[0] * inv), (int)floorf(bbox_max[1] * inv), (int)floorf(bbox_max[2] * inv)};
    for (int k = 0; k < 3; ++k)
        PLADE_REQUIRE((int64_t)lmax[k] - lmin[k] < (1 << 18), PLADE_ELIMIT,
                      "voxel: more than 2^18 leaves along one axis (PCL would refuse this leaf size too)");
    // voxel_grid.hpp:236-245: dx*dy*dz must fit int32 (per cloud; checked on the union bbox, which is
    // the tighter of the two only for the whole-cloud call -- per-plane clouds are subsets)
    {
        int64_t dx = (int64_t)((bbox_max[0] - bbox_min[0]) * inv) + 1, dy = (int64_t)((bbox_max[1] - bbox_min[1]) * inv) + 1,
                dz = (int64_t)((bbox_max[2] - bbox_min[2]) * inv) + 1;
        PLADE_REQUIRE(dx * dy * dz <= (int64_t)INT32_MAX, PLADE_ELIMIT, "voxel: leaf size too small for the data extent");
    }
    keys.ensure(n_items); keys2.ensure(n_items); vals.ensure(n_items); vals2.ensure(n_items);
    flags.ensure((size_t)n_items + 1); seg.ensure((size_t)n_items + 1);
    const unsigned nb = cdiv(n_items, 256);
    hipLaunchKernelGGL(k_voxel_keys, dim3(nb), dim3(256), 0, ctx->stream, d_xyz, stride, d_item_point, d_item_group,
                       n_items, inv, lmin[0], lmin[1], lmin[2], keys.p, vals.p);
    int gbits = 0;
    while ((1u << gbits) < n_groups) ++gbits;
    sort_pairs_u64(ctx, keys.p, keys2.p, vals.p, vals2.p, n_items, 54 + gbits);
    hipLaunchKernelGGL(k_head_flags, dim3(nb), dim3(256), 0, ctx->stream, keys2.p, n_items, flags.p);
    HIP_TRY(hipMemsetAsync(flags.p + n_items, 0, 4, ctx->stream));
    exclusive_scan_u32(ctx, flags.p, seg.p, (size_t)n_items + 1);
    uint32_t n_seg = 0;
    HIP_TRY(hipMemcpyAsync(&n_seg, seg.p + n_items, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    heads.ensure((size_t)n_seg + 1);
    hipLaunchKernelGGL(k_heads, dim3(nb), dim3(256), 0, ctx->stream, flags.p, seg.p, n_items, heads.p);
    out_xyz.ensure((size_t)n_seg * 3 + 4);
    seg_group.ensure((size_t)n_seg + 1);
    hipLaunchKernelGGL(k_voxel_centroids, dim3(cdiv(n_seg, 128)), dim3(128), 0, ctx->stream, d_xyz, stride, d_item_point,
                       keys2.p, vals2.p, heads.p, n_seg, n_items, out_xyz.p, seg_group.p);
    group_offsets.ensure((size_t)n_groups + 2);
    std::vector<uint32_t> init(n_groups + 1, n_seg);
    HIP_TRY(hipMemcpyAsync(group_offsets.p, init.data(), (n_groups + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_group_offsets, dim3(cdiv(n_seg, 256)), dim3(256), 0, ctx->stream, seg_group.p, n_seg, n_groups,
                       group_offsets.p);
    hipLaunchKernelGGL(k_fix_offsets, dim3(1), dim3(1), 0, ctx->stream, group_offsets.p, n_groups, n_seg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // `init` must outlive the async copy
    n_out = n_seg;
    return n_seg;
}

// ------------------------------------------------------------------------------------------------
// average spacing
constexpr int SP_TPB = 256;
constexpr int SP_QB = 8;    // queries per block
constexpr int SP_K = 8;     // max k supported

__global__ __launch_bounds__(SP_TPB) void k_knn_spacing(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ z, uint32_t n, uint32_t step,
                                                        uint32_t nq, int k, double *__restrict__ avg_out,
                                                        uint32_t *__restrict__ nbs_out) {
    __shared__ float s_q[SP_QB][3];
    const uint32_t q0 = blockIdx.x * SP_QB;
    if (threadIdx.x < SP_QB * 3) {
        uint32_t qi = q0 + threadIdx.x / 3;
        uint32_t pi = (qi < nq) ? qi * step : 0;
        const float *src = (threadIdx.x % 3 == 0) ? x : (threadIdx.x % 3 == 1 ? y : z);
        s_q[threadIdx.x / 3][threadIdx.x % 3] = src[pi];
    }
    __syncthreads();
    f3 q[SP_QB];
    float best[SP_QB][SP_K];
#pragma unroll
    for (int a = 0; a < SP_QB; ++a) {
        q[a] = f3(s_q[a][0], s_q[a][1], s_q[a][2]);
#pragma unroll
        for (int b = 0; b < SP_K; ++b) best[a][b] = INFINITY;
    }
    for (uint32_t i = threadIdx.x; i < n; i += SP_TPB) {
        const f3 p(x[i], y[i], z[i]);
#pragma unroll
        for (int a = 0; a < SP_QB; ++a) {
            float d = flann_d2(q[a], p);
            if (d < best[a][SP_K - 1]) {
                // insertion into the ascending list (only the first k entries are ever read back)
#pragma unroll
                for (int b = 0; b < SP_K; ++b) {
                    if (d < best[a][b]) { float t = best[a][b]; best[a][b] = d; d = t; }
                }
            }
        }
    }
    // merge: each query is reduced by selecting the global minimum k times.  Lists are ascending, so
    // every thread only ever offers its current head.
    __shared__ float s_min[SP_TPB / 64];
    __shared__ int s_arg[SP_TPB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < SP_QB; ++a) {
        const uint32_t qi = q0 + a;
        int head = 0;
        double avg = 0.0;
        int found = 0;
        for (int r = 0; r < k; ++r) {
            float v = INFINITY;
#pragma unroll
            for (int b = 0; b < SP_K; ++b) if (b == head) v = best[a][b];
            if (head >= SP_K) v = INFINITY;
            // wave arg-min
            float mv = v;
            int mi = threadIdx.x;
            for (int d = 32; d >= 1; d >>= 1) {
                float ov = __shfl_xor(mv, d, 64);
                int oi = __shfl_xor(mi, d, 64);
                if (ov < mv || (ov == mv && oi < mi)) { mv = ov; mi = oi; }
            }
            if (lane == 0) { s_min[wave] = mv; s_arg[wave] = mi; }
            __syncthreads();
            float bv = s_min[0];
            int bi = s_arg[0];
            for (int w = 1; w < SP_TPB / 64; ++w)
                if (s_min[w] < bv || (s_min[w] == bv && s_arg[w] < bi)) { bv = s_min[w]; bi = s_arg[w]; }
            __syncthreads();
            if (bv == INFINITY) break;
            if ((int)threadIdx.x == bi) ++head;
            ++found;
            if (r >= 1) avg += (double)sqrtf(bv);  // util.cpp:1640-1642: starts from 1 to exclude itself
        }
        if (threadIdx.x == 0 && qi < nq) {
            avg_out[qi] = avg;
            nbs_out[qi] = (uint32_t)found;
        }
    }
}

float average_spacing_dev(plade_ctx *ctx, const float *d_x, const float *d_y, const float *d_z, uint32_t n, int k,
                          uint32_t samples) {
    PLADE_REQUIRE(k >= 1 && k <= SP_K, PLADE_ELIMIT, "average_spacing: k must be in [1, 8]");
    if (n == 0) return 0.f;
    size_t step = 1;
    if (n > samples) step = n / samples;
    const uint32_t nq = (uint32_t)((n + step - 1) / step);
    double *d_avg = reinterpret_cast<double *>(ctx->scratch[1].ensure((size_t)nq * 8 + 8));
    uint32_t *d_nbs = reinterpret_cast<uint32_t *>(ctx->scratch[2].ensure((size_t)nq * 4 + 8));
    hipLaunchKernelGGL(k_knn_spacing, dim3(cdiv(nq, SP_QB)), dim3(SP_TPB), 0, ctx->stream, d_x, d_y, d_z, n,
                       (uint32_t)step, nq, k, d_avg, d_nbs);
    HIP_TRY(hipGetLastError());
    std::vector<double> avg(nq);
    std::vector<uint32_t> nbs(nq);
    HIP_TRY(hipMemcpyAsync(avg.data(), d_avg, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(nbs.data(), d_nbs, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // util.cpp:1630-1647: sequential double accumulation in sample order
    double total = 0.0;
    size_t total_count = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        int nb = (int)nbs[i];
        if (nb <= 1) continue;
        total += (avg[i] / nb);
        ++total_count;
    }
    return static_cast<float>(total / total_count);
}

__global__ void k_strided_to_soa(const float *__restrict__ in, uint32_t n, uint32_t stride, float *__restrict__ x,
                                 float *__restrict__ y, float *__restrict__ z) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = in[(size_t)i * stride]; y[i] = in[(size_t)i * stride + 1]; z[i] = in[(size_t)i * stride + 2];
}


__global__ void k_minmax3_v(const float *__restrict__ xyz, uint32_t n, uint32_t stride, int *__restrict__ out6) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 3; ++k) {
            float v = xyz[(size_t)i * stride + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    for (int k = 0; k < 3; ++k) {
        for (int d = 32; d >= 1; d >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], d, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            int a = __float_as_int(mn[k]), b = __float_as_int(mx[k]);
            a = a >= 0 ? a : a ^ 0x7fffffff;
            b = b >= 0 ? b : b ^ 0x7fffffff;
            atomicMin(&out6[k], a);
            atomicMax(&out6[3 + k], b);
        }
    }
}

void bbox_host(plade_ctx *ctx, const float *d_xyz, uint32_t n, uint32_t stride, float mn[3], float mx[3]) {
    int init[6];
    float pinf = INFINITY, ninf = -INFINITY;
    int a, b;
    memcpy(&a, &pinf, 4); memcpy(&b, &ninf, 4);
    for (int k = 0; k < 3; ++k) { init[k] = a; init[3 + k] = b ^ 0x7fffffff; }
    int *d = reinterpret_cast<int *>(ctx->scratch[3].ensure(64));
    HIP_TRY(hipMemcpyAsync(d, init, 24, hipMemcpyHostToDevice, ctx->stream));
    if (n) hipLaunchKernelGGL(k_minmax3_v, dim3(std::min(cdiv(n, 256), 2048u)), dim3(256), 0, ctx->stream, d_xyz, n, stride, d);
    int out[6];
    HIP_TRY(hipMemcpyAsync(out, d, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 6; ++k) {
        int v = out[k] >= 0 ? out[k] : out[k] ^ 0x7fffffff;
        float f;
        memcpy(&f, &v, 4);
        if (k < 3) mn[k] = f; else mx[k - 3] = f;
    }
}
}  // namespace plade

using namespace plade;

// ---- C ABI ---------------------------------------------------------------------------------
extern "C" int plade_average_spacing(plade_ctx *ctx, const float *xyz, uint32_t n, uint32_t stride, uint32_t k,
                                     uint32_t samples, float *spacing_out) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(xyz && spacing_out && stride >= 3 && samples >= 1, PLADE_EINVAL, "plade_average_spacing: bad argument");
        DBuf<float> d_in, d_soa;
        d_in.ensure((size_t)n * stride + 4);
        d_soa.ensure((size_t)n * 3 + 4);
        HIP_TRY(hipMemcpyAsync(d_in.p, xyz, (size_t)n * stride * 4, hipMemcpyHostToDevice, ctx->stream));
        if (n) hipLaunchKernelGGL(k_strided_to_soa, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_in.p, n, stride, d_soa.p,
                                  d_soa.p + n, d_soa.p + 2 * (size_t)n);
        *spacing_out = average_spacing_dev(ctx, d_soa.p, d_soa.p + n, d_soa.p + 2 * (size_t)n, n, (int)k, samples);
        return PLADE_OK;
    });
}

extern "C" int plade_voxel_downsample(plade_ctx *ctx, const float *xyz, uint32_t n, uint32_t stride, float leaf,
                                      float *out_xyz, uint32_t *n_out) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(xyz && out_xyz && n_out && stride >= 3, PLADE_EINVAL, "plade_voxel_downsample: bad argument");
        *n_out = 0;
        if (n == 0) return PLADE_OK;
        DBuf<float> d_in;
        d_in.ensure((size_t)n * stride + 4);
        HIP_TRY(hipMemcpyAsync(d_in.p, xyz, (size_t)n * stride * 4, hipMemcpyHostToDevice, ctx->stream));
        float mn[3], mx[3];
        bbox_host(ctx, d_in.p, n, stride, mn, mx);
        VoxelWork w;
        uint32_t m = w.run(ctx, d_in.p, stride, nullptr, nullptr, n, 1, leaf, mn, mx);
        HIP_TRY(hipMemcpyAsync(out_xyz, w.out_xyz.p, (size_t)m * 12, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        *n_out = m;
        return PLADE_OK;
    });
}
