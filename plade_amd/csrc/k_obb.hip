// plade_amd/csrc/k_obb.hip -- PCA oriented bounding boxes on the GPU.
//
// ComputeBoundingBox (code/PLADE/util.h:186-248) of the voxel-downsampled cloud (plade.cpp:81-84 / :295-299) and of every
// per-plane downsampled cloud, followed per plane by the projection of the first four box corners onto the plane, their
// centre and half diagonal (plade.cpp:106-117 / :320-330, ProjectPoints2Plane util.h:292-340).  One workgroup per
// "unit" (unit 0 = the whole cloud, unit 1 + i = plane i); nothing but a few dozen floats per unit goes back to the host.
//
// PCL adds the points one after the other in fp32 (compute3DCentroid / computeCovarianceMatrixNormalized,
// centroid.hpp:79-121, 250-300): a 150 000-step dependent chain a GPU lane would spend a millisecond on.  Here lane t of
// the unit's 1024-lane workgroup adds the points t, t + 1024, t + 2048, ... one after the other (coalesced reads: at every
// step a wavefront touches 64 consecutive points; contiguous 64-point chunks per lane were measured at 220 us per
// pass for 147 000 points, every lane on its own cache lines), and one lane adds the 1024 lane sums in lane order: the
// same additions re-associated, deterministic, and exactly what the CPU checker's sum mode 1 does (see tests).
// Everything after the sums -- Eigen's 3x3 self-adjoint solver, the frame, min / max, corners -- is PCL's arithmetic
// operation for operation (hostgeom.h).
#include "stages.h"
#include "hostgeom.h"

namespace plade {

namespace {

struct ObbArgs {
    const float *ds; const uint32_t *n_ds_p;   // whole downsampled cloud, n x 3 (its size is still on the device)
    const float *plane_ds; const uint32_t *plane_off; uint32_t P;   // per-plane clouds, concatenated, P + 1 offsets
    const float *coef;                         // P x 4 plane coefficients
    float *out;                                // OBB_OUT_WHOLE + P * OBB_OUT_PLANE floats
};

constexpr int OBB_T = OBB_LANES;

// sums of the K-component per-point terms over the points of one unit in the lane-strided order: lane t adds the terms
// of points t, t + 1024, t + 2048, ... one after the other (sixteen points are fetched ahead of the additions; at every
// step the lanes of a wavefront read 64 consecutive points), the lane sums go to LDS, lane q < K adds the 1024 lane sums
// of component q in lane order
template <int K, class Term>
__device__ void strided_sums(const float *__restrict__ pts, uint32_t n, float (*s_ch)[6], float *s_out, Term term) {
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.f;
    constexpr int AHEAD = 24;   // points in flight per lane (what 128 registers hold without spilling): a pass over a 130 000-point unit is 6 load latencies, not 130
    for (uint32_t i0 = threadIdx.x; i0 < n; i0 += AHEAD * OBB_T) {
        float x[AHEAD], y[AHEAD], z[AHEAD];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) {
            const uint32_t i = min(i0 + j * OBB_T, n - 1);
            const float3 p3 = *reinterpret_cast<const float3 *>(pts + 3 * (size_t)i);   // one 12-byte load
            x[j] = p3.x; y[j] = p3.y; z[j] = p3.z;
        }
#pragma unroll
        for (int j = 0; j < AHEAD; ++j)
            if (i0 + j * OBB_T < n) term(acc, f3(x[j], y[j], z[j]));
    }
#pragma unroll
    for (int q = 0; q < K; ++q) s_ch[threadIdx.x][q] = acc[q];
    __syncthreads();
    if (threadIdx.x < K) {   // lane sums in lane order; eight LDS reads ahead of the additions
        float total = 0.f;
        for (int t = 0; t < OBB_T; t += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s_ch[t + j][threadIdx.x];
#pragma unroll
            for (int j = 0; j < 8; ++j) total += v[j];
        }
        s_out[threadIdx.x] = total;
    }
    __syncthreads();
}

__device__ __forceinline__ void obb_unit(const ObbArgs &A, const uint32_t u) {
    __shared__ float s_ch[OBB_T][6];
    __shared__ float s_c[3], s_cov[6], s_P[12], s_E[9];
    __shared__ float s_mm[6][OBB_T / 64];
    const float *pts;
    uint32_t n;
    if (u == 0) { pts = A.ds; n = *A.n_ds_p; }
    else {
        const uint32_t b = A.plane_off[u - 1], e = A.plane_off[u];
        pts = A.plane_ds + 3 * (size_t)b;
        n = e - b;
    }
    float *out = u == 0 ? A.out : A.out + OBB_OUT_WHOLE + (size_t)(u - 1) * OBB_OUT_PLANE;
    if (n == 0) {   // empty plane: its boxes stay zero (plade.cpp:106-117 skips it)
        if (u > 0) for (uint32_t i = threadIdx.x; i < OBB_OUT_PLANE; i += blockDim.x) out[i] = 0.f;
        else if (threadIdx.x == 0) out[OBB_OUT_WHOLE - 1] = 0.f;
        return;
    }
    const float nf = (float)n;
    // ---- centroid (compute3DCentroid, centroid.hpp:79-121)
    strided_sums<3>(pts, n, s_ch, s_c, [](float *acc, f3 p) { acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; });
    const float c[3] = {s_c[0] / nf, s_c[1] / nf, s_c[2] / nf};
    // ---- covariance about the centroid (computeCovarianceMatrixNormalized, centroid.hpp:250-300), same chunking
    strided_sums<6>(pts, n, s_ch, s_cov, [c](float *acc, f3 p) { cov_add_point(acc, p, c); });
    if (threadIdx.x == 0) {
        const float cov6[6] = {s_cov[0], s_cov[1], s_cov[2], s_cov[3], s_cov[4], s_cov[5]};
        m3 E;
        float P[12];
        obb_frame(cov6, nf, c, E, P);
        for (int q = 0; q < 12; ++q) s_P[q] = P[q];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) s_E[3 * r + k] = E.m[r][k];
    }
    __syncthreads();
    // ---- min / max in the eigen frame (order-free: exact)
    float P[12];
    for (int q = 0; q < 12; ++q) P[q] = s_P[q];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    constexpr int MM_AHEAD = 24;
    for (uint32_t i0 = threadIdx.x; i0 < n; i0 += MM_AHEAD * OBB_T) {   // twenty-four points in flight per lane
        float x[MM_AHEAD], y[MM_AHEAD], z[MM_AHEAD];
#pragma unroll
        for (int j = 0; j < MM_AHEAD; ++j) {
            const uint32_t i = min(i0 + j * OBB_T, n - 1);
            const float3 p3 = *reinterpret_cast<const float3 *>(pts + 3 * (size_t)i);   // one 12-byte load
            x[j] = p3.x; y[j] = p3.y; z[j] = p3.z;
        }
#pragma unroll
        for (int j = 0; j < MM_AHEAD; ++j) {   // a clamped repeat of the last point changes no minimum / maximum
            const f3 q = pcl_xform(P, f3(x[j], y[j], z[j]));
            mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
            mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
        }
    }
    for (int q = 0; q < 3; ++q)
        for (int d = 32; d >= 1; d >>= 1) {
            mn[q] = fminf(mn[q], __shfl_xor(mn[q], d, 64));
            mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], d, 64));
        }
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 3; ++q) { s_mm[q][threadIdx.x >> 6] = mn[q]; s_mm[3 + q][threadIdx.x >> 6] = mx[q]; }
    __syncthreads();
    if (threadIdx.x) return;
    for (int q = 0; q < 3; ++q)
        for (int w = 0; w < OBB_T / 64; ++w) { mn[q] = fminf(mn[q], s_mm[q][w]); mx[q] = fmaxf(mx[q], s_mm[3 + q][w]); }
    m3 E;
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) E.m[r][k] = s_E[3 * r + k];
    f3 center, corners[8];
    double whd[3];
    obb_finish(E, c, f3(mn[0], mn[1], mn[2]), f3(mx[0], mx[1], mx[2]), center, whd, corners);
    if (u == 0) {
        // plade.cpp:83-84: centre of the box, radius = max(width, height, depth) / 2 (double)
        out[0] = center.x; out[1] = center.y; out[2] = center.z;
        const double r = fmax(fmax(whd[0], whd[1]), whd[2]) / 2;
        memcpy(out + 4, &r, 8);
        out[OBB_OUT_WHOLE - 1] = 1.f;
        return;
    }
    const float *pl = A.coef + 4 * (size_t)(u - 1);
    f3 four[4];
    for (int k = 0; k < 4; ++k) {
        four[k] = project_to_plane(corners[k], pl);
        out[3 * k] = four[k].x; out[3 * k + 1] = four[k].y; out[3 * k + 2] = four[k].z;
    }
    const f3 cen = (four[0] + four[2]) / 2.f;
    out[12] = cen.x; out[13] = cen.y; out[14] = cen.z;
    out[15] = norm_e(four[0] - four[2]) / 2.f;
}

__global__ __launch_bounds__(OBB_T) void k_obb_units(const ObbArgs A) { obb_unit(A, blockIdx.x); }

// the units of up to 16 clouds in one launch: these workgroups are latency chains (three dependent passes over their points
// by ONE workgroup each), so sixteen clouds take as long as one -- and a millisecond of such queue time per registration costs
// the batch 14 % (profiles/r4_experiments.md 2c)
constexpr int OBB_BATCH = 16;
struct ObbBatchArgs { ObbArgs c[OBB_BATCH]; uint32_t ncl; uint32_t unit_start[OBB_BATCH + 1]; };
__global__ __launch_bounds__(OBB_T) void k_obb_units_batch(const ObbBatchArgs B) {
    uint32_t g = 0;
#pragma unroll
    for (int q = 1; q < OBB_BATCH; ++q) g += (q < (int)B.ncl && blockIdx.x >= B.unit_start[q]) ? 1u : 0u;
    obb_unit(B.c[g], blockIdx.x - B.unit_start[g]);
}

}  // namespace

bool obb_units_batch(plade_ctx *ctx, int count, const ObbBatchItem *items, DBuf<float> &coef_scratch) {
    if (count < 1 || count > OBB_BATCH) return false;
    ObbBatchArgs B;
    memset(&B, 0, sizeof(B));
    B.ncl = (uint32_t)count;
    size_t n_coef = 0;
    for (int g = 0; g < count; ++g) n_coef += 4 * (size_t)items[g].P;
    coef_scratch.ensure(n_coef + 4);
    std::vector<float> h(n_coef);
    size_t o = 0;
    for (int g = 0; g < count; ++g) {
        const ObbBatchItem &it = items[g];
        ObbWork &W = *it.work;
        W.out.ensure(OBB_OUT_WHOLE + (size_t)it.P * OBB_OUT_PLANE + 4);
        memcpy(h.data() + o, it.coef_host, 16 * (size_t)it.P);
        B.c[g] = ObbArgs{it.d_ds, it.d_n_ds, it.d_plane_ds, it.d_plane_off, it.P, coef_scratch.p + o, W.out.p};
        o += 4 * (size_t)it.P;
        B.unit_start[g + 1] = B.unit_start[g] + it.P + 1;
    }
    for (int g = count; g < OBB_BATCH; ++g) { B.c[g] = B.c[0]; B.unit_start[g + 1] = B.unit_start[g]; }
    if (n_coef) { const bool staged = ctx->h2d(coef_scratch.p, h.data(), 4 * n_coef); if (!staged) ctx->sync(); }
    launch_raw(ctx, k_obb_units_batch, dim3(B.unit_start[count]), dim3(OBB_T), 0, B);
    HIP_TRY(hipGetLastError());
    return true;
}

void obb_adopt_batch(plade_ctx *ctx, ObbWork &W, uint32_t P) {
    W.host.resize(OBB_OUT_WHOLE + (size_t)P * OBB_OUT_PLANE);
    ctx->d2h(W.host.data(), W.out.p, 4 * W.host.size());   // valid after the next sync of the stream
}

void obb_units(plade_ctx *ctx, ObbWork &W, const float *d_ds, const uint32_t *d_n_ds, uint32_t max_ds, const float *d_plane_ds,
               const uint32_t *d_plane_off, uint32_t max_plane_pts, uint32_t P, const float *coef_host, const float *d_coef) {
    if (!d_coef) {   // the caller has not uploaded the coefficients with something else
        W.d_coef.ensure(4 * (size_t)P + 4);
        if (P) ctx->h2d(W.d_coef.p, coef_host, 16 * (size_t)P);
        d_coef = W.d_coef.p;
    }
    (void)max_ds; (void)max_plane_pts;
    W.out.ensure(OBB_OUT_WHOLE + (size_t)P * OBB_OUT_PLANE + 4);
    W.host.resize(OBB_OUT_WHOLE + (size_t)P * OBB_OUT_PLANE);
    ObbArgs A{d_ds, d_n_ds, d_plane_ds, d_plane_off, P, d_coef, W.out.p};
    launch_raw(ctx, k_obb_units, dim3(P + 1), dim3(OBB_T), 0, A);
    HIP_TRY(hipGetLastError());
    ctx->d2h(W.host.data(), W.out.p, 4 * W.host.size());   // valid after the next sync of the stream
}

}  // namespace plade
