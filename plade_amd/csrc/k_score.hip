// plade_amd/csrc/k_score.hip -- K1: per-hypothesis point-to-plane inlier counting and ordered
// inlier compaction for gfx950 (wave64).
//
// Reference semantics: FlatNormalThreshPointCompatibilityFunc::operator()
// (code/3rd_party/ransac/FlatNormalThreshPointCompatibilityFunc.h:14-23) with
// Plane::Distance = fabs(m_dist - n.p) (ransac/Plane.h:31) and Vec3f::dot accumulated left to
// right (ransac/basic.h:80-86); visitor filter shapeIndex[i] == -1
// (ransac/ScorePrimitiveShapeVisitor.h:39-46).  Output order = ascending point index, which is the
// order the reference's octree visitor produces.
//
// Layout: SoA planes x|y|z|nx|ny|nz, 4 points per lane via 16-byte loads (fully coalesced:
// 1 KiB per wave instruction); plane coefficients staged in LDS; inlier counts reduced with
// __ballot + s_bcnt (one LDS write per wave per hypothesis, one global atomic per block per
// hypothesis).  HBM-bound: 28 B per point per pass (12 pos + 12 normal + 4 shapeIndex).
#include "score.h"
#include "voxel.h"

namespace plade {

constexpr int TPB = 256;
constexpr int PPT = 4;
constexpr int TILE = TPB * PPT;  // 1024 points per block
constexpr int HCHUNK = 64;   // >= the RANSAC candidate pool (48): a re-score pass reads the cloud once

__device__ __forceinline__ bool compatible(float4 pl, float px, float py, float pz, float qx, float qy, float qz,
                                           float eps, float cos_t) {
    float d = pl.x * px;
    d += pl.y * py;
    d += pl.z * pz;
    float dist = fabsf(pl.w - d);
    float nd = pl.x * qx;
    nd += pl.y * qy;
    nd += pl.z * qz;
    return (dist < eps) && (fabsf(nd) >= cos_t);
}

struct Tile {
    float px[PPT], py[PPT], pz[PPT], qx[PPT], qy[PPT], qz[PPT];
    bool valid[PPT];
};

__device__ __forceinline__ void load_tile(Tile &t, const float *x, const float *y, const float *z, const float *nx,
                                          const float *ny, const float *nz, const int32_t *assigned,
                                          const uint32_t *sub_index, uint32_t n, uint32_t base) {
    if (base + PPT <= n) {
        float4 a = *reinterpret_cast<const float4 *>(x + base);
        float4 b = *reinterpret_cast<const float4 *>(y + base);
        float4 c = *reinterpret_cast<const float4 *>(z + base);
        float4 d = *reinterpret_cast<const float4 *>(nx + base);
        float4 e = *reinterpret_cast<const float4 *>(ny + base);
        float4 f = *reinterpret_cast<const float4 *>(nz + base);
        t.px[0] = a.x; t.px[1] = a.y; t.px[2] = a.z; t.px[3] = a.w;
        t.py[0] = b.x; t.py[1] = b.y; t.py[2] = b.z; t.py[3] = b.w;
        t.pz[0] = c.x; t.pz[1] = c.y; t.pz[2] = c.z; t.pz[3] = c.w;
        t.qx[0] = d.x; t.qx[1] = d.y; t.qx[2] = d.z; t.qx[3] = d.w;
        t.qy[0] = e.x; t.qy[1] = e.y; t.qy[2] = e.z; t.qy[3] = e.w;
        t.qz[0] = f.x; t.qz[1] = f.y; t.qz[2] = f.z; t.qz[3] = f.w;
        if (assigned && !sub_index) {
            int4 s = *reinterpret_cast<const int4 *>(assigned + base);
            t.valid[0] = s.x == -1; t.valid[1] = s.y == -1; t.valid[2] = s.z == -1; t.valid[3] = s.w == -1;
        } else if (assigned) {
            uint4 si = *reinterpret_cast<const uint4 *>(sub_index + base);
            t.valid[0] = assigned[si.x] == -1; t.valid[1] = assigned[si.y] == -1;
            t.valid[2] = assigned[si.z] == -1; t.valid[3] = assigned[si.w] == -1;
        } else {
            t.valid[0] = t.valid[1] = t.valid[2] = t.valid[3] = true;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            uint32_t i = base + k;
            bool in = i < n;
            t.px[k] = in ? x[i] : 0.f; t.py[k] = in ? y[i] : 0.f; t.pz[k] = in ? z[i] : 0.f;
            t.qx[k] = in ? nx[i] : 0.f; t.qy[k] = in ? ny[i] : 0.f; t.qz[k] = in ? nz[i] : 0.f;
            bool un = true;
            if (in && assigned) un = (sub_index ? assigned[sub_index[i]] : assigned[i]) == -1;
            t.valid[k] = in && un;
        }
    }
}

// grid: (tiles, hypothesis chunks).  SUB: the cloud is the gathered RANSAC subset (shapeIndex looked up
// through sub_index); a separate symbol so that profiles keep the two call shapes apart.
template <bool SUB>
__global__ __launch_bounds__(TPB) void k_score_multi(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, const float *__restrict__ nx,
                                                     const float *__restrict__ ny, const float *__restrict__ nz,
                                                     const int32_t *__restrict__ assigned,
                                                     const uint32_t *__restrict__ sub_index, uint32_t n,
                                                     const float4 *__restrict__ planes, uint32_t h, float eps,
                                                     float cos_t, uint32_t *__restrict__ counts) {
    __shared__ float4 s_pl[HCHUNK];
    __shared__ uint32_t s_cnt[TPB / 64][HCHUNK];
    const uint32_t h0 = blockIdx.y * HCHUNK;
    const uint32_t hc = min((uint32_t)HCHUNK, h - h0);
    if (threadIdx.x < hc) s_pl[threadIdx.x] = planes[h0 + threadIdx.x];
    Tile t;
    load_tile(t, x, y, z, nx, ny, nz, assigned, SUB ? sub_index : nullptr, n, blockIdx.x * TILE + threadIdx.x * PPT);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t hh = 0; hh < hc; ++hh) {
        const float4 pl = s_pl[hh];
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            bool in = t.valid[k] && compatible(pl, t.px[k], t.py[k], t.pz[k], t.qx[k], t.qy[k], t.qz[k], eps, cos_t);
            c += (uint32_t)__popcll(__ballot(in));
        }
        if (lane == 0) s_cnt[wave][hh] = c;
    }
    __syncthreads();
    if (threadIdx.x < hc) {
        uint32_t tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        if (tot) atomicAdd(&counts[h0 + threadIdx.x], tot);
    }
}

// single hypothesis (read from device memory): 4-bit inlier mask per lane + per-block count
__global__ __launch_bounds__(TPB) void k_score_mark(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, const float *__restrict__ nx,
                                                    const float *__restrict__ ny, const float *__restrict__ nz,
                                                    const int32_t *__restrict__ assigned, uint32_t n,
                                                    const float4 *__restrict__ plane, float eps, float cos_t,
                                                    uint8_t *__restrict__ masks, uint32_t *__restrict__ block_counts,
                                                    const uint32_t *__restrict__ skip) {
    __shared__ uint32_t s_w[TPB / 64];
    if (skip && *skip) return;
    const float4 pl = plane[0];
    Tile t;
    load_tile(t, x, y, z, nx, ny, nz, assigned, nullptr, n, blockIdx.x * TILE + threadIdx.x * PPT);
    uint32_t m = 0, c = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        bool in = t.valid[k] && compatible(pl, t.px[k], t.py[k], t.pz[k], t.qx[k], t.qy[k], t.qz[k], eps, cos_t);
        m |= (in ? 1u : 0u) << k;
        c += (uint32_t)__popcll(__ballot(in));
    }
    masks[blockIdx.x * TPB + threadIdx.x] = (uint8_t)m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// ordered compaction; every block derives its output offset from the preceding blocks' counts itself
// (nb is ~1e3, the count array is L2 resident), which saves a dependent launch per compaction
__global__ __launch_bounds__(TPB) void k_compact(const uint8_t *__restrict__ masks,
                                                 const uint32_t *__restrict__ block_counts, uint32_t nb,
                                                 const uint32_t *__restrict__ values, uint32_t *__restrict__ out,
                                                 uint32_t *__restrict__ total, const uint32_t *__restrict__ skip) {
    __shared__ uint32_t s_w[TPB / 64];
    __shared__ uint32_t s_base[TPB / 64];
    if (skip && *skip) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t pre = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += TPB) pre += block_counts[b];
    for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d, 64);
    if (lane == 0) s_base[wave] = pre;
    const uint32_t m = masks[blockIdx.x * TPB + threadIdx.x];
    const uint32_t c = __popc(m);
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    const uint32_t base = s_base[0] + s_base[1] + s_base[2] + s_base[3];
    uint32_t off = base + incl - c;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    if (blockIdx.x == nb - 1 && threadIdx.x == TPB - 1) *total = off + c;
    const uint32_t first = blockIdx.x * TILE + threadIdx.x * PPT;
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (m & (1u << k)) {
            uint32_t i = first + k;
            out[off++] = values ? values[i] : i;
        }
}

void score_multi(plade_ctx *ctx, const float *x, const float *y, const float *z, const float *nx, const float *ny,
                 const float *nz, const int32_t *assigned, const uint32_t *sub_index, uint32_t n,
                 const float4 *planes_dev, uint32_t h, float eps, float cos_thresh, uint32_t *counts_dev,
                 bool counts_are_zero) {
    if (!counts_are_zero) HIP_TRY(hipMemsetAsync(counts_dev, 0, sizeof(uint32_t) * h, ctx->stream));
    if (n == 0 || h == 0) return;
    dim3 grid(cdiv(n, TILE), cdiv(h, HCHUNK));
    // algorithmic bytes: 12 pos + 12 normal + 4 shapeIndex per point per pass
    ctx->ev_begin(sub_index ? "score_subset" : "score_multi", 28.0 * n);
    if (sub_index)
        hipLaunchKernelGGL(k_score_multi<true>, grid, dim3(TPB), 0, ctx->stream, x, y, z, nx, ny, nz, assigned, sub_index, n,
                           planes_dev, h, eps, cos_thresh, counts_dev);
    else
        hipLaunchKernelGGL(k_score_multi<false>, grid, dim3(TPB), 0, ctx->stream, x, y, z, nx, ny, nz, assigned, sub_index, n,
                           planes_dev, h, eps, cos_thresh, counts_dev);
    ctx->ev_end();
    HIP_TRY(hipGetLastError());
}

void compact_masks(plade_ctx *ctx, CompactScratch &s, uint32_t n, const uint32_t *values, uint32_t *idx_out_dev,
                   uint32_t *count_dev, const uint32_t *skip_flag) {
    const uint32_t nb = cdiv(n, TILE);
    if (nb == 0) { HIP_TRY(hipMemsetAsync(count_dev, 0, 4, ctx->stream)); return; }
    hipLaunchKernelGGL(k_compact, dim3(nb), dim3(TPB), 0, ctx->stream, s.masks.p, s.block_counts.p, nb, values,
                       idx_out_dev, count_dev, skip_flag);
    (void)n;
    HIP_TRY(hipGetLastError());
}

void score_compact(plade_ctx *ctx, CompactScratch &s, const float *x, const float *y, const float *z, const float *nx,
                   const float *ny, const float *nz, const int32_t *assigned, uint32_t n, const float4 *plane_dev,
                   float eps, float cos_thresh, uint32_t *idx_out_dev, uint32_t *count_dev, const uint32_t *skip_flag) {
    const uint32_t nb = cdiv(n, TILE);
    if (nb == 0) { HIP_TRY(hipMemsetAsync(count_dev, 0, 4, ctx->stream)); return; }
    s.masks.ensure((size_t)nb * TPB);
    s.block_counts.ensure(nb);
    ctx->ev_begin("score_mark", 28.0 * n);
    hipLaunchKernelGGL(k_score_mark, dim3(nb), dim3(TPB), 0, ctx->stream, x, y, z, nx, ny, nz, assigned, n, plane_dev,
                       eps, cos_thresh, s.masks.p, s.block_counts.p, skip_flag);
    ctx->ev_end();
    compact_masks(ctx, s, n, nullptr, idx_out_dev, count_dev, skip_flag);
}

// ---------------------------------------------------------------------------------------------
// AoS (N x 6, the PLY vertex layout) -> SoA planes
__global__ void k_aos_to_soa(const float *__restrict__ aos, uint32_t n, size_t pitch, float *__restrict__ soa) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 *p = reinterpret_cast<const float2 *>(aos + 6 * (size_t)i);
    float2 a = p[0], b = p[1], c = p[2];
    soa[i] = a.x; soa[pitch + i] = a.y; soa[2 * pitch + i] = b.x;
    soa[3 * pitch + i] = b.y; soa[4 * pitch + i] = c.x; soa[5 * pitch + i] = c.y;
}

void cloud_upload(plade_ctx *ctx, const float *pos_nrm, uint32_t n, CloudDev &out) {
    out.n = n;
    out.pitch = ((size_t)n + 3) & ~(size_t)3;
    out.soa.ensure(6 * out.pitch + 4);
    if (n == 0) return;
    float *stage = out.aos.ensure((size_t)n * 6 + 8);
    ctx->h2d(stage, pos_nrm, (size_t)n * 24);
    hipLaunchKernelGGL(k_aos_to_soa, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, stage, n, out.pitch, out.soa.p);
    HIP_TRY(hipGetLastError());
    bbox_host(ctx, stage, n, 6, out.bbmin, out.bbmax);
}

}  // namespace plade

using namespace plade;

// ---- C ABI: seam S1a -----------------------------------------------------------------------
extern "C" int plade_score_planes(plade_ctx *ctx, const float *pos_nrm, const int32_t *shape_index, uint32_t n,
                                  const float *planes, uint32_t h, float eps, float cos_thresh, uint32_t *counts,
                                  uint32_t *idx_out, uint32_t cap) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(pos_nrm && planes && counts, PLADE_EINVAL, "plade_score_planes: null argument");
        CloudDev cloud;
        cloud_upload(ctx, pos_nrm, n, cloud);
        DBuf<int32_t> d_assigned;
        if (shape_index) {
            d_assigned.ensure((size_t)n + 4);
            ctx->h2d(d_assigned.p, shape_index, (size_t)n * 4);
        }
        DBuf<float4> d_planes;
        d_planes.ensure(h);
        ctx->h2d(d_planes.p, planes, (size_t)h * 16);
        DBuf<uint32_t> d_counts;
        d_counts.ensure(h + 1);
        score_multi(ctx, cloud.x(), cloud.y(), cloud.z(), cloud.nx(), cloud.ny(), cloud.nz(),
                    shape_index ? d_assigned.p : nullptr, nullptr, n, d_planes.p, h, eps, cos_thresh, d_counts.p);
        ctx->d2h(counts, d_counts.p, (size_t)h * 4);
        ctx->sync();
        if (idx_out && cap) {
            CompactScratch cs;
            DBuf<uint32_t> d_idx, d_cnt;
            d_idx.ensure((size_t)n + 4);
            d_cnt.ensure(1);
            for (uint32_t j = 0; j < h; ++j) {
                score_compact(ctx, cs, cloud.x(), cloud.y(), cloud.z(), cloud.nx(), cloud.ny(), cloud.nz(),
                              shape_index ? d_assigned.p : nullptr, n, d_planes.p + j, eps, cos_thresh, d_idx.p,
                              d_cnt.p);
                uint32_t c = 0;
                ctx->d2h(&c, d_cnt.p, 4);
                ctx->sync();
                PLADE_REQUIRE(c == counts[j], PLADE_EDEVICE, "plade_score_planes: count/compaction disagreement");
                uint32_t w = c < cap ? c : cap;
                if (w) HIP_TRY(hipMemcpy(idx_out + (size_t)j * cap, d_idx.p, (size_t)w * 4, hipMemcpyDeviceToHost));
            }
        }
        return PLADE_OK;
    });
}
