// plade_amd/csrc/k1_point_test.h -- K1: THE point-to-plane inlier test and the tile loader of every scan kernel
// (SURVEY.md A3).  One definition for the whole library: the RANSAC loop's kernels (k_r_score_sub, k_r_rescore,
// k_r_mark in ransac.hip) and the seams that pin them against the reference (plade_score_planes,
// plade_score_planes_subset) run exactly this code.
//
// Reference semantics: FlatNormalThreshPointCompatibilityFunc::operator()
// (code/3rd_party/ransac/FlatNormalThreshPointCompatibilityFunc.h:14-23) with Plane::Distance = fabs(m_dist - n.p)
// (ransac/Plane.h:31) and Vec3f::dot accumulated left to right (ransac/basic.h:80-86); the visitor's filter
// shapeIndex[i] == -1 (ransac/ScorePrimitiveShapeVisitor.h:39-46).  strict `<` on the distance, non-strict `>=` on the
// normal test, everything in fp32 without FMA contraction (the library is built with -ffp-contract=off).
//
// Layout: SoA planes x|y|z|nx|ny|nz, 4 consecutive points per lane via 16-byte loads (1 KiB per wave instruction).
// HBM-bound: 28 B per point and pass (12 position + 12 normal + 4 shapeIndex).
#pragma once
#include "common.h"

namespace plade {

constexpr int K1_TPB = 256, K1_PPT = 4, K1_TILE = K1_TPB * K1_PPT;   // 1024 points per workgroup

__device__ __forceinline__ bool compatible(float4 pl, float px, float py, float pz, float qx, float qy, float qz, float eps,
                                           float cos_t) {
    float d = pl.x * px;
    d += pl.y * py;
    d += pl.z * pz;
    const float dist = fabsf(pl.w - d);
    float nd = pl.x * qx;
    nd += pl.y * qy;
    nd += pl.z * qz;
    return (dist < eps) && (fabsf(nd) >= cos_t);
}

struct Tile {
    float px[K1_PPT], py[K1_PPT], pz[K1_PPT], qx[K1_PPT], qy[K1_PPT], qz[K1_PPT];
    bool valid[K1_PPT];
};

// 4 consecutive points per lane via 16-byte loads; `assigned` (nullable: every point counts) is indexed directly or through
// sub_index.  The extraction loop passes nullptr: its full passes read a view that holds exactly the unassigned points
// (ransac.hip, scan view), 24 bytes per point; the seams hand over the caller's int32 shapeIndex array (28 bytes per point).
__device__ __forceinline__ void load_tile(Tile &t, const float *x, const float *y, const float *z, const float *nx, const float *ny,
                                          const float *nz, const int32_t *assigned, const uint32_t *sub_index, uint32_t n,
                                          uint32_t base) {
    constexpr int PPT = K1_PPT;
    if (base + PPT <= n) {
        const float4 a = *reinterpret_cast<const float4 *>(x + base), b = *reinterpret_cast<const float4 *>(y + base),
                     c = *reinterpret_cast<const float4 *>(z + base), d = *reinterpret_cast<const float4 *>(nx + base),
                     e = *reinterpret_cast<const float4 *>(ny + base), f = *reinterpret_cast<const float4 *>(nz + base);
        t.px[0] = a.x; t.px[1] = a.y; t.px[2] = a.z; t.px[3] = a.w;
        t.py[0] = b.x; t.py[1] = b.y; t.py[2] = b.z; t.py[3] = b.w;
        t.pz[0] = c.x; t.pz[1] = c.y; t.pz[2] = c.z; t.pz[3] = c.w;
        t.qx[0] = d.x; t.qx[1] = d.y; t.qx[2] = d.z; t.qx[3] = d.w;
        t.qy[0] = e.x; t.qy[1] = e.y; t.qy[2] = e.z; t.qy[3] = e.w;
        t.qz[0] = f.x; t.qz[1] = f.y; t.qz[2] = f.z; t.qz[3] = f.w;
        if (assigned && !sub_index) {
            const int4 s = *reinterpret_cast<const int4 *>(assigned + base);
            t.valid[0] = s.x == -1; t.valid[1] = s.y == -1; t.valid[2] = s.z == -1; t.valid[3] = s.w == -1;
        } else if (assigned) {
            const uint4 si = *reinterpret_cast<const uint4 *>(sub_index + base);
            t.valid[0] = assigned[si.x] == -1; t.valid[1] = assigned[si.y] == -1;
            t.valid[2] = assigned[si.z] == -1; t.valid[3] = assigned[si.w] == -1;
        } else {
            t.valid[0] = t.valid[1] = t.valid[2] = t.valid[3] = true;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const uint32_t i = base + k;
            const bool in = i < n;
            t.px[k] = in ? x[i] : 0.f; t.py[k] = in ? y[i] : 0.f; t.pz[k] = in ? z[i] : 0.f;
            t.qx[k] = in ? nx[i] : 0.f; t.qy[k] = in ? ny[i] : 0.f; t.qz[k] = in ? nz[i] : 0.f;
            bool un = true;
            if (in && assigned) un = (sub_index ? assigned[sub_index[i]] : assigned[i]) == -1;
            t.valid[k] = in && un;
        }
    }
}

}  // namespace plade
