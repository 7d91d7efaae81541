// plade_amd/csrc/k_penetration.hip -- penetration filter over the <= 201 surviving candidates
// (SURVEY.md A11): the loop code/PLADE/util.cpp:450-519 around AreTwoPlanesPenetrable
// (util.cpp:1279-1458).
//
// The reference walks i1 over source planes and j1 over target planes and rejects a candidate at the
// first penetrating pair; the verdict is therefore the OR over all (i1, j1) that are not gated out,
// which is what the GPU evaluates, all triples (candidate, i1, j1) in parallel:
//   k_pen_setup : one lane per triple -- gate (util.cpp:487-492), plane/plane intersection line,
//                 clipping against both plane rectangles, overlap interval -> compact work list
//   k_pen_walk  : one workgroup per surviving triple -- the two kd-tree walks along the intersection
//                 segment.  A radiusSearch(p, r, max_nn=2) < 2 gate is "fewer than two cloud points
//                 within r/2 of the step point"; the classified set is the union over gated steps of
//                 the points within r, each counted once (checkIndex).  Both are order independent,
//                 so brute-force fp32 distance tests (FLANN L2_Simple, strict <) reproduce the counts.
#include "stages.h"

namespace plade {

struct PenItem {
    uint32_t k, i1, j1;
    float plane1[4];
    float sx, sy, sz, dx, dy, dz, length;
};

struct PenTables {
    const float *cand;     // K x 12: R row-major (9), T (3)
    const float *s_coef;   // Ps x 4
    const float *s_center; // Ps x 3
    const float *s_four;   // Ps x 12
    const float *t_coef;
    const float *t_center;
    const float *t_four;
    uint32_t K, ps, pt;
};

__device__ __forceinline__ int edge_hits(f3 lineVec, f3 linePoint, const f3 *c, f3 *out) {
    int n = 0;
    for (int i = 1; i <= 4; ++i) {
        const f3 a = c[(i - 1) % 4], b = c[i % 4];
        const f3 tl = normalized_e(b - a);
        f3 ip;
        if (!lines_meet(lineVec, linePoint, tl, a, ip)) continue;
        if (dot_e(a - ip, b - ip) > 0) continue;
        if (n < 4) out[n] = ip;
        ++n;
    }
    return n;
}

__global__ __launch_bounds__(256) void k_pen_setup(PenTables tb, float len_th, float ang_th, PenItem *__restrict__ items,
                                                   uint32_t *__restrict__ n_items, uint32_t cap) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)tb.K * tb.ps * tb.pt;
    if (idx >= total) return;
    const uint32_t j1 = (uint32_t)(idx % tb.pt), i1 = (uint32_t)((idx / tb.pt) % tb.ps), k = (uint32_t)(idx / ((size_t)tb.pt * tb.ps));
    const float *c = tb.cand + 12 * (size_t)k;
    m3 R;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) R.m[r][q] = c[3 * r + q];
    const f3 T(c[9], c[10], c[11]);
    const float T12[12] = {c[0], c[1], c[2], c[9], c[3], c[4], c[5], c[10], c[6], c[7], c[8], c[11]};
    const float *sc = tb.s_coef + 4 * (size_t)i1;
    const f3 pn = mul_e(R, f3(sc[0], sc[1], sc[2]));
    float plane1[4] = {pn.x, pn.y, pn.z, 0.f};
    plane1[3] = -(-sc[3] + dot_s(pn, T));
    const f3 c2m = mul_e(R, f3(tb.s_center[3 * i1], tb.s_center[3 * i1 + 1], tb.s_center[3 * i1 + 2])) + T;
    const float *tc = tb.t_coef + 4 * (size_t)j1;
    const f3 plane_A(tc[0], tc[1], tc[2]);
    const f3 tcen(tb.t_center[3 * j1], tb.t_center[3 * j1 + 1], tb.t_center[3 * j1 + 2]);
    const double c2p = (double)((fabsf(dot_e(plane_A, c2m) + tc[3]) + fabsf(dot_e(pn, tcen) + plane1[3])) / 2);
    if (c2p < (double)len_th && dot_e(pn, plane_A) > ang_th) return;  // util.cpp:489
    f3 lineVec, linePoint;
    if (!plane_plane_line(plane1, tc, lineVec, linePoint)) return;    // util.cpp:1296
    f3 c1[4], c2[4];
    for (int q = 0; q < 4; ++q) {
        const float *f = tb.s_four + 12 * (size_t)i1 + 3 * q;
        c1[q] = pcl_xform(T12, f3(f[0], f[1], f[2]));
        const float *g = tb.t_four + 12 * (size_t)j1 + 3 * q;
        c2[q] = f3(g[0], g[1], g[2]);
    }
    f3 ip1[4], ip2[4];
    const int n1 = edge_hits(lineVec, linePoint, c1, ip1);
    const int n2 = edge_hits(lineVec, linePoint, c2, ip2);
    if (n1 != 2 || n2 != 2) return;  // empty -> not penetrable; any other count -> "-1, continue"
    const f3 direc = normalized_e(ip1[1] - ip1[0]);
    const f3 inter[4] = {ip1[0], ip1[1], ip2[0], ip2[1]};
    float len[4];
    int ord[4] = {0, 1, 2, 3};
    for (int q = 0; q < 4; ++q) len[q] = dot_e(inter[q] - inter[0], direc);
    for (int a = 1; a < 4; ++a) {  // insertion sort == std::sort for n <= 16
        int oi = ord[a];
        float lv = len[oi];
        int b = a - 1;
        while (b >= 0 && lv < len[ord[b]]) { ord[b + 1] = ord[b]; --b; }
        ord[b + 1] = oi;
    }
    if (0 == (ord[0] / 2 - ord[1] / 2)) return;  // no overlap of the two clipped segments
    const f3 sp = inter[ord[1]], ep = inter[ord[2]];
    const float length = norm_e(ep - sp);
    const uint32_t slot = atomicAdd(n_items, 1u);
    if (slot >= cap) return;
    PenItem it;
    it.k = k; it.i1 = i1; it.j1 = j1;
    for (int q = 0; q < 4; ++q) it.plane1[q] = plane1[q];
    it.sx = sp.x; it.sy = sp.y; it.sz = sp.z; it.dx = direc.x; it.dy = direc.y; it.dz = direc.z; it.length = length;
    items[slot] = it;
}

constexpr int PEN_MAXS = 1024;
constexpr int PEN_TPB = 64;   // one wavefront per surviving triple

__global__ __launch_bounds__(PEN_TPB) void k_pen_walk(const PenItem *__restrict__ items, uint32_t n_items, PenTables tb,
                                                  const float *__restrict__ s_xyz, const uint32_t *__restrict__ s_off,
                                                  const float *__restrict__ t_xyz, const uint32_t *__restrict__ t_off,
                                                  float search_radius, int min_points, float min_distance,
                                                  uint32_t *__restrict__ cand_flags, uint32_t *__restrict__ overflow) {
    __shared__ float s_dist[PEN_MAXS];
    __shared__ uint32_t s_cnt[PEN_MAXS];
    __shared__ int s_n, s_pos, s_neg;
    if (blockIdx.x >= n_items) return;
    const PenItem it = items[blockIdx.x];
    // the verdict per candidate is an OR over its items: once one item has rejected the candidate the
    // others cannot change it
    if (__atomic_load_n(&cand_flags[it.k], __ATOMIC_RELAXED)) return;
    const f3 start(it.sx, it.sy, it.sz), direc(it.dx, it.dy, it.dz);
    if (threadIdx.x == 0) {
        int n = 0;
        for (float dist = 0; dist < it.length; dist += search_radius) {  // util.cpp:1383 (fp32 accumulation)
            if (n < PEN_MAXS) s_dist[n] = dist;
            ++n;
            if (n > PEN_MAXS) break;
        }
        if (n > PEN_MAXS) { atomicExch(overflow, 1u); n = PEN_MAXS; }
        s_n = n;
    }
    __syncthreads();
    const int nsteps = s_n;
    if (nsteps == 0) return;  // length <= 0: both walks see nothing -> positive/negative < minPoints
    const float half_r2 = (float)((double)(search_radius / 2) * (double)(search_radius / 2));
    const float full_r2 = (float)((double)search_radius * (double)search_radius);
    const float *c = tb.cand + 12 * (size_t)it.k;
    const float T12[12] = {c[0], c[1], c[2], c[9], c[3], c[4], c[5], c[10], c[6], c[7], c[8], c[11]};
    const uint32_t sb = s_off[it.i1], se = s_off[it.i1 + 1], tb0 = t_off[it.j1], te = t_off[it.j1 + 1];
    const float *tc = tb.t_coef + 4 * (size_t)it.j1;
    const float inv_r = 1.f / search_radius;

    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: gate = target plane cloud, classified = transformed source plane cloud vs plane2
        // pass 1: gate = transformed source plane cloud, classified = target plane cloud vs plane1
        for (int i = threadIdx.x; i < nsteps; i += blockDim.x) s_cnt[i] = 0u;
        if (threadIdx.x == 0) { s_pos = 0; s_neg = 0; }
        __syncthreads();
        const uint32_t gb = pass == 0 ? tb0 : sb, ge = pass == 0 ? te : se;
        for (uint32_t i = gb + threadIdx.x; i < ge; i += blockDim.x) {
            f3 p;
            if (pass == 0) p = f3(t_xyz[3 * (size_t)i], t_xyz[3 * (size_t)i + 1], t_xyz[3 * (size_t)i + 2]);
            else p = pcl_xform(T12, f3(s_xyz[3 * (size_t)i], s_xyz[3 * (size_t)i + 1], s_xyz[3 * (size_t)i + 2]));
            const f3 d = p - start;
            const float t = d.x * direc.x + d.y * direc.y + d.z * direc.z;
            // conservative reject: farther than r/2 (+2 %) from the line => within r/2 of no step point
            if ((d.x * d.x + d.y * d.y + d.z * d.z) - t * t > half_r2 * 1.02f + 1e-12f) continue;
            const int kc = (int)floorf(t * inv_r);
            for (int kk = max(kc - 2, 0); kk <= min(kc + 3, nsteps - 1); ++kk) {
                const float dist = s_dist[kk];
                const f3 spt(start.x + dist * direc.x, start.y + dist * direc.y, start.z + dist * direc.z);
                if (flann_d2(spt, p) < half_r2) atomicAdd(&s_cnt[kk], 1u);
            }
        }
        __syncthreads();
        const float pl0 = pass == 0 ? tc[0] : it.plane1[0], pl1 = pass == 0 ? tc[1] : it.plane1[1],
                    pl2 = pass == 0 ? tc[2] : it.plane1[2], pl3 = pass == 0 ? tc[3] : it.plane1[3];
        const uint32_t ab = pass == 0 ? sb : tb0, ae = pass == 0 ? se : te;
        int lpos = 0, lneg = 0;
        for (uint32_t i = ab + threadIdx.x; i < ae; i += blockDim.x) {
            f3 p;
            if (pass == 0) p = pcl_xform(T12, f3(s_xyz[3 * (size_t)i], s_xyz[3 * (size_t)i + 1], s_xyz[3 * (size_t)i + 2]));
            else p = f3(t_xyz[3 * (size_t)i], t_xyz[3 * (size_t)i + 1], t_xyz[3 * (size_t)i + 2]);
            const f3 d = p - start;
            const float t = d.x * direc.x + d.y * direc.y + d.z * direc.z;
            if ((d.x * d.x + d.y * d.y + d.z * d.z) - t * t > full_r2 * 1.02f + 1e-12f) continue;
            const int kc = (int)floorf(t * inv_r);
            bool hit = false;
            for (int kk = max(kc - 2, 0); kk <= min(kc + 3, nsteps - 1) && !hit; ++kk) {
                if (s_cnt[kk] < 2u) continue;
                const float dist = s_dist[kk];
                const f3 spt(start.x + dist * direc.x, start.y + dist * direc.y, start.z + dist * direc.z);
                if (flann_d2(spt, p) < full_r2) hit = true;
            }
            if (hit) {
                const float td = pl0 * p.x + pl1 * p.y + pl2 * p.z + pl3;
                if (fabsf(td) > min_distance) { if (td >= 0) ++lpos; else ++lneg; }
            }
        }
        if (lpos) atomicAdd(&s_pos, lpos);
        if (lneg) atomicAdd(&s_neg, lneg);
        __syncthreads();
        const int pos = s_pos, neg = s_neg;
        __syncthreads();
        if (pass == 0) {
            if (pos < min_points || neg < min_points) return;
        } else {
            if (pos < min_points && neg < min_points) return;
        }
        if ((double)max(pos, neg) / (double)min(pos, neg + 1) > 5) return;
    }
    if (threadIdx.x == 0) atomicOr(&cand_flags[it.k], 1u);
}

void penetration_filter(plade_ctx *ctx, const float *cand_rt_host, uint32_t K, const PlaneGeomHost &src,
                        const PlaneGeomHost &tgt, const PlaneCloudsDev &src_pts, const PlaneCloudsDev &tgt_pts,
                        float length_threshold, float angle_threshold, std::vector<int32_t> &flags_out) {
    flags_out.assign(K, 0);
    if (!K || !src.P || !tgt.P) return;
    // upload tables
    const size_t nf = 12 * (size_t)K + (4 + 3 + 12) * ((size_t)src.P + tgt.P);
    std::vector<float> h(nf);
    float *p = h.data();
    memcpy(p, cand_rt_host, 48 * (size_t)K);
    float *o_cand = p; p += 12 * (size_t)K;
    float *o_sc = p; memcpy(p, src.coef.data(), 16 * (size_t)src.P); p += 4 * (size_t)src.P;
    float *o_scen = p; memcpy(p, src.center.data(), 12 * (size_t)src.P); p += 3 * (size_t)src.P;
    float *o_sf = p; memcpy(p, src.four.data(), 48 * (size_t)src.P); p += 12 * (size_t)src.P;
    float *o_tc = p; memcpy(p, tgt.coef.data(), 16 * (size_t)tgt.P); p += 4 * (size_t)tgt.P;
    float *o_tcen = p; memcpy(p, tgt.center.data(), 12 * (size_t)tgt.P); p += 3 * (size_t)tgt.P;
    float *o_tf = p; memcpy(p, tgt.four.data(), 48 * (size_t)tgt.P); p += 12 * (size_t)tgt.P;
    float *d = reinterpret_cast<float *>(ctx->scratch[4].ensure(nf * 4 + 64));
    HIP_TRY(hipMemcpyAsync(d, h.data(), nf * 4, hipMemcpyHostToDevice, ctx->stream));
    PenTables tb;
    tb.cand = d + (o_cand - h.data()); tb.s_coef = d + (o_sc - h.data()); tb.s_center = d + (o_scen - h.data());
    tb.s_four = d + (o_sf - h.data()); tb.t_coef = d + (o_tc - h.data()); tb.t_center = d + (o_tcen - h.data());
    tb.t_four = d + (o_tf - h.data());
    tb.K = K; tb.ps = src.P; tb.pt = tgt.P;
    const size_t total = (size_t)K * src.P * tgt.P;
    PLADE_REQUIRE(total < (1ull << 31), PLADE_ELIMIT, "penetration: too many (candidate, plane, plane) triples");
    PenItem *d_items = reinterpret_cast<PenItem *>(ctx->scratch[5].ensure(total * sizeof(PenItem) + 64));
    uint32_t *d_ctr = reinterpret_cast<uint32_t *>(ctx->scratch[6].ensure(((size_t)K + 4) * 4));
    HIP_TRY(hipMemsetAsync(d_ctr, 0, ((size_t)K + 4) * 4, ctx->stream));
    uint32_t *d_n = d_ctr, *d_over = d_ctr + 1, *d_flags = d_ctr + 2;
    hipLaunchKernelGGL(k_pen_setup, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, tb, length_threshold,
                       angle_threshold, d_items, d_n, (uint32_t)total);
    uint32_t n_items = 0;
    HIP_TRY(hipMemcpyAsync(&n_items, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stats.add("pen_items", (double)n_items);
    if (n_items) {
        // AreTwoPlanesPenetrable(..., searchRadius = lengthThreshold, minPointsNum = 10, minDistance = lengthThreshold / 2)
        const float search_radius = (float)(double)length_threshold;
        const float min_distance = (float)((double)length_threshold / 2);
        hipLaunchKernelGGL(k_pen_walk, dim3(n_items), dim3(PEN_TPB), 0, ctx->stream, d_items, n_items, tb, src_pts.xyz.p,
                           src_pts.d_off.p, tgt_pts.xyz.p, tgt_pts.d_off.p, search_radius, 10, min_distance, d_flags,
                           d_over);
    }
    std::vector<uint32_t> out((size_t)K + 1);
    HIP_TRY(hipMemcpyAsync(out.data(), d_over, ((size_t)K + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    PLADE_REQUIRE(out[0] == 0, PLADE_ELIMIT, "penetration: intersection segment longer than 1024 search steps");
    for (uint32_t k = 0; k < K; ++k) flags_out[k] = out[k + 1] ? 1 : 0;
}

}  // namespace plade
