// plade_amd/csrc/k_cluster.hip -- K6 candidate transforms, candidate clustering, K7 plane consistency
// (SURVEY.md A8, A9, A10).
//
// K6: ComputeTransformationUsingTwoVecAndOnePoint (code/PLADE/util.cpp:604-624) per match in the
//     order of the loop util.cpp:303-327: R = umeyama rotation of (l1, l2, l1 x l2) triples,
//     T = p_target - R p_source.  One lane per match, fp32 Jacobi SVD in registers.
// A9: ClusterTransformation (util.cpp:1245-1277) = pcl::ConditionalEuclideanClustering::segment
//     (pcl-1.8.1/segmentation/include/pcl/segmentation/impl/conditional_euclidean_clustering.hpp:
//     42-138) with EnforceSimilarity (util.cpp:1232-1243).  Region growing under a symmetric
//     predicate yields the connected components of the graph
//        edge(a,b) <=> |T_a-T_b|^2 < float(r*r)  &&  |euler_a-euler_b|^2 < gate ,
//     clusters come out ordered by their smallest member and that member is the representative
//     (util.cpp:355-357).  GPU: lock-free union-find (smaller index wins, so the root IS the seed)
//     over a sorted-cell spatial hash of the translations.
// K7: plane-consistency count per cluster seed (util.cpp:359-401), one lane per cluster, plane
//     tables in LDS.
#include "stages.h"
#include "prims.h"

namespace plade {

__device__ void k_transforms(const VB &vb, const float *__restrict__ q_lv1, const float *__restrict__ q_lv2,
                                                    const float *__restrict__ q_p1, const float *__restrict__ t_lv1,
                                                    const float *__restrict__ t_lv2, const float *__restrict__ t_p1,
                                                    const uint32_t *__restrict__ q_idx, const uint32_t *__restrict__ t_idx,
                                                    uint32_t m, float4 *__restrict__ rt, float *__restrict__ t_minmax_part) {
    __shared__ float s_lds[6][8];
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < m) {
    const uint32_t q = q_idx[i], t = t_idx[i];
    f3 s[3], d[3];
    s[0] = f3(q_lv1[3 * q], q_lv1[3 * q + 1], q_lv1[3 * q + 2]);
    s[1] = f3(q_lv2[3 * q], q_lv2[3 * q + 1], q_lv2[3 * q + 2]);
    s[2] = cross(s[0], s[1]);
    d[0] = f3(t_lv1[3 * t], t_lv1[3 * t + 1], t_lv1[3 * t + 2]);
    d[1] = f3(t_lv2[3 * t], t_lv2[3 * t + 1], t_lv2[3 * t + 2]);
    d[2] = cross(d[0], d[1]);
    const m3 R = umeyama_rot3(s, d);
    const f3 sp(q_p1[3 * q], q_p1[3 * q + 1], q_p1[3 * q + 2]), tp(t_p1[3 * t], t_p1[3 * t + 1], t_p1[3 * t + 2]);
    const f3 T = tp - mul_e(R, sp);
    float roll, pitch, yaw;
    euler_zyx(R, roll, pitch, yaw);
    rt[4 * (size_t)i + 0] = make_float4(R.m[0][0], R.m[0][1], R.m[0][2], T.x);
    rt[4 * (size_t)i + 1] = make_float4(R.m[1][0], R.m[1][1], R.m[1][2], T.y);
    rt[4 * (size_t)i + 2] = make_float4(R.m[2][0], R.m[2][1], R.m[2][2], T.z);
    rt[4 * (size_t)i + 3] = make_float4(roll, pitch, yaw, 0.f);
    mn[0] = mx[0] = T.x; mn[1] = mx[1] = T.y; mn[2] = mx[2] = T.z;
    }
    // bounding box of the translations for the clustering grid: one partial per workgroup, reduced by the host
    // (it needs the box anyway to size the grid), no shared counters
    for (int q = 0; q < 3; ++q)
        for (int d = 32; d >= 1; d >>= 1) {
            mn[q] = fminf(mn[q], __shfl_xor(mn[q], d, 64));
            mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], d, 64));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int q = 0; q < 3; ++q) { s_lds[q][wave] = mn[q]; s_lds[3 + q][wave] = mx[q]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_lds[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, s_lds[threadIdx.x][w]) : fmaxf(v, s_lds[threadIdx.x][w]);
        t_minmax_part[6 * (size_t)vb.bx + threadIdx.x] = v;
    }
}

void build_transforms(plade_ctx *ctx, const PairTableDev &src, const PairTableDev &tgt, const uint32_t *d_q_idx,
                      const uint32_t *d_t_idx, uint32_t m, CandidateSet &cs) {
    cs.m = m;
    cs.rt.ensure(4 * (size_t)m + 4);
    if (!m) return;
    cs.t_minmax.ensure(6 * (size_t)cdiv(m, 256) + 8);
    launch<k_transforms, 256>(ctx, dim3(cdiv(m, 256)), 0, src.lv1.p, src.lv2.p, src.p1.p,
                       tgt.lv1.p, tgt.lv2.p, tgt.p1.p, d_q_idx, d_t_idx, m, cs.rt.p, cs.t_minmax.p);
    HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// cells of edge >= the cluster radius over the translations' bounding box; keys packed to the bits the
// extents need (x lowest, so the three x-neighbours of a cell row are one contiguous key range)
struct HashGrid {
    float mnx, mny, mnz, inv;
    int bx, bxy;   // shifts of the y and z fields
};

__device__ __forceinline__ uint64_t cell_key(const HashGrid &g, int cx, int cy, int cz) {
    return ((uint64_t)(uint32_t)cz << g.bxy) | ((uint64_t)(uint32_t)cy << g.bx) | (uint64_t)(uint32_t)cx;
}

// keys + union-find / size initialisation
__device__ void k_t_keys(const VB &vb, const float4 *__restrict__ rt, uint32_t m, HashGrid g, uint64_t *__restrict__ keys,
                         uint32_t *__restrict__ vals, uint32_t *__restrict__ parent, uint32_t *__restrict__ sizes) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int cx = (int)floorf((rt[4 * (size_t)i].w - g.mnx) * g.inv) + 1;
    const int cy = (int)floorf((rt[4 * (size_t)i + 1].w - g.mny) * g.inv) + 1;
    const int cz = (int)floorf((rt[4 * (size_t)i + 2].w - g.mnz) * g.inv) + 1;
    keys[i] = cell_key(g, cx, cy, cz);
    vals[i] = i;
    parent[i] = i;
    sizes[i] = 0;
}

__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    uint32_t p = parent[x];
    while (p != x) {
        uint32_t gp = parent[p];
        if (gp != p) parent[x] = gp;  // path halving (benign race: only ever moves towards the root)
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { uint32_t t = a; a = b; b = t; }  // a > b: hang the larger root under the smaller
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *a, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// After the sort: nodes gathered into cell order (translation + original index, Euler angles), and for the
// first node of every cell the nine spans [lo, hi) of sorted positions covering its 27-neighbourhood.
__device__ void k_cell_spans(const VB &vb, const float4 *__restrict__ rt, const uint32_t *__restrict__ order,
                                                    const uint64_t *__restrict__ skeys, uint32_t m, HashGrid g,
                                                    float4 *__restrict__ st, float4 *__restrict__ se,
                                                    uint2 *__restrict__ spans /* m x 9, rows of cell heads only */) {
    const uint32_t si = vb.bx * blockDim.x + threadIdx.x;
    if (si >= m) return;
    const uint32_t a = order[si];
    st[si] = make_float4(rt[4 * (size_t)a].w, rt[4 * (size_t)a + 1].w, rt[4 * (size_t)a + 2].w, __uint_as_float(a));
    se[si] = rt[4 * (size_t)a + 3];
    const uint64_t key = skeys[si];
    if (si && skeys[si - 1] == key) return;
    const uint64_t mx_ = (1ull << g.bx) - 1, my_ = (1ull << (g.bxy - g.bx)) - 1;
    const int cx = (int)(key & mx_), cy = (int)((key >> g.bx) & my_), cz = (int)(key >> g.bxy);
    int r = 0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy, ++r) {
            const uint64_t k0 = cell_key(g, cx - 1, cy + dy, cz + dz), k1 = cell_key(g, cx + 1, cy + dy, cz + dz);
            const uint32_t lo = lower_bound_u64(skeys, m, k0);
            uint32_t hi = lo;
            if (lo < m && skeys[lo] <= k1) hi = lower_bound_u64(skeys, m, k1 + 1);
            spans[(size_t)si * 9 + r] = make_uint2(lo, hi);
        }
}

// One lane per (node in cell order, neighbour row): nine lanes share a node, lanes of a wave share cells.  The first
// EDGE_QUICK entries of a span are walked by the lane itself; what is left of the long spans -- the dense cells around
// the true transformation hold thousands of candidates, and a lane walking 3 x 2000 entries alone set the kernel's
// duration -- is shared out: the wavefront takes the unfinished (node, span) pairs one after the other, 64 consecutive
// entries per step (coalesced 16 B reads).  The union-find always hangs the larger root under the smaller one, so the
// components and their roots (smallest member) do not depend on the order the edges are found in.
constexpr uint32_t EDGE_QUICK = 16;
__device__ __forceinline__ void edge_test(const float4 *__restrict__ st, const float4 *__restrict__ se, float4 pa, float4 ea,
                                          uint32_t j, float r2, float gate, uint32_t *__restrict__ parent) {
    const float4 pb = st[j];
    const uint32_t a = __float_as_uint(pa.w), b = __float_as_uint(pb.w);
    if (b >= a) return;  // every undirected edge once
    if (!(flann_d2(f3(pa.x, pa.y, pa.z), f3(pb.x, pb.y, pb.z)) < r2)) return;
    const float4 eb = se[j];
    const float t0 = ea.x - eb.x, t1 = ea.y - eb.y, t2 = ea.z - eb.z;
    const float sq = (t0 * t0 + t1 * t1) + t2 * t2;  // Eigen::VectorXf(3).squaredNorm()
    if (sq < gate) uf_union(parent, a, b);
}
__device__ void k_cluster_edges(const VB &vb, const float4 *__restrict__ st, const float4 *__restrict__ se,
                                                       const uint64_t *__restrict__ skeys, const uint2 *__restrict__ spans,
                                                       uint32_t m, float r2, float gate, uint32_t *__restrict__ parent) {
    const uint32_t tid = vb.bx * blockDim.x + threadIdx.x;
    const uint32_t si = tid / 9u, r = tid - 9u * si;
    const bool live = si < m;
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), ea = pa;
    uint32_t j = 0, end = 0;
    if (live) {
        pa = st[si];
        ea = se[si];
        const uint32_t head = lower_bound_u64(skeys, m, skeys[si]);
        const uint2 sp = spans[(size_t)head * 9 + r];
        j = sp.x; end = sp.y;
    }
    for (const uint32_t quick_end = min(end, j + EDGE_QUICK); j < quick_end; ++j) edge_test(st, se, pa, ea, j, r2, gate, parent);
    unsigned long long pend = __ballot(j < end);
    const uint32_t lane = threadIdx.x & 63u;
    while (pend) {
        const int src = __ffsll((long long)pend) - 1;
        pend &= pend - 1;
        const float4 pa_s = make_float4(__shfl(pa.x, src, 64), __shfl(pa.y, src, 64), __shfl(pa.z, src, 64), __shfl(pa.w, src, 64));
        const float4 ea_s = make_float4(__shfl(ea.x, src, 64), __shfl(ea.y, src, 64), __shfl(ea.z, src, 64), __shfl(ea.w, src, 64));
        const uint32_t j_s = __shfl(j, src, 64), end_s = __shfl(end, src, 64);
        for (uint32_t jj = j_s + lane; jj < end_s; jj += 64) edge_test(st, se, pa_s, ea_s, jj, r2, gate, parent);
    }
}

__device__ void k_flatten(const VB &vb, uint32_t *__restrict__ parent, uint32_t n, uint32_t *__restrict__ sizes,
                          uint32_t *__restrict__ root_flags) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) { if (i == n) root_flags[n] = 0; return; }
    uint32_t r = i;
    while (parent[r] != r) r = parent[r];
    atomicAdd(&sizes[r], 1u);
    root_flags[i] = (r == i) ? 1u : 0u;
}
__device__ void k_gather_sizes(const VB &vb, const uint32_t *__restrict__ seeds, uint32_t n, const uint32_t *__restrict__ sizes_all,
                               uint32_t *__restrict__ out) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sizes_all[seeds[i]];
}

static int bits_for(double cells) {
    int b = 1;
    while ((double)(1ull << b) < cells) ++b;
    return b;
}

void cluster_transforms(plade_ctx *ctx, CandidateSet &cs, float dist_threshold, float angle_gate) {
    const uint32_t m = cs.m;
    cs.n_clusters = 0;
    if (!m) return;
    // bbox of the translations (per-workgroup partials left by k_transforms)
    const uint32_t nbp = cdiv(m, 256);
    std::vector<float> part(6 * (size_t)nbp);
    ctx->d2h(part.data(), cs.t_minmax.p, 24 * (size_t)nbp);
    ctx->sync();
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t b = 0; b < nbp; ++b)
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], part[6 * (size_t)b + k]); mx[k] = std::max(mx[k], part[6 * (size_t)b + 3 + k]); }
    const float r2 = pcl_r2((double)dist_threshold);  // setClusterTolerance -> radiusSearch(double) -> float(r*r)
    float cell = dist_threshold * 1.001f;
    if (!(cell > 0.f)) cell = 1.f;
    for (int k = 0; k < 3; ++k)
        PLADE_REQUIRE(std::isfinite(mn[k]) && std::isfinite(mx[k]), PLADE_EFAIL, "candidate transforms are not finite");
    for (;;) {  // at most 21 bits per axis (+1 offset, +1 probe margin)
        double ext = std::max({(double)mx[0] - mn[0], (double)mx[1] - mn[1], (double)mx[2] - mn[2]});
        if (ext / cell + 4 < (double)(1 << 21)) break;
        cell *= 2.f;
    }
    const int bx = bits_for(((double)mx[0] - mn[0]) / cell + 4), by = bits_for(((double)mx[1] - mn[1]) / cell + 4),
              bz = bits_for(((double)mx[2] - mn[2]) / cell + 4);
    HashGrid g{mn[0], mn[1], mn[2], 1.f / cell, bx, bx + by};
    cs.ckeys.ensure(m); cs.ckeys2.ensure(m); cs.cvals.ensure(m); cs.cvals2.ensure(m);
    cs.parent.ensure(m); cs.sizes_all.ensure(m); cs.flags.ensure((size_t)m + 1);
    cs.st.ensure(m); cs.se.ensure(m); cs.spans.ensure((size_t)m * 9);
    const unsigned nb = cdiv(m, 256);
    launch<k_t_keys, 256>(ctx, dim3(nb), 0, cs.rt.p, m, g, cs.ckeys.p, cs.cvals.p, cs.parent.p,
                       cs.sizes_all.p);
    sort_pairs_u64(ctx, cs.ckeys.p, cs.ckeys2.p, cs.cvals.p, cs.cvals2.p, m, bx + by + bz);
    launch<k_cell_spans, 256>(ctx, dim3(nb), 0, cs.rt.p, cs.cvals2.p, cs.ckeys2.p, m, g, cs.st.p,
                       cs.se.p, cs.spans.p);
    ctx->ev_begin("cluster_edges", 0.0);   // latency / atomics bound, no HBM figure
    launch<k_cluster_edges, 256>(ctx, dim3(cdiv((size_t)m * 9, 256)), 0, cs.st.p, cs.se.p, cs.ckeys2.p,
                       cs.spans.p, m, r2, angle_gate, cs.parent.p);
    ctx->ev_end();
    launch<k_flatten, 256>(ctx, dim3(cdiv(m + 1, 256)), 0, cs.parent.p, m, cs.sizes_all.p, cs.flags.p);
    cs.n_clusters = compact_flags(ctx, cs.flags.p, m, cs.pos, cs.seeds);
    cs.sizes.ensure((size_t)cs.n_clusters + 1);
    if (cs.n_clusters)
        launch<k_gather_sizes, 256>(ctx, dim3(cdiv(cs.n_clusters, 256)), 0, cs.seeds.p,
                           cs.n_clusters, cs.sizes_all.p, cs.sizes.p);
    HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
__device__ void k_plane_consistency(const VB &vb, const float4 *__restrict__ rt, const uint32_t *__restrict__ seeds,
                                                           uint32_t n_clusters, const float *__restrict__ s_tab,
                                                           uint32_t ps, const float *__restrict__ t_tab, uint32_t pt,
                                                           f3 src_bc, f3 tgt_bc, float max_radius, float cos_th,
                                                           float len_th, int32_t *__restrict__ counts) {
    // tables: per plane 8 floats = coef(4), centre(3), radius(1)
    extern __shared__ float sh[];
    float *S = sh, *Tt = sh + 8 * ps;
    for (uint32_t i = threadIdx.x; i < 8 * ps; i += blockDim.x) S[i] = s_tab[i];
    for (uint32_t i = threadIdx.x; i < 8 * pt; i += blockDim.x) Tt[i] = t_tab[i];
    __syncthreads();
    const uint32_t c = vb.bx * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    const uint32_t k = seeds[c];
    const float4 r0 = rt[4 * (size_t)k], r1 = rt[4 * (size_t)k + 1], r2v = rt[4 * (size_t)k + 2];
    m3 R;
    R.m[0][0] = r0.x; R.m[0][1] = r0.y; R.m[0][2] = r0.z;
    R.m[1][0] = r1.x; R.m[1][1] = r1.y; R.m[1][2] = r1.z;
    R.m[2][0] = r2v.x; R.m[2][1] = r2v.y; R.m[2][2] = r2v.z;
    const f3 T(r0.w, r1.w, r2v.w);
    // util.cpp:359-363: transformed source centre must stay within maxRadius of the target centre
    const f3 tc = mul_e(R, src_bc) + T;
    if (norm_e(tc - tgt_bc) > max_radius) { counts[c] = -1; return; }
    int matched = 0;
    for (uint32_t i1 = 0; i1 < ps; ++i1) {
        const float *sp = S + 8 * i1;
        const f3 plane1 = mul_e(R, f3(sp[0], sp[1], sp[2]));
        const float d = -(-sp[3] + dot_s(plane1, T));
        const f3 c2d = mul_e(R, f3(sp[4], sp[5], sp[6])) + T;
        for (uint32_t j1 = 0; j1 < pt; ++j1) {
            const float *tp = Tt + 8 * j1;
            const f3 plane_A(tp[0], tp[1], tp[2]);
            if (dot_e(plane1, plane_A) < cos_th) continue;
            const f3 tcen(tp[4], tp[5], tp[6]);
            const double c2p = (double)((fabsf(dot_e(plane_A, c2d) + tp[3]) + fabsf(dot_e(plane1, tcen) + d)) / 2);
            if (c2p > (double)len_th) continue;
            const double dist = (double)norm_e(c2d - tcen);
            if (dist / (double)(sp[7] + tp[7]) > 1) continue;
            ++matched;
            break;
        }
    }
    counts[c] = matched;
}

void plane_consistency(plade_ctx *ctx, CandidateSet &cs, const PlaneGeomHost &src, const PlaneGeomHost &tgt,
                       const float src_bcenter[3], const float tgt_bcenter[3], float max_radius, float cos_angle_th,
                       float length_threshold) {
    const uint32_t n = cs.n_clusters;
    cs.plane_counts.ensure((size_t)n + 1);
    if (!n) return;
    PLADE_REQUIRE(src.P <= 2048 && tgt.P <= 2048, PLADE_ELIMIT, "plane_consistency: too many planes for LDS tables");
    std::vector<float> tab(8 * ((size_t)src.P + tgt.P));
    auto fill = [&](const PlaneGeomHost &g, float *o) {
        for (uint32_t i = 0; i < g.P; ++i) {
            for (int k = 0; k < 4; ++k) o[8 * i + k] = g.coef[4 * i + k];
            for (int k = 0; k < 3; ++k) o[8 * i + 4 + k] = g.center[3 * i + k];
            o[8 * i + 7] = g.radius[i];
        }
    };
    fill(src, tab.data());
    fill(tgt, tab.data() + 8 * (size_t)src.P);
    float *d_tab = reinterpret_cast<float *>(ctx->scratch[4].ensure(tab.size() * 4 + 16));
    const bool staged = ctx->h2d(d_tab, tab.data(), tab.size() * 4);
    const size_t shmem = 32 * ((size_t)src.P + tgt.P);
    launch<k_plane_consistency, 256>(ctx, dim3(cdiv(n, 256)), shmem, cs.rt.p, cs.seeds.p, n, d_tab,
                       src.P, d_tab + 8 * (size_t)src.P, tgt.P, f3(src_bcenter[0], src_bcenter[1], src_bcenter[2]),
                       f3(tgt_bcenter[0], tgt_bcenter[1], tgt_bcenter[2]), max_radius, cos_angle_th, length_threshold,
                       cs.plane_counts.p);
    HIP_TRY(hipGetLastError());
    if (!staged) ctx->sync();  // `tab` must outlive the copy
}

}  // namespace plade

using namespace plade;

// ---- C ABI: seam of the clustering stage ---------------------------------------------------------
// ClusterTransformation (code/PLADE/util.cpp:1245-1277): the candidates come as translations + Euler angles (what
// PointXYZINormal carries there); the stage's own kernels (cluster_transforms above) run on them.
extern "C" int plade_cluster_transforms(plade_ctx *ctx, const float *t_xyz, const float *euler, uint32_t m, float dist_threshold,
                                        float angle_gate, int32_t *cluster_of, uint32_t *n_clusters) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(n_clusters && (m == 0 || (t_xyz && euler && cluster_of)), PLADE_EINVAL, "plade_cluster_transforms: null argument");
        *n_clusters = 0;
        if (m == 0) return PLADE_OK;
        CandidateSet cs;
        cs.m = m;
        cs.rt.ensure(4 * (size_t)m);
        std::vector<float4> rt(4 * (size_t)m);
        const uint32_t nbp = cdiv(m, 256);
        std::vector<float> part(6 * (size_t)nbp);
        for (uint32_t b = 0; b < nbp; ++b)
            for (int k = 0; k < 3; ++k) { part[6 * (size_t)b + k] = INFINITY; part[6 * (size_t)b + 3 + k] = -INFINITY; }
        for (uint32_t i = 0; i < m; ++i) {
            const float *t = t_xyz + 3 * (size_t)i, *e = euler + 3 * (size_t)i;
            rt[4 * (size_t)i] = make_float4(1.f, 0.f, 0.f, t[0]);
            rt[4 * (size_t)i + 1] = make_float4(0.f, 1.f, 0.f, t[1]);
            rt[4 * (size_t)i + 2] = make_float4(0.f, 0.f, 1.f, t[2]);
            rt[4 * (size_t)i + 3] = make_float4(e[0], e[1], e[2], 0.f);
            for (int k = 0; k < 3; ++k) {
                part[6 * (size_t)(i / 256) + k] = std::min(part[6 * (size_t)(i / 256) + k], t[k]);
                part[6 * (size_t)(i / 256) + 3 + k] = std::max(part[6 * (size_t)(i / 256) + 3 + k], t[k]);
            }
        }
        cs.t_minmax.ensure(6 * (size_t)nbp);
        HIP_TRY(hipMemcpyAsync(cs.rt.p, rt.data(), 64 * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(cs.t_minmax.p, part.data(), 24 * (size_t)nbp, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        cluster_transforms(ctx, cs, dist_threshold, angle_gate);
        std::vector<uint32_t> parent(m), seeds(cs.n_clusters);
        HIP_TRY(hipMemcpyAsync(parent.data(), cs.parent.p, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
        if (cs.n_clusters) HIP_TRY(hipMemcpyAsync(seeds.data(), cs.seeds.p, 4 * (size_t)cs.n_clusters, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < m; ++i) {
            uint32_t r = i;
            while (parent[r] != r) r = parent[r];
            // the root of a component is its smallest member = the seed PCL's region growing starts it from
            cluster_of[i] = (int32_t)(std::lower_bound(seeds.begin(), seeds.end(), r) - seeds.begin());
        }
        *n_clusters = cs.n_clusters;
        return PLADE_OK;
    });
}

