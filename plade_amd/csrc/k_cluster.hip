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

__global__ __launch_bounds__(256) void k_transforms(const float *__restrict__ q_lv1, const float *__restrict__ q_lv2,
                                                    const float *__restrict__ q_p1, const float *__restrict__ t_lv1,
                                                    const float *__restrict__ t_lv2, const float *__restrict__ t_p1,
                                                    const uint32_t *__restrict__ q_idx, const uint32_t *__restrict__ t_idx,
                                                    uint32_t m, float4 *__restrict__ rt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
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
}

void build_transforms(plade_ctx *ctx, const PairTableDev &src, const PairTableDev &tgt, const uint32_t *d_q_idx,
                      const uint32_t *d_t_idx, uint32_t m, CandidateSet &cs) {
    cs.m = m;
    cs.rt.ensure(4 * (size_t)m + 4);
    if (!m) return;
    hipLaunchKernelGGL(k_transforms, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, src.lv1.p, src.lv2.p, src.p1.p,
                       tgt.lv1.p, tgt.lv2.p, tgt.p1.p, d_q_idx, d_t_idx, m, cs.rt.p);
    HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
struct HashGrid {
    float mnx, mny, mnz, inv;
};

__device__ __forceinline__ uint64_t cell_key(int cx, int cy, int cz) {
    return ((uint64_t)(uint32_t)cz << 42) | ((uint64_t)(uint32_t)cy << 21) | (uint64_t)(uint32_t)cx;
}

__global__ __launch_bounds__(256) void k_t_minmax(const float4 *__restrict__ rt, uint32_t m, int *__restrict__ out6) {
    __shared__ float s_lds[6][8];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const float v[3] = {rt[4 * (size_t)i].w, rt[4 * (size_t)i + 1].w, rt[4 * (size_t)i + 2].w};
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
    }
    block_minmax_commit<3>(mn, mx, out6, s_lds);
}

__global__ void k_t_keys(const float4 *__restrict__ rt, uint32_t m, HashGrid g, uint64_t *__restrict__ keys,
                         uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int cx = (int)floorf((rt[4 * (size_t)i].w - g.mnx) * g.inv) + 1;
    const int cy = (int)floorf((rt[4 * (size_t)i + 1].w - g.mny) * g.inv) + 1;
    const int cz = (int)floorf((rt[4 * (size_t)i + 2].w - g.mnz) * g.inv) + 1;
    keys[i] = cell_key(cx, cy, cz);
    vals[i] = i;
}

__global__ void k_cell_heads(const uint64_t *__restrict__ keys, uint32_t m, uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_cell_unique(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ flags,
                              const uint32_t *__restrict__ pos, uint32_t m, uint64_t *__restrict__ ukeys,
                              uint32_t *__restrict__ ustart) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && flags[i]) { ukeys[pos[i]] = keys[i]; ustart[pos[i]] = i; }
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

// one lane per node (in sorted-cell order so a wave walks neighbouring cells together)
__global__ __launch_bounds__(256) void k_cluster_edges(const float4 *__restrict__ rt, const uint32_t *__restrict__ order,
                                                       uint32_t m, const uint64_t *__restrict__ ukeys,
                                                       const uint32_t *__restrict__ ustart, uint32_t ncell,
                                                       const uint64_t *__restrict__ skeys, float r2, float gate,
                                                       uint32_t *__restrict__ parent) {
    const uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si >= m) return;
    const uint32_t a = order[si];
    const f3 ta(rt[4 * (size_t)a].w, rt[4 * (size_t)a + 1].w, rt[4 * (size_t)a + 2].w);
    const float4 ea = rt[4 * (size_t)a + 3];
    const uint64_t key = skeys[si];
    const int cx = (int)(key & 0x1fffff), cy = (int)((key >> 21) & 0x1fffff), cz = (int)(key >> 42);
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy) {
            const uint64_t k0 = cell_key(cx - 1, cy + dy, cz + dz), k1 = cell_key(cx + 1, cy + dy, cz + dz);
            uint32_t c = lower_bound_u64(ukeys, ncell, k0);
            for (; c < ncell && ukeys[c] <= k1; ++c) {
                const uint32_t b0 = ustart[c], b1 = (c + 1 < ncell) ? ustart[c + 1] : m;
                for (uint32_t j = b0; j < b1; ++j) {
                    const uint32_t b = order[j];
                    if (b >= a) continue;  // every undirected edge once
                    const f3 tb(rt[4 * (size_t)b].w, rt[4 * (size_t)b + 1].w, rt[4 * (size_t)b + 2].w);
                    if (!(flann_d2(ta, tb) < r2)) continue;
                    const float4 eb = rt[4 * (size_t)b + 3];
                    const float t0 = ea.x - eb.x, t1 = ea.y - eb.y, t2 = ea.z - eb.z;
                    const float sq = (t0 * t0 + t1 * t1) + t2 * t2;  // Eigen::VectorXf(3).squaredNorm()
                    if (sq < gate) uf_union(parent, a, b);
                }
            }
        }
}

__global__ void k_iota(uint32_t *p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
__global__ void k_flatten(uint32_t *__restrict__ parent, uint32_t n, uint32_t *__restrict__ sizes,
                          uint32_t *__restrict__ root_flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = i;
    while (parent[r] != r) r = parent[r];
    atomicAdd(&sizes[r], 1u);
    root_flags[i] = (r == i) ? 1u : 0u;
}
__global__ void k_gather_sizes(const uint32_t *__restrict__ seeds, uint32_t n, const uint32_t *__restrict__ sizes_all,
                               uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sizes_all[seeds[i]];
}

void cluster_transforms(plade_ctx *ctx, CandidateSet &cs, float dist_threshold, float angle_gate) {
    const uint32_t m = cs.m;
    cs.n_clusters = 0;
    if (!m) return;
    // bbox of the translations
    int init[6];
    {
        float pinf = INFINITY, ninf = -INFINITY;
        int a, b;
        memcpy(&a, &pinf, 4); memcpy(&b, &ninf, 4);
        for (int k = 0; k < 3; ++k) { init[k] = a; init[3 + k] = b ^ 0x7fffffff; }
    }
    int *d6 = reinterpret_cast<int *>(ctx->scratch[3].ensure(64));
    HIP_TRY(hipMemcpyAsync(d6, init, 24, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_t_minmax, dim3(std::min(cdiv(m, 1024), 512u)), dim3(256), 0, ctx->stream, cs.rt.p, m, d6);
    int out[6];
    HIP_TRY(hipMemcpyAsync(out, d6, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float mn[3], mx[3];
    for (int k = 0; k < 6; ++k) {
        int v = out[k] >= 0 ? out[k] : out[k] ^ 0x7fffffff;
        float f;
        memcpy(&f, &v, 4);
        if (k < 3) mn[k] = f; else mx[k - 3] = f;
    }
    const float r2 = pcl_r2((double)dist_threshold);  // setClusterTolerance -> radiusSearch(double) -> float(r*r)
    float cell = dist_threshold * 1.001f;
    if (!(cell > 0.f)) cell = 1.f;
    for (;;) {  // 21-bit cell coordinates (+1 offset, +1 probe margin)
        double ext = std::max({(double)mx[0] - mn[0], (double)mx[1] - mn[1], (double)mx[2] - mn[2]});
        if (ext / cell + 4 < (double)(1 << 21)) break;
        cell *= 2.f;
    }
    HashGrid g{mn[0], mn[1], mn[2], 1.f / cell};
    cs.ckeys.ensure(m); cs.ckeys2.ensure(m); cs.cvals.ensure(m); cs.cvals2.ensure(m);
    cs.cflags.ensure((size_t)m + 1); cs.cpos.ensure((size_t)m + 1);
    const unsigned nb = cdiv(m, 256);
    hipLaunchKernelGGL(k_t_keys, dim3(nb), dim3(256), 0, ctx->stream, cs.rt.p, m, g, cs.ckeys.p, cs.cvals.p);
    sort_pairs_u64(ctx, cs.ckeys.p, cs.ckeys2.p, cs.cvals.p, cs.cvals2.p, m, 63);
    hipLaunchKernelGGL(k_cell_heads, dim3(nb), dim3(256), 0, ctx->stream, cs.ckeys2.p, m, cs.cflags.p);
    HIP_TRY(hipMemsetAsync(cs.cflags.p + m, 0, 4, ctx->stream));
    exclusive_scan_u32(ctx, cs.cflags.p, cs.cpos.p, (size_t)m + 1);
    uint32_t ncell = 0;
    HIP_TRY(hipMemcpyAsync(&ncell, cs.cpos.p + m, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    cs.ucell_keys.ensure((size_t)ncell + 1); cs.ucell_start.ensure((size_t)ncell + 1);
    hipLaunchKernelGGL(k_cell_unique, dim3(nb), dim3(256), 0, ctx->stream, cs.ckeys2.p, cs.cflags.p, cs.cpos.p, m,
                       cs.ucell_keys.p, cs.ucell_start.p);
    cs.parent.ensure(m); cs.sizes_all.ensure(m); cs.flags.ensure((size_t)m + 1);
    hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, ctx->stream, cs.parent.p, m);
    hipLaunchKernelGGL(k_cluster_edges, dim3(nb), dim3(256), 0, ctx->stream, cs.rt.p, cs.cvals2.p, m, cs.ucell_keys.p,
                       cs.ucell_start.p, ncell, cs.ckeys2.p, r2, angle_gate, cs.parent.p);
    HIP_TRY(hipMemsetAsync(cs.sizes_all.p, 0, (size_t)m * 4, ctx->stream));
    hipLaunchKernelGGL(k_flatten, dim3(nb), dim3(256), 0, ctx->stream, cs.parent.p, m, cs.sizes_all.p, cs.flags.p);
    HIP_TRY(hipMemsetAsync(cs.flags.p + m, 0, 4, ctx->stream));
    cs.n_clusters = compact_flags(ctx, cs.flags.p, m, cs.pos, cs.seeds);
    cs.sizes.ensure((size_t)cs.n_clusters + 1);
    if (cs.n_clusters)
        hipLaunchKernelGGL(k_gather_sizes, dim3(cdiv(cs.n_clusters, 256)), dim3(256), 0, ctx->stream, cs.seeds.p,
                           cs.n_clusters, cs.sizes_all.p, cs.sizes.p);
    HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
constexpr int PC_MAXP = 128;

struct PlaneTab {
    float coef[PC_MAXP][4];
    float cen[PC_MAXP][3];
    float rad[PC_MAXP];
};

__global__ __launch_bounds__(256) void k_plane_consistency(const float4 *__restrict__ rt, const uint32_t *__restrict__ seeds,
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
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
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
    HIP_TRY(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    const size_t shmem = 32 * ((size_t)src.P + tgt.P);
    hipLaunchKernelGGL(k_plane_consistency, dim3(cdiv(n, 256)), dim3(256), shmem, ctx->stream, cs.rt.p, cs.seeds.p, n, d_tab,
                       src.P, d_tab + 8 * (size_t)src.P, tgt.P, f3(src_bcenter[0], src_bcenter[1], src_bcenter[2]),
                       f3(tgt_bcenter[0], tgt_bcenter[1], tgt_bcenter[2]), max_radius, cos_angle_th, length_threshold,
                       cs.plane_counts.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // `tab` must outlive the copy
}

}  // namespace plade
