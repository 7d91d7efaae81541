// plade_amd/csrc/ransac.hip -- plane extraction on one MI355X (seam S1b).
//
// Replaces PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:61-200) and the
// RansacShapeDetector::Detect loop it drives (code/3rd_party/ransac/RansacShapeDetector.cpp:455-907).
// The reference is a sequential, time-seeded Efficient-RANSAC over octrees; it cannot reproduce its
// own output run to run, so parity is pinned per kernel (same hypothesis => identical inliers,
// identical connected component, LS plane within the reference's own accumulation noise) and at the
// plane-set level.  This is a GPU-native driver with the same semantics per accepted plane:
//
//   sampling   : DrawSamplesStratified (RansacShapeDetector.cpp:909-954) -- first point anywhere,
//                the other two from the same octree cell at a random level.  Here the cloud is
//                Morton-sorted once and "the cell at level l" is a contiguous key range.
//   hypothesis : Plane::Init(p1,p2,p3) (ransac/Plane.cpp:29-38) + the 3-sample verification
//                (RansacShapeDetector.cpp:143-153).
//   scoring    : K1 (k_score.hip).  Schnabel scores lazily on nested random subsets to save CPU
//                time; on the GPU a batch of H hypotheses is scored on a stratified subset in one
//                launch and the leaders are re-scored on ALL unassigned points in one HBM pass.
//   acceptance : exactly the reference's sequence (RansacShapeDetector.cpp:618-675):
//                GlobalScore(3 eps) -> ConnectedComponent(bitmap eps) -> up to 3 x
//                { LSFit -> GlobalWeightedScore(3 eps) } keeping a refit only if its weighted score
//                and size improve -> points marked assigned, drawnCandidates scaled by (1-|S|/n)^3.
//   stopping   : CandidateFailureProbability(minSupport, n_remaining, drawn, levels) <= p
//                (RansacShapeDetector.h:61-67, .cpp:856-858).
#include "ransac.h"
#include <condition_variable>
#include <mutex>
#include "score.h"
#include "prims.h"
#include "voxel.h"
#include <algorithm>
#include <map>
#include <memory>

namespace plade {

// ------------------------------------------------------------------------------------------------
// Morton order
// 8 bits per axis = the 8 octree levels the sampler draws from (24-bit keys: three radix passes)
__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // up to 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FF;
    v = (v | (v << 8)) & 0x0300F00F;
    v = (v | (v << 4)) & 0x030C30C3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

__global__ void k_morton(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                         uint32_t n, float mnx, float mny, float mnz, float inv_cube, uint32_t *__restrict__ keys,
                         uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t qx = min(255u, (uint32_t)max(0.f, (x[i] - mnx) * inv_cube * 256.f));
    const uint32_t qy = min(255u, (uint32_t)max(0.f, (y[i] - mny) * inv_cube * 256.f));
    const uint32_t qz = min(255u, (uint32_t)max(0.f, (z[i] - mnz) * inv_cube * 256.f));
    keys[i] = (spread3(qz) << 2) | (spread3(qy) << 1) | spread3(qx);
    vals[i] = i;
}

// Morton-order gather from the AoS copy (24 contiguous bytes per point: one or two 64 B sectors per
// point instead of six scattered 4 B reads from the SoA planes)
__global__ void k_gather_cloud(const float *__restrict__ aos, const uint32_t *__restrict__ perm, uint32_t n,
                               float *__restrict__ dst, size_t dpitch) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 *p = reinterpret_cast<const float2 *>(aos + 6 * (size_t)perm[i]);
    const float2 a = p[0], b = p[1], c = p[2];
    dst[i] = a.x; dst[dpitch + i] = a.y; dst[2 * dpitch + i] = b.x;
    dst[3 * dpitch + i] = b.y; dst[4 * dpitch + i] = c.x; dst[5 * dpitch + i] = c.y;
}

__global__ void k_make_subset(const float *__restrict__ src, size_t spitch, uint32_t n, uint32_t stride, uint32_t n_sub,
                              float *__restrict__ dst, size_t dpitch, uint32_t *__restrict__ sub_index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sub) return;
    const uint32_t p = min(n - 1, i * stride);
    sub_index[i] = p;
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[k * dpitch + i] = src[k * spitch + p];
}

// ------------------------------------------------------------------------------------------------
// hypothesis sampling
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s;
    __device__ uint32_t next() { s = mix64(s); return (uint32_t)(s >> 32); }
};

__device__ __forceinline__ uint32_t lb_u32(const uint32_t *a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

struct CloudView {
    const float *x, *y, *z, *nx, *ny, *nz;
    uint32_t n;
};

__global__ __launch_bounds__(256) void k_sample(CloudView c, const uint32_t *__restrict__ codes,
                                                const int32_t *__restrict__ assigned, uint64_t seed, uint32_t round,
                                                uint32_t h, int min_level, int max_level, float eps, float cos_t,
                                                float4 *__restrict__ hyp, float4 *__restrict__ hyp_pos,
                                                uint32_t *__restrict__ counts, uint32_t *__restrict__ misc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= h) return;
    Rng rng{mix64(seed ^ ((uint64_t)round << 32) ^ t)};
    const float nanv = __int_as_float(0x7fc00000);
    hyp[t] = make_float4(0.f, 0.f, 0.f, nanv);
    hyp_pos[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    counts[t] = 0;                 // the scoring pass that follows accumulates with atomics
    if (t == 0) *misc = 0;         // and so does the unassigned-point count of the subset
    // draws are made eight at a time so that their shapeIndex look-ups are in flight together (late rounds
    // have few unassigned points left and most draws miss)
    uint32_t i0 = 0;
    bool ok = false;
    for (int tr = 0; tr < 64 && !ok; tr += 8) {
        uint32_t cand[8];
        int32_t av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cand[u] = rng.next() % c.n;
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = assigned[cand[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (!ok && av[u] == -1) { i0 = cand[u]; ok = true; }
    }
    if (!ok) return;
    const int level = min_level + (int)(rng.next() % (uint32_t)(max_level - min_level + 1));
    const uint32_t low_bits = 24 - 3 * level;
    const uint32_t mask = low_bits >= 32 ? 0u : ~((1u << low_bits) - 1u);
    const uint32_t lo_key = codes[i0] & mask, hi_key = lo_key | ~mask;
    const uint32_t lo = lb_u32(codes, c.n, lo_key);
    uint32_t hi = (hi_key == 0xffffffffu || hi_key >= 0xffffffu) ? c.n : lb_u32(codes, c.n, hi_key + 1u);
    if (hi - lo < 3) return;
    uint32_t s[3] = {i0, 0, 0};
    int got = 1;
    for (int tr = 0; tr < 40 && got < 3; tr += 8) {
        uint32_t cand[8];
        int32_t av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cand[u] = lo + rng.next() % (hi - lo);
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = assigned[cand[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (got >= 3 || av[u] != -1) continue;
            bool dup = false;
            for (int q = 0; q < got; ++q) dup = dup || (s[q] == cand[u]);
            if (!dup) s[got++] = cand[u];
        }
    }
    if (got < 3) return;
    // three samples drawn: this counts as a generated candidate (genCands, RansacShapeDetector.cpp:122-125)
    // whether or not the plane survives construction / verification below
    hyp_pos[t] = make_float4(0.f, 0.f, 0.f, 2.f);
    // Plane::Init (ransac/Plane.cpp:29-38)
    const float p1[3] = {c.x[s[0]], c.y[s[0]], c.z[s[0]]}, p2[3] = {c.x[s[1]], c.y[s[1]], c.z[s[1]]},
                p3[3] = {c.x[s[2]], c.y[s[2]], c.z[s[2]]};
    const float a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, b[3] = {p3[0] - p2[0], p3[1] - p2[1], p3[2] - p2[2]};
    float nr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    float sq = nr[0] * nr[0];
    sq += nr[1] * nr[1];
    sq += nr[2] * nr[2];
    if (sq < 1E-6f) return;
    const float len = sqrtf(sq);
    nr[0] /= len; nr[1] /= len; nr[2] /= len;
    float dist = p1[0] * nr[0];
    dist += p1[1] * nr[1];
    dist += p1[2] * nr[2];
    // verify the three samples (RansacShapeDetector.cpp:143-153)
    for (int k = 0; k < 3; ++k) {
        float d = nr[0] * c.x[s[k]];
        d += nr[1] * c.y[s[k]];
        d += nr[2] * c.z[s[k]];
        float nd = nr[0] * c.nx[s[k]];
        nd += nr[1] * c.ny[s[k]];
        nd += nr[2] * c.nz[s[k]];
        if (!(fabsf(dist - d) < eps && fabsf(nd) >= cos_t)) return;
    }
    hyp[t] = make_float4(nr[0], nr[1], nr[2], dist);
    hyp_pos[t] = make_float4(p1[0], p1[1], p1[2], 1.f);
}

__global__ void k_count_unassigned(const int32_t *__restrict__ assigned, const uint32_t *__restrict__ sub_index,
                                   uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool un = false;
    if (i < n) un = assigned[sub_index ? sub_index[i] : i] == -1;
    const uint32_t c = (uint32_t)__popcll(__ballot(un));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ------------------------------------------------------------------------------------------------
// plane state on the device: everything the acceptance sequence needs without a host round trip
struct PlaneState {
    float n[3], pos[3], dist;        // Plane (ransac/Plane.h: m_normal, m_pos, m_dist)
    float a0[3], a1[3];              // HyperplaneCoordinateSystem axes (GfxTL/HyperplaneCoordinateSystem.h:81-93)
    int bb[4];                       // ordered-int (min u, min v, max u, max v)
    uint32_t ue, ve;
    uint32_t best_root, n_fg;
    double wscore;
    float nsum[3];                   // sum of inlier point normals (orientation)
    uint32_t err;
    uint32_t converged;              // refit chain: this slot's plane is bitwise the previous slot's
};

constexpr int CHAIN_MAX = BATCH_MAXJ;   // chains of a launch: 8 per cloud, two clouds (table passed by value in the kernarg)

// Device-visible buffers of one acceptance chain.  Every kernel of the acceptance sequence takes the
// table of chains and handles chain blockIdx.y (slot k of it): one launch serves the whole batch of
// candidates that are accepted together.
struct ChainDev {
    PlaneState *st;        // 4 slots: candidate + 3 refits
    uint32_t *cntS;        // per-slot result counts
    float *nsum;           // 4 x 3
    float4 *top;           // hypothesis (n, dist), position
    float4 *plane_cur;     // 4
    uint32_t *idxA, *cntA; // score(3 eps) list before the connected component
    uint32_t *idxS[4];     // per-slot result lists
    float2 *uv;
    uint32_t *bidx, *label, *sizes;
    uint8_t *bmp, *tmp;
    uint8_t *masks2;       // connected-component selection masks
    uint32_t *bc2;
    const uint32_t *bcA;   // per-tile counts of the score list
    float *bbpart;         // per-tile (u, v) bounding boxes of the score list
    double *part;
};
// A batch may hold the chains of two clouds (the two scans of a pair are extracted in lock-step): chains
// [0, split) belong to group 0, the rest to group 1.
struct ChainGroup { CloudView cloud; float eps3, bitmap_eps; uint32_t nb4; };
struct ChainTab { ChainDev c[CHAIN_MAX]; ChainGroup g[2]; uint32_t split; };


__device__ __forceinline__ void hcs_axes(const float *n, float *a0, float *a1) {
    float t[3];
    if (fabsf(n[0]) < 0.015625f && fabsf(n[1]) < 0.015625f) {  // (0,1,0) x n
        t[0] = 1.f * n[2] - 0.f * n[1]; t[1] = 0.f * n[0] - 0.f * n[2]; t[2] = 0.f * n[1] - 1.f * n[0];
    } else {                                                      // (0,0,1) x n
        t[0] = 0.f * n[2] - 1.f * n[1]; t[1] = 1.f * n[0] - 0.f * n[2]; t[2] = 0.f * n[1] - 0.f * n[0];
    }
    float l = t[0] * t[0];
    l += t[1] * t[1];
    l += t[2] * t[2];
    l = sqrtf(l);
    a0[0] = t[0] / l; a0[1] = t[1] / l; a0[2] = t[2] / l;
    float u[3] = {n[1] * a0[2] - n[2] * a0[1], n[2] * a0[0] - n[0] * a0[2], n[0] * a0[1] - n[1] * a0[0]};
    l = u[0] * u[0];
    l += u[1] * u[1];
    l += u[2] * u[2];
    l = sqrtf(l);
    a1[0] = u[0] / l; a1[1] = u[1] / l; a1[2] = u[2] / l;
}

__device__ __forceinline__ int ord_i(float f) { int v = __float_as_int(f); return v >= 0 ? v : v ^ 0x7fffffff; }
__device__ __forceinline__ float ord_f(int v) { return __int_as_float(v >= 0 ? v : v ^ 0x7fffffff); }

// initialise the state from a hypothesis (n, dist) + position
__global__ void k_state_from_hyp(const ChainTab chains) {
    if (threadIdx.x) return;
    const ChainDev &C = chains.c[blockIdx.x];
    const float4 *hyp = C.top, *pos = C.top + 1;
    PlaneState *st = C.st;
    float4 *plane_out = C.plane_cur;
    st->n[0] = hyp->x; st->n[1] = hyp->y; st->n[2] = hyp->z; st->dist = hyp->w;
    st->pos[0] = pos->x; st->pos[1] = pos->y; st->pos[2] = pos->z;
    hcs_axes(st->n, st->a0, st->a1);
    st->err = 0;
    st->converged = 0;
    st->bb[0] = st->bb[1] = ord_i(INFINITY);
    st->bb[2] = st->bb[3] = ord_i(-INFINITY);
    *plane_out = *hyp;
}

constexpr uint32_t CC_MAXPIX = 1u << 20;

// BitmapExtent (PlanePrimitiveShape.cpp:185-191)
__device__ __forceinline__ bool cc_dims(const float bb[4], uint32_t count, float eps, uint32_t &ue, uint32_t &ve) {
    ue = 2; ve = 2;
    if (count) {
        const float mnu = bb[0], mnv = bb[1], mxu = bb[2], mxv = bb[3];
        const float fu = ceilf((mxu - mnu) / eps), fv = ceilf((mxv - mnv) / eps);
        ue = (fu < 4.0e6f ? (uint32_t)fu : 4000000u) + 1;
        ve = (fv < 4.0e6f ? (uint32_t)fv : 4000000u) + 1;
        if (ue < 2) ue = 2;
        if (ve < 2) ve = 2;
    }
    if ((uint64_t)ue * ve > CC_MAXPIX) { ue = ve = 2; return false; }
    return true;
}

// BuildBitmap (BitmapPrimitiveShape.h:139-150) with InBitmap (PlanePrimitiveShape.cpp:193-199).
// The bitmap is all-zero on entry (k_cc_label clears what it used).
__global__ __launch_bounds__(256) void k_cc_raster(const ChainTab chains, int k) {
    const ChainDev &C = chains.c[blockIdx.y];
    const ChainGroup &G = chains.g[blockIdx.y >= chains.split ? 1 : 0];
    const float eps = G.bitmap_eps;
    const uint32_t n_tiles = G.nb4;
    PlaneState *st = C.st + k;
    if (st->converged) return;
    const float2 *__restrict__ uv = C.uv;
    const uint32_t *__restrict__ count = C.cntA;
    uint32_t *__restrict__ bidx = C.bidx;
    uint8_t *__restrict__ bmp = C.bmp;
    const uint32_t m = *count;
    // bounding box of the list's (u, v) parameters (BitmapPrimitiveShape.h:113-126): every block reduces the
    // per-tile boxes the compaction left behind (a few thousand floats from L2)
    __shared__ float s_bb[4][4];
    float bbv[4] = {INFINITY, INFINITY, -INFINITY, -INFINITY};
    for (uint32_t b = threadIdx.x; b < n_tiles; b += blockDim.x)
        if (C.bcA[b]) {
            const float4 t = *reinterpret_cast<const float4 *>(C.bbpart + 4 * (size_t)b);
            bbv[0] = fminf(bbv[0], t.x); bbv[1] = fminf(bbv[1], t.y); bbv[2] = fmaxf(bbv[2], t.z); bbv[3] = fmaxf(bbv[3], t.w);
        }
    for (int q = 0; q < 4; ++q)
        for (int d = 32; d >= 1; d >>= 1) {
            const float o = __shfl_xor(bbv[q], d, 64);
            bbv[q] = q < 2 ? fminf(bbv[q], o) : fmaxf(bbv[q], o);
        }
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 4; ++q) s_bb[q][threadIdx.x >> 6] = bbv[q];
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        float v = s_bb[q][0];
        for (int w = 1; w < 4; ++w) v = q < 2 ? fminf(v, s_bb[q][w]) : fmaxf(v, s_bb[q][w]);
        bbv[q] = v;
    }
    uint32_t ue, ve;
    const bool ok = cc_dims(bbv, m, eps, ue, ve);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->ue = ue; st->ve = ve;
        if (!ok) st->err = 1;
        for (int q = 0; q < 4; ++q) st->bb[q] = ord_i(bbv[q]);
    }
    if (!ok) return;
    const float mnu = bbv[0], mnv = bbv[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        int bu = (int)floorf((uv[i].x - mnu) / eps), bv = (int)floorf((uv[i].y - mnv) / eps);
        bu = min(max(bu, 0), (int)ue - 1);
        bv = min(max(bv, 0), (int)ve - 1);
        const uint32_t b = (uint32_t)bu + (uint32_t)bv * ue;
        bidx[i] = b;
        bmp[b] = 1;
    }
}

// closing (DilateCross + ErodeCross, ransac/Bitmap.cpp:154-260, 459-570; no wrapping for planes),
// 8-connected labelling (Components, Bitmap.cpp:633-834) and selection of the component with most
// pixels, first in raster order on ties (BitmapPrimitiveShape.cpp:168-173).  One workgroup; bitmaps of
// up to CC_LDS_PIX pixels (every realistic plane at bitmap eps = 2 % of the scene) live in LDS.
constexpr int CC_LDS_PIX = 8192;

// The labelling proper on a bitmap that lives either in LDS or in global memory.  Force-inlined into both branches
// of k_cc_label so that the LDS instance is compiled to ds_* instructions (a pointer selected at run time would
// make every access a flat_* one).
__device__ __forceinline__ void cc_label_body(uint8_t *bmp, uint8_t *tmp, uint32_t *label, uint32_t *sizes, int ue, int ve, int npx,
                                              int do_filter, unsigned long long *s_best_p, PlaneState *st) {
    unsigned long long &s_best = *s_best_p;
    if (do_filter) {
        for (int p = threadIdx.x; p < npx; p += blockDim.x) {
            const int u = p % ue, v = p / ue;
            bool r = bmp[p];
            if (u > 0) r = r || bmp[p - 1];
            if (u < ue - 1) r = r || bmp[p + 1];
            if (v > 0) r = r || bmp[p - ue];
            if (v < ve - 1) r = r || bmp[p + ue];
            tmp[p] = r;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < npx; p += blockDim.x) {
            const int u = p % ue, v = p / ue;
            bool r = tmp[p];
            if (u > 0) r = r && tmp[p - 1];
            if (u < ue - 1) r = r && tmp[p + 1];
            if (v > 0) r = r && tmp[p - ue];
            if (v < ve - 1) r = r && tmp[p + ue];
            bmp[p] = r;
        }
        __syncthreads();
    }
    // 8-connected labelling: (1) every pixel gets the first pixel of its horizontal run as label (one lane
    // per row, sequential along the row), (2) runs are united with the runs they touch in the row above
    // (N, NW, NE) by lock-free union-find on the run heads; the smaller index always becomes the parent, so
    // a component's root is its first pixel in raster order
    {   // one wavefront per row, 64 columns at a time: run head = prefix maximum of the run-start columns
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
        for (int v = wave; v < ve; v += nwaves) {
            int carry = -1;   // head column of the run that reaches the end of the previous 64-column chunk
            for (int u0 = 0; u0 < ue; u0 += 64) {
                const int u = u0 + lane;
                const int p = v * ue + u;
                const bool fg = u < ue && bmp[p];
                const bool left_fg = (u > 0 && u < ue) ? (bool)bmp[p - 1] : false;
                int h = (fg && !left_fg) ? u : -1;   // a run starts here
                if (lane == 0 && fg && left_fg) h = carry;   // the run continues from the previous chunk
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(h, d, 64);
                    if (lane >= d) h = max(h, o);
                }
                if (u < ue) { label[p] = fg ? (uint32_t)(v * ue + h) : 0xffffffffu; sizes[p] = 0; }
                const int last_h = __shfl(h, 63, 64);
                const bool last_fg = __shfl((int)fg, 63, 64) != 0;
                carry = last_fg ? last_h : -1;
            }
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        if (!bmp[p] || p < ue) continue;
        const int u = p % ue;
        const int nb[3] = {u > 0 ? p - ue - 1 : -1, p - ue, u < ue - 1 ? p - ue + 1 : -1};
        for (int e = 0; e < 3; ++e) {
            if (nb[e] < 0 || !bmp[nb[e]]) continue;
            // N also covers NW / NE whenever N is set (same run above): skip the redundant unions
            if (e != 1 && bmp[p - ue]) continue;
            uint32_t a = label[p], b = label[nb[e]];
            for (;;) {
                while (label[a] != a) a = label[a];
                while (label[b] != b) b = label[b];
                if (a == b) break;
                if (a < b) { const uint32_t t = a; a = b; b = t; }   // a > b: hang a under b
                const uint32_t old = atomicMin(&label[a], b);
                if (old == a) break;
                a = old;
            }
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        if (!bmp[p]) continue;
        uint32_t r = label[p];
        while (label[r] != r) r = label[r];
        label[p] = r;   // races only write the final root or an ancestor: harmless
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x)
        if (bmp[p]) atomicAdd(&sizes[label[p]], 1u);
    if (threadIdx.x == 0) s_best = 0ull;
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x)
        if (bmp[p] && label[p] == (uint32_t)p) {
            // max size, then smallest raster-first pixel
            const unsigned long long key = ((unsigned long long)sizes[p] << 32) | (0xffffffffu - (uint32_t)p);
            atomicMax(&s_best, key);
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_best == 0ull) { st->best_root = 0xffffffffu; st->n_fg = 0; }
        else { st->best_root = 0xffffffffu - (uint32_t)(s_best & 0xffffffffu); st->n_fg = (uint32_t)(s_best >> 32); }
    }
}

__global__ __launch_bounds__(1024) void k_cc_label(const ChainTab chains, int k, int do_filter) {
    const ChainDev &C = chains.c[blockIdx.x];
    PlaneState *st = C.st + k;
    uint8_t *__restrict__ g_bmp = C.bmp, *__restrict__ g_tmp = C.tmp;
    uint32_t *__restrict__ g_label = C.label, *__restrict__ g_sizes = C.sizes;
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_label[CC_LDS_PIX];
    __shared__ uint32_t s_sizes[CC_LDS_PIX];
    __shared__ uint8_t s_bmp[CC_LDS_PIX], s_tmp[CC_LDS_PIX];
    if (st->converged) return;
    const int ue = (int)st->ue, ve = (int)st->ve, npx = ue * ve;
    if (npx <= CC_LDS_PIX) {
        for (int p = threadIdx.x; p < npx; p += blockDim.x) { s_bmp[p] = g_bmp[p]; g_bmp[p] = 0; }  // also leaves it clean
        __syncthreads();
        cc_label_body(s_bmp, s_tmp, s_label, s_sizes, ue, ve, npx, do_filter, &s_best, st);
        // k_cc_select reads the labels from global memory; the bitmap is left all-zero for the next raster
        for (int p = threadIdx.x; p < npx; p += blockDim.x) g_label[p] = s_label[p];
    } else {
        cc_label_body(g_bmp, g_tmp, g_label, g_sizes, ue, ve, npx, do_filter, &s_best, st);
        __syncthreads();
        for (int p = threadIdx.x; p < npx; p += blockDim.x) g_bmp[p] = 0;
    }
}

// mask layout of k_compact: one byte per lane covering 4 consecutive items, block counts per 1024.
// The same pass accumulates, over the kept points, the LS-fit moments (12 sums) and Candidate::WeightedScore
// (ransac/Candidate.cpp:77-87 with weigh(), ScoreComputer.h:10-16) of the slot's plane: one row of FIT_COLS
// doubles per 1024 list positions, reduced by k_fit_final in a fixed order.
constexpr int FIT_COLS = 13;

__global__ __launch_bounds__(256) void k_cc_select(const ChainTab chains, int k) {
    __shared__ uint32_t s_w[4];
    __shared__ double s[4][FIT_COLS];
    const ChainDev &C = chains.c[blockIdx.y];
    const ChainGroup &G = chains.g[blockIdx.y >= chains.split ? 1 : 0];
    if (blockIdx.x >= G.nb4) return;   // the grid covers the larger cloud of the batch
    const CloudView &c = G.cloud;
    const float eps = G.eps3;
    const PlaneState *st = C.st + k;
    if (st->converged) return;
    const uint32_t *__restrict__ bidx = C.bidx, *__restrict__ count = C.cntA, *__restrict__ label = C.label;
    const uint32_t *__restrict__ idx = C.idxA;
    uint8_t *__restrict__ masks = C.masks2;
    uint32_t *__restrict__ block_counts = C.bc2;
    const uint32_t m = *count, best = st->best_root;
    if (blockIdx.x * 1024u >= m) {   // past the list: no kept points, no partial row (k_fit_final reads ceil(m / 1024) rows)
        masks[blockIdx.x * 256 + threadIdx.x] = 0;
        if (threadIdx.x == 0) block_counts[blockIdx.x] = 0;
        return;
    }
    const float n0 = st->n[0], n1 = st->n[1], n2 = st->n[2], dist = st->dist;
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t mk = 0, cnt = 0;
    double a[FIT_COLS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // two rounds of independent loads (list entries, then everything they point to) instead of a chain of four:
    // nearly every listed point is kept, so the coordinates are fetched before the label test is known
    uint32_t bi[4], pi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t i = min(base + q, m - 1);
        bi[q] = bidx[i];
        pi[q] = idx[i];
    }
    uint32_t lb[4];
    float fx[4], fy[4], fz[4], gx[4], gy[4], gz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        lb[q] = label[bi[q]];
        const uint32_t p = pi[q];
        fx[q] = c.x[p]; fy[q] = c.y[p]; fz[q] = c.z[p];
        gx[q] = c.nx[p]; gy[q] = c.ny[p]; gz[q] = c.nz[p];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool in = base + q < m && best != 0xffffffffu && lb[q] == best;
        mk |= (in ? 1u : 0u) << q;
        cnt += (uint32_t)__popcll(__ballot(in));
        if (in) {
            const double x = fx[q], y = fy[q], z = fz[q];
            a[0] += x; a[1] += y; a[2] += z;
            a[3] += x * x; a[4] += x * y; a[5] += x * z; a[6] += y * y; a[7] += y * z; a[8] += z * z;
            a[9] += gx[q]; a[10] += gy[q]; a[11] += gz[q];
            float d = n0 * fx[q];
            d += n1 * fy[q];
            d += n2 * fz[q];
            d = fabsf(dist - d);
            a[12] += (double)expf(-d * d / (2.f / 9.f * eps * eps));
        }
    }
    masks[blockIdx.x * 256 + threadIdx.x] = (uint8_t)mk;
    for (int q = 0; q < FIT_COLS; ++q)
        for (int d = 32; d >= 1; d >>= 1) a[q] += __shfl_xor(a[q], d, 64);
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = cnt;
        for (int q = 0; q < FIT_COLS; ++q) s[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (threadIdx.x < FIT_COLS)
        C.part[(size_t)blockIdx.x * FIT_COLS + threadIdx.x] = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// LS refit (PlanePrimitiveShape::LSFit -> Plane::LeastSquaresFit, ransac/Plane.cpp:169-176,
// Plane.h:65-74: mean + covariance about the mean + smallest-|eigenvalue| eigenvector).
// Accumulated in fp64 with a fixed reduction tree (deterministic); the reference accumulates in
// fp32 sequentially, which is the noisier of the two.
__device__ void jacobi3_d(double a[3][3], double d[3], double v[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = i == j;
    const double scale = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off <= 1e-24 * scale) break;   // far below fp64 resolution of the eigenvectors (quadratic convergence)
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
                const double t = (theta >= 0 ? 1 : -1) / (fabs(theta) + sqrt(theta * theta + 1));
                const double cs = 1 / sqrt(t * t + 1), sn = t * cs;
                for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = cs * vkp - sn * vkq; v[k][q] = sn * vkp + cs * vkq; }
            }
    }
    for (int i = 0; i < 3; ++i) d[i] = a[i][i];
}

// after the last slot: slots that were skipped because the chain had converged repeat their predecessor's results
__device__ void wscore_final(const ChainDev &C) {
    PlaneState *st = C.st;
    uint32_t *__restrict__ cnt = C.cntS;
    if (threadIdx.x == 0)
        for (int k = 1; k < 4; ++k)
            if (st[k].converged) { st[k].wscore = st[k - 1].wscore; cnt[k] = cnt[k - 1]; st[k].ue = st[k - 1].ue; st[k].ve = st[k - 1].ve; }
}

// Sums the per-block partials of the index list `count` belongs to (fixed tree => deterministic);
// nsum_out receives the sum of the list's point normals (orientation).  mode 0 additionally writes the
// fitted plane into `st` (the NEXT slot's state) / plane_out.  One workgroup of 256 lanes.
// `cur` is the state of the slot whose list was just reduced; when the new plane is bitwise equal to
// cur's plane the chain has converged: every later slot would reproduce cur's results, so they are
// flagged and their kernels return immediately.
__device__ void fit_final(const ChainDev &C, int k, double (*s_red)[FIT_COLS]) {
    const int kn = k < 3 ? k + 1 : 3, mode = k < 3 ? 0 : 1;
    const double *__restrict__ part = C.part;
    const uint32_t *__restrict__ count = C.cntS + k;
    const PlaneState *cur = C.st + k;
    PlaneState *st = C.st + kn;
    float4 *plane_out = C.plane_cur + kn;
    float *__restrict__ nsum_out = C.nsum + 3 * k;
    if (cur->converged) {
        if (threadIdx.x == 0) {
            nsum_out[0] = nsum_out[-3]; nsum_out[1] = nsum_out[-2]; nsum_out[2] = nsum_out[-1];
            if (mode == 0) { st->converged = 1; st->err = 0; *plane_out = plane_out[-1]; }
        }
        return;
    }
    // rows written by k_cc_select: one per 1024 positions of the score list; lane t adds rows t, t + 256, ... in
    // order, then the fixed shuffle / wave tree
    const uint32_t rows = (*C.cntA + 1023u) / 1024u;
    double a[FIT_COLS];
    for (int q = 0; q < FIT_COLS; ++q) a[q] = 0.0;
    for (uint32_t r = threadIdx.x; r < rows; r += blockDim.x)
        for (int q = 0; q < FIT_COLS; ++q) a[q] += part[(size_t)r * FIT_COLS + q];
    for (int q = 0; q < FIT_COLS; ++q)
        for (int d = 32; d >= 1; d >>= 1) a[q] += __shfl_xor(a[q], d, 64);
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < FIT_COLS; ++q) s_red[threadIdx.x >> 6][q] = a[q];
    __syncthreads();
    if (threadIdx.x) return;
    for (int q = 0; q < FIT_COLS; ++q) a[q] = (s_red[0][q] + s_red[1][q]) + (s_red[2][q] + s_red[3][q]);
    C.st[k].wscore = a[12];
    nsum_out[0] = (float)a[9]; nsum_out[1] = (float)a[10]; nsum_out[2] = (float)a[11];
    if (mode == 1) return;
    st->err = 0;
    st->bb[0] = st->bb[1] = ord_i(INFINITY);
    st->bb[2] = st->bb[3] = ord_i(-INFINITY);
    const double m = (double)*count;
    if (m < 3) { st->err = 2; *plane_out = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fc00000)); return; }
    const double mx = a[0] / m, my = a[1] / m, mz = a[2] / m;
    double cv[3][3];
    cv[0][0] = a[3] / m - mx * mx; cv[0][1] = a[4] / m - mx * my; cv[0][2] = a[5] / m - mx * mz;
    cv[1][1] = a[6] / m - my * my; cv[1][2] = a[7] / m - my * mz; cv[2][2] = a[8] / m - mz * mz;
    cv[1][0] = cv[0][1]; cv[2][0] = cv[0][2]; cv[2][1] = cv[1][2];
    double ev[3], vec[3][3];
    jacobi3_d(cv, ev, vec);
    int mi = 0;
    for (int i = 1; i < 3; ++i) if (fabs(ev[i]) < fabs(ev[mi])) mi = i;
    const float n[3] = {(float)vec[0][mi], (float)vec[1][mi], (float)vec[2][mi]};
    st->n[0] = n[0]; st->n[1] = n[1]; st->n[2] = n[2];
    st->pos[0] = (float)mx; st->pos[1] = (float)my; st->pos[2] = (float)mz;
    float dist = st->pos[0] * n[0];   // Plane(p1, normal): m_dist = m_pos.dot(m_normal) (Plane.cpp:21-26)
    dist += st->pos[1] * n[1];
    dist += st->pos[2] * n[2];
    st->dist = dist;
    hcs_axes(st->n, st->a0, st->a1);
    *plane_out = make_float4(n[0], n[1], n[2], dist);
    st->converged = (n[0] == cur->n[0] && n[1] == cur->n[1] && n[2] == cur->n[2] && dist == cur->dist &&
                     st->pos[0] == cur->pos[0] && st->pos[1] == cur->pos[1] && st->pos[2] == cur->pos[2]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_fit_final(const ChainTab chains, int k) {
    __shared__ double s_red[4][FIT_COLS];
    const ChainDev &C = chains.c[blockIdx.x];
    fit_final(C, k, s_red);
    if (k == 3) {   // last slot: skipped slots repeat their predecessor
        __syncthreads();
        wscore_final(C);
    }
}

// point removal + output index lists of all accepted candidates of a batch in one launch (job = blockIdx.y)
struct AssignJobs {
    const uint32_t *idx[16];
    uint32_t m[16];
    int32_t id[16];
    int32_t *out[16];   // nullptr: support below min_support, points are removed but no plane is reported
};
__global__ void k_assign_batch(AssignJobs jobs, const uint32_t *__restrict__ orig, int32_t *__restrict__ assigned) {
    const uint32_t j = blockIdx.y, m = jobs.m[j];
    const uint32_t *__restrict__ idx = jobs.idx[j];
    int32_t *__restrict__ out = jobs.out[j];
    const int32_t id = jobs.id[j];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t p = idx[i];
        assigned[p] = id;
        if (out) out[i] = (int32_t)orig[p];
    }
}

__global__ void k_fill_i32(int32_t *p, uint32_t n, int32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// One acceptance chain: everything the reference's per-candidate sequence touches.  Several candidates
// whose supports provably cannot overlap are accepted together: every kernel of the sequence is
// launched once for the whole batch (chain = blockIdx.y), the scoring pass reads the cloud once.
struct Chain {
    DBuf<float4> plane_cur;
    CompactScratch cs, cs2;
    DBuf<uint32_t> idxA, cntA;       // score(3 eps) list before the connected component
    DBuf<uint32_t> idxS[4];          // per-slot result lists
    DBuf<float2> uv;
    DBuf<uint32_t> bidx, label, sizes;
    DBuf<uint8_t> bmp, tmp;
    DBuf<double> part;
    DBuf<float> bbpart;
};

constexpr size_t ACCEPT_BYTES = 4 * sizeof(PlaneState) + 16 + 48;   // states, counts, normal sums
constexpr size_t ACCEPT_STRIDE = (ACCEPT_BYTES + 63) & ~(size_t)63;

struct PairAccept;
struct RansacWork {
    CloudDev sorted;
    DBuf<uint32_t> codes, codes_in, vals_in, orig;
    DBuf<int32_t> assigned;
    DBuf<float> sub;
    size_t sub_pitch = 0;
    DBuf<uint32_t> sub_index;
    uint32_t n_sub = 0;
    DBuf<char> round_block;          // contiguous hypotheses / positions / counts: one D2H per round
    float4 *hyp = nullptr, *hyp_pos = nullptr;
    uint32_t *hyp_counts = nullptr, *misc = nullptr;
    DBuf<float4> top;
    DBuf<int32_t> out_idx;
    HBuf<char> pinned;
    // acceptance chains
    std::vector<std::unique_ptr<Chain>> chains;
    DBuf<char> accept_block;         // B x ACCEPT_STRIDE: one D2H copy per batch
    DBuf<float4> cand_in;            // B x (hypothesis, position): one H2D copy per batch
    std::vector<ChainDev> h_tab;     // the chains' device pointers (host copy; batches pass them by value)
    std::vector<MarkJob> mark_jobs;      // [slot][chain]   (host; launches copy one slot's row into the kernargs)
    std::vector<CompactJob> compact_jobs;   // [slot][A|S][chain]
    HBuf<char> pinned_accept;
    uint32_t B = 0;
    std::vector<uint64_t> tab_key;
    uint64_t tab_hash = 0;           // changes whenever a pointer of the tables moves (keys the captured graphs)
    PairAccept *solo = nullptr;      // runs this work area's batches when no pair coordinator is given
    hipEvent_t ev_ready = nullptr;
    ~RansacWork();
};

RansacWork *ransac_work_create() { return new RansacWork; }
void ransac_work_destroy(RansacWork *w) { delete w; }

namespace {

struct Accepted {
    float coef[4];
    uint32_t support;
    uint32_t offset;   // into out_idx
};

// One cloud's share of an acceptance batch.
struct AcceptSide {
    RansacWork *W = nullptr;
    uint32_t nc = 0;              // chains of this cloud in the batch
    CloudView cv{};               // the Morton-ordered cloud
    const int32_t *assigned = nullptr;
    float eps3 = 0.f, cos_t = 0.f, bitmap_eps = 0.f;
    hipEvent_t ready = nullptr;   // recorded on the owner's stream after it queued the batch's inputs
};

// The whole per-candidate sequence of RansacShapeDetector.cpp:618-656 for the chains of one or two clouds x four
// slots, no host round trip: slot 0 = the candidate (GlobalScore(3 eps) + ConnectedComponent; its clone's first
// GlobalWeightedScore is the same computation), slot k = k-th LS refit of slot k-1's points.  Per slot:
// GlobalWeightedScore (Candidate.h:293-302) = score(3 eps) -> ConnectedComponent -> weighted score,
// then the LS fit of the result list.
void enqueue_accept(plade_ctx *ctx, const AcceptSide *sides, int ns) {
    hipStream_t st = ctx->stream;
    ChainTab tab;
    uint32_t nc = 0, nb4_max = 0;
    for (int i = 0; i < ns; ++i) {
        const AcceptSide &S = sides[i];
        for (uint32_t b = 0; b < S.nc; ++b) tab.c[nc + b] = S.W->h_tab[b];
        tab.g[i] = ChainGroup{S.cv, S.eps3, S.bitmap_eps, cdiv(S.cv.n, 1024)};
        nb4_max = std::max(nb4_max, cdiv(S.cv.n, 1024));
        nc += S.nc;
    }
    PLADE_REQUIRE(nc >= 1 && nc <= (uint32_t)CHAIN_MAX, PLADE_EINVAL, "ransac: batch size");
    for (uint32_t b = nc; b < (uint32_t)CHAIN_MAX; ++b) tab.c[b] = tab.c[0];
    if (ns == 1) tab.g[1] = tab.g[0];
    tab.split = sides[0].nc;
    hipLaunchKernelGGL(k_state_from_hyp, dim3(nc), dim3(64), 0, st, tab);
    std::vector<MarkJob> mj(nc);
    std::vector<CompactJob> cj(nc);
    for (int k = 0; k < 4; ++k) {
        ScanGroup groups[2];
        uint32_t o = 0;
        for (int i = 0; i < ns; ++i) {
            const AcceptSide &S = sides[i];
            for (uint32_t b = 0; b < S.nc; ++b) mj[o + b] = S.W->mark_jobs[(size_t)k * S.W->B + b];
            groups[i] = ScanGroup{S.cv.x, S.cv.y, S.cv.z, S.cv.nx, S.cv.ny, S.cv.nz, S.assigned, S.cv.n, 0, 0, S.nc, S.eps3, S.cos_t};
            o += S.nc;
        }
        score_mark_batch(ctx, st, mj.data(), groups, (uint32_t)ns);
        for (int ab = 0; ab < 2; ++ab) {
            if (ab == 1) {
                hipLaunchKernelGGL(k_cc_raster, dim3(128, nc), dim3(256), 0, st, tab, k);
                hipLaunchKernelGGL(k_cc_label, dim3(nc), dim3(1024), 0, st, tab, k, 1);
                hipLaunchKernelGGL(k_cc_select, dim3(nb4_max, nc), dim3(256), 0, st, tab, k);
            }
            o = 0;
            for (int i = 0; i < ns; ++i) {
                const AcceptSide &S = sides[i];
                for (uint32_t b = 0; b < S.nc; ++b) cj[o + b] = S.W->compact_jobs[(size_t)(2 * k + ab) * S.W->B + b];
                o += S.nc;
            }
            compact_batch(ctx, st, cj.data(), nc);
        }
        hipLaunchKernelGGL(k_fit_final, dim3(nc), dim3(256), 0, st, tab, k);
    }
}

void chains_prepare(plade_ctx *ctx, RansacWork &W, uint32_t B, uint32_t n, float eps3, float cos_t, float bitmap_eps) {
    PLADE_REQUIRE(B >= 1 && B <= (uint32_t)CHAIN_MAX, PLADE_EINVAL, "ransac: batch size");
    W.B = B;
    while (W.chains.size() < B) W.chains.emplace_back(new Chain);
    char *ab = W.accept_block.ensure(B * ACCEPT_STRIDE);
    W.cand_in.ensure(2 * B);
    W.pinned_accept.ensure(B * ACCEPT_STRIDE + 64);
    std::vector<ChainDev> tab(B);
    std::vector<MarkJob> mj(4 * (size_t)B);
    std::vector<CompactJob> cj(8 * (size_t)B);
    const uint32_t nb4 = cdiv(n, 1024);
    bool fresh_bitmap = false;
    for (uint32_t b = 0; b < B; ++b) {
        Chain &C = *W.chains[b];
        ChainDev &D = tab[b];
        char *base = ab + b * ACCEPT_STRIDE;
        D.st = reinterpret_cast<PlaneState *>(base);
        D.cntS = reinterpret_cast<uint32_t *>(base + 4 * sizeof(PlaneState));
        D.nsum = reinterpret_cast<float *>(base + 4 * sizeof(PlaneState) + 16);
        D.top = W.cand_in.p + 2 * b;
        D.plane_cur = C.plane_cur.ensure(4);
        D.idxA = C.idxA.ensure((size_t)n + 4);
        D.cntA = C.cntA.ensure(8);
        for (int k = 0; k < 4; ++k) D.idxS[k] = C.idxS[k].ensure((size_t)n + 4);
        D.uv = C.uv.ensure((size_t)n + 4);
        D.bidx = C.bidx.ensure((size_t)n + 4);
        fresh_bitmap = fresh_bitmap || C.bmp.cap < CC_MAXPIX;
        const bool fresh = C.bmp.cap < CC_MAXPIX;
        D.label = C.label.ensure(CC_MAXPIX); D.sizes = C.sizes.ensure(CC_MAXPIX);
        D.bmp = C.bmp.ensure(CC_MAXPIX); D.tmp = C.tmp.ensure(CC_MAXPIX);
        if (fresh) HIP_TRY(hipMemsetAsync(C.bmp.p, 0, C.bmp.cap, ctx->stream));
        D.part = C.part.ensure((size_t)nb4 * FIT_COLS + 16);
        C.cs.masks.ensure((size_t)nb4 * 256 + 16); C.cs.block_counts.ensure(nb4 + 4);
        D.bcA = C.cs.block_counts.p;
        D.bbpart = C.bbpart.ensure(4 * (size_t)nb4 + 16);
        D.masks2 = C.cs2.masks.ensure((size_t)nb4 * 256 + 16);
        D.bc2 = C.cs2.block_counts.ensure(nb4 + 4);
        for (int k = 0; k < 4; ++k) {
            const uint32_t *skip = &D.st[k].converged;
            mj[(size_t)k * B + b] = MarkJob{D.plane_cur + k, C.cs.masks.p, C.cs.block_counts.p, skip};
            // score list + its (u, v) parameters in slot k's plane frame (PlaneState: pos, dist, a0, a1, bb)
            const CloudDev &sc = W.sorted;
            cj[(size_t)(2 * k) * B + b] = CompactJob{C.cs.masks.p, C.cs.block_counts.p, nullptr, D.idxA, D.cntA, skip,
                                                     D.st[k].pos, D.uv, D.bbpart, nb4, sc.x(), sc.y(), sc.z()};
            cj[(size_t)(2 * k + 1) * B + b] = CompactJob{D.masks2, D.bc2, D.idxA, D.idxS[k], D.cntS + k, skip, nullptr, nullptr, nullptr,
                                                         nb4, nullptr, nullptr, nullptr};
        }
    }
    // (re)upload the tables only when a pointer moved
    std::vector<uint64_t> key;
    key.reserve(tab.size() * sizeof(ChainDev) / 8 + 8);
    for (const ChainDev &D : tab) {
        const uint64_t *w = reinterpret_cast<const uint64_t *>(&D);
        key.insert(key.end(), w, w + sizeof(ChainDev) / 8);
    }
    {   // ... and every pointer of the job tables
        const uint64_t *w = reinterpret_cast<const uint64_t *>(mj.data());
        key.insert(key.end(), w, w + mj.size() * sizeof(MarkJob) / 8);
        w = reinterpret_cast<const uint64_t *>(cj.data());
        key.insert(key.end(), w, w + cj.size() * sizeof(CompactJob) / 8);
    }
    W.h_tab = tab;
    W.mark_jobs = mj;
    W.compact_jobs = cj;
    if (key != W.tab_key) {   // the tables are baked into the captured launches' arguments
        W.tab_key = key;
        uint64_t h = 1469598103934665603ull;
        for (uint64_t w : key) { h ^= w; h *= 1099511628211ull; }
        W.tab_hash = h;
    }
    (void)ctx; (void)eps3; (void)cos_t; (void)bitmap_eps; (void)fresh_bitmap;
}

}  // namespace

// Runs acceptance batches.  With two participants (the target's and the source's extraction threads of one
// registration) the two clouds' batches are merged: every kernel of the 29-launch sequence is latency-bound, so
// one sequence serving the chains of both clouds costs little more than one cloud's and the registration issues
// half as many of them.  A thread that reaches its next batch waits for the other one (or for it to finish its
// extraction); whoever arrives last launches for both on its own stream and wakes the other when the GPU is done.
struct PairAccept {
    std::mutex m;
    std::condition_variable cv;
    int participants = 0, waiting = 0;
    uint64_t generation = 0;
    AcceptSide side[2];
    bool present[2] = {false, false};
    Err err{0, ""};
    std::map<uint64_t, hipGraphExec_t> graphs;   // keyed on everything baked into the captured launches
    ~PairAccept() { for (auto &kv : graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second); }

    void join() { std::lock_guard<std::mutex> lk(m); ++participants; }
    void leave() { { std::lock_guard<std::mutex> lk(m); --participants; } cv.notify_all(); }

    static uint64_t mix(uint64_t h, uint64_t v) { h ^= v; h *= 1099511628211ull; return h; }
    static uint64_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

    void run(plade_ctx *ctx, const AcceptSide *sides, int ns) {
        if (ctx->profiling() || getenv("PLADE_NO_GRAPH")) { enqueue_accept(ctx, sides, ns); return; }
        uint64_t key = 1469598103934665603ull;
        for (int i = 0; i < ns; ++i) {
            const AcceptSide &S = sides[i];
            key = mix(key, S.W->tab_hash); key = mix(key, S.nc); key = mix(key, S.cv.n); key = mix(key, (uint64_t)S.cv.x);
            key = mix(key, (uint64_t)S.assigned); key = mix(key, fbits(S.eps3)); key = mix(key, fbits(S.cos_t));
            key = mix(key, fbits(S.bitmap_eps)); key = mix(key, (uint64_t)S.W);
        }
        hipGraphExec_t &g = graphs[key];
        if (!g) {
            if (graphs.size() > 96) {   // bounded cache (a batch of differently sized clouds)
                for (auto &kv : graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
                graphs.clear();
            }
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            try { enqueue_accept(ctx, sides, ns); }
            catch (...) { (void)hipStreamEndCapture(ctx->stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
            HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
            hipGraphExec_t e = nullptr;
            HIP_TRY(hipGraphInstantiate(&e, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            graphs[key] = e;
            HIP_TRY(hipGraphLaunch(e, ctx->stream));
            return;
        }
        HIP_TRY(hipGraphLaunch(g, ctx->stream));
    }

    // `who`: 0 = target, 1 = source.  Returns when the chains of `s` have run (the launching thread has waited for
    // the GPU); everything the caller queued for them on its own stream must have completed before the call.
    void submit(int who, plade_ctx *ctx, const AcceptSide &s) {
        std::unique_lock<std::mutex> lk(m);
        side[who] = s;
        present[who] = true;
        ++waiting;
        const uint64_t gen = generation;
        for (;;) {
            if (generation != gen) break;
            if (waiting >= participants) {   // everybody who is still extracting is here: launch for all
                AcceptSide both[2];
                int ns = 0;
                for (int i = 0; i < 2; ++i) if (present[i]) both[ns++] = side[i];
                err = Err{0, ""};
                ctx->stats.add(ns == 2 ? "ransac_launches_merged" : "ransac_launches_single", 1);
                try {
                    for (int i = 0; i < ns; ++i)
                        if (both[i].W != s.W && both[i].ready) HIP_TRY(hipStreamWaitEvent(ctx->stream, both[i].ready, 0));
                    run(ctx, both, ns);
                    for (int i = 0; i < ns; ++i)   // results of every cloud: one block per chain, into pinned memory
                        HIP_TRY(hipMemcpyAsync(both[i].W->pinned_accept.p, both[i].W->accept_block.p, both[i].nc * ACCEPT_STRIDE,
                                               hipMemcpyDeviceToHost, ctx->stream));
                    ctx->sync();
                }
                catch (const Err &e) { err = e; }
                catch (const std::exception &e) { err = Err{PLADE_EDEVICE, e.what()}; }
                present[0] = present[1] = false;
                waiting = 0;
                ++generation;
                cv.notify_all();
                break;
            }
            cv.wait(lk);
        }
        if (err.code) throw err;
    }
};

PairAccept *pair_accept_create() { return new PairAccept; }
void pair_accept_destroy(PairAccept *p) { delete p; }
RansacWork::~RansacWork() { delete solo; if (ev_ready) (void)hipEventDestroy(ev_ready); }

namespace {

inline bool same_plane(const float4 &a, const float4 &b, float eps) {
    const float c = a.x * b.x + a.y * b.y + a.z * b.z;
    if (std::fabs(c) < 0.995f) return false;
    const float db = c >= 0 ? b.w : -b.w;
    return std::fabs(a.w - db) < 2 * eps;
}

// Can a point be an inlier (|dist| < 3 eps AND |n.n_p| >= cos_t) of both planes?  Provably not when
//  (a) the normals are far apart: n_p within acos(cos_t) of both +-n_a and +-n_b needs
//      angle(n_a, n_b) <= 2 acos(cos_t) (or its supplement); a 5 degree margin covers the refits; or
//  (b) the planes are nearly parallel and, everywhere inside the cloud's bounding box, more than
//      8 eps apart (the two 3 eps bands plus refit drift cannot meet): the difference of the signed
//      distances is linear in p, so constant sign at the 8 corners + min |.| at a corner decide it.
// Anything else counts as a conflict and the two candidates are accepted one after the other.
inline bool conflict_free(const float4 &a, const float4 &b, float eps, float cos_t, const float *bbmin, const float *bbmax) {
    const float c = std::fabs(a.x * b.x + a.y * b.y + a.z * b.z);
    const float two_theta = 2.f * std::acos(std::min(1.f, cos_t)) + 0.0873f;
    if (two_theta < 1.5707f && c < std::cos(two_theta)) return true;
    if (c < 0.97f) return false;
    const float s = (a.x * b.x + a.y * b.y + a.z * b.z) >= 0 ? 1.f : -1.f;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = 0; k < 8; ++k) {
        const float px = (k & 1) ? bbmax[0] : bbmin[0], py = (k & 2) ? bbmax[1] : bbmin[1], pz = (k & 4) ? bbmax[2] : bbmin[2];
        const float da = a.x * px + a.y * py + a.z * pz - a.w, db = s * (b.x * px + b.y * py + b.z * pz) - s * b.w;
        mn = std::min(mn, da - db);
        mx = std::max(mx, da - db);
    }
    return (mn > 8 * eps) || (mx < -8 * eps);
}

}  // namespace

__global__ void k_list_masks(uint32_t m, uint32_t nb, uint8_t *__restrict__ masks, uint32_t *__restrict__ block_counts) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // one mask byte per 4 list positions
    if (t < nb * 256) {
        const uint32_t first = t * 4;
        masks[t] = first >= m ? 0 : (m - first >= 4 ? 0xF : (uint8_t)((1u << (m - first)) - 1));
    }
    if (t < nb) { const uint32_t b0 = t * 1024; block_counts[t] = b0 >= m ? 0 : min(1024u, m - b0); }
}

void plane_component(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const float normal[3], const float point[3],
                     const int32_t *idx, uint32_t m, float bitmap_eps, bool closing_filter, float w_eps, ComponentOut &out) {
    out.kept.clear();
    out.wscore = 0;
    out.err = 0;
    for (float &f : out.fit) f = 0.f;
    const uint32_t n = cloud.n;
    PLADE_REQUIRE(m <= n && bitmap_eps > 0.f, PLADE_EINVAL, "plane_component: bad argument");
    if (m == 0) return;
    chains_prepare(ctx, W, 1, n, 0.f, 0.f, 0.f);
    const ChainDev &D = W.h_tab[0];
    Chain &C = *W.chains[0];
    hipStream_t st = ctx->stream;
    CloudView cv{cloud.x(), cloud.y(), cloud.z(), cloud.nx(), cloud.ny(), cloud.nz(), n};
    // Plane(point, normal): dist = point . normal with Vec3f::dot's left-to-right sum (Plane.cpp:21-26)
    float dist = point[0] * normal[0];
    dist += point[1] * normal[1];
    dist += point[2] * normal[2];
    const float4 two[2] = {make_float4(normal[0], normal[1], normal[2], dist), make_float4(point[0], point[1], point[2], 0.f)};
    ctx->h2d(W.cand_in.p, two, 32);
    const uint32_t nb4 = cdiv(n, 1024);
    ChainTab tab;
    for (int b = 0; b < CHAIN_MAX; ++b) tab.c[b] = D;
    tab.g[0] = tab.g[1] = ChainGroup{cv, w_eps, bitmap_eps, nb4};
    tab.split = 1;
    hipLaunchKernelGGL(k_state_from_hyp, dim3(1), dim3(64), 0, st, tab);
    // the list as an all-ones mask over list positions, values = the caller's indices
    DBuf<uint32_t> d_idx;
    d_idx.ensure((size_t)n + 4);
    ctx->h2d(d_idx.p, idx, 4 * (size_t)m);
    hipLaunchKernelGGL(k_list_masks, dim3(cdiv(nb4 * 256, 256)), dim3(256), 0, st, m, nb4, C.cs.masks.p, C.cs.block_counts.p);
    const CompactJob hj{C.cs.masks.p, C.cs.block_counts.p, d_idx.p, D.idxA, D.cntA, nullptr, D.st[0].pos, D.uv, D.bbpart,
                        nb4, cv.x, cv.y, cv.z};
    compact_batch(ctx, st, &hj, 1);
    hipLaunchKernelGGL(k_cc_raster, dim3(128, 1), dim3(256), 0, st, tab, 0);
    hipLaunchKernelGGL(k_cc_label, dim3(1), dim3(1024), 0, st, tab, 0, closing_filter ? 1 : 0);
    hipLaunchKernelGGL(k_cc_select, dim3(nb4, 1), dim3(256), 0, st, tab, 0);
    compact_batch(ctx, st, W.compact_jobs.data() + (size_t)1 * W.B, 1);
    hipLaunchKernelGGL(k_fit_final, dim3(1), dim3(256), 0, st, tab, 0);
    PlaneState hst[2];
    uint32_t nk = 0;
    ctx->d2h(hst, D.st, 2 * sizeof(PlaneState));
    ctx->d2h(&nk, D.cntS, 4);
    ctx->sync(st);
    HIP_TRY(hipGetLastError());
    out.err = hst[0].err;
    if (out.err) return;
    out.kept.resize(nk);
    if (nk) HIP_TRY(hipMemcpy(out.kept.data(), D.idxS[0], 4 * (size_t)nk, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) { out.fit[k] = hst[1].n[k]; out.fit[3 + k] = hst[1].pos[k]; }
    out.fit[6] = hst[1].dist;
    out.wscore = hst[0].wscore;
}

void ransac_detect(plade_ctx *ctx, RansacWork &W, const CloudDev &cloud, const RansacParams &rp, PlaneSetOut &out, PairAccept *pair,
                   int who) {
    // a member of a pair takes part in the lock-step batches from here until it returns
    struct Membership {
        PairAccept *p;
        explicit Membership(PairAccept *q) : p(q) { if (p) p->join(); }
        ~Membership() { if (p) p->leave(); }
    } membership(pair);
    const uint32_t n = cloud.n;
    Clock::time_point t_setup0 = Clock::now();
    out.coef.clear(); out.offsets.assign(1, 0); out.idx.clear(); out.d_idx = nullptr;
    out.n_score_passes = 0; out.remaining = n;
    if (n < 3) return;
    // ---- scale exactly as plane_extraction.cpp:71-80 + PointCloud.h:94-98 (Z bug: maxZ stays
    //      -FLT_MAX, so the Z extent never wins the max) ------------------------------------------
    const float scale = std::max(cloud.bbmax[0] - cloud.bbmin[0], cloud.bbmax[1] - cloud.bbmin[1]);
    const float eps = rp.dist_rel * scale, bitmap_eps = rp.bitmap_rel * scale, cos_t = rp.cos_thresh;
    const float eps3 = 3 * eps;   // RansacShapeDetector.cpp:471-473
    PLADE_REQUIRE(eps > 0.f && bitmap_eps > 0.f, PLADE_EINVAL, "plane extraction: degenerate bounding box");

    // ---- Morton order -------------------------------------------------------------------------
    W.codes_in.ensure(n); W.vals_in.ensure(n); W.codes.ensure(n); W.orig.ensure(n);
    const float cube = std::max({cloud.bbmax[0] - cloud.bbmin[0], cloud.bbmax[1] - cloud.bbmin[1], cloud.bbmax[2] - cloud.bbmin[2], 1e-30f});
    const unsigned nb = cdiv(n, 256);
    hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, ctx->stream, cloud.x(), cloud.y(), cloud.z(), n, cloud.bbmin[0],
                       cloud.bbmin[1], cloud.bbmin[2], 1.f / cube, W.codes_in.p, W.vals_in.p);
    sort_pairs_u32(ctx, W.codes_in.p, W.codes.p, W.vals_in.p, W.orig.p, n, 24);
    W.sorted.n = n;
    W.sorted.pitch = cloud.pitch;
    W.sorted.soa.ensure(6 * cloud.pitch + 4);
    hipLaunchKernelGGL(k_gather_cloud, dim3(nb), dim3(256), 0, ctx->stream, cloud.aos.p, W.orig.p, n, W.sorted.soa.p, W.sorted.pitch);
    const CloudDev &c = W.sorted;
    CloudView cv{c.x(), c.y(), c.z(), c.nx(), c.ny(), c.nz(), n};
    W.assigned.ensure((size_t)n + 4);
    hipLaunchKernelGGL(k_fill_i32, dim3(nb), dim3(256), 0, ctx->stream, W.assigned.p, n, -1);
    // ---- stratified subset (every stride-th point of the Morton order) --------------------------
    const uint32_t stride = std::max(1u, n / 16384u);
    W.n_sub = (n + stride - 1) / stride;
    W.sub_pitch = ((size_t)W.n_sub + 3) & ~(size_t)3;
    W.sub.ensure(6 * W.sub_pitch + 4);
    W.sub_index.ensure(W.sub_pitch + 4);
    hipLaunchKernelGGL(k_make_subset, dim3(cdiv(W.n_sub, 256)), dim3(256), 0, ctx->stream, c.soa.p, c.pitch, n, stride, W.n_sub,
                       W.sub.p, W.sub_pitch, W.sub_index.p);
    const float *sx = W.sub.p, *sy = sx + W.sub_pitch, *sz = sy + W.sub_pitch, *snx = sz + W.sub_pitch, *sny = snx + W.sub_pitch,
                *snz = sny + W.sub_pitch;

    const uint32_t H = 4096, TOP = 48;
    const size_t round_bytes = (size_t)H * 36 + 64;
    char *rb = W.round_block.ensure(round_bytes);
    W.hyp = reinterpret_cast<float4 *>(rb);
    W.hyp_pos = W.hyp + H;
    W.hyp_counts = reinterpret_cast<uint32_t *>(W.hyp_pos + H);
    W.misc = W.hyp_counts + H;
    W.top.ensure(TOP + TOP / 4 + 4);   // TOP planes followed by TOP counts: one upload brings planes + zeroed counts
    char *pin = W.pinned.ensure(round_bytes + 256);
    W.out_idx.ensure((size_t)n + 4);
    // ---- acceptance chains ------------------------------------------------------------------------
    uint32_t B = 8;
    if (const char *e = getenv("PLADE_RANSAC_CHAINS")) B = (uint32_t)std::max(1, std::min(CHAIN_MAX / 2, atoi(e)));
    chains_prepare(ctx, W, B, n, eps3, cos_t, bitmap_eps);

    ctx->sync();
    ctx->stats.add("ransac_t_setup", secs_since(t_setup0));
    const int min_level = 1, max_level = 8;
    const float levels = (float)(max_level - min_level + 1);
    auto fail_prob = [&](float cand_size, float n_pts, float drawn) {  // RansacShapeDetector.h:61-67 (reqSamples = 3)
        return std::min(std::pow(1.f - cand_size / (n_pts * levels * 4.f), drawn), 1.f);
    };

    double t_sample = 0, t_rescore = 0, t_accept = 0;
    uint32_t n_rounds = 0, n_accepts = 0, n_batches = 0;
    std::vector<Accepted> accepted;
    std::vector<float4> h_hyp(H), h_pos(H);
    std::vector<uint32_t> h_counts(H);
    struct Cand { float4 pl, pos; uint32_t count; };
    std::vector<Cand> pool;
    uint32_t n_remaining = n;
    float drawn = 0.f;
    uint32_t out_off = 0;
    uint32_t n_full_passes = 0;
    const uint32_t max_rounds = 4000;
    const bool dbg = getenv("PLADE_DEBUG_RANSAC") != nullptr;
    for (uint32_t round = 0; round < max_rounds; ++round) {
        if (n_remaining < rp.min_support) break;
        if (round > 0 && fail_prob((float)rp.min_support, (float)n_remaining, drawn) <= rp.overlook_p && pool.empty()) break;
        // ---- draw and score a batch ------------------------------------------------------------
        ++n_rounds;
        Clock::time_point t_s0 = Clock::now();
        hipLaunchKernelGGL(k_sample, dim3(cdiv(H, 256)), dim3(256), 0, ctx->stream, cv, W.codes.p, W.assigned.p, rp.seed, round, H,
                           min_level, max_level, eps, cos_t, W.hyp, W.hyp_pos, W.hyp_counts, W.misc);
        score_multi(ctx, sx, sy, sz, snx, sny, snz, W.assigned.p, W.sub_index.p, W.n_sub, W.hyp, H, eps, cos_t, W.hyp_counts, true);
        hipLaunchKernelGGL(k_count_unassigned, dim3(cdiv(W.n_sub, 256)), dim3(256), 0, ctx->stream, W.assigned.p, W.sub_index.p,
                           W.n_sub, W.misc);
        uint32_t sub_un = 0;
        ctx->d2h(pin, rb, (size_t)H * 36 + 4);
        ctx->sync();
        memcpy(h_hyp.data(), pin, (size_t)H * 16);
        memcpy(h_pos.data(), pin + (size_t)H * 16, (size_t)H * 16);
        memcpy(h_counts.data(), pin + (size_t)H * 32, (size_t)H * 4);
        memcpy(&sub_un, pin + (size_t)H * 36, 4);
        t_sample += secs_since(t_s0);
        uint32_t valid = 0;
        for (uint32_t i = 0; i < H; ++i) valid += h_pos[i].w != 0.f;   // w: 0 no samples, 1 verified plane, 2 drawn but rejected
        drawn += (float)valid;
        if (dbg)
            fprintf(stderr, "[ransac] round %u valid %u drawn %.0f n_rem %u accepted %zu failprob %.4g\n", round, valid, drawn,
                    n_remaining, accepted.size(), fail_prob((float)rp.min_support, (float)n_remaining, drawn));
        // leaders of this batch by estimated support, one representative per distinct plane
        const double ratio = sub_un ? (double)n_remaining / sub_un : 0.0;
        std::vector<uint32_t> order;
        order.reserve(H);
        for (uint32_t i = 0; i < H; ++i)
            if (h_pos[i].w == 1.f && h_counts[i] * ratio >= 0.5 * rp.min_support) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return h_counts[a] != h_counts[b] ? h_counts[a] > h_counts[b] : a < b;
        });
        for (uint32_t i : order) {
            if (pool.size() >= TOP) break;
            bool dup = false;
            for (const Cand &pc : pool) if (same_plane(pc.pl, h_hyp[i], eps)) { dup = true; break; }
            if (!dup) pool.push_back(Cand{h_hyp[i], h_pos[i], 0});
        }
        // ---- harvest: re-score the pool on all unassigned points, accept the leaders, repeat ----------
        while (!pool.empty()) {
            Clock::time_point t_r0 = Clock::now();
            const uint32_t np = (uint32_t)pool.size();
            std::vector<float4> pl(TOP + TOP / 4, make_float4(0.f, 0.f, 0.f, 0.f));
            for (uint32_t i = 0; i < np; ++i) pl[i] = pool[i].pl;
            uint32_t *top_counts = reinterpret_cast<uint32_t *>(W.top.p + TOP);
            ctx->h2d(W.top.p, pl.data(), pl.size() * 16);
            score_multi(ctx, c.x(), c.y(), c.z(), c.nx(), c.ny(), c.nz(), W.assigned.p, nullptr, n, W.top.p, np, eps, cos_t,
                        top_counts, true);
            ++n_full_passes;
            std::vector<uint32_t> cnts(np);
            ctx->d2h(cnts.data(), top_counts, np * 4);
            ctx->sync();
            t_rescore += secs_since(t_r0);
            for (uint32_t i = 0; i < np; ++i) pool[i].count = cnts[i];
            // candidates that can no longer reach min_support are dropped (RansacShapeDetector.cpp:826-832)
            pool.erase(std::remove_if(pool.begin(), pool.end(), [&](const Cand &a) { return a.count < rp.min_support; }), pool.end());
            if (pool.empty()) break;
            std::stable_sort(pool.begin(), pool.end(), [](const Cand &a, const Cand &b) { return a.count > b.count; });
            // the best candidate plus every further one whose support provably cannot touch the supports
            // already in the batch: accepting them concurrently equals accepting them one by one
            std::vector<uint32_t> batch{0};
            for (uint32_t i = 1; i < pool.size() && batch.size() < B; ++i) {
                bool ok = true;
                for (uint32_t j : batch) ok = ok && conflict_free(pool[i].pl, pool[j].pl, eps, cos_t, cloud.bbmin, cloud.bbmax);
                if (ok) batch.push_back(i);
            }
            Clock::time_point t_a0 = Clock::now();
            ++n_batches;
            const uint32_t nc = (uint32_t)batch.size();
            std::vector<float4> h_in(2 * nc);
            for (uint32_t b = 0; b < nc; ++b) { h_in[2 * b] = pool[batch[b]].pl; h_in[2 * b + 1] = pool[batch[b]].pos; }
            ctx->h2d(W.cand_in.p, h_in.data(), 32 * nc);
            {
                if (!W.ev_ready) HIP_TRY(hipEventCreateWithFlags(&W.ev_ready, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(W.ev_ready, ctx->stream));
                AcceptSide S;
                S.W = &W; S.nc = nc; S.cv = cv; S.assigned = W.assigned.p;
                S.eps3 = eps3; S.cos_t = cos_t; S.bitmap_eps = bitmap_eps; S.ready = W.ev_ready;
                if (!pair && !W.solo) { W.solo = new PairAccept; W.solo->participants = 1; }
                // launches the batch (merged with the other cloud's when both are ready), reads the accept blocks
                // back into pinned memory and waits for the GPU
                (pair ? pair : W.solo)->submit(pair ? who : 0, ctx, S);
            }
            HIP_TRY(hipGetLastError());
            n_full_passes += 4 * (uint32_t)batch.size();
            t_accept += secs_since(t_a0);
            AssignJobs aj{};
            uint32_t n_aj = 0, max_m = 0;
            for (size_t b = 0; b < batch.size(); ++b) {
                Chain &C = *W.chains[b];
                const Cand &bc = pool[batch[b]];
                ++n_accepts;
                PlaneState hst[4];
                uint32_t hcnt[4];
                float hns[12];
                const char *hb = W.pinned_accept.p + b * ACCEPT_STRIDE;
                memcpy(hst, hb, 4 * sizeof(PlaneState));
                memcpy(hcnt, hb + 4 * sizeof(PlaneState), 16);
                memcpy(hns, hb + 4 * sizeof(PlaneState) + 16, 48);
                PLADE_REQUIRE(hst[0].err != 1, PLADE_ELIMIT, "plane extraction: connected-component bitmap too large");
                // replay of the reference's refit loop on the four results
                int final_slot = 0;
                {
                    double newScore = hst[0].wscore;
                    for (int fittingIter = 1; fittingIter <= 3; ++fittingIter) {
                        const double oldScore = newScore;
                        if (hcnt[fittingIter - 1] < 3 || hst[fittingIter].err) break;   // LSFit impossible
                        newScore = hst[fittingIter].wscore;
                        const uint32_t newSize = hcnt[fittingIter];
                        if (newScore > oldScore && newSize > rp.min_support) final_slot = fittingIter;  // clone.Clone(&candidates.back())
                        if (!(newScore > oldScore)) break;
                    }
                }
                const PlaneState &cand_state = hst[final_slot];
                const uint32_t cand_size = hcnt[final_slot];
                uint32_t *cand_idx = C.idxS[final_slot].p;
                if (dbg)
                    fprintf(stderr, "[ransac]   accept[%zu/%zu]: eps-count %u sizes %u %u %u %u wscore %.1f %.1f %.1f %.1f final slot %d bitmap %ux%u\n",
                            b, batch.size(), bc.count, hcnt[0], hcnt[1], hcnt[2], hcnt[3], hst[0].wscore, hst[1].wscore, hst[2].wscore,
                            hst[3].wscore, final_slot, hst[final_slot].ue, hst[final_slot].ve);
                // ---- remove the points (RansacShapeDetector.cpp:666-675) ---------------------------------
                if (cand_size == 0) continue;
                const int32_t shape_id = (int32_t)accepted.size();
                aj.idx[n_aj] = cand_idx; aj.m[n_aj] = cand_size; aj.id[n_aj] = shape_id; aj.out[n_aj] = nullptr;
                max_m = std::max(max_m, cand_size);
                drawn = std::pow(1.f - (cand_size / float(n_remaining)), 3.f) * drawn;
                n_remaining -= std::min(n_remaining, cand_size);
                // plane_extraction.cpp:134-149: shapes below min_support are skipped, d = -n.p with n re-normalised
                Accepted a{};
                a.support = 0; a.offset = out_off;
                if (cand_size >= rp.min_support) {
                    float nn[3] = {cand_state.n[0], cand_state.n[1], cand_state.n[2]};
                    float l = nn[0] * nn[0];
                    l += nn[1] * nn[1];
                    l += nn[2] * nn[2];
                    l = std::sqrt(l);
                    if (l > 0) { nn[0] /= l; nn[1] /= l; nn[2] /= l; }
                    float d = -(nn[0] * cand_state.pos[0] + nn[1] * cand_state.pos[1] + nn[2] * cand_state.pos[2]);
                    if (rp.orient_normals) {  // mean inlier normal (the intent of plane_extraction.cpp:43-58)
                        const float *ns = hns + 3 * final_slot;
                        if (ns[0] * nn[0] + ns[1] * nn[1] + ns[2] * nn[2] < 0) { nn[0] = -nn[0]; nn[1] = -nn[1]; nn[2] = -nn[2]; d = -d; }
                    }
                    a.coef[0] = nn[0]; a.coef[1] = nn[1]; a.coef[2] = nn[2]; a.coef[3] = d;
                    a.support = cand_size;
                    aj.out[n_aj] = W.out_idx.p + out_off;
                    out_off += cand_size;
                }
                ++n_aj;
                accepted.push_back(a);
            }
            if (n_aj)   // the candidates of a batch have disjoint supports: one launch removes them all
                hipLaunchKernelGGL(k_assign_batch, dim3(std::min(cdiv(max_m, 256), 256u), n_aj), dim3(256), 0, ctx->stream, aj, W.orig.p,
                                   W.assigned.p);
            // drop the batch from the pool (indices are ascending)
            for (size_t b = batch.size(); b-- > 0;) pool.erase(pool.begin() + batch[b]);
            if (n_remaining < rp.min_support) { pool.clear(); break; }
        }
    }
    ctx->stats.add("ransac_t_sample", t_sample);
    ctx->stats.add("ransac_t_rescore", t_rescore);
    ctx->stats.add("ransac_t_accept", t_accept);
    ctx->stats.add("ransac_rounds", n_rounds);
    ctx->stats.add("ransac_accepts", n_accepts);
    ctx->stats.add("ransac_batches", n_batches);
    // ---- output ---------------------------------------------------------------------------------
    out.idx.clear();
    if (rp.host_indices) {
        out.idx.resize(out_off);
        if (out_off) ctx->d2h(out.idx.data(), W.out_idx.p, 4 * (size_t)out_off);
    }
    ctx->sync();
    for (auto &a : accepted) {
        if (!a.support) continue;
        out.coef.insert(out.coef.end(), a.coef, a.coef + 4);
        out.offsets.push_back((int32_t)(a.offset + a.support));
    }
    out.d_idx = reinterpret_cast<const uint32_t *>(W.out_idx.p);
    out.n_score_passes = n_full_passes;
    out.remaining = n_remaining;
}

}  // namespace plade
