// plade_amd/csrc/k_overlap.hip -- K8: per-candidate overlap counting on gfx950.
//
// Reference: the verification loop code/PLADE/plade.cpp:547-564 calling
// ComputeOverlap<PointXYZ> (code/PLADE/util.h:611-647):
//   U_k   = { t in tgt_ds : |c_k - t|^2 < float(R_s^2) }                    (coarse sphere)
//   cnt_k = #{ p in src_ds : exists t in U_k with |T_k p - t|^2 < float(leaf^2) }
// with FLANN's fp32 L2_Simple accumulation and strict `<` (flann/algorithms/dist.h:84-90,
// flann/util/result_set.h:479,582), pcl::transformPointCloud's expression order
// (pcl-1.8.1/common/include/pcl/common/impl/transforms.hpp:69-71) and the squared radii formed as
// float(double(r)*double(r)) (pcl-1.8.1/kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:193).
//
// GPU mapping: the two kd-trees per candidate are replaced by ONE uniform grid over tgt_ds
// (cell >= leaf, so a 27-cell probe is exhaustive); the predicate is evaluated on the same fp32
// distances, so counts are bit-exact.  One lane = one source point, candidates staged in LDS
// (12 floats of T + centre), source stream fully coalesced, hit flags reduced with
// __ballot/s_bcnt and one atomicAdd per wave per candidate.  The grid (a few MB) lives in L2/MALL.
#include "overlap.h"
#include "prims.h"

namespace plade {

struct GridParams {
    float mnx, mny, mnz, inv;
    int dx, dy, dz;
};

__device__ void k_minmax3(const VB &vb, const float *__restrict__ xyz, uint32_t n, uint32_t stride, float *__restrict__ out6) {
    // out6 initialised to (+inf x3, -inf x3) as ordered ints by the host
    __shared__ float s_lds[6][8];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = vb.bx * blockDim.x + threadIdx.x; i < n; i += vb.gx * blockDim.x)
        for (int k = 0; k < 3; ++k) {
            float v = xyz[(size_t)i * stride + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    block_minmax_commit<3>(mn, mx, reinterpret_cast<int *>(out6), s_lds);
}

// blocked numbering: 4 x 4 x 4 cells per block, blocks x-major (see overlap.h)
__device__ __forceinline__ uint32_t block_of(int cx, int cy, int cz, int dx, int dy) {
    const int bdx = (dx + 3) >> 2, bdy = (dy + 3) >> 2;
    return (uint32_t)((cx >> 2) + bdx * ((cy >> 2) + bdy * (cz >> 2)));
}
__device__ __forceinline__ uint32_t local_of(int cx, int cy, int cz) { return (uint32_t)((cx & 3) | ((cy & 3) << 2) | ((cz & 3) << 4)); }

__device__ void k_cell_ids(const VB &vb, const float *__restrict__ xyz, uint32_t n, uint32_t stride, GridParams g, int blocked,
                           uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = xyz[(size_t)i * stride], y = xyz[(size_t)i * stride + 1], z = xyz[(size_t)i * stride + 2];
    int cx = min(max((int)floorf((x - g.mnx) * g.inv), 0), g.dx - 1);
    int cy = min(max((int)floorf((y - g.mny) * g.inv), 0), g.dy - 1);
    int cz = min(max((int)floorf((z - g.mnz) * g.inv), 0), g.dz - 1);
    // blocked = 2: the dense row index -- linear id in the grid padded by two cells on every side
    if (blocked == 2) keys[i] = (uint32_t)(cx + 2) + (uint32_t)(g.dx + 4) * ((uint32_t)(cy + 2) + (uint32_t)(g.dy + 4) * (uint32_t)(cz + 2));
    else keys[i] = blocked ? (block_of(cx, cy, cz, g.dx, g.dy) << 6) | local_of(cx, cy, cz) : (uint32_t)(cx + g.dx * (cy + g.dy * cz));
    vals[i] = i;
}

// dense row index, three small kernels behind the sort (no fill, no atomics, no scan):
// (1) the points in cell order
__device__ void k_dense_points(const VB &vb, const float *__restrict__ xyz, uint32_t stride, const uint32_t *__restrict__ vals, uint32_t n,
                               float4 *__restrict__ sorted) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = vals[i];
    sorted[i] = make_float4(xyz[(size_t)v * stride], xyz[(size_t)v * stride + 1], xyz[(size_t)v * stride + 2], __uint_as_float(v));
}
// (2) row_start[L] = number of sorted keys < L, for 256 consecutive L per workgroup: one uniform binary search for the first of them
// (scalar loads), then the few keys that fall into the workgroup's 256 cells are counted in LDS and prefix-summed
__device__ void k_row_table(const VB &vb, const uint32_t *__restrict__ keys, uint32_t n, uint32_t table_n, uint32_t *__restrict__ row_start) {
    __shared__ uint32_t s_hist[256], s_w[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t L0 = vb.bx * 256u;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < L0) lo = mid + 1; else hi = mid; }
    s_hist[tid] = 0u;
    __syncthreads();
    for (uint32_t j0 = lo;; j0 += 256u) {   // uniform
        const uint32_t j = j0 + tid;
        const uint32_t k = j < n ? keys[j] : 0xffffffffu;
        const bool in = k - L0 < 256u;      // (k >= L0: the keys are sorted)
        if (in) atomicAdd(&s_hist[k - L0], 1u);
        if (!__syncthreads_and(in ? 1 : 0)) break;
    }
    const uint32_t c = s_hist[tid];
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= (uint32_t)d) incl += o; }
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    uint32_t off = lo;
    for (uint32_t w = 0; w < wave; ++w) off += s_w[w];
    if (L0 + tid < table_n) row_start[L0 + tid] = off + incl - c;
}
// (3) the near mask: one lane per block of (1 << ms)^3 padded cells -- is any cell of the block widened by one cell occupied?
// (rows of the widened block = differences of row_start)
__device__ void k_near_mask(const VB &vb, const uint32_t *__restrict__ row_start, GridParams g, int ms, uint32_t n_blocks,
                            uint32_t *__restrict__ mask) {
    const uint32_t b = vb.bx * blockDim.x + threadIdx.x;
    const int DX = g.dx + 4, DY = g.dy + 4, DZ = g.dz + 4, S = 1 << ms;
    const int MBX = ((DX - 1) >> ms) + 1, MBY = ((DY - 1) >> ms) + 1;
    bool occ = false;
    if (b < n_blocks) {
        const int bx = (int)(b % (uint32_t)MBX), by = (int)((b / (uint32_t)MBX) % (uint32_t)MBY), bz = (int)(b / ((uint32_t)MBX * (uint32_t)MBY));
        const int x0 = max(bx * S - 1, 0), x1 = min(bx * S + S, DX - 1), y0 = max(by * S - 1, 0), y1 = min(by * S + S, DY - 1),
                  z0 = max(bz * S - 1, 0), z1 = min(bz * S + S, DZ - 1);
        for (int z = z0; z <= z1 && !occ; ++z)
            for (int y = y0; y <= y1 && !occ; ++y) {
                const uint32_t a = (uint32_t)DX * ((uint32_t)y + (uint32_t)DY * (uint32_t)z);
                occ = row_start[a + (uint32_t)x1 + 1u] != row_start[a + (uint32_t)x0];
            }
    }
    const unsigned long long m = __ballot(occ);
    const uint32_t lane = threadIdx.x & 63u, w0 = (vb.bx * blockDim.x + (threadIdx.x & ~63u)) >> 5;
    if (lane == 0) mask[w0] = (uint32_t)m;
    if (lane == 32) mask[w0 + 1u] = (uint32_t)(m >> 32);
}

// (4, r6) row occupancy: bit a = "the run of three cells that starts at cell a holds a point" (row_start[a + 3] != row_start[a]).
// 1 bit per padded cell = 256 KB at the bench's shape: unlike the 8 MB of row words it stays in every XCD's L2, and six of the nine
// runs around a probe on a surface are empty -- their row words need not be fetched at all.
__device__ void k_row_occ(const VB &vb, const uint32_t *__restrict__ row_start, uint32_t table_n, uint32_t *__restrict__ occ) {
    const uint32_t a = vb.bx * blockDim.x + threadIdx.x;
    const bool on = a + 3u < table_n && row_start[a + 3u] != row_start[a];
    const unsigned long long m = __ballot(on);
    const uint32_t lane = threadIdx.x & 63u, w0 = (vb.bx * blockDim.x + (threadIdx.x & ~63u)) >> 5;
    if (lane == 0) occ[w0] = (uint32_t)m;
    if (lane == 32) occ[w0 + 1u] = (uint32_t)(m >> 32);
}

__device__ void k_gather_cells(const VB &vb, const float *__restrict__ xyz, uint32_t stride, const uint32_t *__restrict__ keys,
                               const uint32_t *__restrict__ vals, uint32_t n, float4 *__restrict__ sorted,
                               uint32_t *__restrict__ cell_start, uint32_t *__restrict__ cell_end) {
    uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = vals[i];
    sorted[i] = make_float4(xyz[(size_t)v * stride], xyz[(size_t)v * stride + 1], xyz[(size_t)v * stride + 2],
                            __uint_as_float(v));
    uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) cell_start[k] = i;
    if (i == n - 1 || keys[i + 1] != k) cell_end[k] = i + 1;
}

// compact occupancy index over the sorted cell keys
__device__ void k_occ_bits(const VB &vb, const uint32_t *__restrict__ keys, uint32_t n, unsigned long long *__restrict__ bits) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) atomicOr(&bits[k >> 6], 1ull << (k & 63));
}
// ... and the BLOCK mask: one bit per 4 x 4 x 4-cell block, set when the block holds any occupied cell (bit b of blk[b >> 6]; the
// 64 lanes of a wavefront look at 64 consecutive blocks, one ballot is one word).  The verification kernel keeps it in LDS.
__device__ void k_occ_pop(const VB &vb, const unsigned long long *__restrict__ bits, uint32_t nw, uint32_t *__restrict__ pop,
                          unsigned long long *__restrict__ blk) {
    const uint32_t w = vb.bx * blockDim.x + threadIdx.x;
    const unsigned long long b = w < nw ? bits[w] : 0ull;
    if (w < nw) pop[w] = (uint32_t)__popcll(b);
    if (w == nw) pop[w] = 0;
    const unsigned long long any = __ballot(b != 0ull);
    if ((threadIdx.x & 63) == 0 && w <= nw) blk[w >> 6] = any;
}
__device__ void k_occ_start(const VB &vb, const float *__restrict__ xyz, uint32_t stride, const uint32_t *__restrict__ keys,
                            const uint32_t *__restrict__ vals, uint32_t n, const unsigned long long *__restrict__ bits,
                            const uint32_t *__restrict__ rank, uint32_t nw, float4 *__restrict__ sorted,
                            uint32_t *__restrict__ occ_start, float4 *__restrict__ cell_first) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = vals[i];
    const float4 pt = make_float4(xyz[(size_t)v * stride], xyz[(size_t)v * stride + 1], xyz[(size_t)v * stride + 2], __uint_as_float(v));
    sorted[i] = pt;
    const uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) {
        const uint32_t rk = rank[k >> 6] + (uint32_t)__popcll(bits[k >> 6] & ((1ull << (k & 63)) - 1ull));
        occ_start[rk] = i;
        // the cell's first point next to its position in the sorted array, bit 31 set when it is the cell's ONLY point (nearly
        // every cell of a voxel-downsampled cloud): the verification kernel then needs one load per occupied cell
        const bool single = i == n - 1 || keys[i + 1] != k;
        cell_first[rk] = make_float4(pt.x, pt.y, pt.z, __uint_as_float(i | (single ? 0x80000000u : 0u)));
    }
    if (i == n - 1) occ_start[rank[nw]] = n;
}

void TargetGrid::build(plade_ctx *ctx, const float *d_xyz, uint32_t n_pts, uint32_t stride, float min_cell,
                       const float *bbox_min, const float *bbox_max, bool compact_index) {
    n = n_pts;
    compact = compact_index;
    if (n == 0) return;
    float init[6];
    int iinit[6];
    if (bbox_min && bbox_max) {
        for (int k = 0; k < 3; ++k) { init[k] = bbox_min[k]; init[3 + k] = bbox_max[k]; }
    } else {
    for (int k = 0; k < 3; ++k) {
        float a = INFINITY, b = -INFINITY;
        int ia, ib;
        memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
        iinit[k] = ia >= 0 ? ia : ia ^ 0x7fffffff;
        iinit[3 + k] = ib >= 0 ? ib : ib ^ 0x7fffffff;
    }
    bbox.ensure(6);
    ctx->h2d(bbox.p, iinit, 24);
    launch<k_minmax3, 256>(ctx, dim3(std::min(cdiv(n, 256), 1024u)), 0, d_xyz, n, stride,
                       bbox.p);
    int ih[6];
    ctx->d2h(ih, bbox.p, 24);
    ctx->sync();
    for (int k = 0; k < 6; ++k) {
        int v = ih[k] >= 0 ? ih[k] : ih[k] ^ 0x7fffffff;
        memcpy(&init[k], &v, 4);
    }
    }
    // cell strictly larger than the probe radius so a +-1 cell probe is exhaustive even with the
    // fp32 rounding of the cell coordinate
    float cell = min_cell * 1.001f;
    if (!(cell > 0.f)) cell = 1.f;
    PLADE_REQUIRE(std::isfinite(init[0]) && std::isfinite(init[1]) && std::isfinite(init[2]) && std::isfinite(init[3]) &&
                      std::isfinite(init[4]) && std::isfinite(init[5]), PLADE_EINVAL, "grid: non-finite bounding box");
    static const bool old_index = getenv("PLADE_OVERLAP_INDEX_COMPACT") != nullptr;   // A/B hook: the bitmap + rank index of rounds 3-5
    dense = compact && !old_index;
    // the dense row index has one row start per cell of the grid PADDED by two cells on every side: the cap is on that
    // product (a flat or elongated target -- 6900 x 6900 x 1 cells -- would otherwise get 2.4e8 rows, 1 GB; advisor r5)
    const double pad = dense ? 4.0 : 0.0;
    for (;;) {
        double ex = std::floor((init[3] - init[0]) / cell) + 1, ey = std::floor((init[4] - init[1]) / cell) + 1,
               ez = std::floor((init[5] - init[2]) / cell) + 1;
        if ((ex + pad) * (ey + pad) * (ez + pad) <= 48.0e6) { gp.dx = (int)ex; gp.dy = (int)ey; gp.dz = (int)ez; break; }
        cell *= 1.26f;
    }
    gp.mnx = init[0]; gp.mny = init[1]; gp.mnz = init[2];
    gp.inv = 1.f / cell;
    keys.ensure(n); keys2.ensure(n); vals.ensure(n); vals2.ensure(n);
    sorted.ensure(n);
    GridParams g{gp.mnx, gp.mny, gp.mnz, gp.inv, gp.dx, gp.dy, gp.dz};
    if (dense) {
        DX = gp.dx + 4; DY = gp.dy + 4; DZ = gp.dz + 4;
        ncells = (size_t)DX * DY * DZ;                       // <= 48e6 (the cap above): a table of <= 192 MB
        const size_t table = (ncells + 8 + 15) & ~(size_t)15;   // row_start[a + 3] of the last row; whole 64-byte fills
        for (mask_shift = 2;; ++mask_shift) {                // the near mask must fit 32 KB of LDS
            const size_t mb = (size_t)(((DX - 1) >> mask_shift) + 1) * (((DY - 1) >> mask_shift) + 1) * (((DZ - 1) >> mask_shift) + 1);
            mask_words = (uint32_t)((mb + 31) / 32);
            if (mask_words * 4u <= (32u << 10)) break;
        }
        const uint32_t n_blocks = (uint32_t)((size_t)(((DX - 1) >> mask_shift) + 1) * (((DY - 1) >> mask_shift) + 1) * (((DZ - 1) >> mask_shift) + 1));
        row_start.ensure(table + 256); near_mask.ensure((size_t)mask_words + 64);
        launch<k_cell_ids, 256>(ctx, dim3(cdiv(n, 256)), 0, d_xyz, n, stride, g, 2, keys.p, vals.p);
        int bits = 1;
        while (((size_t)1 << bits) < ncells) ++bits;
        sort_pairs_u32(ctx, keys.p, keys2.p, vals.p, vals2.p, n, bits);
        launch<k_dense_points, 256>(ctx, dim3(cdiv(n, 256)), 0, d_xyz, stride, vals2.p, n, sorted.p);
        launch<k_row_table, 256>(ctx, dim3(cdiv(table, 256)), 0, keys2.p, n, (uint32_t)table, row_start.p);
        launch<k_near_mask, 256>(ctx, dim3(cdiv(n_blocks, 256)), 0, row_start.p, g, mask_shift, n_blocks, near_mask.p);
        static const bool no_occ = getenv("PLADE_OVERLAP_NO_ROW_OCC") != nullptr;   // A/B timing hook
        row_occ_on = !no_occ;
        if (row_occ_on) {
            row_occ.ensure(table / 32 + 64);
            launch<k_row_occ, 256>(ctx, dim3(cdiv(table, 256)), 0, row_start.p, (uint32_t)table, row_occ.p);
        }
        HIP_TRY(hipGetLastError());
        return;
    }
    ncells = compact ? (size_t)((gp.dx + 3) >> 2) * ((gp.dy + 3) >> 2) * ((gp.dz + 3) >> 2) * 64 : (size_t)gp.dx * gp.dy * gp.dz;
    launch<k_cell_ids, 256>(ctx, dim3(cdiv(n, 256)), 0, d_xyz, n, stride, g, compact ? 1 : 0, keys.p,
                       vals.p);
    int bits = 1;
    while (((size_t)1 << bits) < ncells) ++bits;
    sort_pairs_u32(ctx, keys.p, keys2.p, vals.p, vals2.p, n, bits);
    if (compact) {
        const uint32_t nw = (uint32_t)((ncells + 63) / 64);
        occ_bits.ensure(nw + 8); occ_pop.ensure(nw + 2); occ_rank.ensure(nw + 2); occ_start.ensure((size_t)n + 2);
        occ_blk.ensure(nw / 64 + 2);
        cell_first.ensure((size_t)n + 2);
        // (a multiple of 64 bytes: the runtime splits any other size into an aligned fill and a second command for the tail)
        ctx->fill_async(occ_bits.p, 0, (((size_t)nw + 1 + 7) & ~(size_t)7) * 8);
        launch<k_occ_bits, 256>(ctx, dim3(cdiv(n, 256)), 0, keys2.p, n, occ_bits.p);
        launch<k_occ_pop, 256>(ctx, dim3(cdiv(nw + 1, 256)), 0, occ_bits.p, nw, occ_pop.p, occ_blk.p);
        exclusive_scan_u32(ctx, occ_pop.p, occ_rank.p, (size_t)nw + 1);
        launch<k_occ_start, 256>(ctx, dim3(cdiv(n, 256)), 0, d_xyz, stride, keys2.p, vals2.p, n,
                           occ_bits.p, occ_rank.p, nw, sorted.p, occ_start.p, cell_first.p);
        HIP_TRY(hipGetLastError());
        return;
    }
    cell_start.ensure(ncells); cell_end.ensure(ncells);
    ctx->fill_async(cell_start.p, 0, ncells * 4);
    ctx->fill_async(cell_end.p, 0, ncells * 4);
    launch<k_gather_cells, 256>(ctx, dim3(cdiv(n, 256)), 0, d_xyz, stride, keys2.p, vals2.p, n,
                       sorted.p, cell_start.p, cell_end.p);
    HIP_TRY(hipGetLastError());
}

constexpr int OV_TPB = 256;
#ifndef OVD_W
#define OVD_W 5                      // wavefronts per SIMD the dense kernel is compiled for (= persistent workgroups per CU)
#endif
constexpr int OV_KCH = 16;             // candidates per chunk at most (few candidates: chunks of 2, so that ~10 candidates fill the GPU)
constexpr uint32_t OV_MASK_MAX = 48u << 10;   // bytes of LDS the block mask may take (3.9e5 blocks = 2.5e7 cells); above: no mask

// One work item = one tile of 256 source points x one chunk of `ch` candidates.  A workgroup takes a contiguous range of items
// (chunk-major, so that it changes chunk at most once or twice), keeps the chunk's transforms and its hit counters in LDS and
// adds the counters to the global counts when the chunk changes: the source tile is read once per chunk instead of once per
// pair of candidates (r3), and a counter shared by thousands of wavefronts on eight XCDs is touched once per workgroup and
// candidate.  With `mask_words` != 0 the workgroup first copies the target grid's BLOCK mask (k_occ_pop) into LDS: a probe whose
// <= 8 blocks are all empty -- nearly every probe of a wrong candidate, and everything outside the target -- ends without a
// global load; only the words of non-empty blocks are fetched.
// (five wavefronts per SIMD: 96 registers; the probe is a chain of three load rounds, latency hidden by occupancy -- 4 waves / 100
//  registers: 186 us at the bench's shape, 0.62 s at the stress shape; 5: 178 us, 0.575 s; 6 and 8 spill and are slower)
__device__ void k_overlap(const VB &vb, const float *__restrict__ sx, const float *__restrict__ sy,
                                                    const float *__restrict__ sz, uint32_t n_s,
                                                    const float4 *__restrict__ tgt,
                                                    const unsigned long long *__restrict__ occ_bits,
                                                    const uint32_t *__restrict__ occ_rank,
                                                    const uint32_t *__restrict__ occ_start,
                                                    const float4 *__restrict__ cell_first,
                                                    const uint32_t *__restrict__ blk_mask, uint32_t mask_words, GridParams g,
                                                    const float *__restrict__ T /*K x 16*/,
                                                    const float *__restrict__ centers /*K x 3*/, uint32_t K, float R2,
                                                    float r2, int32_t *__restrict__ counts, uint32_t ch, uint32_t items_per_wg) {
    __shared__ float s_T[OV_KCH][12];
    __shared__ float s_c[OV_KCH][3];
    __shared__ uint32_t s_cnt[OV_KCH];
    extern __shared__ uint32_t s_mask[];
    for (uint32_t i = threadIdx.x; i < mask_words; i += OV_TPB) s_mask[i] = blk_mask[i];
    const int lane = threadIdx.x & 63;
    const uint32_t ntiles = (n_s + OV_TPB - 1) / OV_TPB, nchunks = (K + ch - 1) / ch;
    const uint64_t nitems = (uint64_t)ntiles * nchunks;
    const uint64_t it0 = (uint64_t)vb.bx * items_per_wg, it1 = min(nitems, it0 + items_per_wg);
    uint32_t cur = 0xffffffffu, k0 = 0, kc = 0;
    for (uint64_t item = it0; item < it1; ++item) {
        const uint32_t chunk = (uint32_t)(item / ntiles), tile = (uint32_t)(item % ntiles);
        if (chunk != cur) {   // uniform
            __syncthreads();
            if (cur != 0xffffffffu && threadIdx.x < kc && s_cnt[threadIdx.x]) atomicAdd(&counts[k0 + threadIdx.x], (int32_t)s_cnt[threadIdx.x]);
            __syncthreads();
            cur = chunk; k0 = chunk * ch; kc = min(ch, K - k0);
            for (uint32_t i = threadIdx.x; i < kc * 12; i += OV_TPB) s_T[i / 12][i % 12] = T[(size_t)(k0 + i / 12) * 16 + i % 12];
            for (uint32_t i = threadIdx.x; i < kc * 3; i += OV_TPB) s_c[i / 3][i % 3] = centers[(size_t)(k0 + i / 3) * 3 + i % 3];
            if (threadIdx.x < OV_KCH) s_cnt[threadIdx.x] = 0u;
            __syncthreads();
        }
        const uint32_t i = tile * OV_TPB + threadIdx.x;
        const bool live = i < n_s;
        const f3 p = live ? f3(sx[i], sy[i], sz[i]) : f3();
        for (uint32_t kk = 0; kk < kc; ++kk) {
            bool hit = false;
            if (live) {
                const f3 q_ = pcl_xform(s_T[kk], p);
                const f3 c(s_c[kk][0], s_c[kk][1], s_c[kk][2]);
                const int cx = (int)floorf((q_.x - g.mnx) * g.inv), cy = (int)floorf((q_.y - g.mny) * g.inv),
                          cz = (int)floorf((q_.z - g.mnz) * g.inv);
                const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
                const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.dy - 1);
                const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.dz - 1);
                // the <= 27 cells live in <= 8 blocks of 4 x 4 x 4: one word + one rank per block.  The words of all non-empty blocks
                // are fetched together (one load latency instead of up to eight dependent ones), then looked at one after the other
                if (x0 <= x1 && y0 <= y1 && z0 <= z1) {
                    const int bx0 = x0 >> 2, by0 = y0 >> 2, bz0 = z0 >> 2;
                    const int nbx = (x1 >> 2) - bx0 + 1, nby = (y1 >> 2) - by0 + 1, nbz = (z1 >> 2) - bz0 + 1;   // 1 or 2 each
                    // Three rounds of INDEPENDENT loads instead of a chain of ~20 dependent ones (r3: block word -> rank -> start ->
                    // point, cell after cell: ~20 round trips to L2 per probe of a point near a surface): (1) the words of the
                    // non-empty blocks, (2) the ranks of the blocks that hold wanted cells, (3) the first points of up to four
                    // wanted cells at a time (cell_first: point + "only point of its cell").
                    unsigned long long w8[8], m8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int ix = q & 1, iy = (q >> 1) & 1, iz = q >> 2;
                        w8[q] = 0ull;
                        if (ix < nbx && iy < nby && iz < nbz) {
                            const uint32_t blk = block_of((bx0 + ix) << 2, (by0 + iy) << 2, (bz0 + iz) << 2, g.dx, g.dy);
                            if (!mask_words || ((s_mask[blk >> 5] >> (blk & 31u)) & 1u)) w8[q] = occ_bits[blk];
                        }
                    }
                    uint32_t r8[8];
                    bool any = false;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        m8[q] = 0ull;
                        r8[q] = 0u;
                        if (!w8[q]) continue;
                        const int bx = bx0 + (q & 1), by = by0 + ((q >> 1) & 1), bz = bz0 + (q >> 2);
                        // wanted cells of this block: the part of [x0,x1] x [y0,y1] x [z0,z1] inside it
                        // (ranges lo .. hi of local coordinates 0 .. 3 per axis as bit patterns: bits lo .. hi of every nibble / nibbles
                        //  lo .. hi of every 16-bit group / groups lo .. hi -- three multiplications instead of three loops)
                        const int lx = max(x0, bx << 2) & 3, hx = min(x1, (bx << 2) + 3) & 3;
                        const int ly = max(y0, by << 2) & 3, hy = min(y1, (by << 2) + 3) & 3;
                        const int lz = max(z0, bz << 2) & 3, hz = min(z1, (bz << 2) + 3) & 3;
                        const unsigned long long mx = 0x1111111111111111ull * (unsigned long long)((2u << hx) - (1u << lx));
                        const unsigned long long my = 0x0001000100010001ull * (unsigned long long)((16u << (hy << 2)) - (1u << (ly << 2)));
                        const unsigned long long mz = ((hz == 3 ? 0ull : (1ull << ((hz + 1) << 4))) - 1ull) & ~((1ull << (lz << 4)) - 1ull);
                        m8[q] = w8[q] & mx & my & mz;
                        if (m8[q]) { r8[q] = occ_rank[block_of(bx << 2, by << 2, bz << 2, g.dx, g.dy)]; any = true; }
                    }
                    while (any && !hit) {
                        uint32_t rk0 = 0xffffffffu, rk1 = 0xffffffffu, rk2 = 0xffffffffu, rk3 = 0xffffffffu;
                        int cnt = 0;
                        any = false;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            while (m8[q] && cnt < 4) {
                                const int bit = __ffsll((long long)m8[q]) - 1;
                                m8[q] &= m8[q] - 1;
                                const uint32_t rk = r8[q] + (uint32_t)__popcll(w8[q] & ((1ull << bit) - 1ull));
                                if (cnt == 0) rk0 = rk; else if (cnt == 1) rk1 = rk; else if (cnt == 2) rk2 = rk; else rk3 = rk;
                                ++cnt;
                            }
                            any = any || m8[q] != 0ull;
                        }
                        float4 c0, c1, c2, c3;
                        if (rk0 != 0xffffffffu) c0 = cell_first[rk0];
                        if (rk1 != 0xffffffffu) c1 = cell_first[rk1];
                        if (rk2 != 0xffffffffu) c2 = cell_first[rk2];
                        if (rk3 != 0xffffffffu) c3 = cell_first[rk3];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t rk = e == 0 ? rk0 : e == 1 ? rk1 : e == 2 ? rk2 : rk3;
                            if (rk == 0xffffffffu || hit) continue;
                            const float4 cf = e == 0 ? c0 : e == 1 ? c1 : e == 2 ? c2 : c3;
                            const f3 t(cf.x, cf.y, cf.z);
                            if (flann_d2(q_, t) < r2 && flann_d2(c, t) < R2) { hit = true; continue; }
                            const uint32_t tag = __float_as_uint(cf.w);
                            if (tag & 0x80000000u) continue;              // that was the cell's only point
                            const uint32_t pe = occ_start[rk + 1];
                            for (uint32_t j2 = (tag & 0x7fffffffu) + 1; j2 < pe; ++j2) {
                                const float4 t4 = tgt[j2];
                                const f3 t2(t4.x, t4.y, t4.z);
                                if (flann_d2(q_, t2) < r2 && flann_d2(c, t2) < R2) { hit = true; break; }
                            }
                        }
                    }
                }
            }
            const uint32_t h = (uint32_t)__popcll(__ballot(hit));
            if (lane == 0 && h) atomicAdd(&s_cnt[kk], h);
        }
    }
    __syncthreads();
    if (cur != 0xffffffffu && threadIdx.x < kc && s_cnt[threadIdx.x]) atomicAdd(&counts[k0 + threadIdx.x], (int32_t)s_cnt[threadIdx.x]);
}

// (r5) The same counts over the dense row index (TargetGrid::dense).  The bitmap + rank form above spends ~1 900 vector instructions per
// wavefront and candidate (rocprofv3 SQ_INSTS_VALU: eight predicated block words, eight wanted-cell masks in 64-bit arithmetic, cells
// taken four at a time in nested loops that run as long as the slowest lane) and the SIMDs' vector pipes were ~80 % busy while it ran:
// it was bound by instruction issue, not by memory.  Here a probe is: one LDS bit (the near mask), nine 4-word table reads (row a of
// three cells = the contiguous run [row_start[a], row_start[a + 3]) of the sorted points), the first point of every run fetched
// together, the rest of a run one after the other -- a row of three cells of a voxel-downsampled surface holds 0-4 points.  Same
// points, same fp32 distances, same strict comparisons: the counts are the bits of the other form (and of the oracle).
__device__ void k_overlap_dense(const VB &vb, const float *__restrict__ sx, const float *__restrict__ sy, const float *__restrict__ sz,
                                uint32_t n_s, const float4 *__restrict__ tgt, const uint32_t *__restrict__ row_start,
                                const uint32_t *__restrict__ near_mask, uint32_t mask_words, int ms,
                                const uint32_t *__restrict__ row_occ /* or null */, GridParams g,
                                const float *__restrict__ T /*K x 16*/, const float *__restrict__ centers /*K x 3*/, uint32_t K, float R2,
                                float r2, int32_t *__restrict__ counts, uint32_t ch, uint32_t items_per_wg) {
    __shared__ float s_T[OV_KCH][12];
    __shared__ float s_c[OV_KCH][3];
    __shared__ uint32_t s_cnt[OV_KCH];
    extern __shared__ uint32_t s_mask[];
    for (uint32_t i = threadIdx.x; i < mask_words; i += OV_TPB) s_mask[i] = near_mask[i];
    const int lane = threadIdx.x & 63;
    const uint32_t ntiles = (n_s + OV_TPB - 1) / OV_TPB, nchunks = (K + ch - 1) / ch;
    const uint64_t nitems = (uint64_t)ntiles * nchunks;
    const uint64_t it0 = (uint64_t)vb.bx * items_per_wg, it1 = min(nitems, it0 + items_per_wg);
    const uint32_t DX = (uint32_t)g.dx + 4u, DY = (uint32_t)g.dy + 4u;
    const uint32_t MBX = ((DX - 1u) >> ms) + 1u, MBY = ((DY - 1u) >> ms) + 1u;
    const uint32_t RS = DX, SS = DX * DY;                 // next row, next slice
    const float fdx = (float)(g.dx + 1), fdy = (float)(g.dy + 1), fdz = (float)(g.dz + 1);
    uint32_t cur = 0xffffffffu, k0 = 0, kc = 0;
    for (uint64_t item = it0; item < it1; ++item) {
        const uint32_t chunk = (uint32_t)(item / ntiles), tile = (uint32_t)(item % ntiles);
        if (chunk != cur) {   // uniform
            __syncthreads();
            if (cur != 0xffffffffu && threadIdx.x < kc && s_cnt[threadIdx.x]) atomicAdd(&counts[k0 + threadIdx.x], (int32_t)s_cnt[threadIdx.x]);
            __syncthreads();
            cur = chunk; k0 = chunk * ch; kc = min(ch, K - k0);
            for (uint32_t i = threadIdx.x; i < kc * 12; i += OV_TPB) s_T[i / 12][i % 12] = T[(size_t)(k0 + i / 12) * 16 + i % 12];
            for (uint32_t i = threadIdx.x; i < kc * 3; i += OV_TPB) s_c[i / 3][i % 3] = centers[(size_t)(k0 + i / 3) * 3 + i % 3];
            if (threadIdx.x < OV_KCH) s_cnt[threadIdx.x] = 0u;
            __syncthreads();
        }
        const uint32_t i = tile * OV_TPB + threadIdx.x;
        const bool live = i < n_s;
        const f3 p = live ? f3(sx[i], sy[i], sz[i]) : f3();
        for (uint32_t kk = 0; kk < kc; ++kk) {
            bool hit = false;
            const f3 q_ = pcl_xform(s_T[kk], p);
            // the probe's own cell, -1 .. d per axis (beyond that no cell of the grid is within one cell of it; a NaN fails the test)
            const float fx = (q_.x - g.mnx) * g.inv, fy = (q_.y - g.mny) * g.inv, fz = (q_.z - g.mnz) * g.inv;
            bool probe = live && fx >= -1.f && fy >= -1.f && fz >= -1.f && fx < fdx && fy < fdy && fz < fdz;
            uint32_t base = 0;
            if (probe) {
                const uint32_t px = (uint32_t)((int)floorf(fx) + 2), py = (uint32_t)((int)floorf(fy) + 2), pz = (uint32_t)((int)floorf(fz) + 2);
                const uint32_t b = (px >> ms) + MBX * ((py >> ms) + MBY * (pz >> ms));
                probe = (s_mask[b >> 5] >> (b & 31u)) & 1u;
                base = (px - 1u) + DX * ((py - 1u) + DY * (pz - 1u));
            }
            if (probe) {
                const f3 c(s_c[kk][0], s_c[kk][1], s_c[kk][2]);
                // nine rows, the probe's own first: (row, slice) offsets
                uint32_t s9[9], e9[9], a9[9];
                uint32_t occ9 = 0x1ffu;
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const int o = (r + 4) % 9;                         // 4 = the centre row (dy = 0, dz = 0)
                    a9[r] = base + (uint32_t)(o % 3) * RS + (uint32_t)(o / 3) * SS;
                }
                if (row_occ) occ9 = (row_occ[a9[0] >> 5] >> (a9[0] & 31u)) & 1u;     // the own run's bit now, the others' if it misses
#pragma unroll
                for (int r = 0; r < 9; ++r) { s9[r] = 0u; e9[r] = 0u; }
                if (occ9 & 1u) { s9[0] = row_start[a9[0]]; e9[0] = row_start[a9[0] + 3]; }
                // (r6) The probe's OWN run first -- its bit, its row words, its points --: where the clouds overlap most hits are found
                // there and nothing of the other eight runs is then requested.  The lanes that missed go on: the other runs' bits,
                // the row words of the non-empty ones, then point t of every remaining run together, t = 0, 1, ... (loads in flight;
                // the depth of that chain is the LONGEST run, 1-3 points on a voxel-downsampled surface, not the sum of the runs).
                // (A software pipeline over the chunk -- the next candidate's rows requested while this one's points are on their
                // way -- was built in r5 and is slower: it spills at the 96 registers that keep five wavefronts per SIMD.)
                for (uint32_t j = s9[0]; j < e9[0] && !hit; ++j) {
                    const float4 f = tgt[j];
                    const f3 tp(f.x, f.y, f.z);
                    hit = flann_d2(q_, tp) < r2 && flann_d2(c, tp) < R2;
                }
                e9[0] = s9[0];
                if (!hit) {      // the bits and row words of the other non-empty runs: only for the lanes whose own run did not settle it
                    if (row_occ) {
                        uint32_t w9[9];
#pragma unroll
                        for (int r = 1; r < 9; ++r) w9[r] = row_occ[a9[r] >> 5];
#pragma unroll
                        for (int r = 1; r < 9; ++r) occ9 |= ((w9[r] >> (a9[r] & 31u)) & 1u) << r;
                    }
#pragma unroll
                    for (int r = 1; r < 9; ++r)
                        if ((occ9 >> r) & 1u) { s9[r] = row_start[a9[r]]; e9[r] = row_start[a9[r] + 3]; }
                }
                uint32_t longest = 0;
#pragma unroll
                for (int r = 0; r < 9; ++r) longest = max(longest, e9[r] - s9[r]);
                if (hit) longest = 0;
                for (uint32_t t = 0; t < longest && !hit; ++t) {
                    float4 f9[9];
#pragma unroll
                    for (int r = 0; r < 9; ++r)
                        if (s9[r] + t < e9[r]) f9[r] = tgt[s9[r] + t];
#pragma unroll
                    for (int r = 0; r < 9; ++r)
                        if (s9[r] + t < e9[r] && !hit) {
                            const f3 tp(f9[r].x, f9[r].y, f9[r].z);
                            hit = flann_d2(q_, tp) < r2 && flann_d2(c, tp) < R2;
                        }
                }
            }
            const uint32_t h = (uint32_t)__popcll(__ballot(hit));
            if (lane == 0 && h) atomicAdd(&s_cnt[kk], h);
        }
    }
    __syncthreads();
    if (cur != 0xffffffffu && threadIdx.x < kc && s_cnt[threadIdx.x]) atomicAdd(&counts[k0 + threadIdx.x], (int32_t)s_cnt[threadIdx.x]);
}

// source points into a spatially blocked order: key = blocked cell id in the source's own frame
__device__ void k_init_minmax6(const VB &, int *__restrict__ out6) {   // (+inf x3, -inf x3) as ordered ints
    if (threadIdx.x < 3) out6[threadIdx.x] = ordered_int(INFINITY);
    else if (threadIdx.x < 6) out6[threadIdx.x] = ordered_int(-INFINITY);
}
__device__ void k_src_minmax(const VB &vb, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, uint32_t n,
                             int *__restrict__ out6) {
    __shared__ float s_lds[6][8];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = vb.bx * blockDim.x + threadIdx.x; i < n; i += vb.gx * blockDim.x) {
        const float v[3] = {x[i], y[i], z[i]};
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
    }
    block_minmax_commit<3>(mn, mx, out6, s_lds);
}
__device__ void k_src_keys(const VB &vb, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, uint32_t n,
                           const int *__restrict__ bbox, float inv, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mnx = ordered_float(bbox[0]), mny = ordered_float(bbox[1]), mnz = ordered_float(bbox[2]);
    // 10 bits per axis (coarsened if the extent needs more): 8-bit block coordinates + 2 local bits
    int cx = (int)((x[i] - mnx) * inv), cy = (int)((y[i] - mny) * inv), cz = (int)((z[i] - mnz) * inv);
    const float ext = fmaxf(fmaxf(ordered_float(bbox[3]) - mnx, ordered_float(bbox[4]) - mny), ordered_float(bbox[5]) - mnz) * inv;
    int sh = 0;
    while ((ext / (float)(1 << sh)) >= 1023.f) ++sh;
    cx >>= sh; cy >>= sh; cz >>= sh;
    cx = min(max(cx, 0), 1023); cy = min(max(cy, 0), 1023); cz = min(max(cz, 0), 1023);
    // the block alone is the key (24 bits = three radix passes instead of the four that 30 bits took): inside a block of 4 x 4 x 4
    // cells the points keep their order -- any order gives the same counts, the sort only makes a wavefront's probes local
    keys[i] = (uint32_t)(cx >> 2) | ((uint32_t)(cy >> 2) << 8) | ((uint32_t)(cz >> 2) << 16);
    vals[i] = i;
}
__device__ void k_src_gather(const VB &vb, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, uint32_t n,
                             const uint32_t *__restrict__ perm, float *__restrict__ out) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = perm[i];
    out[i] = x[p]; out[(size_t)n + i] = y[p]; out[2 * (size_t)n + i] = z[p];
}

// does the coarse sphere of candidate k contain any target point?  (util.h:621-625)
__device__ void k_sphere_any(const VB &vb, const float4 *__restrict__ tgt, uint32_t n_t,
                                                    const float *__restrict__ centers, uint32_t K, float R2,
                                                    uint32_t *__restrict__ any) {
    extern __shared__ float s_cc[];
    for (uint32_t i = threadIdx.x; i < K * 3; i += blockDim.x) s_cc[i] = centers[i];
    __syncthreads();
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    const bool live = i < n_t;
    const float4 t4 = live ? tgt[i] : make_float4(0, 0, 0, 0);
    const f3 t(t4.x, t4.y, t4.z);
    const int lane = threadIdx.x & 63;
    for (uint32_t k = 0; k < K; ++k) {
        const f3 c(s_cc[3 * k], s_cc[3 * k + 1], s_cc[3 * k + 2]);
        bool in = live && flann_d2(c, t) < R2;
        if (__ballot(in) && lane == 0) any[k] = 1u;
    }
}

void overlap_sort_source(plade_ctx *ctx, OverlapWork &work, const float *d_sx, const float *d_sy, const float *d_sz, uint32_t n_s,
                         float cell) {
    if (!n_s) return;
    // any order gives the same counts; this one makes a wavefront's probes local
    work.bbox.ensure(8);
    launch<k_init_minmax6, 64>(ctx, dim3(1), 0, work.bbox.p);
    launch<k_src_minmax, 256>(ctx, dim3(std::min(cdiv(n_s, 256), 512u)), 0, d_sx, d_sy, d_sz, n_s,
                       work.bbox.p);
    work.keys.ensure(n_s); work.keys2.ensure(n_s); work.vals.ensure(n_s); work.vals2.ensure(n_s);
    work.sorted.ensure(3 * (size_t)n_s + 4);
    launch<k_src_keys, 256>(ctx, dim3(cdiv(n_s, 256)), 0, d_sx, d_sy, d_sz, n_s, work.bbox.p, 1.f / cell,
                       work.keys.p, work.vals.p);
    sort_pairs_u32(ctx, work.keys.p, work.keys2.p, work.vals.p, work.vals2.p, n_s, 24);
    launch<k_src_gather, 256>(ctx, dim3(cdiv(n_s, 256)), 0, d_sx, d_sy, d_sz, n_s, work.vals2.p,
                       work.sorted.p);
    HIP_TRY(hipGetLastError());
}

void overlap_counts(plade_ctx *ctx, OverlapWork &work, const float *d_sx, const float *d_sy, const float *d_sz, uint32_t n_s,
                    const TargetGrid &grid, const float *d_T, const float *d_centers, uint32_t K, float src_radius,
                    float inlier_dist, int32_t *d_counts, uint32_t *d_any, bool counts_are_zero) {
    if (counts_are_zero) {}
    else if (reinterpret_cast<uint32_t *>(d_counts) + K == d_any) ctx->fill_async(d_counts, 0, (size_t)K * 8);   // one block
    else {
        ctx->fill_async(d_counts, 0, (size_t)K * 4);
        ctx->fill_async(d_any, 0, (size_t)K * 4);
    }
    if (K == 0 || grid.n == 0) return;
    const float R2 = pcl_r2((double)src_radius), r2 = pcl_r2((double)inlier_dist);
    GridParams g{grid.gp.mnx, grid.gp.mny, grid.gp.mnz, grid.gp.inv, grid.gp.dx, grid.gp.dy, grid.gp.dz};
    // sphere test in chunks of <= 2048 candidates (LDS)
    for (uint32_t k0 = 0; k0 < K; k0 += 2048) {
        uint32_t kc = std::min(2048u, K - k0);
        launch<k_sphere_any, 256>(ctx, dim3(cdiv(grid.n, 256)), kc * 12, grid.sorted.p, grid.n,
                           d_centers + (size_t)k0 * 3, kc, R2, d_any + k0);
    }
    if (n_s) {
        // chunks of 2 candidates while there are few (the default <= 201, of which ~20 reach this stage: the items must fill the
        // GPU), of OV_KCH from a few hundred candidates on (BASELINE configs[4]: 10^4)
        uint32_t ch = K <= 64 ? 2u : (K <= 512 ? 4u : (uint32_t)OV_KCH);
        if (grid.dense && K > 512) ch = 8u;   // (stress size, 2 launches of: chunks of 1 / 2 / 4 / 8 / 16 candidates: 178 / 160 / 157 / 150 / 160 ms)
        const uint64_t nitems = (uint64_t)cdiv(n_s, OV_TPB) * cdiv(K, ch);
        // (in a group the launch is merged with those of up to seven other pairs: a quarter of the workgroups per pair fill the
        //  part as well, and each of them stages the block mask -- up to 48 KB -- for four times the items)
        // (dense form: the kernel waits on memory three quarters of its time, so what counts is that every SIMD holds its five
        //  wavefronts from the first to the last item -- 1 280 persistent workgroups on the part's 256 CUs, shared out over the pairs
        //  of a merged launch; 2 048 were 1.6 rounds)
        const uint32_t fill = 256u * OVD_W / (uint32_t)std::max(1, ctx->comb ? ctx->comb->size() : 1);
        const uint32_t wgs = (uint32_t)std::min<uint64_t>(nitems, grid.dense ? std::max(64u, fill) : (ctx->comb ? 512u : 2048u));
        const uint32_t per = (uint32_t)((nitems + wgs - 1) / wgs);
        // algorithmic bytes: K * n_s * 12 B source stream + n_t * 12 B target
        ctx->ev_begin("overlap", (double)K * n_s * 12.0 + (double)grid.n * 12.0);
        PLADE_REQUIRE(grid.compact, PLADE_EINVAL, "overlap: the target grid needs the compact occupancy index");
        if (grid.dense) {
            launch<k_overlap_dense, OV_TPB, OVD_W>(ctx, dim3(cdiv(nitems, per)), grid.mask_words * 4, d_sx, d_sy, d_sz, n_s, grid.sorted.p,
                               grid.row_start.p, grid.near_mask.p, grid.mask_words, grid.mask_shift,
                               grid.row_occ_on ? (const uint32_t *)grid.row_occ.p : (const uint32_t *)nullptr, g, d_T, d_centers, K, R2, r2, d_counts,
                               ch, per);
            ctx->ev_end();
            HIP_TRY(hipGetLastError());
            return;
        }
        const uint32_t nw = (uint32_t)((grid.ncells + 63) / 64);
        const uint32_t mask_words = (nw + 31) / 32 * 4 <= OV_MASK_MAX ? (nw + 63) / 64 * 2 : 0u;   // whole 64-bit words of occ_blk
        launch<k_overlap, OV_TPB, 5>(ctx, dim3(cdiv(nitems, per)), mask_words * 4, d_sx, d_sy, d_sz, n_s, grid.sorted.p,
                           grid.occ_bits.p, grid.occ_rank.p, grid.occ_start.p, grid.cell_first.p, reinterpret_cast<const uint32_t *>(grid.occ_blk.p), mask_words, g,
                           d_T, d_centers, K, R2, r2, d_counts, ch, per);
        ctx->ev_end();
    }
    HIP_TRY(hipGetLastError());
}

__global__ void k_deinterleave3(const float *__restrict__ xyz, uint32_t n, float *__restrict__ x, float *__restrict__ y,
                                float *__restrict__ z) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = xyz[3 * (size_t)i]; y[i] = xyz[3 * (size_t)i + 1]; z[i] = xyz[3 * (size_t)i + 2];
}

void deinterleave3(plade_ctx *ctx, const float *d_xyz, uint32_t n, float *d_x, float *d_y, float *d_z) {
    if (!n) return;
    launch_raw(ctx, k_deinterleave3, dim3(cdiv(n, 256)), dim3(256), 0, d_xyz, n, d_x, d_y, d_z);
    HIP_TRY(hipGetLastError());
}

}  // namespace plade

using namespace plade;

// ---- C ABI: seam S3 ------------------------------------------------------------------------
extern "C" int plade_overlap_counts(plade_ctx *ctx, const float *src_ds, uint32_t n_s, const float *tgt_ds,
                                    uint32_t n_t, const float *T, uint32_t k, const float *centers, float src_radius,
                                    float inlier_dist, int32_t *counts) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(src_ds && tgt_ds && T && centers && counts, PLADE_EINVAL, "plade_overlap_counts: null argument");
        DBuf<float> d_src, d_tgt, d_soa, d_T, d_c;
        DBuf<int32_t> d_counts;
        DBuf<uint32_t> d_any;
        d_src.ensure((size_t)n_s * 3 + 4); d_tgt.ensure((size_t)n_t * 3 + 4); d_soa.ensure((size_t)n_s * 3 + 4);
        d_T.ensure((size_t)k * 16 + 4); d_c.ensure((size_t)k * 3 + 4); d_counts.ensure(k + 1); d_any.ensure(k + 1);
        ctx->h2d(d_src.p, src_ds, (size_t)n_s * 12);
        ctx->h2d(d_tgt.p, tgt_ds, (size_t)n_t * 12);
        ctx->h2d(d_T.p, T, (size_t)k * 64);
        ctx->h2d(d_c.p, centers, (size_t)k * 12);
        deinterleave3(ctx, d_src.p, n_s, d_soa.p, d_soa.p + n_s, d_soa.p + 2 * (size_t)n_s);
        TargetGrid grid;
        grid.build(ctx, d_tgt.p, n_t, 3, inlier_dist, nullptr, nullptr, true);
        OverlapWork ow;
        if (n_t && n_s) overlap_sort_source(ctx, ow, d_soa.p, d_soa.p + n_s, d_soa.p + 2 * (size_t)n_s, n_s, 1.f / grid.gp.inv);
        else ow.sorted.ensure(3 * (size_t)n_s + 4);
        overlap_counts(ctx, ow, ow.sorted.p, ow.sorted.p + n_s, ow.sorted.p + 2 * (size_t)n_s, n_s, grid, d_T.p, d_c.p, k, src_radius,
                       inlier_dist, d_counts.p, d_any.p);
        std::vector<uint32_t> any(k);
        ctx->d2h(counts, d_counts.p, (size_t)k * 4);
        ctx->d2h(any.data(), d_any.p, (size_t)k * 4);
        ctx->sync();
        for (uint32_t i = 0; i < k; ++i)
            if (!any[i]) counts[i] = -1;
        return PLADE_OK;
    });
}
