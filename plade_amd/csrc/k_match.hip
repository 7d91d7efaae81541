// plade_amd/csrc/k_match.hip -- K5: all-pairs 8-D descriptor radius match.
//
// Reference: KdTreeSearchNDim<VectorXf,8>::find_neighbors(p, 0, 0.04, ...)
// (code/3rd_party/ann_1.1.2/include/ANN/ANN.h:978-1029; leaf test
// ann_1.1.2/src/kd_fix_rad_search.cpp:148-183; call site code/PLADE/util.cpp:163):
// coordinates widened to double, squared distance accumulated in dimension order, in range
// <=> dist <= double(float(r*r)); results ascending by distance.  Ties between value-identical
// distances come out in ascending target index here (ANN's order for exact ties is kd-tree
// traversal dependent: documented in DESIGN.md).
//
// GPU mapping: exact fp64 brute force.  One lane owns one query (8 doubles in VGPRs), target
// tiles are staged through LDS and read as wave-wide broadcasts; a count pass sizes the output,
// an exclusive scan places it, a fill pass writes (target, dist2) in ascending target order, and
// two stable radix sorts (by dist2 bits, then by query) give the reference order.
#include "match.h"
#include "prims.h"

namespace plade {

constexpr int MT_TPB = 64;    // one wave of queries per workgroup
constexpr int MT_TILE = 128;   // targets per LDS tile
constexpr int MT_CHUNK = 512;  // targets per block (grid.y)

template <bool FILL>
__device__ void k_match(const VB &vb, const float *__restrict__ qry, uint32_t dq,
                                                  const float *__restrict__ tgt, uint32_t dt, double sq_rad,
                                                  uint32_t nch, uint32_t chunk /* targets per blockIdx.y, multiple of MT_TILE */,
                                                  uint32_t *__restrict__ cnt /* dq*nch */,
                                                  const uint32_t *__restrict__ offs /* dq*nch */,
                                                  uint32_t *__restrict__ t_idx, double *__restrict__ d2_out,
                                                  uint32_t *__restrict__ q_idx, uint32_t *__restrict__ info) {
    __shared__ float s_t[MT_TILE][8];
    if (!FILL && vb.bx == 0 && vb.by == 0 && threadIdx.x == 0) {   // instead of two memset commands
        cnt[(size_t)dq * nch] = 0u;       // the scan's extra slot
        info[0] = 0u; info[1] = 0u;       // k_query_offsets: total, longest list (atomicMax)
    }
    const uint32_t q = vb.bx * MT_TPB + threadIdx.x;
    const bool live = q < dq;
    double qd[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) qd[d] = live ? (double)qry[(size_t)q * 8 + d] : 0.0;
    const uint32_t t_begin = vb.by * chunk;
    const uint32_t t_end = min(dt, t_begin + chunk);
    uint32_t c = 0;
    uint32_t wpos = (FILL && live) ? offs[(size_t)q * nch + vb.by] : 0u;
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += MT_TILE) {
        const uint32_t tn = min((uint32_t)MT_TILE, t_end - t0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tn * 8; i += MT_TPB) (&s_t[0][0])[i] = tgt[(size_t)t0 * 8 + i];
        __syncthreads();
        if (live)
            for (uint32_t j = 0; j < tn; ++j) {
                double dist = 0.0;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    double t = qd[d] - (double)s_t[j][d];
                    dist += t * t;
                }
                if (dist <= sq_rad) {
                    if (FILL) {
                        t_idx[wpos] = t0 + j;
                        d2_out[wpos] = dist;
                        q_idx[wpos] = q;
                        ++wpos;
                    } else ++c;
                }
            }
    }
    if (!FILL && live) cnt[(size_t)q * nch + vb.by] = c;
}

__device__ void k_d2_keys(const VB &vb, const double *__restrict__ d2, uint32_t m, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= m) return;
    keys[i] = (uint64_t)__double_as_longlong(d2[i]);  // non-negative doubles order as their bit patterns
    vals[i] = i;
}
__global__ void k_iota_u32(uint32_t *__restrict__ p, uint32_t m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) p[i] = i;
}
__global__ void k_d2_keys_perm(const double *__restrict__ d2, const uint32_t *__restrict__ perm, uint32_t m,
                               uint64_t *__restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) keys[i] = (uint64_t)__double_as_longlong(d2[perm[i]]);
}
__device__ void k_gather_u32(const VB &vb, const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t m,
                             uint32_t *__restrict__ dst) {
    uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i < m) dst[i] = src[perm[i]];
}
__device__ void k_gather_final(const VB &vb, const uint32_t *__restrict__ t_idx, const double *__restrict__ d2,
                               const uint32_t *__restrict__ perm, uint32_t m, uint32_t *__restrict__ t_out,
                               double *__restrict__ d2_out) {
    uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t p = perm[i];
    t_out[i] = t_idx[p];
    d2_out[i] = d2[p];
}
// per-query offsets + {total, longest per-query list} for the host
__device__ void k_query_offsets(const VB &vb, const uint32_t *__restrict__ offs, uint32_t dq, uint32_t nch, int64_t *__restrict__ out,
                                uint32_t *__restrict__ info /* [0] total, [1] max list (zeroed) */) {
    uint32_t q = vb.bx * blockDim.x + threadIdx.x;
    const uint32_t total = offs[(size_t)dq * nch];
    uint32_t len = 0;
    if (q < dq) {
        const uint32_t b = offs[(size_t)q * nch], e = offs[(size_t)(q + 1) * nch];
        out[q] = b;
        len = e - b;
    }
    for (int d = 32; d >= 1; d >>= 1) len = max(len, (uint32_t)__shfl_xor((int)len, d, 64));
    if ((threadIdx.x & 63) == 0 && len > info[1]) atomicMax(&info[1], len);   // one (conditional) atomic per wavefront
    if (q == dq) { out[q] = total; info[0] = total; }
}

// Order inside each query's list: (dist2, target index).  The fill pass already wrote the lists query by
// query in ascending target order, so each entry's final place is its rank by (dist2, position) inside
// its own list: one workgroup per query, the list in LDS, O(len^2 / 256) compares (lists are tens to a few
// thousand entries).  Ties are broken on the target index itself, so the fill order inside a list is free.
constexpr uint32_t RANK_MAX_LIST = 4096;
__device__ void k_rank_lists(const VB &vb, const int64_t *__restrict__ offsets, uint32_t dq,
                                                    const uint32_t *__restrict__ t_raw, const double *__restrict__ d2_raw,
                                                    uint32_t *__restrict__ t_out, double *__restrict__ d2_out, uint32_t cap) {
    // one workgroup per query: its list staged in LDS -- as much of it as the LONGEST list of the call needs (`cap` entries of
    // 12 bytes, dynamic: with room for 4096 entries reserved per workgroup only three fitted a CU, for lists of a few dozen)
    extern __shared__ double s_rank[];
    double *s_d2 = s_rank;
    uint32_t *s_t = reinterpret_cast<uint32_t *>(s_rank + cap);
    const uint32_t q = vb.bx;
    if (q >= dq) return;
    const uint32_t b = (uint32_t)offsets[q], e = (uint32_t)offsets[q + 1], len = e - b;
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) { s_d2[i] = d2_raw[b + i]; s_t[i] = t_raw[b + i]; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
        const double di = s_d2[i];
        const uint32_t ti = s_t[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < len; ++j) {
            const double dj = s_d2[j];
            rank += (dj < di || (dj == di && s_t[j] < ti)) ? 1u : 0u;   // ties: ascending target index
        }
        t_out[b + rank] = ti;
        d2_out[b + rank] = di;
    }
}

// ---- windowed variant for large tables ----------------------------------------------------------------
// |q - t|^2 <= r^2 needs |q[0] - t[0]| <= r: with both tables sorted by the first component (the
// closest-point distance of the line pair) a group of 64 neighbouring queries only meets the targets of one
// contiguous window.  Same exact fp64 test as k_match; only the enumeration differs.
__device__ __forceinline__ uint32_t ord_u32(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

__global__ void k_len_keys(const float *__restrict__ desc, uint32_t n, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = ord_u32(desc[8 * (size_t)i]); vals[i] = i; }
}
__global__ void k_gather_desc(const float *__restrict__ desc, const uint32_t *__restrict__ perm, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 2) return;   // two float4 per descriptor
    const float4 *src = reinterpret_cast<const float4 *>(desc + 8 * (size_t)perm[i >> 1]) + (i & 1);
    reinterpret_cast<float4 *>(out)[i] = *src;
}
// window [lo, lo + cnt) of sorted targets for every group of MT_TPB sorted queries; info[1] = max window
__global__ void k_windows(const float *__restrict__ q_sorted, uint32_t dq, const uint32_t *__restrict__ t_keys_sorted, uint32_t dt,
                          float radius, uint32_t *__restrict__ w_lo, uint32_t *__restrict__ w_cnt, uint32_t *__restrict__ info) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, ng = (dq + MT_TPB - 1) / MT_TPB;
    if (g >= ng) return;
    const float qmin = q_sorted[8 * (size_t)(g * MT_TPB)];
    const float qmax = q_sorted[8 * (size_t)min(dq - 1, g * MT_TPB + MT_TPB - 1)];
    // conservative float bounds of [qmin - r, qmax + r] (the exact fp64 test decides membership)
    const float pad = radius * 1.0001f + 1e-30f;
    const uint32_t klo = ord_u32(qmin - pad - fabsf(qmin) * 1e-6f), khi = ord_u32(qmax + pad + fabsf(qmax) * 1e-6f);
    uint32_t lo = 0, hi = dt;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (t_keys_sorted[mid] < klo) lo = mid + 1; else hi = mid; }
    uint32_t lo2 = lo, hi2 = dt;
    while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2) >> 1; if (t_keys_sorted[mid] <= khi) lo2 = mid + 1; else hi2 = mid; }
    w_lo[g] = lo;
    w_cnt[g] = lo2 - lo;
    if (lo2 - lo > info[1]) atomicMax(&info[1], lo2 - lo);
}
template <bool FILL>
__global__ __launch_bounds__(MT_TPB) void k_match_win(const float *__restrict__ qry, uint32_t dq, const float *__restrict__ tgt,
                                                      const uint32_t *__restrict__ w_lo, const uint32_t *__restrict__ w_cnt,
                                                      double sq_rad, uint32_t nch, uint32_t *__restrict__ cnt,
                                                      const uint32_t *__restrict__ base /* dq*nch: final positions */,
                                                      const uint32_t *__restrict__ q_perm, const uint32_t *__restrict__ t_perm,
                                                      uint32_t *__restrict__ t_idx, double *__restrict__ d2_out,
                                                      uint32_t *__restrict__ q_idx) {
    __shared__ float s_t[MT_TILE][8];
    const uint32_t q = blockIdx.x * MT_TPB + threadIdx.x;
    const bool live = q < dq;
    const uint32_t wl = w_lo[blockIdx.x], wc = w_cnt[blockIdx.x];
    const uint32_t t_begin = wl + blockIdx.y * MT_CHUNK;
    const uint32_t t_end = min(wl + wc, t_begin + MT_CHUNK);
    if (blockIdx.y * MT_CHUNK >= wc) {   // chunk beyond this group's window
        if (!FILL && live) cnt[(size_t)q * nch + blockIdx.y] = 0;
        return;
    }
    double qd[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) qd[d] = live ? (double)qry[(size_t)q * 8 + d] : 0.0;
    uint32_t c = 0;
    uint32_t wpos = (FILL && live) ? base[(size_t)q * nch + blockIdx.y] : 0u;
    const uint32_t q_orig = (FILL && live) ? q_perm[q] : 0u;
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += MT_TILE) {
        const uint32_t tn = min((uint32_t)MT_TILE, t_end - t0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tn * 8; i += MT_TPB) (&s_t[0][0])[i] = tgt[(size_t)t0 * 8 + i];
        __syncthreads();
        if (live)
            for (uint32_t j = 0; j < tn; ++j) {
                double dist = 0.0;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    double t = qd[d] - (double)s_t[j][d];
                    dist += t * t;
                }
                if (dist <= sq_rad) {
                    if (FILL) {
                        t_idx[wpos] = t_perm[t0 + j];
                        d2_out[wpos] = dist;
                        q_idx[wpos] = q_orig;
                        ++wpos;
                    } else ++c;
                }
            }
    }
    if (!FILL && live) cnt[(size_t)q * nch + blockIdx.y] = c;
}
// per sorted query (of a slab): its total, scattered to the original query index
__global__ void k_row_totals(const uint32_t *__restrict__ offs, uint32_t dq, uint32_t nch, const uint32_t *__restrict__ q_perm,
                             uint32_t *__restrict__ row_tot /* original order */) {
    const uint32_t sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq < dq) row_tot[q_perm[sq]] = offs[(size_t)(sq + 1) * nch] - offs[(size_t)sq * nch];
}
// final position of every (sorted query, chunk) cell of a slab
__global__ void k_win_bases(const uint32_t *__restrict__ offs, uint32_t dq, uint32_t nch, const uint32_t *__restrict__ q_perm,
                            const uint32_t *__restrict__ row_off, uint32_t *__restrict__ base) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)dq * nch) {
        const uint32_t sq = (uint32_t)(i / nch);
        base[i] = row_off[q_perm[sq]] + (offs[i] - offs[(size_t)sq * nch]);
    }
}
// offsets of the original queries; total and longest list for the host
__global__ void k_win_offsets(const uint32_t *__restrict__ row_off, uint32_t dq, int64_t *__restrict__ out, uint32_t *__restrict__ info) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= dq) out[i] = row_off[i];
    uint32_t len = i < dq ? row_off[i + 1] - row_off[i] : 0u;
    for (int d = 32; d >= 1; d >>= 1) len = max(len, (uint32_t)__shfl_xor((int)len, d, 64));
    if ((threadIdx.x & 63) == 0 && len > info[1]) atomicMax(&info[1], len);   // one conditional atomic per wavefront
    if (i == dq) info[0] = row_off[dq];
}

uint64_t MatchResult::run_windowed(plade_ctx *ctx, const float *d_qry, uint32_t dq, const float *d_tgt, uint32_t dt, float radius) {
    const float sq_rad_f = radius * radius;
    const double sq_rad = sq_rad_f;
    // both tables by ascending first component
    wk_a.ensure(std::max(dq, dt)); wk_b.ensure(std::max(dq, dt)); wv_a.ensure(std::max(dq, dt));
    q_perm.ensure(dq); t_perm.ensure(dt); q_sorted.ensure(8 * (size_t)dq + 8); t_sorted.ensure(8 * (size_t)dt + 8);
    launch_raw(ctx, k_len_keys, dim3(cdiv(dq, 256)), dim3(256), 0, d_qry, dq, wk_a.p, wv_a.p);
    sort_pairs_u32(ctx, wk_a.p, wk_b.p, wv_a.p, q_perm.p, dq, 32);
    launch_raw(ctx, k_gather_desc, dim3(cdiv(2 * (size_t)dq, 256)), dim3(256), 0, d_qry, q_perm.p, dq, q_sorted.p);
    launch_raw(ctx, k_len_keys, dim3(cdiv(dt, 256)), dim3(256), 0, d_tgt, dt, wk_a.p, wv_a.p);
    sort_pairs_u32(ctx, wk_a.p, wk_b.p, wv_a.p, t_perm.p, dt, 32);   // wk_b = sorted target keys
    launch_raw(ctx, k_gather_desc, dim3(cdiv(2 * (size_t)dt, 256)), dim3(256), 0, d_tgt, t_perm.p, dt, t_sorted.p);
    const uint32_t ng = cdiv(dq, MT_TPB);
    w_lo.ensure(ng); w_cnt.ensure(ng); info.ensure(2);
    ctx->fill_async(info.p, 0, 8);
    launch_raw(ctx, k_windows, dim3(cdiv(ng, 256)), dim3(256), 0, q_sorted.p, dq, wk_b.p, dt, radius, w_lo.p, w_cnt.p,
                       info.p);
    uint32_t h_info[2] = {0, 0};
    ctx->d2h(h_info, info.p, 8);
    ctx->sync();
    const uint32_t nch = std::max(1u, cdiv(h_info[1], MT_CHUNK));
    // The (query, chunk) cells of the count / fill passes are laid out per SLAB of consecutive sorted queries, at most
    // 2^26 cells (two u32 arrays: 512 MB) at a time: BASELINE configs[4] (~90 planes per cloud: 6e6 x 1.2e7 descriptors)
    // has ~2e9 of them.  With more than one slab the count pass runs twice (totals of all slabs first -- they place the
    // lists in the order of the ORIGINAL queries -- then again in front of each slab's fill pass).
    uint64_t cell_budget = 1ull << 26;
    if (ctx->params.match_cell_budget) cell_budget = std::max<uint64_t>((uint64_t)MT_TPB * nch, ctx->params.match_cell_budget);   // tests: small slabs
    PLADE_REQUIRE((uint64_t)MT_TPB * nch <= cell_budget, PLADE_ELIMIT, "match: a window of more than 5e8 targets");
    const uint32_t groups_per_slab = (uint32_t)std::min<uint64_t>(ng, cell_budget / ((uint64_t)MT_TPB * nch));
    const uint32_t n_slabs = cdiv(ng, groups_per_slab);
    const size_t slab_cells = (size_t)groups_per_slab * MT_TPB * nch;
    cnt.ensure(slab_cells + 1); offs.ensure(slab_cells + 1);
    row_tot.ensure((size_t)dq + 1); row_off.ensure((size_t)dq + 1);
    auto count_slab = [&](uint32_t s, uint32_t &q0, uint32_t &dqs, uint32_t &g0, uint32_t &ngs) {
        g0 = s * groups_per_slab;
        ngs = std::min(groups_per_slab, ng - g0);
        q0 = g0 * MT_TPB;
        dqs = std::min(dq - q0, ngs * MT_TPB);
        const size_t ncnt = (size_t)dqs * nch;
        launch_raw(ctx, k_match_win<false>, dim3(ngs, nch), dim3(MT_TPB), 0, q_sorted.p + 8 * (size_t)q0, dqs, t_sorted.p,
                           w_lo.p + g0, w_cnt.p + g0, sq_rad, nch, cnt.p, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr, (uint32_t *)nullptr);
        ctx->fill_async(cnt.p + ncnt, 0, 4);
        exclusive_scan_u32(ctx, cnt.p, offs.p, ncnt + 1);
    };
    for (uint32_t s = 0; s < n_slabs; ++s) {
        uint32_t q0, dqs, g0, ngs;
        count_slab(s, q0, dqs, g0, ngs);
        launch_raw(ctx, k_row_totals, dim3(cdiv(dqs, 256)), dim3(256), 0, offs.p, dqs, nch, q_perm.p + q0, row_tot.p);
    }
    ctx->fill_async(row_tot.p + dq, 0, 4);
    exclusive_scan_u32(ctx, row_tot.p, row_off.p, (size_t)dq + 1);
    ctx->fill_async(info.p, 0, 8);
    launch_raw(ctx, k_win_offsets, dim3(cdiv((size_t)dq + 1, 256)), dim3(256), 0, row_off.p, dq, offsets.p, info.p);
    ctx->d2h(h_info, info.p, 8);
    ctx->sync();
    total = h_info[0];
    const uint32_t max_list = h_info[1];
    if (total == 0) return 0;
    const uint32_t m = (uint32_t)total;
    t_raw.ensure(m); d2_raw.ensure(m); q_raw.ensure(m);
    for (uint32_t s = 0; s < n_slabs; ++s) {
        uint32_t q0, dqs, g0, ngs;
        if (n_slabs > 1) count_slab(s, q0, dqs, g0, ngs);   // one slab: its offsets are still in place
        else { g0 = 0; ngs = ng; q0 = 0; dqs = dq; }
        const size_t ncnt = (size_t)dqs * nch;
        // `cnt` is free again: it receives the final write positions
        launch_raw(ctx, k_win_bases, dim3(cdiv(ncnt, 256)), dim3(256), 0, offs.p, dqs, nch, q_perm.p + q0, row_off.p, cnt.p);
        launch_raw(ctx, k_match_win<true>, dim3(ngs, nch), dim3(MT_TPB), 0, q_sorted.p + 8 * (size_t)q0, dqs, t_sorted.p,
                           w_lo.p + g0, w_cnt.p + g0, sq_rad, nch, (uint32_t *)nullptr, cnt.p, q_perm.p + q0, t_perm.p, t_raw.p, d2_raw.p,
                           q_raw.p);
    }
    t_idx.ensure(m); dist2.ensure(m);
    q_idx_sorted = q_raw.p;
    if (max_list <= RANK_MAX_LIST) {
        const uint32_t cap = std::max(64u, (max_list + 63u) & ~63u);   // entries of LDS per workgroup
        launch<k_rank_lists, 256>(ctx, dim3(dq), cap * 12, offsets.p, dq, t_raw.p, d2_raw.p, t_idx.p, dist2.p, cap);
        HIP_TRY(hipGetLastError());
        return total;
    }
    // very long lists: stable sorts by target index, then dist2, then query
    k64a.ensure(m); k64b.ensure(m); v32a.ensure(m); v32b.ensure(m); k32a.ensure(m); k32b.ensure(m);
    launch_raw(ctx, k_iota_u32, dim3(cdiv(m, 256)), dim3(256), 0, v32a.p, m);
    int tbits = 1;
    while ((1ull << tbits) < dt) ++tbits;
    sort_pairs_u32(ctx, t_raw.p, k32b.p, v32a.p, v32b.p, m, tbits);                  // v32b: entries by target index
    launch_raw(ctx, k_d2_keys_perm, dim3(cdiv(m, 256)), dim3(256), 0, d2_raw.p, v32b.p, m, k64a.p);
    sort_pairs_u64(ctx, k64a.p, k64b.p, v32b.p, v32a.p, m, 64);                       // v32a: by (dist2, target)
    launch<k_gather_u32, 256>(ctx, dim3(cdiv(m, 256)), 0, q_raw.p, v32a.p, m, k32a.p);
    int qbits = 1;
    while ((1ull << qbits) < dq) ++qbits;
    sort_pairs_u32(ctx, k32a.p, k32b.p, v32a.p, v32b.p, m, qbits);                    // v32b: by (query, dist2, target)
    launch<k_gather_final, 256>(ctx, dim3(cdiv(m, 256)), 0, t_raw.p, d2_raw.p, v32b.p, m, t_idx.p,
                       dist2.p);
    q_idx_sorted = k32b.p;
    HIP_TRY(hipGetLastError());
    return total;
}

uint64_t MatchResult::run(plade_ctx *ctx, const float *d_qry, uint32_t dq, const float *d_tgt, uint32_t dt,
                          float radius) {
    total = 0;
    offsets.ensure((size_t)dq + 1);
    if (dq == 0) { ctx->fill_async(offsets.p, 0, 8); return 0; }
    if (radius < 0.f || dt == 0) { ctx->fill_async(offsets.p, 0, ((size_t)dq + 1) * 8); return 0; }
    const float sq_rad_f = radius * radius;  // ANN.h:987 `float sqRad = radius*radius`
    const double sq_rad = sq_rad_f;
    // large tables (or plade_params.match_window = 1): enumerate only inside the length windows
    {
        const bool force = ctx->params.match_window > 0, never = ctx->params.match_window < 0;
        if (!never && (force || (double)dq * (double)dt > 2.0e10)) return run_windowed(ctx, d_qry, dq, d_tgt, dt, radius);
    }
    // A workgroup is one wave of 64 queries against one chunk of targets (fp64 distances, ~50 cycles per target): with
    // 512 targets per chunk the usual table (2 000 x 6 000 descriptors) is 420 waves of 50 us on a part with 1024 SIMDs;
    // chunks of one LDS tile give four times the waves at a quarter of the length (96 + 82 -> see profiles/r3).  The
    // (query, chunk) count table grows with it, so large tables keep the long chunks.
    const uint32_t chunk = (size_t)dq * cdiv(dt, MT_TILE) <= (1u << 22) ? (uint32_t)MT_TILE : (uint32_t)MT_CHUNK;
    const uint32_t nch = cdiv(dt, chunk);
    const size_t ncnt = (size_t)dq * nch;
    PLADE_REQUIRE(ncnt < (1ull << 31), PLADE_ELIMIT, "match: too many (query, chunk) cells");
    cnt.ensure(ncnt + 1); offs.ensure(ncnt + 1);
    dim3 grid(cdiv(dq, MT_TPB), nch);
    info.ensure(2);
    launch<k_match<false>, MT_TPB>(ctx, grid, 0, d_qry, dq, d_tgt, dt, sq_rad, nch, chunk, cnt.p,
                       (const uint32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr, (uint32_t *)nullptr, info.p);
    exclusive_scan_u32(ctx, cnt.p, offs.p, ncnt + 1);
    launch<k_query_offsets, 256>(ctx, dim3(cdiv(dq + 1, 256)), 0, offs.p, dq, nch, offsets.p, info.p);
    uint32_t h_info[2] = {0, 0};
    ctx->d2h(h_info, info.p, 8);
    ctx->sync();
    const uint32_t tot32 = h_info[0], max_list = h_info[1];
    total = tot32;
    if (total == 0) return 0;
    const uint32_t m = tot32;
    t_raw.ensure(m); d2_raw.ensure(m); q_raw.ensure(m);
    launch<k_match<true>, MT_TPB>(ctx, grid, 0, d_qry, dq, d_tgt, dt, sq_rad, nch, chunk, cnt.p, offs.p,
                       t_raw.p, d2_raw.p, q_raw.p, (uint32_t *)nullptr);
    t_idx.ensure(m); dist2.ensure(m);
    q_idx_sorted = q_raw.p;   // lists are contiguous per query
    if (max_list <= RANK_MAX_LIST) {
        const uint32_t cap = std::max(64u, (max_list + 63u) & ~63u);   // entries of LDS per workgroup
        launch<k_rank_lists, 256>(ctx, dim3(dq), cap * 12, offsets.p, dq, t_raw.p, d2_raw.p, t_idx.p,
                           dist2.p, cap);
        HIP_TRY(hipGetLastError());
        return total;
    }
    // very long lists: a stable sort by dist2 followed by a stable sort by query
    k64a.ensure(m); k64b.ensure(m); v32a.ensure(m); v32b.ensure(m); k32a.ensure(m); k32b.ensure(m);
    launch<k_d2_keys, 256>(ctx, dim3(cdiv(m, 256)), 0, d2_raw.p, m, k64a.p, v32a.p);
    sort_pairs_u64(ctx, k64a.p, k64b.p, v32a.p, v32b.p, m, 64);
    launch<k_gather_u32, 256>(ctx, dim3(cdiv(m, 256)), 0, q_raw.p, v32b.p, m, k32a.p);
    int qbits = 1;
    while ((1ull << qbits) < dq) ++qbits;
    sort_pairs_u32(ctx, k32a.p, k32b.p, v32b.p, v32a.p, m, qbits);
    launch<k_gather_final, 256>(ctx, dim3(cdiv(m, 256)), 0, t_raw.p, d2_raw.p, v32a.p, m,
                       t_idx.p, dist2.p);
    q_idx_sorted = k32b.p;
    HIP_TRY(hipGetLastError());
    return total;
}

}  // namespace plade

using namespace plade;

// ---- C ABI: seam S2 ------------------------------------------------------------------------
extern "C" int plade_match_descriptors(plade_ctx *ctx, const float *src, uint32_t ds, const float *tgt, uint32_t dt,
                                       float radius, int64_t *offsets, uint32_t *t_idx, double *dist2, uint64_t cap,
                                       uint64_t *n_pairs) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(offsets && n_pairs && (src || !ds) && (tgt || !dt), PLADE_EINVAL, "plade_match_descriptors: null argument");
        DBuf<float> d_q, d_t;
        d_q.ensure((size_t)ds * 8 + 8); d_t.ensure((size_t)dt * 8 + 8);
        if (ds) ctx->h2d(d_q.p, src, (size_t)ds * 32);
        if (dt) ctx->h2d(d_t.p, tgt, (size_t)dt * 32);
        MatchResult r;
        uint64_t total = r.run(ctx, d_q.p, ds, d_t.p, dt, radius);
        *n_pairs = total;
        ctx->d2h(offsets, r.offsets.p, ((size_t)ds + 1) * 8);
        uint64_t w = total < cap ? total : cap;
        if (w && t_idx) ctx->d2h(t_idx, r.t_idx.p, w * 4);
        if (w && dist2) ctx->d2h(dist2, r.dist2.p, w * 8);
        ctx->sync();
        return (total > cap && (t_idx || dist2) && cap) ? PLADE_ECAP : PLADE_OK;
    });
}
