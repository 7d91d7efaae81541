// plade_amd/csrc/k_lines.hip -- K4: line-pair tables and "22" descriptors.
//
// Reference: ConstructPairLinesKdTree (code/PLADE/util.cpp:706-1165; only the "22" table is ever
// matched), the source-side twin code/PLADE/plade.cpp:451-482 + :511-521 and the
// query descriptors of MatchingLines (code/PLADE/util.cpp:133-168), built on
// ComputeNearstTwoPointsOfTwo3DLine (util.cpp:1167-1229) and
// ComputeDescriptorVectorForPairLines (util.cpp:533-577).
//
// One lane per ordered line pair (L <= P(P-1)/2, so L^2 is at most a few 1e5..1e7 pairs); the
// sequential in-place re-normalisation of the stored direction vectors is reproduced from the
// per-line iterate table (see stages.h).
#include "stages.h"
#include "prims.h"

namespace plade {

struct LinesView {
    const float *pt;       // L x 3
    const int32_t *sp;     // L x 2
    const float *iter;     // iterates
    const int32_t *it;     // L x 4: off, len, start, period
    const float *normals;  // P x 3
    uint32_t L;
};

__device__ __forceinline__ f3 line_iter(const LinesView &v, uint32_t line, int k /* normalisation count >= 3 */) {
    const int off = v.it[4 * line], len = v.it[4 * line + 1], start = v.it[4 * line + 2], period = v.it[4 * line + 3];
    int idx = k - 3;
    if (idx >= len) idx = start + (idx - start) % period;
    const float *p = v.iter + 3 * (size_t)(off + idx);
    return f3(p[0], p[1], p[2]);
}

__global__ __launch_bounds__(256) void k_pair_table(LinesView v, float scale, float angle_thresh, int target,
                                                    uint32_t *__restrict__ flags, float *__restrict__ desc,
                                                    float *__restrict__ lv1, float *__restrict__ lv2,
                                                    float *__restrict__ p1out) {
    const uint32_t L = v.L;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)L * L) return;
    const uint32_t i = (uint32_t)(idx / L), j = (uint32_t)(idx % L);
    flags[idx] = 0u;
    if (idx == 0) flags[(size_t)L * L] = 0u;   // the scan's extra slot (was a memset command of its own)
    if (i == j) return;
    if (!target && i > j) return;
    const f3 pti(v.pt[3 * i], v.pt[3 * i + 1], v.pt[3 * i + 2]), ptj(v.pt[3 * j], v.pt[3 * j + 1], v.pt[3 * j + 2]);
    f3 p1, p2, vi, vj;
    double len;
    bool ok;
    if (i < j) {
        const f3 ci = line_iter(v, i, 3 + (int)j), cj = line_iter(v, j, 4 + (int)i);
        ok = closest_points(ci, pti, cj, ptj, p1, p2, len);
        if (target) { vi = ci; vj = cj; }
        else { vi = line_iter(v, i, 2 + (int)L); vj = line_iter(v, j, 2 + (int)L); }
    } else {
        // copied from entry (j, i) (util.cpp:784-788)
        f3 q1, q2;
        ok = closest_points(line_iter(v, j, 3 + (int)i), ptj, line_iter(v, i, 4 + (int)j), pti, q1, q2, len);
        p1 = q2; p2 = q1;
        vi = line_iter(v, i, 3 + (int)i);
        vj = line_iter(v, j, 2 + (int)L);
    }
    if (!ok) len = -1;
    len = len / scale;
    if (fabsf(dot_e(vi, vj)) > angle_thresh) return;
    const int a0 = v.sp[2 * i], a1 = v.sp[2 * i + 1], b0 = v.sp[2 * j], b1 = v.sp[2 * j + 1];
    const float *N = v.normals;
    float d[8];
    f3 n1, n2;
    descriptor22(vi, vj, f3(N[3 * a0], N[3 * a0 + 1], N[3 * a0 + 2]), f3(N[3 * a1], N[3 * a1 + 1], N[3 * a1 + 2]),
                 f3(N[3 * b0], N[3 * b0 + 1], N[3 * b0 + 2]), f3(N[3 * b1], N[3 * b1 + 1], N[3 * b1 + 2]), d, n1, n2);
    d[0] = (float)len;
    flags[idx] = 1u;
    for (int k = 0; k < 8; ++k) desc[idx * 8 + k] = d[k];
    lv1[idx * 3] = n1.x; lv1[idx * 3 + 1] = n1.y; lv1[idx * 3 + 2] = n1.z;
    lv2[idx * 3] = n2.x; lv2[idx * 3 + 1] = n2.y; lv2[idx * 3 + 2] = n2.z;
    p1out[idx * 3] = p1.x; p1out[idx * 3 + 1] = p1.y; p1out[idx * 3 + 2] = p1.z;
    (void)p2;
}

__global__ void k_scatter_pairs(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, size_t n,
                                const float *__restrict__ desc, const float *__restrict__ lv1,
                                const float *__restrict__ lv2, const float *__restrict__ p1, float *__restrict__ o_desc,
                                float *__restrict__ o_lv1, float *__restrict__ o_lv2, float *__restrict__ o_p1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const size_t o = pos[i];
    for (int k = 0; k < 8; ++k) o_desc[o * 8 + k] = desc[i * 8 + k];
    for (int k = 0; k < 3; ++k) { o_lv1[o * 3 + k] = lv1[i * 3 + k]; o_lv2[o * 3 + k] = lv2[i * 3 + k]; o_p1[o * 3 + k] = p1[i * 3 + k]; }
}

__global__ void k_flag_positions(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, uint32_t n,
                                 uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) out[pos[i]] = i;
}

uint32_t compact_flags(plade_ctx *ctx, const uint32_t *d_flags, uint32_t n, DBuf<uint32_t> &pos, DBuf<uint32_t> &out_idx) {
    if (n == 0) return 0;
    pos.ensure((size_t)n + 1);
    // scan n + 1 entries so that pos[n] is the total (the flag array must have n + 1 slots, last = 0)
    exclusive_scan_u32(ctx, d_flags, pos.p, (size_t)n + 1);
    uint32_t total = 0;
    ctx->d2h(&total, pos.p + n, 4);
    ctx->sync();
    out_idx.ensure((size_t)total + 1);
    hipLaunchKernelGGL(k_flag_positions, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_flags, pos.p, n, out_idx.p);
    HIP_TRY(hipGetLastError());
    return total;
}

void build_pair_table(plade_ctx *ctx, const LineTableHost &lt, const float *normals, uint32_t P, float scale, bool target,
                      PairTableDev &out) {
    out.count = 0;
    const uint32_t L = lt.L;
    if (L == 0) return;
    const size_t n = (size_t)L * L;
    PLADE_REQUIRE(n < (1ull << 31), PLADE_ELIMIT, "line-pair table too large");
    // one staging buffer, one upload: pt (3L) | sp (2L) | iterates | it (4L) | normals (3P)
    const size_t o_pt = 0, o_sp = o_pt + 3 * (size_t)L, o_iter = o_sp + 2 * (size_t)L, o_it = o_iter + lt.iter.size(),
                 o_nrm = o_it + 4 * (size_t)L, words = o_nrm + 3 * (size_t)P;
    std::vector<uint32_t> blob(words);
    memcpy(&blob[o_pt], lt.pt.data(), 12 * (size_t)L);
    memcpy(&blob[o_sp], lt.sp.data(), 8 * (size_t)L);
    memcpy(&blob[o_iter], lt.iter.data(), 4 * lt.iter.size());
    {
        int32_t *it = reinterpret_cast<int32_t *>(&blob[o_it]);
        for (uint32_t l = 0; l < L; ++l) { it[4 * l] = lt.it_off[l]; it[4 * l + 1] = lt.it_len[l]; it[4 * l + 2] = lt.it_start[l]; it[4 * l + 3] = lt.it_period[l]; }
    }
    memcpy(&blob[o_nrm], normals, 12 * (size_t)P);
    uint32_t *d = out.d_blob.ensure(words + 4);
    const bool staged = ctx->h2d(d, blob.data(), 4 * words);
    out.all_desc.ensure(n * 8); out.all_lv1.ensure(n * 3); out.all_lv2.ensure(n * 3); out.all_p1.ensure(n * 3);
    out.flags.ensure(n + 1); out.pos.ensure(n + 1);
    LinesView v{reinterpret_cast<const float *>(d + o_pt), reinterpret_cast<const int32_t *>(d + o_sp),
                reinterpret_cast<const float *>(d + o_iter), reinterpret_cast<const int32_t *>(d + o_it),
                reinterpret_cast<const float *>(d + o_nrm), L};
    const float angle_thresh = (float)cos(10.0 / 180 * M_PI);  // util.cpp:773, plade.cpp:513
    hipLaunchKernelGGL(k_pair_table, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, v, scale, angle_thresh, target ? 1 : 0,
                       out.flags.p, out.all_desc.p, out.all_lv1.p, out.all_lv2.p, out.all_p1.p);
    exclusive_scan_u32(ctx, out.flags.p, out.pos.p, n + 1);
    if (staged && n <= (1u << 20)) {
        // the usual table (a few hundred lines): the compacted arrays are sized for all n pairs (68 B each) and the count
        // comes back with the caller's next wait -- no host round trip between the scan and the compaction
        out.desc.ensure(n * 8 + 8); out.lv1.ensure(n * 3 + 4); out.lv2.ensure(n * 3 + 4); out.p1.ensure(n * 3 + 4);
        hipLaunchKernelGGL(k_scatter_pairs, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, out.flags.p, out.pos.p, n,
                           out.all_desc.p, out.all_lv1.p, out.all_lv2.p, out.all_p1.p, out.desc.p, out.lv1.p, out.lv2.p,
                           out.p1.p);
        HIP_TRY(hipGetLastError());
        ctx->d2h(&out.count, out.pos.p + n, 4);   // out.count is valid after the next sync() of this stream
        return;
    }
    uint32_t total = 0;
    ctx->d2h(&total, out.pos.p + n, 4);
    ctx->sync();  // also keeps `blob` alive until the copy is done
    out.count = total;
    out.desc.ensure((size_t)total * 8 + 8); out.lv1.ensure((size_t)total * 3 + 4); out.lv2.ensure((size_t)total * 3 + 4);
    out.p1.ensure((size_t)total * 3 + 4);
    if (total)
        hipLaunchKernelGGL(k_scatter_pairs, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, out.flags.p, out.pos.p, n,
                           out.all_desc.p, out.all_lv1.p, out.all_lv2.p, out.all_p1.p, out.desc.p, out.lv1.p, out.lv2.p,
                           out.p1.p);
    HIP_TRY(hipGetLastError());
}

}  // namespace plade
