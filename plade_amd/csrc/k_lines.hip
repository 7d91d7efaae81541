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
#include "k_svd.h"

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

// ---- closest_point_mode = 1: the closest points of every line pair i < j by the reference's 9 x 9 solve (k_svd.h) -------
// One system per lane, 64 lanes per workgroup (46 KB of LDS hold their matrices); lane t serves the t-th pair of the
// row-major enumeration of i < j and leaves (q1, q2) in slot i * L + j of `cp`; k_pair_table picks them up for (i, j) and,
// mirrored, for (j, i) (util.cpp:784-788 copies the entry).  The direction vectors are the iterates the pair would see in
// the reference's call order, exactly as in the closed-form path.
constexpr int CP_TPB = 64;
__device__ __forceinline__ void unordered_pair(uint32_t t, uint32_t L, uint32_t &i, uint32_t &j) {
    // pairs before row i: f(i) = i (2L - i - 1) / 2
    const double b = 2.0 * L - 1.0;
    long r = (long)floor((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (r < 0) r = 0;
    auto f = [&](long q) { return (unsigned long long)q * (2ull * L - (unsigned long long)q - 1ull) / 2ull; };
    while (r + 1 < (long)L && f(r + 1) <= t) ++r;
    while (r > 0 && f(r) > t) --r;
    i = (uint32_t)r;
    j = (uint32_t)(t - f(r)) + i + 1u;
}

__device__ void k_closest_svd(const VB &vb, LinesView v, uint32_t n_pairs, float *__restrict__ cp) {
    const uint32_t t = vb.bx * CP_TPB + threadIdx.x;
    if (t >= n_pairs) return;
    uint32_t i, j;
    unordered_pair(t, v.L, i, j);
    const f3 ci = line_iter(v, i, 3 + (int)j), cj = line_iter(v, j, 4 + (int)i);
    if (ci.x == cj.x && ci.y == cj.y && ci.z == cj.z) return;     // util.cpp:1173: the caller sees "-1"
    const f3 pti(v.pt[3 * i], v.pt[3 * i + 1], v.pt[3 * i + 2]), ptj(v.pt[3 * j], v.pt[3 * j + 1], v.pt[3 * j + 2]);
    f3 q1, q2;
    closest_points_svd(ci, pti, cj, ptj, q1, q2);
    float *o = cp + 6 * ((size_t)i * v.L + j);
    o[0] = q1.x; o[1] = q1.y; o[2] = q1.z; o[3] = q2.x; o[4] = q2.y; o[5] = q2.z;
}

// closest points of pair (a, b), a < b: from the solver's table when there is one, else the closed form
__device__ __forceinline__ bool pair_closest(const float *__restrict__ cp, uint32_t L, uint32_t a, uint32_t b, f3 ua, f3 pa, f3 ub, f3 pb,
                                             f3 &q1, f3 &q2, double &len) {
    if (!cp) return closest_points(ua, pa, ub, pb, q1, q2, len);
    if (ua.x == ub.x && ua.y == ub.y && ua.z == ub.z) return false;
    const float *o = cp + 6 * ((size_t)a * L + b);
    q1 = f3(o[0], o[1], o[2]); q2 = f3(o[3], o[4], o[5]);
    len = norm_e(q1 - q2);     // (point1 - point2).norm(), util.cpp:1227
    return true;
}

__device__ void k_pair_table(const VB &vb, LinesView v, float scale, float angle_thresh, int target,
                                                    const float *__restrict__ cp,
                                                    uint32_t *__restrict__ flags, float *__restrict__ desc,
                                                    float *__restrict__ lv1, float *__restrict__ lv2,
                                                    float *__restrict__ p1out) {
    const uint32_t L = v.L;
    const size_t idx = (size_t)vb.bx * blockDim.x + threadIdx.x;
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
        ok = pair_closest(cp, L, i, j, ci, pti, cj, ptj, p1, p2, len);
        if (target) { vi = ci; vj = cj; }
        else { vi = line_iter(v, i, 2 + (int)L); vj = line_iter(v, j, 2 + (int)L); }
    } else {
        // copied from entry (j, i) (util.cpp:784-788)
        f3 q1, q2;
        ok = pair_closest(cp, L, j, i, line_iter(v, j, 3 + (int)i), ptj, line_iter(v, i, 4 + (int)j), pti, q1, q2, len);
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

__device__ void k_scatter_pairs(const VB &vb, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, size_t n,
                                const float *__restrict__ desc, const float *__restrict__ lv1,
                                const float *__restrict__ lv2, const float *__restrict__ p1, float *__restrict__ o_desc,
                                float *__restrict__ o_lv1, float *__restrict__ o_lv2, float *__restrict__ o_p1) {
    const size_t i = (size_t)vb.bx * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const size_t o = pos[i];
    for (int k = 0; k < 8; ++k) o_desc[o * 8 + k] = desc[i * 8 + k];
    for (int k = 0; k < 3; ++k) { o_lv1[o * 3 + k] = lv1[i * 3 + k]; o_lv2[o * 3 + k] = lv2[i * 3 + k]; o_p1[o * 3 + k] = p1[i * 3 + k]; }
}

__device__ void k_flag_positions(const VB &vb, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, uint32_t n,
                                 uint32_t *__restrict__ out) {
    uint32_t i = vb.bx * blockDim.x + threadIdx.x;
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
    launch<k_flag_positions, 256>(ctx, dim3(cdiv(n, 256)), 0, d_flags, pos.p, n, out_idx.p);
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
    const float *cp = nullptr;
    if (ctx->params.closest_point_mode == 1 && L > 1) {
        const uint32_t n_pairs = (uint32_t)((size_t)L * (L - 1) / 2);
        cp = out.cp.ensure(n * 6);
        launch<k_closest_svd, CP_TPB>(ctx, dim3(cdiv(n_pairs, CP_TPB)), 0, v, n_pairs, out.cp.p);
    }
    launch<k_pair_table, 256>(ctx, dim3(cdiv(n, 256)), 0, v, scale, angle_thresh, target ? 1 : 0, cp,
                       out.flags.p, out.all_desc.p, out.all_lv1.p, out.all_lv2.p, out.all_p1.p);
    exclusive_scan_u32(ctx, out.flags.p, out.pos.p, n + 1);
    if (staged && n <= (1u << 20)) {
        // the usual table (a few hundred lines): the compacted arrays are sized for all n pairs (68 B each) and the count
        // comes back with the caller's next wait -- no host round trip between the scan and the compaction
        out.desc.ensure(n * 8 + 8); out.lv1.ensure(n * 3 + 4); out.lv2.ensure(n * 3 + 4); out.p1.ensure(n * 3 + 4);
        launch<k_scatter_pairs, 256>(ctx, dim3(cdiv(n, 256)), 0, out.flags.p, out.pos.p, n,
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
        launch<k_scatter_pairs, 256>(ctx, dim3(cdiv(n, 256)), 0, out.flags.p, out.pos.p, n,
                           out.all_desc.p, out.all_lv1.p, out.all_lv2.p, out.all_p1.p, out.desc.p, out.lv1.p, out.lv2.p,
                           out.p1.p);
    HIP_TRY(hipGetLastError());
}

// ---- seams: n independent line pairs through the same device functions ----------------------------------------------------
// in: u1 | p1 | u2 | p2 (each n x 3); out: q1 | q2 (n x 3 each)
template <int SVD>
__global__ __launch_bounds__(CP_TPB) void k_closest_seam(const float *__restrict__ in, uint32_t n, float *__restrict__ q, double *__restrict__ len,
                                                         int32_t *__restrict__ ok) {
    const uint32_t t = blockIdx.x * CP_TPB + threadIdx.x;
    if (t >= n) return;
    auto ld = [&](int a) { const float *p = in + 3 * ((size_t)a * n + t); return f3(p[0], p[1], p[2]); };
    const f3 u1 = normalized_e(ld(0)), p1 = ld(1), u2 = normalized_e(ld(2)), p2 = ld(3);
    f3 q1, q2;
    double l = -1;
    bool good;
    if (SVD) {
        good = !(u1.x == u2.x && u1.y == u2.y && u1.z == u2.z);
        if (good) {
            closest_points_svd(u1, p1, u2, p2, q1, q2);
            l = norm_e(q1 - q2);
        }
    } else good = closest_points(u1, p1, u2, p2, q1, q2, l);
    ok[t] = good ? 1 : 0;
    len[t] = good ? l : -1.0;
    float *a = q + 3 * (size_t)t, *b = q + 3 * ((size_t)n + t);
    a[0] = q1.x; a[1] = q1.y; a[2] = q1.z; b[0] = q2.x; b[1] = q2.y; b[2] = q2.z;
}

template <int SVD>
__global__ __launch_bounds__(128) void k_meet_seam(const float *__restrict__ in, uint32_t n, float *__restrict__ out, int32_t *__restrict__ ok) {
    const uint32_t t = blockIdx.x * 128 + threadIdx.x;
    if (t >= n) return;
    auto ld = [&](int a) { const float *p = in + 3 * ((size_t)a * n + t); return f3(p[0], p[1], p[2]); };
    const f3 v1 = ld(0), p1 = ld(1), v2 = ld(2), p2 = ld(3);
    f3 o;
    bool good;
    if (SVD) {
        good = !(fabsf(dot_e(v1, v2)) > 0.9999);
        if (good) o = lines_meet_svd(v1, p1, v2, p2);
    } else good = lines_meet(v1, p1, v2, p2, o);
    ok[t] = good ? 1 : 0;
    out[3 * (size_t)t] = o.x; out[3 * (size_t)t + 1] = o.y; out[3 * (size_t)t + 2] = o.z;
}

}  // namespace plade

using namespace plade;

static int line_seam(plade_ctx *ctx, int kind, int32_t mode, const float *a, const float *b, const float *c, const float *d, uint32_t n,
                     float *o1, float *o2, double *len, int32_t *ok) {
    return guarded(ctx, [&]() -> int {
        PLADE_REQUIRE(mode == 0 || mode == 1, PLADE_EINVAL, "line seam: mode must be 0 (closed form) or 1 (svd_fp32)");
        PLADE_REQUIRE(n == 0 || (a && b && c && d && o1 && ok && (kind == 1 || (o2 && len))), PLADE_EINVAL, "line seam: null argument");
        if (!n) return PLADE_OK;
        float *d_in = reinterpret_cast<float *>(ctx->scratch[0].ensure(48 * (size_t)n + 64));
        float *d_q = reinterpret_cast<float *>(ctx->scratch[1].ensure(24 * (size_t)n + 64));
        double *d_len = reinterpret_cast<double *>(ctx->scratch[2].ensure(8 * (size_t)n + 64));
        int32_t *d_ok = reinterpret_cast<int32_t *>(ctx->scratch[3].ensure(4 * (size_t)n + 64));
        const float *src[4] = {a, b, c, d};
        for (int k = 0; k < 4; ++k) HIP_TRY(hipMemcpyAsync(d_in + 3 * (size_t)k * n, src[k], 12 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        if (kind == 0) {
            if (mode) launch_raw(ctx, k_closest_seam<1>, dim3(cdiv(n, CP_TPB)), dim3(CP_TPB), 0, d_in, n, d_q, d_len, d_ok);
            else launch_raw(ctx, k_closest_seam<0>, dim3(cdiv(n, CP_TPB)), dim3(CP_TPB), 0, d_in, n, d_q, d_len, d_ok);
        } else {
            if (mode) launch_raw(ctx, k_meet_seam<1>, dim3(cdiv(n, 128)), dim3(128), 0, d_in, n, d_q, d_ok);
            else launch_raw(ctx, k_meet_seam<0>, dim3(cdiv(n, 128)), dim3(128), 0, d_in, n, d_q, d_ok);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(o1, d_q, 12 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        if (kind == 0) {
            HIP_TRY(hipMemcpyAsync(o2, d_q + 3 * (size_t)n, 12 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(len, d_len, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(hipMemcpyAsync(ok, d_ok, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return PLADE_OK;
    });
}

// Host seam of the register form of the reference's solver (k_svd.h, RegSolver): the same functions the kernels inline,
// instantiated for the host.  ok: 1 solved, 0 rank-deficient (the kernels hand such a system to LaneSolver), -1 the guard of
// the caller fired (identical directions, util.cpp:1173 / |v1.v2| > 0.9999, util.cpp:1463).  No context, no GPU.
extern "C" int plade_diag_line_solver_host(int32_t kind, const float *a, const float *b, const float *c, const float *d, uint32_t n,
                                           float *o1, float *o2, int32_t *ok) {
    if (n && !(a && b && c && d && o1 && ok && (kind == 1 || o2)) || (kind != 0 && kind != 1)) return PLADE_EINVAL;
    for (uint32_t t = 0; t < n; ++t) {
        auto ld = [&](const float *p) { return f3(p[3 * (size_t)t], p[3 * (size_t)t + 1], p[3 * (size_t)t + 2]); };
        if (kind == 0) {
            const f3 u1 = normalized_e(ld(a)), u2 = normalized_e(ld(c));
            f3 q1, q2;
            if (u1.x == u2.x && u1.y == u2.y && u1.z == u2.z) { ok[t] = -1; continue; }
            ok[t] = closest_points_regs(u1, ld(b), u2, ld(d), q1, q2) ? 1 : 0;
            o1[3 * (size_t)t] = q1.x; o1[3 * (size_t)t + 1] = q1.y; o1[3 * (size_t)t + 2] = q1.z;
            o2[3 * (size_t)t] = q2.x; o2[3 * (size_t)t + 1] = q2.y; o2[3 * (size_t)t + 2] = q2.z;
        } else {
            const f3 v1 = ld(a), v2 = ld(c);
            f3 o;
            if (fabsf(dot_e(v1, v2)) > 0.9999) { ok[t] = -1; continue; }
            ok[t] = lines_meet_regs(v1, ld(b), v2, ld(d), o) ? 1 : 0;
            o1[3 * (size_t)t] = o.x; o1[3 * (size_t)t + 1] = o.y; o1[3 * (size_t)t + 2] = o.z;
        }
    }
    return PLADE_OK;
}

extern "C" int plade_closest_points(plade_ctx *ctx, int32_t mode, const float *u1, const float *p1, const float *u2, const float *p2, uint32_t n,
                                    float *q1, float *q2, double *len, int32_t *ok) {
    return line_seam(ctx, 0, mode, u1, p1, u2, p2, n, q1, q2, len, ok);
}
extern "C" int plade_lines_meet(plade_ctx *ctx, int32_t mode, const float *v1, const float *p1, const float *v2, const float *p2, uint32_t n,
                                float *out, int32_t *ok) {
    return line_seam(ctx, 1, mode, v1, p1, v2, p2, n, out, nullptr, nullptr, ok);
}
