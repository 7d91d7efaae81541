// plade_amd/csrc/k_penetration.hip -- penetration filter over the <= 201 surviving candidates
// (SURVEY.md A11): the loop code/PLADE/util.cpp:450-519 around AreTwoPlanesPenetrable
// (util.cpp:1279-1458).
//
// The reference walks i1 over source planes and j1 over target planes and rejects a candidate at the
// first penetrating pair; the verdict is therefore the OR over all (i1, j1) that are not gated out,
// which is what the GPU evaluates, all triples (candidate, i1, j1) in parallel:
//   k_pen_setup : one lane per triple -- gate (util.cpp:487-492), plane/plane intersection line,
//                 clipping against both plane rectangles, overlap interval -> compact work list
//   k_pen_walk  : one wavefront per surviving triple, four triples of the same plane pair per
//                 workgroup -- the two kd-tree walks along the intersection segment.  A radiusSearch(p, r, max_nn=2) < 2 gate is "fewer than two cloud points
//                 within r/2 of the step point"; the classified set is the union over gated steps of
//                 the points within r, each counted once (checkIndex).  Both are order independent,
//                 so brute-force fp32 distance tests (FLANN L2_Simple, strict <) reproduce the counts.
#include "stages.h"
#include "prims.h"
#include "k_svd.h"
#include <algorithm>

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

// `meet(v1, p1, v2, p2, out)`: ComputeIntersectionPointOf23DLine (util.cpp:1461-1500) in the arithmetic the context asked for
template <class Meet>
__device__ __forceinline__ void edge_hit(f3 lineVec, f3 linePoint, f3 a, f3 b, f3 *out, int &n, Meet meet) {
    const f3 tl = normalized_e(b - a);
    f3 ip;
    if (!meet(lineVec, linePoint, tl, a, ip)) return;
    if (dot_e(a - ip, b - ip) > 0) return;
    if (n < 4) out[n] = ip;
    ++n;
}
template <bool ROLLED, class Meet>
__device__ __forceinline__ int edge_hits(f3 lineVec, f3 linePoint, const f3 *c, f3 *out, Meet meet) {
    int n = 0;
    if (ROLLED) {     // the solver's body once, not four times, in the instruction stream
#pragma unroll 1
        for (int i = 1; i <= 4; ++i) edge_hit(lineVec, linePoint, c[(i - 1) % 4], c[i % 4], out, n, meet);
    } else {
        for (int i = 1; i <= 4; ++i) edge_hit(lineVec, linePoint, c[(i - 1) % 4], c[i % 4], out, n, meet);
    }
    return n;
}

// items are grouped by plane pair: pair (i1, j1) owns slots [pair * K, pair * K + pair_count[pair])
struct PenTriple {     // what a triple (candidate k, source plane i1, target plane j1) carries from the gate to its item
    uint32_t k, i1, j1;
    float plane1[4];
    f3 lineVec, linePoint;
    f3 c[8];           // rectangle of the transformed source plane (0..3), of the target plane (4..7)
};

// gate (util.cpp:487-492), plane/plane intersection line (util.cpp:1296), the two rectangles; false: the triple is out
__device__ __forceinline__ bool pen_triple(const PenTables &tb, size_t idx, float len_th, float ang_th, PenTriple &t) {
    const uint32_t j1 = (uint32_t)(idx % tb.pt), i1 = (uint32_t)((idx / tb.pt) % tb.ps), k = (uint32_t)(idx / ((size_t)tb.pt * tb.ps));
    t.k = k; t.i1 = i1; t.j1 = j1;
    const float *c = tb.cand + 12 * (size_t)k;
    m3 R;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) R.m[r][q] = c[3 * r + q];
    const f3 T(c[9], c[10], c[11]);
    const float T12[12] = {c[0], c[1], c[2], c[9], c[3], c[4], c[5], c[10], c[6], c[7], c[8], c[11]};
    const float *sc = tb.s_coef + 4 * (size_t)i1;
    const f3 pn = mul_e(R, f3(sc[0], sc[1], sc[2]));
    t.plane1[0] = pn.x; t.plane1[1] = pn.y; t.plane1[2] = pn.z;
    t.plane1[3] = -(-sc[3] + dot_s(pn, T));
    const f3 c2m = mul_e(R, f3(tb.s_center[3 * i1], tb.s_center[3 * i1 + 1], tb.s_center[3 * i1 + 2])) + T;
    const float *tc = tb.t_coef + 4 * (size_t)j1;
    const f3 plane_A(tc[0], tc[1], tc[2]);
    const f3 tcen(tb.t_center[3 * j1], tb.t_center[3 * j1 + 1], tb.t_center[3 * j1 + 2]);
    const double c2p = (double)((fabsf(dot_e(plane_A, c2m) + tc[3]) + fabsf(dot_e(pn, tcen) + t.plane1[3])) / 2);
    if (c2p < (double)len_th && dot_e(pn, plane_A) > ang_th) return false;  // util.cpp:489
    if (!plane_plane_line(t.plane1, tc, t.lineVec, t.linePoint)) return false;    // util.cpp:1296
    for (int q = 0; q < 4; ++q) {
        const float *f = tb.s_four + 12 * (size_t)i1 + 3 * q;
        t.c[q] = pcl_xform(T12, f3(f[0], f[1], f[2]));
        const float *g = tb.t_four + 12 * (size_t)j1 + 3 * q;
        t.c[4 + q] = f3(g[0], g[1], g[2]);
    }
    return true;
}

// the clipped segments' overlap (util.cpp:1330-1378) -> the triple's item, in its plane pair's slots
__device__ __forceinline__ void pen_emit(const PenTables &tb, const PenTriple &t, const f3 *ip1, const f3 *ip2, PenItem *__restrict__ items,
                                         uint32_t *__restrict__ pair_count) {
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
    // (a single counter bumped by every surviving triple serialises at the memory side: one counter per plane pair)
    const uint32_t pair = t.i1 * tb.pt + t.j1;
    const uint32_t slot = pair * tb.K + atomicAdd(&pair_count[pair], 1u);
    PenItem it;
    it.k = t.k; it.i1 = t.i1; it.j1 = t.j1;
    for (int q = 0; q < 4; ++q) it.plane1[q] = t.plane1[q];
    it.sx = sp.x; it.sy = sp.y; it.sz = sp.z; it.dx = direc.x; it.dy = direc.y; it.dz = direc.z; it.length = length;
    items[slot] = it;
}

// closest_point_mode = 0: one lane per triple, the eight line / rectangle-edge meetings by the closed form
template <int TPB>
__device__ void k_pen_setup(const VB &vb, PenTables tb, float len_th, float ang_th, PenItem *__restrict__ items,
                            uint32_t *__restrict__ n_items, uint32_t *__restrict__ pair_count) {
    (void)n_items;
    const size_t idx = (size_t)vb.bx * blockDim.x + threadIdx.x;
    const size_t total = (size_t)tb.K * tb.ps * tb.pt;
    if (idx >= total) return;
    PenTriple t;
    if (!pen_triple(tb, idx, len_th, ang_th, t)) return;
    f3 ip1[4], ip2[4];
    auto meet = [&](f3 v1, f3 p1, f3 v2, f3 p2, f3 &out) -> bool { return lines_meet(v1, p1, v2, p2, out); };
    const int n1 = edge_hits<false>(t.lineVec, t.linePoint, t.c, ip1, meet);
    const int n2 = edge_hits<false>(t.lineVec, t.linePoint, t.c + 4, ip2, meet);
    if (n1 != 2 || n2 != 2) return;  // empty -> not penetrable; any other count -> "-1, continue"
    pen_emit(tb, t, ip1, ip2, items, pair_count);
}

// closest_point_mode = 1 (the default): the eight meetings of a triple are the reference's 6 x 5 float solves (k_svd.h,
// ~15 000 instructions each), and most triples need few of them -- the gate drops a triple, util.cpp:1463 drops an edge that is
// parallel to the intersection line (two of a rectangle's four in a Manhattan scene).  One lane per triple would run eight
// solves in sequence with most lanes idle, so a wavefront POOLS its solves: every lane publishes its line and its edges in
// LDS, the needed (lane, edge) items are numbered by ballots, the 64 lanes take one item each per round (whoever's it is),
// leave the meeting point where the edge was, and every lane then reads its own eight results and finishes as before.
// Same systems, same arithmetic, same results -- only which lane evaluates which system changes.
constexpr int PS_TPB = 64;
__device__ void k_pen_setup_svd(const VB &vb, PenTables tb, float len_th, float ang_th, PenItem *__restrict__ items,
                                uint32_t *__restrict__ n_items, uint32_t *__restrict__ pair_count) {
    (void)n_items;
    __shared__ float pub[54 * PS_TPB];        // word w of lane l at w * PS_TPB + l: line (6); per edge e: direction (3), corner (3)
    __shared__ uint16_t queue[8 * PS_TPB];
    const uint32_t lane = threadIdx.x;
    const size_t idx = (size_t)vb.bx * PS_TPB + lane;
    const size_t total = (size_t)tb.K * tb.ps * tb.pt;
    PenTriple t;
    const bool active = idx < total && pen_triple(tb, idx, len_th, ang_th, t);
    uint32_t need = 0;
    auto put = [&](int w, f3 v) { pub[(w + 0) * PS_TPB + lane] = v.x; pub[(w + 1) * PS_TPB + lane] = v.y; pub[(w + 2) * PS_TPB + lane] = v.z; };
    auto get = [&](int w, uint32_t l) { return f3(pub[(w + 0) * PS_TPB + l], pub[(w + 1) * PS_TPB + l], pub[(w + 2) * PS_TPB + l]); };
    if (active) {
        put(0, t.lineVec); put(3, t.linePoint);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int base = e & 4, i = (e & 3) + 1;
            const f3 a = t.c[base + (i - 1) % 4], b = t.c[base + i % 4];
            const f3 tl = normalized_e(b - a);
            if (!(fabsf(dot_e(t.lineVec, tl)) > 0.9999)) {    // util.cpp:1463
                need |= 1u << e;
                put(6 + 6 * e, tl); put(9 + 6 * e, a);
            }
        }
    }
    // number the items edge by edge, lane by lane
    uint32_t n_items_wave = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned long long m = __ballot((need >> e) & 1u);
        if ((need >> e) & 1u) queue[n_items_wave + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(lane | (e << 6));
        n_items_wave += (uint32_t)__popcll(m);
    }
    __syncthreads();
    for (uint32_t first = 0; first < n_items_wave; first += PS_TPB) {
        const uint32_t it = first + lane;
        if (it < n_items_wave) {
            const uint32_t q = queue[it], sl = q & 63u, e = q >> 6;
            const f3 o = lines_meet_svd(get(0, sl), get(3, sl), get(6 + 6 * e, sl), get(9 + 6 * e, sl));
            pub[(6 + 6 * e) * PS_TPB + sl] = o.x; pub[(7 + 6 * e) * PS_TPB + sl] = o.y; pub[(8 + 6 * e) * PS_TPB + sl] = o.z;
        }
    }
    __syncthreads();
    if (!active) return;
    f3 ip1[4], ip2[4];
    int n1 = 0, n2 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (!((need >> e) & 1u)) continue;
        const int base = e & 4, i = (e & 3) + 1;
        const f3 a = t.c[base + (i - 1) % 4], b = t.c[base + i % 4];
        const f3 ip = get(6 + 6 * e, lane);
        if (dot_e(a - ip, b - ip) > 0) continue;
        if (e < 4) { if (n1 < 4) ip1[n1] = ip; ++n1; }
        else { if (n2 < 4) ip2[n2] = ip; ++n2; }
    }
    if (n1 != 2 || n2 != 2) return;  // empty -> not penetrable; any other count -> "-1, continue"
    pen_emit(tb, t, ip1, ip2, items, pair_count);
}

constexpr int PEN_MAXS = 1024;   // search steps along one intersection segment
constexpr int PEN_G = 4;         // items (= wavefronts) per workgroup
constexpr int PEN_TPB = 64 * PEN_G;

// ---- in-plane grids ---------------------------------------------------------------------------
// Only the points within the search radius of the intersection line matter to a walk, and that line lies
// in both planes: each plane cloud is binned once into square cells of its own (u, v) frame, and a walk
// visits the cells its segment crosses (+ the radius) instead of streaming the whole plane cloud.
__device__ __forceinline__ void pen_uv(const PenFrame &f, f3 p, float &u, float &v) {
    const float dx = p.x - f.o[0], dy = p.y - f.o[1], dz = p.z - f.o[2];
    u = dx * f.eu[0] + dy * f.eu[1] + dz * f.eu[2];
    v = dx * f.ev[0] + dy * f.ev[1] + dz * f.ev[2];
}

__device__ void k_pen_cell_keys(const VB &vb, const float *__restrict__ xyz, uint32_t n, const uint32_t *__restrict__ off, uint32_t P,
                                const PenFrame *__restrict__ frames, float inv_cell, uint32_t *__restrict__ keys,
                                uint32_t *__restrict__ vals) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0, hi = P;   // plane of point i: last g with off[g] <= i
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    const PenFrame f = frames[lo];
    float u, v;
    pen_uv(f, f3(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]), u, v);
    const int cu = min(max((int)floorf(u * inv_cell) + 1, 0), f.nu - 1);   // one cell of margin on every side
    const int cv = min(max((int)floorf(v * inv_cell) + 1, 0), f.nv - 1);
    keys[i] = f.base + (uint32_t)cv * (uint32_t)f.nu + (uint32_t)cu;
    vals[i] = i;
}

__device__ void k_pen_cell_fill(const VB &vb, const float *__restrict__ xyz, const uint32_t *__restrict__ skeys,
                                const uint32_t *__restrict__ svals, uint32_t n, uint32_t n_cells,
                                float4 *__restrict__ pts, uint32_t *__restrict__ cell_start) {
    const uint32_t i = vb.bx * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t p = svals[i];
        pts[i] = make_float4(xyz[3 * (size_t)p], xyz[3 * (size_t)p + 1], xyz[3 * (size_t)p + 2], 0.f);
    }
    if (i <= n_cells) {   // first sorted position with key >= i
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skeys[mid] < i) lo = mid + 1; else hi = mid; }
        cell_start[i] = lo;
    }
}

void build_pen_grid(plade_ctx *ctx, PlaneCloudsDev &pc, const PlaneGeomHost &geom, float cell) {
    const uint32_t P = geom.P, n = pc.off[P];
    std::vector<PenFrame> fr(P);
    uint32_t total = 0;
    for (uint32_t g = 0; g < P; ++g) {
        const float *c = geom.four.data() + 12 * (size_t)g;
        PenFrame &f = fr[g];
        double eu[3], ev[3], lu = 0, lv = 0;
        for (int k = 0; k < 3; ++k) { eu[k] = (double)c[3 + k] - c[k]; ev[k] = (double)c[9 + k] - c[k]; lu += eu[k] * eu[k]; lv += ev[k] * ev[k]; }
        lu = std::sqrt(lu); lv = std::sqrt(lv);
        for (int k = 0; k < 3; ++k) {
            f.o[k] = c[k];
            f.eu[k] = lu > 0 ? (float)(eu[k] / lu) : (k == 0);
            f.ev[k] = lv > 0 ? (float)(ev[k] / lv) : (k == 1);
        }
        f.nu = (int)std::min(4096.0, std::floor(lu / cell) + 3);
        f.nv = (int)std::min(4096.0, std::floor(lv / cell) + 3);
        f.base = total;
        total += (uint32_t)f.nu * (uint32_t)f.nv;
    }
    pc.n_cells = total;
    pc.grid_cell = cell;
    pc.frames.ensure(P);
    const bool staged = ctx->h2d(pc.frames.p, fr.data(), P * sizeof(PenFrame));
    pc.cell_pts.ensure((size_t)n + 1);
    pc.cell_start.ensure((size_t)total + 2);
    pc.ckeys.ensure((size_t)n + 1); pc.ckeys2.ensure((size_t)n + 1); pc.cvals.ensure((size_t)n + 1); pc.cvals2.ensure((size_t)n + 1);
    if (n)
        launch<k_pen_cell_keys, 256>(ctx, dim3(cdiv(n, 256)), 0, pc.xyz.p, n, pc.d_off.p, P, pc.frames.p,
                           1.f / cell, pc.ckeys.p, pc.cvals.p);
    int bits = 1;
    while ((1ull << bits) < total) ++bits;
    sort_pairs_u32(ctx, pc.ckeys.p, pc.ckeys2.p, pc.cvals.p, pc.cvals2.p, n, bits);
    launch<k_pen_cell_fill, 256>(ctx, dim3(cdiv(std::max(n, total + 1), 256)), 0, pc.xyz.p, pc.ckeys2.p,
                       pc.cvals2.p, n, total, pc.cell_pts.p, pc.cell_start.p);
    if (!staged) ctx->sync();   // `fr` must outlive the copy
}

// Cells of one plane grid that can hold a point within `rr` of the segment start + t direc, t in [0, L]
// (all in the plane cloud's own frame).  The line is walked along its major in-plane axis: one lane per column (or row),
// in each the cell range the line covers, widened by the radius.  fn(point) is called once for every point of those
// cells.  The POINTS are then shared out over the wavefront: the q-th cell of every column contributes a range of points,
// the ranges of the 64 columns are laid end to end (wave prefix sum) and lane l takes entries l, l + 64, ... of that
// list (owner column by binary search over the prefix sums).  A lane per column walking its own cells -- the first
// version -- is a chain of ~50 dependent loads for the usual 5-column segment with 5 of 64 lanes busy; this way a
// segment costs two dependent loads per cell row and all lanes load points at once.  Same set of points, every point
// once; fn only counts, so the order does not matter.
template <class Fn>
__device__ __forceinline__ void pen_visit(const PenFrame &f, const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                          f3 start, f3 direc, float L, float rr, float cell, int lane, Fn fn) {
    float su, sv, eu_, ev_;
    pen_uv(f, start, su, sv);
    pen_uv(f, f3(start.x + L * direc.x, start.y + L * direc.y, start.z + L * direc.z), eu_, ev_);
    const float inv = 1.f / cell;
    const bool major_u = fabsf(eu_ - su) >= fabsf(ev_ - sv);
    // a = coordinate along the major axis, b = along the minor one
    const float a0 = major_u ? su : sv, a1 = major_u ? eu_ : ev_, b0 = major_u ? sv : su, b1 = major_u ? ev_ : eu_;
    const int na = major_u ? f.nu : f.nv, nbm = major_u ? f.nv : f.nu;
    const float pad = rr * 1.4143f + 0.02f * cell;   // perpendicular rr seen along the minor axis (slope <= 1) + rounding
    const float amin = fminf(a0, a1) - rr - 0.02f * cell, amax = fmaxf(a0, a1) + rr + 0.02f * cell;
    const int ca0 = max((int)floorf(amin * inv) + 1, 0), ca1 = min((int)floorf(amax * inv) + 1, na - 1);
    const float da = a1 - a0;
    const float slope = fabsf(da) > 1e-12f ? (b1 - b0) / da : 0.f;
    for (int base = ca0; base <= ca1; base += 64) {          // wave-uniform
        const int ca = base + lane;
        int cb0 = 0, ncell = 0;
        if (ca <= ca1) {
            // the part of the segment (extended by rr at both ends) inside this column
            float lo = fmaxf((float)(ca - 1) * cell, amin), hi = fminf((float)ca * cell, amax);
            if (hi < lo) { const float t = lo; lo = hi; hi = t; }
            const float bl = b0 + (lo - a0) * slope, bh = b0 + (hi - a0) * slope;
            cb0 = max((int)floorf((fminf(bl, bh) - pad) * inv) + 1, 0);
            const int cb1 = min((int)floorf((fmaxf(bl, bh) + pad) * inv) + 1, nbm - 1);
            ncell = max(cb1 - cb0 + 1, 0);
        }
        int most = ncell;
        for (int d = 32; d >= 1; d >>= 1) most = max(most, __shfl_xor(most, d, 64));
        constexpr int QN = 4;                                // cells of a column per round (a column spans <= 4: 2 pad + slope)
        for (int q0 = 0; q0 < most; q0 += QN) {              // wave-uniform
            uint32_t pb[QN], len[QN], mine = 0;
#pragma unroll
            for (int q = 0; q < QN; ++q) {                   // all cell_start loads of the round are in flight together
                pb[q] = 0; len[q] = 0;
                if (q0 + q < ncell) {
                    const int cb = cb0 + q0 + q;
                    const uint32_t c = f.base + (major_u ? (uint32_t)cb * (uint32_t)f.nu + (uint32_t)ca
                                                         : (uint32_t)ca * (uint32_t)f.nu + (uint32_t)cb);
                    pb[q] = cell_start[c];
                    len[q] = cell_start[c + 1] - pb[q];
                }
                mine += len[q];
            }
            uint32_t incl = mine;                            // inclusive prefix sum of the columns' point counts
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            const uint32_t total = __shfl(incl, 63, 64), excl = incl - mine;
            for (uint32_t t0 = 0; t0 < total; t0 += 64) {    // wave-uniform
                const uint32_t t = t0 + (uint32_t)lane;
                int lo = 0, hi = 63;                         // owner column = first lane whose inclusive sum exceeds t
                for (int it = 0; it < 6; ++it) {
                    const int mid = (lo + hi) >> 1;
                    const bool right = __shfl(incl, mid, 64) <= t;
                    lo = right ? mid + 1 : lo;
                    hi = right ? hi : mid;
                }
                uint32_t r = t - __shfl(excl, lo, 64), at = 0;   // position inside the owner's cells, then inside one cell
                bool found = false;
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    const uint32_t o_pb = __shfl(pb[q], lo, 64), o_len = __shfl(len[q], lo, 64);
                    if (!found && r < o_len) { at = o_pb + r; found = true; }
                    if (!found) r -= o_len;
                }
                if (t < total && found) {
                    const float4 p4 = pts[at];
                    fn(f3(p4.x, p4.y, p4.z));
                }
            }
        }
    }
}

struct PenSide {
    const PenFrame *frames;
    const float4 *pts;
    const uint32_t *cell_start;
};

// One wavefront per item.  step_dist: the reference's `for (dist = 0; dist < length; dist += r)` sequence
// (util.cpp:1383, fp32 accumulation), PEN_MAXS + 1 entries computed once on the host -- it does not depend
// on the item.  All distance tests are the reference's arithmetic on the reference's operands (source
// points moved by the candidate with pcl_xform); the grids only select which points are looked at.
__device__ void k_pen_walk(const VB &vb, const PenItem *__restrict__ items, const uint32_t *__restrict__ pair_count,
                                                      const uint32_t *__restrict__ pair_order, PenTables tb,
                                                      const float *__restrict__ step_dist, PenSide S, PenSide T_, float cell,
                                                      float search_radius, int min_points, float min_distance,
                                                      uint32_t *__restrict__ cand_flags, uint32_t *__restrict__ overflow) {
    __shared__ uint32_t s_cnt_all[PEN_G][PEN_MAXS];
    __shared__ float s_dist[PEN_MAXS + 1];
    const uint32_t pair = pair_order[vb.bx];   // heaviest plane pairs first (longest-processing-time order)
    const uint32_t cnt = pair_count[pair];
    if (vb.by * PEN_G >= cnt) return;          // whole workgroup idle (uniform)
    for (int i = threadIdx.x; i <= PEN_MAXS; i += blockDim.x) s_dist[i] = step_dist[i];   // the step table in LDS
    __syncthreads();
    // the wave index as a scalar: everything derived from it (the item, its candidate, the two plane frames) is then loaded
    // with scalar loads into SGPRs instead of being replicated over the lanes' VGPRs: 142 -> 99 VGPRs, and the occupancy
    // hint above brings the kernel to 80 (6 waves per SIMD instead of 3; the LDS step counters allow 7)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t slot = vb.by * PEN_G + wave;
    if (slot >= cnt) return;
    uint32_t *s_cnt = s_cnt_all[wave];
    const PenItem it = items[(size_t)pair * tb.K + slot];
    // the verdict per candidate is an OR over its items: once one item has rejected the candidate the
    // others cannot change it
    if (__atomic_load_n(&cand_flags[it.k], __ATOMIC_RELAXED)) return;
    // number of steps with dist < length (the sequence is increasing)
    int nsteps;
    {
        int lo = 0, hi = PEN_MAXS;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_dist[mid] < it.length) lo = mid + 1; else hi = mid; }
        if (lo == PEN_MAXS && s_dist[PEN_MAXS] < it.length && lane == 0) atomicExch(overflow, 1u);
        nsteps = lo;
    }
    if (nsteps == 0) return;  // length <= 0: both walks see nothing -> positive/negative < minPoints
    const f3 start(it.sx, it.sy, it.sz), direc(it.dx, it.dy, it.dz);
    const float half_r2 = (float)((double)(search_radius / 2) * (double)(search_radius / 2));
    const float full_r2 = (float)((double)search_radius * (double)search_radius);
    const float *c = tb.cand + 12 * (size_t)it.k;
    const float T12[12] = {c[0], c[1], c[2], c[9], c[3], c[4], c[5], c[10], c[6], c[7], c[8], c[11]};
    const uint32_t i1 = pair / tb.pt, j1 = pair % tb.pt;
    const float *tc = tb.t_coef + 4 * (size_t)j1;
    const float inv_r = 1.f / search_radius;
    const float L = s_dist[nsteps - 1];   // the last step point
    // the segment in the source frame: p_s = R^T (p - T) (cell selection only)
    const f3 ds(start.x - c[9], start.y - c[10], start.z - c[11]);
    const f3 start_s(c[0] * ds.x + c[3] * ds.y + c[6] * ds.z, c[1] * ds.x + c[4] * ds.y + c[7] * ds.z,
                     c[2] * ds.x + c[5] * ds.y + c[8] * ds.z);
    const f3 direc_s(c[0] * direc.x + c[3] * direc.y + c[6] * direc.z, c[1] * direc.x + c[4] * direc.y + c[7] * direc.z,
                     c[2] * direc.x + c[5] * direc.y + c[8] * direc.z);
    const PenFrame fs = S.frames[i1], ft = T_.frames[j1];

    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: gate = target plane cloud, classified = transformed source plane cloud vs plane2
        // pass 1: gate = transformed source plane cloud, classified = target plane cloud vs plane1
        for (int i = lane; i < nsteps; i += 64) s_cnt[i] = 0u;
        __syncwarp();
        auto gate = [&](f3 raw) {
            const f3 p = pass == 0 ? raw : pcl_xform(T12, raw);
            const f3 d = p - start;
            const float t = d.x * direc.x + d.y * direc.y + d.z * direc.z;
            const int kc = (int)floorf(t * inv_r);
            for (int kk = max(kc - 2, 0); kk <= min(kc + 3, nsteps - 1); ++kk) {
                const float dist = s_dist[kk];
                const f3 spt(start.x + dist * direc.x, start.y + dist * direc.y, start.z + dist * direc.z);
                if (flann_d2(spt, p) < half_r2) atomicAdd(&s_cnt[kk], 1u);
            }
        };
        if (pass == 0) pen_visit(ft, T_.pts, T_.cell_start, start, direc, L, search_radius * 0.5f, cell, lane, gate);
        else pen_visit(fs, S.pts, S.cell_start, start_s, direc_s, L, search_radius * 0.5f, cell, lane, gate);
        __syncwarp();
        if (__atomic_load_n(&cand_flags[it.k], __ATOMIC_RELAXED)) return;   // rejected by another item meanwhile
        const float pl0 = pass == 0 ? tc[0] : it.plane1[0], pl1 = pass == 0 ? tc[1] : it.plane1[1],
                    pl2 = pass == 0 ? tc[2] : it.plane1[2], pl3 = pass == 0 ? tc[3] : it.plane1[3];
        int pos = 0, neg = 0;
        auto classify = [&](f3 raw) {
            const f3 p = pass == 0 ? pcl_xform(T12, raw) : raw;
            const f3 d = p - start;
            const float t = d.x * direc.x + d.y * direc.y + d.z * direc.z;
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
                if (fabsf(td) > min_distance) { if (td >= 0) ++pos; else ++neg; }
            }
        };
        if (pass == 0) pen_visit(fs, S.pts, S.cell_start, start_s, direc_s, L, search_radius, cell, lane, classify);
        else pen_visit(ft, T_.pts, T_.cell_start, start, direc, L, search_radius, cell, lane, classify);
        for (int dlt = 32; dlt >= 1; dlt >>= 1) { pos += __shfl_xor(pos, dlt, 64); neg += __shfl_xor(neg, dlt, 64); }
        __syncwarp();   // all reads of s_cnt done before the next pass clears it
        if (pass == 0) {
            if (pos < min_points || neg < min_points) return;
        } else {
            if (pos < min_points && neg < min_points) return;
        }
        if ((double)max(pos, neg) / (double)min(pos, neg + 1) > 5) return;
    }
    if (lane == 0) atomicOr(&cand_flags[it.k], 1u);
}

void penetration_filter(plade_ctx *ctx, const float *cand_rt_host, uint32_t K, const PlaneGeomHost &src,
                        const PlaneGeomHost &tgt, PlaneCloudsDev &src_pts, PlaneCloudsDev &tgt_pts,
                        float length_threshold, float angle_threshold, std::vector<int32_t> &flags_out, const float *cand_rt_dev,
                        const PenGather *gather) {
    flags_out.assign(K, 0);
    if (!K || !src.P || !tgt.P) {   // (the caller's read-back of the table rides on our wait)
        if (gather && K) {
            uint32_t *d_ids = reinterpret_cast<uint32_t *>(ctx->scratch[4].ensure(4 * (size_t)K + 64));
            ctx->h2d(d_ids, gather->ids, 4 * (size_t)K);
            gather->launch(d_ids);
        }
        if (cand_rt_dev) ctx->sync();
        return;
    }
    // upload tables
    // one upload: candidates | plane tables of both sides | search steps | plane-pair order | the caller's candidate ids |
    // the zeroed counters (a memset of their own would be one or two more commands)
    const uint32_t n_pairs = src.P * tgt.P;
    const size_t n_cand = cand_rt_dev ? 0 : 12 * (size_t)K;   // candidates already on the device: not uploaded again
    const size_t n_tab = n_cand + (4 + 3 + 12) * ((size_t)src.P + tgt.P);
    const size_t n_ids = gather ? K : 0;
    // counters: [0] items, [1] overflow, [2 .. 2+K) candidate flags, then one count per plane pair
    const size_t n_ctr = (size_t)K + 2 + n_pairs;
    const size_t nf = n_tab + (PEN_MAXS + 1) + n_pairs + n_ids + n_ctr;
    std::vector<float> h(nf, 0.f);
    float *p = h.data();
    if (n_cand) memcpy(p, cand_rt_host, 48 * (size_t)K);
    float *o_cand = p; p += n_cand;
    float *o_sc = p; memcpy(p, src.coef.data(), 16 * (size_t)src.P); p += 4 * (size_t)src.P;
    float *o_scen = p; memcpy(p, src.center.data(), 12 * (size_t)src.P); p += 3 * (size_t)src.P;
    float *o_sf = p; memcpy(p, src.four.data(), 48 * (size_t)src.P); p += 12 * (size_t)src.P;
    float *o_tc = p; memcpy(p, tgt.coef.data(), 16 * (size_t)tgt.P); p += 4 * (size_t)tgt.P;
    float *o_tcen = p; memcpy(p, tgt.center.data(), 12 * (size_t)tgt.P); p += 3 * (size_t)tgt.P;
    float *o_tf = p; memcpy(p, tgt.four.data(), 48 * (size_t)tgt.P); p += 12 * (size_t)tgt.P;
    float *d = reinterpret_cast<float *>(ctx->scratch[4].ensure(nf * 4 + 64));
    PenTables tb;
    tb.cand = cand_rt_dev ? cand_rt_dev : d + (o_cand - h.data()); tb.s_coef = d + (o_sc - h.data()); tb.s_center = d + (o_scen - h.data());
    tb.s_four = d + (o_sf - h.data()); tb.t_coef = d + (o_tc - h.data()); tb.t_center = d + (o_tcen - h.data());
    tb.t_four = d + (o_tf - h.data());
    tb.K = K; tb.ps = src.P; tb.pt = tgt.P;
    const size_t total = (size_t)K * src.P * tgt.P;
    PLADE_REQUIRE(total < (1ull << 31), PLADE_ELIMIT, "penetration: too many (candidate, plane, plane) triples");
    PenItem *d_items = reinterpret_cast<PenItem *>(ctx->scratch[5].ensure(total * sizeof(PenItem) + 64));
    // AreTwoPlanesPenetrable(..., searchRadius = lengthThreshold, minPointsNum = 10, minDistance = lengthThreshold / 2)
    const float search_radius = (float)(double)length_threshold;
    const float min_distance = (float)((double)length_threshold / 2);
    uint32_t *d_ctr = reinterpret_cast<uint32_t *>(d + n_tab + PEN_MAXS + 1 + n_pairs + n_ids);
    uint32_t *d_n = d_ctr, *d_over = d_ctr + 1, *d_flags = d_ctr + 2, *d_pair = d_ctr + 2 + K;
    std::vector<uint32_t> order(n_pairs);
    {
        // an item's cost is proportional to the two plane clouds it streams
        std::vector<uint32_t> w(n_pairs);
        for (uint32_t i = 0; i < src.P; ++i)
            for (uint32_t j = 0; j < tgt.P; ++j)
                w[i * tgt.P + j] = (src_pts.off[i + 1] - src_pts.off[i]) + (tgt_pts.off[j + 1] - tgt_pts.off[j]);
        for (uint32_t i = 0; i < n_pairs; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
    }
    float *h_steps = h.data() + n_tab;
    {
        float dist = 0;
        for (int i = 0; i <= PEN_MAXS; ++i) { h_steps[i] = dist; dist += search_radius; }  // util.cpp:1383
    }
    memcpy(h.data() + n_tab + PEN_MAXS + 1, order.data(), 4 * (size_t)n_pairs);
    if (n_ids) memcpy(h.data() + n_tab + PEN_MAXS + 1 + n_pairs, gather->ids, 4 * n_ids);
    ctx->h2d(d, h.data(), nf * 4);
    if (gather) gather->launch(reinterpret_cast<const uint32_t *>(d + n_tab + PEN_MAXS + 1 + n_pairs));
    const float *d_steps = d + n_tab;
    const uint32_t *d_order = reinterpret_cast<const uint32_t *>(d + n_tab + PEN_MAXS + 1);
    if (ctx->params.closest_point_mode == 1)
        launch<k_pen_setup_svd, PS_TPB>(ctx, dim3(cdiv(total, PS_TPB)), 0, tb, length_threshold,
                           angle_threshold, d_items, d_n, d_pair);
    else
        launch<k_pen_setup<256>, 256>(ctx, dim3(cdiv(total, 256)), 0, tb, length_threshold,
                           angle_threshold, d_items, d_n, d_pair);
    // in-plane grids of both sides (cell = 2 r)
    const float cell = pen_grid_cell(length_threshold);
    if (src_pts.grid_cell != cell) build_pen_grid(ctx, src_pts, src, cell);   // normally built by prepare_side
    if (tgt_pts.grid_cell != cell) build_pen_grid(ctx, tgt_pts, tgt, cell);
    const PenSide sS{src_pts.frames.p, src_pts.cell_pts.p, src_pts.cell_start.p}, sT{tgt_pts.frames.p, tgt_pts.cell_pts.p, tgt_pts.cell_start.p};
    // a plane pair holds at most K items: grid.y covers the worst case, empty groups exit at once
    ctx->ev_begin("pen_walk", 0.0);
    launch<k_pen_walk, PEN_TPB, 6>(ctx, dim3(n_pairs, cdiv(K, PEN_G)), 0, d_items, d_pair, d_order, tb, d_steps,
                       sS, sT, cell, search_radius, 10, min_distance, d_flags, d_over);
    ctx->ev_end();
    std::vector<uint32_t> out(n_ctr);   // items, overflow, K candidate flags, per-pair item counts
    ctx->d2h(out.data(), d_ctr, n_ctr * 4);
    ctx->sync();
    HIP_TRY(hipGetLastError());
    double items = 0;
    for (uint32_t pr = 0; pr < n_pairs; ++pr) items += out[(size_t)K + 2 + pr];
    ctx->stats.add("pen_items", items);
    PLADE_REQUIRE(out[1] == 0, PLADE_ELIMIT, "penetration: intersection segment longer than 1024 search steps");
    for (uint32_t k = 0; k < K; ++k) flags_out[k] = out[k + 2] ? 1 : 0;
}

}  // namespace plade
