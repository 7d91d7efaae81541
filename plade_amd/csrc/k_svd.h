// plade_amd/csrc/k_svd.h -- plade_params.closest_point_mode = 1 ("svd_fp32"): the reference's own arithmetic for the
// closest points of two lines and for the meeting point of two lines.
//
// The reference does not evaluate a formula there; it hands a 9 x 9 (ComputeNearstTwoPointsOfTwo3DLine,
// code/PLADE/util.cpp:1183-1226) and a 6 x 5 (ComputeIntersectionPointOf23DLine, util.cpp:1467-1497) float system to
// cv::solve(A, B, X, cv::DECOMP_SVD) (code/3rd_party/opencv/modules/core/src/lapack.cpp:1335-1460), which works on the
// transpose: the columns of A are made mutually orthogonal by Hestenes' one-sided Jacobi rotations (lapack.cpp:533-710;
// float data, every inner product and squared norm accumulated in double, tolerance 2 FLT_EPSILON, at most max(m, 30)
// sweeps, rotations also applied to an identity that becomes V^T), the column norms are the singular values (ordered by a
// selection sort), rows that ended up empty are replaced by sign vectors made orthogonal to the others, and the solution
// is V diag(1/w) U^T b with float products summed in double (lapack.cpp:751-812).  On axis-aligned scenes the result of
// those solves is ill-conditioned, so reproducing the REFERENCE's transform there needs every rounding of that
// procedure -- this file performs them in the same order, one system per lane, in two forms:
//
//   * RegSolver (below, host + device): the whole system in registers, every loop over compile-time bounds -- what the kernels
//     run (k_closest_svd, k_pen_setup_svd) and what plade_diag_line_solver_host instantiates on the host;
//   * LaneSolver (device): the general form with run-time row indices -- the selection sort and the sign-vector completion of a
//     vanished column (lapack.cpp:650-699) address rows by index, which registers cannot do -- on a lane's own words of LDS or of
//     scratch memory.  Since round 6 it only serves the rare systems RegSolver hands back (a column of norm <= FLT_MIN).
//   * lanes converge after different numbers of sweeps; a lane that is done simply leaves the loop.
//
// hypot(): the reference calls the C library's.  glibc 2.35's dbl-64 kernel (no FMA, no rescaling between 2^-459 and
// 2^511) is the correction step below; tests/test_host_logic.py checks the oracle's copy of it against libm's hypot on
// 5e6 random arguments (bit for bit), and kernel and oracle share the formula so that they cannot differ by a libm.
#pragma once
#include "common.h"

namespace plade {

HD double hypot_corrected(double x, double y) {
    double big = fabs(x), small = fabs(y);
    if (big < small) { const double t = big; big = small; small = t; }
    double h = sqrt(big * big + small * small);
    double e1, e2;   // h^2 - big^2 - small^2, split so that every product is exact or nearly so
    if (h <= 2.0 * small) {
        const double d = h - small;
        e1 = big * (2.0 * d - big);
        e2 = (d - 2.0 * (big - small)) * d;
    } else {
        const double d = h - big;
        e1 = 2.0 * d * (big - 2.0 * small);
        e2 = (4.0 * d - small) * small + d * d;
    }
    h -= (e1 + e2) / (2.0 * h);
    return h;
}

// ---- the same procedure with the whole system in REGISTERS (round 6) -------------------------------------------------------
// Every loop of LaneSolver runs over compile-time bounds, so with all of them unrolled each (row, column) access names a
// fixed register: A^T (N x M), V^T (N x N) and the N squared norms (fp64) of the 9 x 9 system are ~200 VGPRs, two
// wavefronts per SIMD and no LDS traffic (the LDS form: 46 KB per wavefront, 72 LDS operations per rotation, < 1 wavefront per
// SIMD).  The arithmetic is LaneSolver's, rounding for rounding, with two licences that cannot change a bit:
//   * double accumulations of float x float products use fma(): the product of two floats is exact in double (48 significant
//     bits), so round(acc + a b) is what the separate multiply and add give;
//   * the selection sort exchanges columns by predicated moves (same comparisons, same sequence of exchanges).
// What stays with LaneSolver is the sign-vector completion of a column whose norm vanished (lapack.cpp:650-699: run-time row
// indices, a random generator with state): ok() is false for such a lane and the caller hands that system to LaneSolver --
// the 9 x 9 systems of line pairs have full rank unless the two directions are parallel to the last bit, so this is rare.
template <int M, int N>
struct RegSolver {
    float at[N][M];     // at[c][k] = A[k][c]
    float vt[N][N];
    double sq[N];
    bool full_rank;

    HD void clear() {
#pragma unroll
        for (int c = 0; c < N; ++c)
#pragma unroll
            for (int k = 0; k < M; ++k) at[c][k] = 0.f;
    }

    HD double column_energy(int c) const {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) { const double t = (double)at[c][k]; acc = fma(t, t, acc); }
        return acc;
    }

    // one Hestenes rotation test of the column pair (a, b), lapack.cpp:558-610; true if it rotated
    HD bool rotate_pair(int a, int b) {
        const float tol = FLT_EPSILON * 2;
        double inner = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) inner = fma((double)at[a][k], (double)at[b][k], inner);
        double ea = sq[a], eb = sq[b];
        if (fabs(inner) <= tol * sqrt(ea * eb)) return false;
        inner *= 2;
        const double gap = ea - eb, hyp = hypot_corrected(inner, gap);
        // lapack.cpp:583-594: gap < 0: s = sqrt((hyp - gap) / 2 / hyp), c = inner / (hyp s 2); else c = sqrt((hyp + gap) / (2 hyp)),
        // s = inner / (hyp c 2).  One square root and two divisions on operands picked per lane -- the operations of the lane's own
        // branch, in its order -- instead of both branches one after the other whenever the lanes of a wavefront disagree
        const bool neg = gap < 0;
        const double num = neg ? (hyp - gap) * 0.5 : hyp + gap, den = neg ? hyp : hyp * 2;
        const float first = (float)sqrt(num / den);
        const float second = (float)(inner / (hyp * first * 2));
        const float cs = neg ? second : first, sn = neg ? first : second;
        ea = 0; eb = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            const float ra = at[a][k], rb = at[b][k];
            const float na = cs * ra + sn * rb;
            const float nb = -sn * ra + cs * rb;
            at[a][k] = na; at[b][k] = nb;
            const double da = (double)na, db = (double)nb;
            ea = fma(da, da, ea); eb = fma(db, db, eb);
        }
        sq[a] = ea; sq[b] = eb;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float va = vt[a][k], vb = vt[b][k];
            vt[a][k] = cs * va + sn * vb;
            vt[b][k] = -sn * va + cs * vb;
        }
        return true;
    }

    HD void orthogonalise_columns() {
#pragma unroll
        for (int c = 0; c < N; ++c) {
            sq[c] = column_energy(c);
#pragma unroll
            for (int k = 0; k < N; ++k) vt[c][k] = k == c ? 1.f : 0.f;
        }
        const int sweeps = M > 30 ? M : 30;
#pragma unroll 1
        for (int sweep = 0; sweep < sweeps; ++sweep) {
            bool rotated = false;
#pragma unroll
            for (int a = 0; a < N - 1; ++a)
#pragma unroll
                for (int b = a + 1; b < N; ++b) rotated |= rotate_pair(a, b);
            if (!rotated) break;
        }
    }

    HD void order_singular_values() {
#pragma unroll
        for (int c = 0; c < N; ++c) sq[c] = sqrt(column_energy(c));   // from here on sq[] holds the NORMS
#pragma unroll
        for (int c = 0; c < N - 1; ++c) {
            int top = c;
            double wtop = sq[c];
#pragma unroll
            for (int k = c + 1; k < N; ++k) if (wtop < sq[k]) { top = k; wtop = sq[k]; }
#pragma unroll
            for (int k = c + 1; k < N; ++k) {
                const bool sw = top == k;
                { const double x = sq[c], y = sq[k]; sq[c] = sw ? y : x; sq[k] = sw ? x : y; }
#pragma unroll
                for (int e = 0; e < M; ++e) { const float x = at[c][e], y = at[k][e]; at[c][e] = sw ? y : x; at[k][e] = sw ? x : y; }
#pragma unroll
                for (int e = 0; e < N; ++e) { const float x = vt[c][e], y = vt[k][e]; vt[c][e] = sw ? y : x; vt[k][e] = sw ? x : y; }
            }
        }
    }

    HD void normalise_left_vectors() {
        full_rank = true;
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const double len = sq[c];
            if (len <= (double)FLT_MIN) full_rank = false;     // lapack.cpp:654: this column wants the sign-vector completion
            const float inv = (float)(1 / len);
#pragma unroll
            for (int k = 0; k < M; ++k) at[c][k] *= inv;
        }
    }

    HD void back_substitute(const float (&rhs)[M], float (&x)[N]) const {
        const float tol = (float)(DBL_EPSILON * 2);
        constexpr int R = M < N ? M : N;
        double cut = 0;
#pragma unroll
        for (int c = 0; c < R; ++c) cut += (float)sq[c];
        cut *= tol;
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = 0.f;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            double w = (float)sq[c];
            if (fabs(w) <= cut) continue;
            w = 1 / w;
            double amp = 0;
#pragma unroll
            for (int k = 0; k < M; ++k) amp += at[c][k] * rhs[k];                     // float product, double sum
            amp *= w;
#pragma unroll
            for (int k = 0; k < N; ++k) x[k] = (float)(x[k] + amp * vt[c][k]);
        }
    }

    // the caller has written A (at[c][k] = A[k][c]); false: a column vanished, x is not valid (see above)
    HD bool solve(const float (&rhs)[M], float (&x)[N]) {
        orthogonalise_columns();
        order_singular_values();
        normalise_left_vectors();
        back_substitute(rhs, x);
        return full_rank;
    }
};

// util.cpp:1183-1226 on RegSolver; false = rank-deficient system, (q1, q2) not valid
HD bool closest_points_regs(f3 u1, f3 p1, f3 u2, f3 p2, f3 &q1, f3 &q2) {
    const f3 dir = normalized_e(cross(u1, u2));
    RegSolver<9, 9> s;
    s.clear();
    s.at[0][0] = 1.f; s.at[1][1] = 1.f; s.at[2][2] = 1.f;
    s.at[3][0] = -u1.x; s.at[3][1] = -u1.y; s.at[3][2] = -u1.z;
    s.at[4][3] = 1.f; s.at[5][4] = 1.f; s.at[6][5] = 1.f;
    s.at[7][3] = -u2.x; s.at[7][4] = -u2.y; s.at[7][5] = -u2.z;
    s.at[0][6] = -1.f; s.at[1][7] = -1.f; s.at[2][8] = -1.f;
    s.at[4][6] = 1.f; s.at[5][7] = 1.f; s.at[6][8] = 1.f;
    s.at[8][6] = -dir.x; s.at[8][7] = -dir.y; s.at[8][8] = -dir.z;
    const float rhs[9] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, 0.f, 0.f, 0.f};
    float x[9];
    const bool ok = s.solve(rhs, x);
    q1 = f3(x[0], x[1], x[2]);
    q2 = f3(x[4], x[5], x[6]);
    return ok;
}

// util.cpp:1467-1497 on RegSolver
HD bool lines_meet_regs(f3 v1, f3 p1, f3 v2, f3 p2, f3 &out) {
    RegSolver<6, 5> s;
    s.clear();
    s.at[0][0] = 1.f; s.at[1][1] = 1.f; s.at[2][2] = 1.f;
    s.at[0][3] = 1.f; s.at[1][4] = 1.f; s.at[2][5] = 1.f;
    s.at[3][0] = -v1.x; s.at[3][1] = -v1.y; s.at[3][2] = -v1.z;
    s.at[4][3] = -v2.x; s.at[4][4] = -v2.y; s.at[4][5] = -v2.z;
    const float rhs[6] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
    float x[5];
    const bool ok = s.solve(rhs, x);
    out = f3(x[0], x[1], x[2]);
    return ok;
}

#ifdef __HIPCC__
// One M x N least-squares system per lane (M equations, N unknowns; M >= N is not required by the procedure).
template <int M, int N, int TPB>
struct LaneSolver {
    static constexpr int NORM_WORD = (N * M + N * N + 1) / 2 * 2;     // (even: the norms are fp64, also when TPB = 1)
    static constexpr int WORDS_PER_LANE = NORM_WORD + 2 * N;          // A^T, V^T, squared column norms (fp64)
    static constexpr size_t LDS_BYTES = (size_t)WORDS_PER_LANE * TPB * 4;
    float *at, *vt;     // this lane's word 0 of the two matrices
    double *nrm;        // this lane's first norm

    // `mem`: LDS shared by the TPB lanes of a workgroup, or (TPB = 1, lane 0) an 8-byte aligned array of the lane's own
    __device__ LaneSolver(float *mem, int lane) : at(mem + lane), vt(mem + (size_t)N * M * TPB + lane),
                                                  nrm(reinterpret_cast<double *>(mem + (size_t)NORM_WORD * TPB) + lane) {}
    __device__ __forceinline__ float &col(int c, int k) { return at[(c * M + k) * TPB]; }     // entry k of column c of A
    __device__ __forceinline__ float &vrow(int c, int k) { return vt[(c * N + k) * TPB]; }
    __device__ __forceinline__ double &sq(int c) { return nrm[c * TPB]; }

    __device__ void clear() { for (int e = 0; e < N * M; ++e) at[e * TPB] = 0.f; }

    __device__ double column_energy(int c) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) { const float t = col(c, k); acc += (double)t * t; }
        return acc;
    }

    // lapack.cpp:547-614: sweeps over all column pairs until a whole sweep rotates nothing
    __device__ void orthogonalise_columns() {
        const float tol = FLT_EPSILON * 2;
        for (int c = 0; c < N; ++c) {
            sq(c) = column_energy(c);
            for (int k = 0; k < N; ++k) vrow(c, k) = k == c ? 1.f : 0.f;
        }
        const int sweeps = M > 30 ? M : 30;
        for (int sweep = 0; sweep < sweeps; ++sweep) {
            bool rotated = false;
#pragma unroll 1
            for (int a = 0; a < N - 1; ++a)
#pragma unroll 1
                for (int b = a + 1; b < N; ++b) {
                    float ra[M], rb[M];
                    double inner = 0;
#pragma unroll
                    for (int k = 0; k < M; ++k) { ra[k] = col(a, k); rb[k] = col(b, k); inner += (double)ra[k] * rb[k]; }
                    double ea = sq(a), eb = sq(b);
                    if (fabs(inner) <= tol * sqrt(ea * eb)) continue;
                    inner *= 2;
                    const double gap = ea - eb, hyp = hypot_corrected(inner, gap);
                    float cs, sn;
                    if (gap < 0) {
                        const double half = (hyp - gap) * 0.5;
                        sn = (float)sqrt(half / hyp);
                        cs = (float)(inner / (hyp * sn * 2));
                    } else {
                        cs = (float)sqrt((hyp + gap) / (hyp * 2));
                        sn = (float)(inner / (hyp * cs * 2));
                    }
                    ea = 0; eb = 0;
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const float na = cs * ra[k] + sn * rb[k];
                        const float nb = -sn * ra[k] + cs * rb[k];
                        col(a, k) = na; col(b, k) = nb;
                        ea += (double)na * na; eb += (double)nb * nb;
                    }
                    sq(a) = ea; sq(b) = eb;
                    rotated = true;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const float va = vrow(a, k), vb = vrow(b, k);
                        vrow(a, k) = cs * va + sn * vb;
                        vrow(b, k) = -sn * va + cs * vb;
                    }
                }
            if (!rotated) break;
        }
    }

    __device__ void exchange_columns(int a, int b) {
        const double t = sq(a); sq(a) = sq(b); sq(b) = t;
        for (int k = 0; k < M; ++k) { const float x = col(a, k); col(a, k) = col(b, k); col(b, k) = x; }
        for (int k = 0; k < N; ++k) { const float x = vrow(a, k); vrow(a, k) = vrow(b, k); vrow(b, k) = x; }
    }

    // lapack.cpp:616-648: singular values = column norms, largest first (selection sort: the first of equal maxima stays)
    __device__ void order_singular_values() {
        for (int c = 0; c < N; ++c) sq(c) = sqrt(column_energy(c));   // from here on sq() holds the NORMS
        for (int c = 0; c < N - 1; ++c) {
            int top = c;
            for (int k = c + 1; k < N; ++k) if (sq(top) < sq(k)) top = k;
            if (top != c) exchange_columns(c, top);
        }
    }

    // lapack.cpp:650-699: every column becomes a unit vector; one whose norm vanished is rebuilt from a +-1/m sign
    // pattern (cv::RNG(0x12345678): x <- (uint32)x * 4164903690 + (x >> 32), bit 8 of the low word picks the sign),
    // made orthogonal to the columns before it, twice, and L1-normalised in between
    __device__ void complete_left_vectors() {
        const float tol = FLT_EPSILON * 2;
        uint64_t rng = 0x12345678ull;
        for (int c = 0; c < N; ++c) {
            double len = sq(c);
            while (len <= (double)FLT_MIN) {
                const float unit = (float)(1. / M);
                for (int k = 0; k < M; ++k) {
                    rng = (uint64_t)(uint32_t)rng * 4164903690u + (uint32_t)(rng >> 32);
                    col(c, k) = ((uint32_t)rng & 256u) != 0 ? unit : -unit;
                }
                for (int pass = 0; pass < 2; ++pass)
                    for (int j = 0; j < c; ++j) {
                        double proj = 0;
                        for (int k = 0; k < M; ++k) proj += col(c, k) * col(j, k);     // float product, double sum
                        float mass = 0;
                        for (int k = 0; k < M; ++k) {
                            const float t = (float)(col(c, k) - proj * col(j, k));
                            col(c, k) = t;
                            mass += fabsf(t);
                        }
                        mass = mass > tol * 100 ? 1 / mass : 0;
                        for (int k = 0; k < M; ++k) col(c, k) *= mass;
                    }
                len = sqrt(column_energy(c));
            }
            const float inv = (float)(1 / len);
            for (int k = 0; k < M; ++k) col(c, k) *= inv;
        }
    }

    // lapack.cpp:751-795 for one right-hand side: x = sum over the singular triplets above the threshold of
    // v_i (u_i . b) / w_i
    __device__ void back_substitute(const float (&rhs)[M], float (&x)[N]) {
        const float tol = (float)(DBL_EPSILON * 2);
        constexpr int R = M < N ? M : N;
        double cut = 0;
        for (int c = 0; c < R; ++c) cut += (float)sq(c);
        cut *= tol;
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = 0.f;
        for (int c = 0; c < R; ++c) {
            double w = (float)sq(c);
            if (fabs(w) <= cut) continue;
            w = 1 / w;
            double amp = 0;
#pragma unroll
            for (int k = 0; k < M; ++k) amp += col(c, k) * rhs[k];                     // float product, double sum
            amp *= w;
#pragma unroll
            for (int k = 0; k < N; ++k) x[k] = (float)(x[k] + amp * vrow(c, k));
        }
    }

    // the caller has written A (col(c, k) = A[k][c]); `rhs` is b
    __device__ void solve(const float (&rhs)[M], float (&x)[N]) {
        orthogonalise_columns();
        order_singular_values();
        complete_left_vectors();
        back_substitute(rhs, x);
    }
};

// util.cpp:1183-1226: unknowns (point1, t1, point2, t2, s); point1 - t1 u1 = p1, point2 - t2 u2 = p2,
// point2 - point1 - s dir = 0.  `s` is this lane's LaneSolver<9, 9, TPB>.
template <int TPB>
__device__ __forceinline__ void closest_points_solver(LaneSolver<9, 9, TPB> &s, f3 u1, f3 p1, f3 u2, f3 p2, f3 &q1, f3 &q2) {
    const f3 dir = normalized_e(cross(u1, u2));
    s.clear();
    s.col(0, 0) = 1.f; s.col(1, 1) = 1.f; s.col(2, 2) = 1.f;
    s.col(3, 0) = -u1.x; s.col(3, 1) = -u1.y; s.col(3, 2) = -u1.z;
    s.col(4, 3) = 1.f; s.col(5, 4) = 1.f; s.col(6, 5) = 1.f;
    s.col(7, 3) = -u2.x; s.col(7, 4) = -u2.y; s.col(7, 5) = -u2.z;
    s.col(0, 6) = -1.f; s.col(1, 7) = -1.f; s.col(2, 8) = -1.f;
    s.col(4, 6) = 1.f; s.col(5, 7) = 1.f; s.col(6, 8) = 1.f;
    s.col(8, 6) = -dir.x; s.col(8, 7) = -dir.y; s.col(8, 8) = -dir.z;
    const float rhs[9] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, 0.f, 0.f, 0.f};
    float x[9];
    s.solve(rhs, x);
    q1 = f3(x[0], x[1], x[2]);
    q2 = f3(x[4], x[5], x[6]);
}

// util.cpp:1467-1497: unknowns (point, t1, t2); point - t1 v1 = p1, point - t2 v2 = p2
template <int TPB>
__device__ __forceinline__ f3 lines_meet_solver(LaneSolver<6, 5, TPB> &s, f3 v1, f3 p1, f3 v2, f3 p2) {
    s.clear();
    s.col(0, 0) = 1.f; s.col(1, 1) = 1.f; s.col(2, 2) = 1.f;
    s.col(0, 3) = 1.f; s.col(1, 4) = 1.f; s.col(2, 5) = 1.f;
    s.col(3, 0) = -v1.x; s.col(3, 1) = -v1.y; s.col(3, 2) = -v1.z;
    s.col(4, 3) = -v2.x; s.col(4, 4) = -v2.y; s.col(4, 5) = -v2.z;
    const float rhs[6] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
    float x[5];
    s.solve(rhs, x);
    return f3(x[0], x[1], x[2]);
}

// The rare systems RegSolver hands back (a vanished column: lapack.cpp:650-699 completes it with a sign vector): LaneSolver on
// the lane's own scratch memory.  Out of line, so that the fast path's register allocation does not see it.
__device__ __noinline__ void closest_points_completed(f3 u1, f3 p1, f3 u2, f3 p2, f3 &q1, f3 &q2) {
    alignas(8) float mem[LaneSolver<9, 9, 1>::WORDS_PER_LANE];
    LaneSolver<9, 9, 1> s(mem, 0);
    closest_points_solver(s, u1, p1, u2, p2, q1, q2);
}
__device__ __noinline__ f3 lines_meet_completed(f3 v1, f3 p1, f3 v2, f3 p2) {
    alignas(8) float mem[LaneSolver<6, 5, 1>::WORDS_PER_LANE];
    LaneSolver<6, 5, 1> s(mem, 0);
    return lines_meet_solver(s, v1, p1, v2, p2);
}
// the two solves as the kernels call them
__device__ __forceinline__ void closest_points_svd(f3 u1, f3 p1, f3 u2, f3 p2, f3 &q1, f3 &q2) {
    if (!closest_points_regs(u1, p1, u2, p2, q1, q2)) closest_points_completed(u1, p1, u2, p2, q1, q2);
}
__device__ __forceinline__ f3 lines_meet_svd(f3 v1, f3 p1, f3 v2, f3 p2) {
    f3 o;
    if (!lines_meet_regs(v1, p1, v2, p2, o)) o = lines_meet_completed(v1, p1, v2, p2);
    return o;
}
#endif

}  // namespace plade
