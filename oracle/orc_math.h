// oracle/orc_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/plade_oracle.cpp).
//
// Small fixed-size linear algebra written to evaluate in the SAME operation
// order as the Eigen 3.4.0 expressions the reference uses, so that fp32 results
// are bit-identical to the reference's on x86-64 without FMA contraction:
//   * fixed-size 3-vectors reduce as  x0 + (x1 + x2)   (Eigen redux unroller;
//     pinned by tests/test_oracle_vs_ref.py against real Eigen via oracle/_ref)
//   * dynamic-size vectors reduce sequentially ((x0 + x1) + x2)
//   * Matrix3f * Vector3f is a per-row fixed-size dot.
// Also: restatements of Eigen::SelfAdjointEigenSolver<Matrix3f>::compute
// (Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h:420-468, 504-572, 839-900,
//  Tridiagonalization.h:464-504, Jacobi/Jacobi.h:231-268), of JacobiSVD 3x3
// (SVD/JacobiSVD.h:666-790, misc/RealSvd2x2.h:19-49, Jacobi/Jacobi.h:94-125)
// and of Eigen::umeyama without scaling (Geometry/Umeyama.h:94-160).
#pragma once
#include <cmath>
#include <cfloat>
#include <cstring>
#include <algorithm>

namespace orc {

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
// Eigen fixed-size order
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
// Eigen dynamic-size (sequential) order
inline float dot_seq(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float sqnorm(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return std::sqrt(sqnorm(a)); }
inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// MatrixBase::normalize(): z = squaredNorm(); if (z > 0) v /= sqrt(z)
inline void normalize(V3 &a) {
    float z = sqnorm(a);
    if (z > 0.f) {
        float s = std::sqrt(z);
        a = a / s;
    }
}

struct M3 {
    float m[3][3];  // row-major m[r][c]
    float operator()(int r, int c) const { return m[r][c]; }
    float &operator()(int r, int c) { return m[r][c]; }
};
inline V3 mul(const M3 &R, V3 v) {
    return V3(R.m[0][0] * v.x + (R.m[0][1] * v.y + R.m[0][2] * v.z),
              R.m[1][0] * v.x + (R.m[1][1] * v.y + R.m[1][2] * v.z),
              R.m[2][0] * v.x + (R.m[2][1] * v.y + R.m[2][2] * v.z));
}
inline M3 transpose(const M3 &a) {
    M3 t;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) t.m[r][c] = a.m[c][r];
    return t;
}
inline float det3(const M3 &a) {
    // Eigen determinant_impl<Derived,3>: bruteforce_det3_helper
    // helper(a,b,c) = m(0,a) * (m(1,b)*m(2,c) - m(1,c)*m(2,b));  det = h(0,1,2) - h(1,0,2) + h(2,0,1)
    auto h = [&](int i, int j, int k) {
        return a.m[0][i] * (a.m[1][j] * a.m[2][k] - a.m[1][k] * a.m[2][j]);
    };
    return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
}

// --------------------------------------------------------------------------
// Givens / Jacobi rotations (Eigen/src/Jacobi/Jacobi.h)
struct Rot { float c, s; };

inline Rot make_givens(float p, float q) {  // Jacobi.h:231-268
    Rot r;
    if (q == 0.f) { r.c = p < 0.f ? -1.f : 1.f; r.s = 0.f; }
    else if (p == 0.f) { r.c = 0.f; r.s = q < 0.f ? 1.f : -1.f; }
    else if (std::fabs(p) > std::fabs(q)) {
        float t = q / p;
        float u = std::sqrt(1.f + t * t);
        if (p < 0.f) u = -u;
        r.c = 1.f / u;
        r.s = -t * r.c;
    } else {
        float t = p / q;
        float u = std::sqrt(1.f + t * t);
        if (q < 0.f) u = -u;
        r.s = -1.f / u;
        r.c = -t * r.s;
    }
    return r;
}

inline bool make_jacobi(float x, float y, float z, Rot &j) {  // Jacobi.h:94-125
    float deno = 2.f * std::fabs(y);
    if (deno < FLT_MIN) { j.c = 1.f; j.s = 0.f; return false; }
    float tau = (x - z) / deno;
    float w = std::sqrt(tau * tau + 1.f);
    float t = tau > 0.f ? 1.f / (tau + w) : 1.f / (tau - w);
    float sign_t = t > 0.f ? 1.f : -1.f;
    float n = 1.f / std::sqrt(t * t + 1.f);
    j.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
    j.c = n;
    return true;
}

// apply_rotation_in_the_plane(x, y, j): x' = c x + s y ; y' = -s x + c y
inline void rot_apply(float &x, float &y, Rot j) {
    float xi = x, yi = y;
    x = j.c * xi + j.s * yi;
    y = -j.s * xi + j.c * yi;
}
inline Rot rot_transpose(Rot j) { return Rot{j.c, -j.s}; }
// JacobiRotation operator*: c = c1 c2 - s1 s2 ; s = c1 s2 + s1 c2  (real case)
inline Rot rot_mul(Rot a, Rot b) { return Rot{a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }

// hypot() as the reference's OpenCV calls it (lapack.cpp:579: the C library's).  The GPU cannot call glibc, so oracle
// and kernel (plade_amd/csrc/k_svd.h) evaluate the SAME explicit formula: the kernel of glibc 2.35's dbl-64 hypot without
// FMA (sqrt of the plain sum, then one correction step with the residual h^2 - x^2 - y^2 split into two nearly exact
// terms); glibc rescales only above 2^511 / below 2^-459, far outside what the 9 x 9 systems produce.
// tests/test_host_logic.py::test_hypot_formula_is_glibc_s compares it with this machine's libm bit for bit.
inline double hypot_glibc235(double x, double y) {
    double ax = std::fabs(x), ay = std::fabs(y);
    if (ax < ay) std::swap(ax, ay);
    double h = std::sqrt(ax * ax + ay * ay), t1, t2;
    if (h <= 2.0 * ay) {
        double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

// positive_real_hypot (Eigen/src/Core/MathFunctionsImpl.h:80-94) on |x|,|y|
inline float eig_hypot(float x, float y) {
    x = std::fabs(x); y = std::fabs(y);
    float p = std::max(x, y);
    if (p == 0.f) return 0.f;
    float qp = std::min(y, x) / p;
    return p * std::sqrt(1.f + qp * qp);
}

// Eigen::SelfAdjointEigenSolver<Matrix3f>(cov, ComputeEigenvectors): eigenvalues
// ascending in evals, eigenvectors in the COLUMNS of evecs.
inline void selfadjoint_eig3(const M3 &cov, float evals[3], M3 &evecs) {
    // lower triangular view, scaled to [-1,1]  (SelfAdjointEigenSolver.h:451-455)
    float a00 = cov.m[0][0], a10 = cov.m[1][0], a11 = cov.m[1][1], a20 = cov.m[2][0],
          a21 = cov.m[2][1], a22 = cov.m[2][2];
    float scale = 0.f;
    {
        // mat = lower triangular (upper part zero); cwiseAbs().maxCoeff()
        float vals[6] = {a00, a10, a20, a11, a21, a22};
        for (int i = 0; i < 6; ++i) scale = std::max(scale, std::fabs(vals[i]));
    }
    if (scale == 0.f) scale = 1.f;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    float diag[3], sub[2];
    float q[3][3];  // q[r][c]
    // Tridiagonalization.h:464-504
    diag[0] = a00;
    float v1norm2 = a20 * a20;
    if (v1norm2 <= FLT_MIN) {
        diag[1] = a11; diag[2] = a22; sub[0] = a10; sub[1] = a21;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) q[r][c] = (r == c) ? 1.f : 0.f;
    } else {
        float beta = std::sqrt(a10 * a10 + v1norm2);
        float invBeta = 1.f / beta;
        float m01 = a10 * invBeta;
        float m02 = a20 * invBeta;
        float qq = 2.f * m01 * a21 + m02 * (a22 - a11);
        diag[1] = a11 + m02 * qq;
        diag[2] = a22 - m02 * qq;
        sub[0] = beta;
        sub[1] = a21 - m01 * qq;
        float t[3][3] = {{1, 0, 0}, {0, m01, m02}, {0, m02, -m01}};
        memcpy(q, t, sizeof(q));
    }
    // computeFromTridiagonal_impl (SelfAdjointEigenSolver.h:504-572), n = 3, maxIter = 30
    const int n = 3;
    int end = n - 1, start = 0, iter = 0;
    const float considerAsZero = FLT_MIN;
    const float precision_inv = 1.f / FLT_EPSILON;
    bool ok = true;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            if (std::fabs(sub[i]) < considerAsZero) sub[i] = 0.f;
            else {
                const float ss = precision_inv * sub[i];
                if (ss * ss <= (std::fabs(diag[i]) + std::fabs(diag[i + 1]))) sub[i] = 0.f;
            }
        }
        while (end > 0 && sub[end - 1] == 0.f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > 30 * n) { ok = false; break; }
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.f) start--;
        // tridiagonal_qr_step (SelfAdjointEigenSolver.h:839-900)
        float td = (diag[end - 1] - diag[end]) * 0.5f;
        float e = sub[end - 1];
        float mu = diag[end];
        if (td == 0.f) mu -= std::fabs(e);
        else if (e != 0.f) {
            const float e2 = e * e;
            const float h = eig_hypot(td, e);
            if (e2 == 0.f) mu -= e / ((td + (td > 0.f ? h : -h)) / e);
            else mu -= e2 / (td + (td > 0.f ? h : -h));
        }
        float x = diag[start] - mu;
        float z = sub[start];
        for (int k = start; k < end && z != 0.f; ++k) {
            Rot rot = make_givens(x, z);
            float sdk = rot.s * diag[k] + rot.c * sub[k];
            float dkp1 = rot.s * sub[k] + rot.c * diag[k + 1];
            diag[k] = rot.c * (rot.c * diag[k] - rot.s * sub[k]) -
                      rot.s * (rot.c * sub[k] - rot.s * diag[k + 1]);
            diag[k + 1] = rot.s * sdk + rot.c * dkp1;
            sub[k] = rot.c * sdk - rot.s * dkp1;
            if (k > start) sub[k - 1] = rot.c * sub[k - 1] - rot.s * z;
            x = sub[k];
            if (k < end - 1) {
                z = -rot.s * sub[k + 1];
                sub[k + 1] = rot.c * sub[k + 1];
            }
            // q.applyOnTheRight(k, k+1, rot): columns k,k+1 rotated by rot.transpose()
            Rot jt = rot_transpose(rot);
            for (int r = 0; r < 3; ++r) rot_apply(q[r][k], q[r][k + 1], jt);
        }
    }
    if (ok) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0;
            float mn = diag[i];
            for (int j = 1; j < n - i; ++j)
                if (diag[i + j] < mn) { mn = diag[i + j]; k = j; }
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; ++r) std::swap(q[r][i], q[r][k + i]);
            }
        }
    }
    for (int i = 0; i < 3; ++i) evals[i] = diag[i] * scale;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) evecs.m[r][c] = q[r][c];
}

// JacobiSVD<Matrix3f>(sigma, ComputeFullU|ComputeFullV) -- U, V (columns), no sort needed
// for U*S*V^T but Eigen sorts singular values descending and permutes U,V columns; the
// determinant sign test and U S V^T are permutation-consistent, we reproduce the sort too.
inline void jacobi_svd3(const M3 &A, M3 &U, float sv[3], M3 &V) {
    const float precision = 2.f * FLT_EPSILON;
    const float considerAsZero = FLT_MIN;
    float scale = 0.f;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) scale = std::max(scale, std::fabs(A.m[r][c]));
    if (scale == 0.f) scale = 1.f;
    float W[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        W[r][c] = A.m[r][c] / scale;
        U.m[r][c] = V.m[r][c] = (r == c) ? 1.f : 0.f;
    }
    float maxDiag = std::max(std::fabs(W[0][0]), std::max(std::fabs(W[1][1]), std::fabs(W[2][2])));
    bool finished = false;
    while (!finished) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                float threshold = std::max(considerAsZero, precision * maxDiag);
                if (std::fabs(W[p][q]) > threshold || std::fabs(W[q][p]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd (misc/RealSvd2x2.h:19-49)
                    float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                    Rot rot1;
                    float t = m00 + m11;
                    float d = m10 - m01;
                    if (std::fabs(d) < FLT_MIN) { rot1.s = 0.f; rot1.c = 1.f; }
                    else {
                        float u = t / d;
                        float tmp = std::sqrt(1.f + u * u);
                        rot1.s = 1.f / tmp;
                        rot1.c = u / tmp;
                    }
                    // m.applyOnTheLeft(0,1,rot1): rows 0,1 of m: for each column
                    rot_apply(m00, m10, rot1);
                    rot_apply(m01, m11, rot1);
                    Rot j_right;
                    make_jacobi(m00, m01, m11, j_right);
                    Rot j_left = rot_mul(rot1, rot_transpose(j_right));
                    // m_workMatrix.applyOnTheLeft(p,q,j_left): rows p,q
                    for (int c = 0; c < 3; ++c) rot_apply(W[p][c], W[q][c], j_left);
                    // m_matrixU.applyOnTheRight(p,q,j_left.transpose()): columns p,q with j^T^T = j_left
                    for (int r = 0; r < 3; ++r) rot_apply(U.m[r][p], U.m[r][q], j_left);
                    // m_workMatrix.applyOnTheRight(p,q,j_right): columns with j_right.transpose()
                    Rot jrt = rot_transpose(j_right);
                    for (int r = 0; r < 3; ++r) rot_apply(W[r][p], W[r][q], jrt);
                    for (int r = 0; r < 3; ++r) rot_apply(V.m[r][p], V.m[r][q], jrt);
                    maxDiag = std::max(maxDiag, std::max(std::fabs(W[p][p]), std::fabs(W[q][q])));
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        float a = W[i][i];
        sv[i] = std::fabs(a);
        if (a < 0.f) for (int r = 0; r < 3; ++r) U.m[r][i] = -U.m[r][i];
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    // sort descending (JacobiSVD.h:773-790)
    for (int i = 0; i < 3; ++i) {
        int pos = 0;
        float mx = sv[i];
        for (int j = 1; j < 3 - i; ++j)
            if (sv[i + j] > mx) { mx = sv[i + j]; pos = j; }
        if (mx == 0.f) break;
        if (pos) {
            pos += i;
            std::swap(sv[i], sv[pos]);
            for (int r = 0; r < 3; ++r) { std::swap(U.m[r][i], U.m[r][pos]); std::swap(V.m[r][i], V.m[r][pos]); }
        }
    }
}

// Rotation part of Eigen::umeyama(src, dst, false) for 3 points each (columns).
// src/dst given as 3 V3 points.  (Geometry/Umeyama.h:94-160; PCL call site
// registration/impl/transformation_estimation_svd.hpp:118-148)
inline M3 umeyama_rotation3(const V3 s[3], const V3 d[3]) {
    const float one_over_n = 1.f / 3.f;
    // rowwise().sum() over dynamic #cols: sequential
    V3 sm(((s[0].x + s[1].x) + s[2].x) * one_over_n, ((s[0].y + s[1].y) + s[2].y) * one_over_n,
          ((s[0].z + s[1].z) + s[2].z) * one_over_n);
    V3 dm(((d[0].x + d[1].x) + d[2].x) * one_over_n, ((d[0].y + d[1].y) + d[2].y) * one_over_n,
          ((d[0].z + d[1].z) + d[2].z) * one_over_n);
    V3 sd[3], dd[3];
    for (int i = 0; i < 3; ++i) { sd[i] = s[i] - sm; dd[i] = d[i] - dm; }
    // sigma = one_over_n * dst_demean * src_demean^T  (3xN * Nx3, dynamic inner size ->
    // coefficient-based lazy product, sequential inner sum, scalar factor applied last;
    // order pinned experimentally against Eigen 3.4.0, see tests/test_oracle_vs_ref.py)
    M3 sigma;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            sigma.m[r][c] = one_over_n * ((dd[0][r] * sd[0][c] + dd[1][r] * sd[1][c]) + dd[2][r] * sd[2][c]);
        }
    M3 U, V;
    float sv[3];
    jacobi_svd3(sigma, U, sv, V);
    float S[3] = {1.f, 1.f, 1.f};
    if (det3(U) * det3(V) < 0.f) S[2] = -1.f;
    // R = U * S.asDiagonal() * V^T : ((U*S) * V^T), fixed-size 3 inner product (tree order)
    M3 R;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float a0 = (U.m[r][0] * S[0]) * V.m[c][0];
            float a1 = (U.m[r][1] * S[1]) * V.m[c][1];
            float a2 = (U.m[r][2] * S[2]) * V.m[c][2];
            R.m[r][c] = a0 + (a1 + a2);
        }
    return R;
}

}  // namespace orc
