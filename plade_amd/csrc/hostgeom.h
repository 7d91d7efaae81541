// plade_amd/csrc/hostgeom.h -- the small, strictly sequential fp32 pieces the C++ host keeps
// (exactly the work the reference's host does per cloud / per plane): PCA oriented bounding boxes
// of voxel-downsampled clouds.  Sequential accumulation order is part of the result, so these run on
// the host over the (small) downsampled clouds copied back from the GPU.
#pragma once
#include "geom.h"
#include <vector>
#include <algorithm>

namespace plade {

// Eigen::SelfAdjointEigenSolver<Matrix3f>::compute (iterative QR):
// Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h:420-468 (scaling), Tridiagonalization.h:464-504
// (3x3 in-place tridiagonalisation), :504-572 + :839-900 (implicit symmetric QR with Wilkinson
// shift), eigenvalues ascending, eigenvectors in the columns of `vec`.
inline void symmetric_eigen3(const m3 &cov, float val[3], m3 &vec) {
    float a00 = cov.m[0][0], a10 = cov.m[1][0], a11 = cov.m[1][1], a20 = cov.m[2][0], a21 = cov.m[2][1], a22 = cov.m[2][2];
    float scale = std::max({std::fabs(a00), std::fabs(a10), std::fabs(a20), std::fabs(a11), std::fabs(a21), std::fabs(a22)});
    if (scale == 0.f) scale = 1.f;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    float diag[3], sub[2], q[3][3];
    diag[0] = a00;
    const float v1norm2 = a20 * a20;
    if (v1norm2 <= FLT_MIN) {
        diag[1] = a11; diag[2] = a22; sub[0] = a10; sub[1] = a21;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) q[r][c] = r == c ? 1.f : 0.f;
    } else {
        const float beta = std::sqrt(a10 * a10 + v1norm2);
        const float invBeta = 1.f / beta;
        const float m01 = a10 * invBeta, m02 = a20 * invBeta;
        const float qq = 2.f * m01 * a21 + m02 * (a22 - a11);
        diag[1] = a11 + m02 * qq;
        diag[2] = a22 - m02 * qq;
        sub[0] = beta;
        sub[1] = a21 - m01 * qq;
        q[0][0] = 1; q[0][1] = 0; q[0][2] = 0;
        q[1][0] = 0; q[1][1] = m01; q[1][2] = m02;
        q[2][0] = 0; q[2][1] = m02; q[2][2] = -m01;
    }
    int end = 2, start = 0, iter = 0;
    const float precision_inv = 1.f / FLT_EPSILON;
    bool converged = true;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            if (std::fabs(sub[i]) < FLT_MIN) sub[i] = 0.f;
            else {
                const float ss = precision_inv * sub[i];
                if (ss * ss <= (std::fabs(diag[i]) + std::fabs(diag[i + 1]))) sub[i] = 0.f;
            }
        }
        while (end > 0 && sub[end - 1] == 0.f) end--;
        if (end <= 0) break;
        if (++iter > 90) { converged = false; break; }
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.f) start--;
        float td = (diag[end - 1] - diag[end]) * 0.5f;
        const float e = sub[end - 1];
        float mu = diag[end];
        if (td == 0.f) mu -= std::fabs(e);
        else if (e != 0.f) {
            const float e2 = e * e;
            const float ax = std::fabs(td), ay = std::fabs(e);
            const float p = std::max(ax, ay);
            float h = 0.f;
            if (p != 0.f) { const float qp = std::min(ay, ax) / p; h = p * std::sqrt(1.f + qp * qp); }
            if (e2 == 0.f) mu -= e / ((td + (td > 0.f ? h : -h)) / e);
            else mu -= e2 / (td + (td > 0.f ? h : -h));
        }
        float x = diag[start] - mu, z = sub[start];
        for (int k = start; k < end && z != 0.f; ++k) {
            const rot2 g = givens(x, z);
            const float sdk = g.s * diag[k] + g.c * sub[k];
            const float dkp1 = g.s * sub[k] + g.c * diag[k + 1];
            diag[k] = g.c * (g.c * diag[k] - g.s * sub[k]) - g.s * (g.c * sub[k] - g.s * diag[k + 1]);
            diag[k + 1] = g.s * sdk + g.c * dkp1;
            sub[k] = g.c * sdk - g.s * dkp1;
            if (k > start) sub[k - 1] = g.c * sub[k - 1] - g.s * z;
            x = sub[k];
            if (k < end - 1) { z = -g.s * sub[k + 1]; sub[k + 1] = g.c * sub[k + 1]; }
            const rot2 gt = rot_t(g);
            for (int r = 0; r < 3; ++r) rot_apply(q[r][k], q[r][k + 1], gt);
        }
    }
    if (converged)
        for (int i = 0; i < 2; ++i) {
            int k = 0;
            float mn = diag[i];
            for (int j = 1; j < 3 - i; ++j) if (diag[i + j] < mn) { mn = diag[i + j]; k = j; }
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; ++r) std::swap(q[r][i], q[r][k + i]);
            }
        }
    for (int i = 0; i < 3; ++i) val[i] = diag[i] * scale;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) vec.m[r][c] = q[r][c];
}

struct Obb {
    f3 center;
    double width = 0, height = 0, depth = 0;
    f3 corners[8];
};

// ComputeBoundingBox (code/PLADE/util.h:186-248): pcl::compute3DCentroid + computeCovarianceMatrixNormalized
// (pcl-1.8.1/common/include/pcl/common/impl/centroid.hpp:79-121, 250-300) -> eigenvectors ->
// transformPointCloud into the eigen frame -> getMinMax3D -> centre / extents / 8 corners.
inline bool oriented_bbox(const float *xyz, size_t n, Obb &o, bool corners) {
    if (n == 0) return false;
    float c[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) { c[0] += xyz[3 * i]; c[1] += xyz[3 * i + 1]; c[2] += xyz[3 * i + 2]; }
    const float nf = (float)n;
    c[0] /= nf; c[1] /= nf; c[2] /= nf;
    m3 cov;
    memset(&cov, 0, sizeof(cov));
    for (size_t i = 0; i < n; ++i) {
        const float px = xyz[3 * i] - c[0], py = xyz[3 * i + 1] - c[1], pz = xyz[3 * i + 2] - c[2];
        cov.m[1][1] += py * py;
        cov.m[1][2] += py * pz;
        cov.m[2][2] += pz * pz;
        cov.m[0][0] += px * px;
        cov.m[0][1] += py * px;
        cov.m[0][2] += pz * px;
    }
    cov.m[1][0] = cov.m[0][1]; cov.m[2][0] = cov.m[0][2]; cov.m[2][1] = cov.m[1][2];
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) cov.m[r][k] /= nf;
    float ev[3];
    m3 E;
    symmetric_eigen3(cov, ev, E);
    const f3 c0(E.m[0][0], E.m[1][0], E.m[2][0]), c1(E.m[0][1], E.m[1][1], E.m[2][1]);
    const f3 c2 = cross(c0, c1);
    E.m[0][2] = c2.x; E.m[1][2] = c2.y; E.m[2][2] = c2.z;
    m3 Et;
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Et.m[r][k] = E.m[k][r];
    const f3 cen(c[0], c[1], c[2]);
    const f3 t = -1.f * mul_e(Et, cen);
    float P[12] = {Et.m[0][0], Et.m[0][1], Et.m[0][2], t.x, Et.m[1][0], Et.m[1][1], Et.m[1][2], t.y,
                   Et.m[2][0], Et.m[2][1], Et.m[2][2], t.z};
    f3 mn(FLT_MAX, FLT_MAX, FLT_MAX), mx(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (size_t i = 0; i < n; ++i) {
        const f3 q = pcl_xform(P, f3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        mn.x = std::min(mn.x, q.x); mn.y = std::min(mn.y, q.y); mn.z = std::min(mn.z, q.z);
        mx.x = std::max(mx.x, q.x); mx.y = std::max(mx.y, q.y); mx.z = std::max(mx.z, q.z);
    }
    const f3 mean_diag = 0.5f * (mx + mn);
    o.center = mul_e(E, mean_diag) + cen;
    o.width = mx.x - mn.x;
    o.depth = mx.y - mn.y;
    o.height = mx.z - mn.z;
    if (corners) {
        const float x = mn.x, y = mn.y, z = mn.z;
        const double w = o.width, d = o.depth, h = o.height;
        const f3 cs[8] = {mn,
                          f3(x, (float)(y + d), z),
                          f3(x, (float)(y + d), (float)(z + h)),
                          f3(x, y, (float)(z + h)),
                          f3((float)(x + w), y, (float)(z + h)),
                          f3((float)(x + w), (float)(y + d), z),
                          f3((float)(x + w), y, z),
                          f3((float)(x + w), (float)(y + d), (float)(z + h))};
        float Q[12] = {E.m[0][0], E.m[0][1], E.m[0][2], cen.x, E.m[1][0], E.m[1][1], E.m[1][2], cen.y,
                       E.m[2][0], E.m[2][1], E.m[2][2], cen.z};
        for (int i = 0; i < 8; ++i) o.corners[i] = pcl_xform(Q, cs[i]);
    }
    return true;
}

}  // namespace plade
