// plade_amd/csrc/hostgeom.h -- PCA oriented bounding boxes of the voxel-downsampled clouds (ComputeBoundingBox,
// code/PLADE/util.h:186-248): the 3x3 eigen-solver and the box construction as host+device functions (k_obb.hip runs
// them on the GPU, one workgroup per cloud / per plane).
#pragma once
#include "geom.h"

namespace plade {

// Eigen::SelfAdjointEigenSolver<Matrix3f>::compute (iterative QR):
// Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h:420-468 (scaling), Tridiagonalization.h:464-504
// (3x3 in-place tridiagonalisation), :504-572 + :839-900 (implicit symmetric QR with Wilkinson
// shift), eigenvalues ascending, eigenvectors in the columns of `vec`.
HD void symmetric_eigen3(const m3 &cov, float val[3], m3 &vec) {
    float a00 = cov.m[0][0], a10 = cov.m[1][0], a11 = cov.m[1][1], a20 = cov.m[2][0], a21 = cov.m[2][1], a22 = cov.m[2][2];
    float scale = fmaxf(fmaxf(fmaxf(fabsf(a00), fabsf(a10)), fmaxf(fabsf(a20), fabsf(a11))), fmaxf(fabsf(a21), fabsf(a22)));
    if (scale == 0.f) scale = 1.f;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    float diag[3], sub[2], q[3][3];
    diag[0] = a00;
    const float v1norm2 = a20 * a20;
    if (v1norm2 <= FLT_MIN) {
        diag[1] = a11; diag[2] = a22; sub[0] = a10; sub[1] = a21;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) q[r][c] = r == c ? 1.f : 0.f;
    } else {
        const float beta = sqrtf(a10 * a10 + v1norm2);
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
            if (fabsf(sub[i]) < FLT_MIN) sub[i] = 0.f;
            else {
                const float ss = precision_inv * sub[i];
                if (ss * ss <= (fabsf(diag[i]) + fabsf(diag[i + 1]))) sub[i] = 0.f;
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
        if (td == 0.f) mu -= fabsf(e);
        else if (e != 0.f) {
            const float e2 = e * e;
            const float ax = fabsf(td), ay = fabsf(e);
            const float p = fmaxf(ax, ay);
            float h = 0.f;
            if (p != 0.f) { const float qp = fminf(ay, ax) / p; h = p * sqrtf(1.f + qp * qp); }
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
                { const float t = diag[i]; diag[i] = diag[k + i]; diag[k + i] = t; }
                for (int r = 0; r < 3; ++r) { const float t = q[r][i]; q[r][i] = q[r][k + i]; q[r][k + i] = t; }
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
// transformPointCloud into the eigen frame -> getMinMax3D -> centre / extents / 8 corners.  Split at the two places
// where all points are visited: the caller supplies the sums and the min / max.

// the six covariance terms of one point about the centroid, in PCL's order of operations (centroid.hpp:263-287)
HD void cov_add_point(float acc[6] /* 11 12 22 00 01 02 */, f3 p, const float c[3]) {
    const float px = p.x - c[0], py = p.y - c[1], pz = p.z - c[2];
    acc[0] += py * py;
    acc[1] += py * pz;
    acc[2] += pz * pz;
    acc[3] += px * px;   // pt *= pt.x()
    acc[4] += py * px;
    acc[5] += pz * px;
}

// eigen frame from the summed covariance terms: E (columns = axes, the third one = col0 x col1) and the 3x4 transform
// P = [E^T | -E^T c] into that frame
HD void obb_frame(const float cov6[6], float nf, const float c[3], m3 &E, float P[12]) {
    m3 cov;
    cov.m[1][1] = cov6[0]; cov.m[1][2] = cov6[1]; cov.m[2][2] = cov6[2];
    cov.m[0][0] = cov6[3]; cov.m[0][1] = cov6[4]; cov.m[0][2] = cov6[5];
    cov.m[1][0] = cov.m[0][1]; cov.m[2][0] = cov.m[0][2]; cov.m[2][1] = cov.m[1][2];
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) cov.m[r][k] /= nf;
    float ev[3];
    symmetric_eigen3(cov, ev, E);
    const f3 c0(E.m[0][0], E.m[1][0], E.m[2][0]), c1(E.m[0][1], E.m[1][1], E.m[2][1]);
    const f3 c2 = cross(c0, c1);
    E.m[0][2] = c2.x; E.m[1][2] = c2.y; E.m[2][2] = c2.z;
    m3 Et;
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Et.m[r][k] = E.m[k][r];
    const f3 cen(c[0], c[1], c[2]);
    const f3 t = -1.f * mul_e(Et, cen);
    P[0] = Et.m[0][0]; P[1] = Et.m[0][1]; P[2] = Et.m[0][2]; P[3] = t.x;
    P[4] = Et.m[1][0]; P[5] = Et.m[1][1]; P[6] = Et.m[1][2]; P[7] = t.y;
    P[8] = Et.m[2][0]; P[9] = Et.m[2][1]; P[10] = Et.m[2][2]; P[11] = t.z;
}

// centre, extents and corners from the min / max of the points in the eigen frame
HD void obb_finish(const m3 &E, const float c[3], f3 mn, f3 mx, f3 &center, double whd[3], f3 corners[8]) {
    const f3 cen(c[0], c[1], c[2]);
    const f3 mean_diag = 0.5f * (mx + mn);
    center = mul_e(E, mean_diag) + cen;
    whd[0] = mx.x - mn.x;   // width  (float subtraction, widened)
    whd[2] = mx.y - mn.y;   // depth
    whd[1] = mx.z - mn.z;   // height
    const float x = mn.x, y = mn.y, z = mn.z;
    const double w = whd[0], d = whd[2], h = whd[1];
    const f3 cs[8] = {mn,
                      f3(x, (float)(y + d), z),
                      f3(x, (float)(y + d), (float)(z + h)),
                      f3(x, y, (float)(z + h)),
                      f3((float)(x + w), y, (float)(z + h)),
                      f3((float)(x + w), (float)(y + d), z),
                      f3((float)(x + w), y, z),
                      f3((float)(x + w), (float)(y + d), (float)(z + h))};
    const float Q[12] = {E.m[0][0], E.m[0][1], E.m[0][2], cen.x, E.m[1][0], E.m[1][1], E.m[1][2], cen.y,
                         E.m[2][0], E.m[2][1], E.m[2][2], cen.z};
    for (int i = 0; i < 8; ++i) corners[i] = pcl_xform(Q, cs[i]);
}

constexpr int OBB_LANES = 1024;   // lanes of the strided summation order (see k_obb.hip)

}  // namespace plade
