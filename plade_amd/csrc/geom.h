// plade_amd/csrc/geom.h -- small fixed-size geometry used by the registration pipeline, callable
// from host and device.  Each routine states the reference expression it evaluates; fp32 op order
// follows Eigen 3.4 (fixed-size reductions x0 + (x1 + x2), dynamic ones sequential) so that the
// discrete decisions downstream (radius tests, gates, cluster membership) agree with the CPU path.
#pragma once
#include "common.h"

namespace plade {

// ---- Jacobi / Givens rotations (Eigen/src/Jacobi/Jacobi.h) -------------------------------------
struct rot2 { float c, s; };

HD rot2 givens(float p, float q) {  // JacobiRotation::makeGivens, real case (Jacobi.h:231-268)
    rot2 r;
    if (q == 0.f) { r.c = p < 0.f ? -1.f : 1.f; r.s = 0.f; }
    else if (p == 0.f) { r.c = 0.f; r.s = q < 0.f ? 1.f : -1.f; }
    else if (fabsf(p) > fabsf(q)) {
        float t = q / p, u = sqrtf(1.f + t * t);
        if (p < 0.f) u = -u;
        r.c = 1.f / u; r.s = -t * r.c;
    } else {
        float t = p / q, u = sqrtf(1.f + t * t);
        if (q < 0.f) u = -u;
        r.s = -1.f / u; r.c = -t * r.s;
    }
    return r;
}
HD rot2 jacobi2(float x, float y, float z) {  // JacobiRotation::makeJacobi (Jacobi.h:94-125)
    rot2 j;
    float deno = 2.f * fabsf(y);
    if (deno < FLT_MIN) { j.c = 1.f; j.s = 0.f; return j; }
    float tau = (x - z) / deno;
    float w = sqrtf(tau * tau + 1.f);
    float t = tau > 0.f ? 1.f / (tau + w) : 1.f / (tau - w);
    float sign_t = t > 0.f ? 1.f : -1.f;
    float n = 1.f / sqrtf(t * t + 1.f);
    j.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    j.c = n;
    return j;
}
HD void rot_apply(float &x, float &y, rot2 j) {  // apply_rotation_in_the_plane
    float xi = x, yi = y;
    x = j.c * xi + j.s * yi;
    y = -j.s * xi + j.c * yi;
}
HD rot2 rot_t(rot2 j) { rot2 r; r.c = j.c; r.s = -j.s; return r; }
HD rot2 rot_mul(rot2 a, rot2 b) { rot2 r; r.c = a.c * b.c - a.s * b.s; r.s = a.c * b.s + a.s * b.c; return r; }

HD float det3_e(const m3 &a) {  // Eigen bruteforce_det3_helper order
    float h0 = a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]);
    float h1 = a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]);
    float h2 = a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
    return h0 - h1 + h2;
}

// JacobiSVD<Matrix3f>(A, ComputeFullU | ComputeFullV) (Eigen/src/SVD/JacobiSVD.h:666-790 with
// misc/RealSvd2x2.h:19-49); U, V hold the singular vectors in their columns.
HD void jacobi_svd3(const m3 &A, m3 &U, float sv[3], m3 &V) {
    const float precision = 2.f * FLT_EPSILON;
    float scale = 0.f;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) scale = fmaxf(scale, fabsf(A.m[r][c]));
    if (scale == 0.f) scale = 1.f;
    float W[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        W[r][c] = A.m[r][c] / scale;
        U.m[r][c] = V.m[r][c] = (r == c) ? 1.f : 0.f;
    }
    float maxDiag = fmaxf(fabsf(W[0][0]), fmaxf(fabsf(W[1][1]), fabsf(W[2][2])));
    bool finished = false;
    int guard = 0;
    while (!finished && guard++ < 200) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                float threshold = fmaxf(FLT_MIN, precision * maxDiag);
                if (fabsf(W[p][q]) > threshold || fabsf(W[q][p]) > threshold) {
                    finished = false;
                    float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                    rot2 rot1;
                    float t = m00 + m11, d = m10 - m01;
                    if (fabsf(d) < FLT_MIN) { rot1.s = 0.f; rot1.c = 1.f; }
                    else {
                        float u = t / d, tmp = sqrtf(1.f + u * u);
                        rot1.s = 1.f / tmp; rot1.c = u / tmp;
                    }
                    rot_apply(m00, m10, rot1);
                    rot_apply(m01, m11, rot1);
                    rot2 jr = jacobi2(m00, m01, m11);
                    rot2 jl = rot_mul(rot1, rot_t(jr));
                    for (int c = 0; c < 3; ++c) rot_apply(W[p][c], W[q][c], jl);
                    for (int r = 0; r < 3; ++r) rot_apply(U.m[r][p], U.m[r][q], jl);
                    rot2 jrt = rot_t(jr);
                    for (int r = 0; r < 3; ++r) rot_apply(W[r][p], W[r][q], jrt);
                    for (int r = 0; r < 3; ++r) rot_apply(V.m[r][p], V.m[r][q], jrt);
                    maxDiag = fmaxf(maxDiag, fmaxf(fabsf(W[p][p]), fabsf(W[q][q])));
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        float a = W[i][i];
        sv[i] = fabsf(a);
        if (a < 0.f) for (int r = 0; r < 3; ++r) U.m[r][i] = -U.m[r][i];
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    for (int i = 0; i < 3; ++i) {  // descending sort with column swaps
        int pos = 0;
        float mx = sv[i];
        for (int j = 1; j < 3 - i; ++j) if (sv[i + j] > mx) { mx = sv[i + j]; pos = j; }
        if (mx == 0.f) break;
        if (pos) {
            pos += i;
            float tv = sv[i]; sv[i] = sv[pos]; sv[pos] = tv;
            for (int r = 0; r < 3; ++r) {
                float a = U.m[r][i]; U.m[r][i] = U.m[r][pos]; U.m[r][pos] = a;
                float b = V.m[r][i]; V.m[r][i] = V.m[r][pos]; V.m[r][pos] = b;
            }
        }
    }
}

// Rotation of Eigen::umeyama(src, dst, false) for three points each
// (Eigen/src/Geometry/Umeyama.h:94-160; reached from ComputeTransformationUsingTwoVecAndOnePoint,
// code/PLADE/util.cpp:604-624 via pcl TransformationEstimationSVD, transformation_estimation_svd.hpp:118-148)
HD m3 umeyama_rot3(const f3 s[3], const f3 d[3]) {
    const float inv_n = 1.f / 3.f;
    f3 sm(((s[0].x + s[1].x) + s[2].x) * inv_n, ((s[0].y + s[1].y) + s[2].y) * inv_n, ((s[0].z + s[1].z) + s[2].z) * inv_n);
    f3 dm(((d[0].x + d[1].x) + d[2].x) * inv_n, ((d[0].y + d[1].y) + d[2].y) * inv_n, ((d[0].z + d[1].z) + d[2].z) * inv_n);
    float sd[3][3], dd[3][3];  // [point][coord]
    for (int i = 0; i < 3; ++i) {
        sd[i][0] = s[i].x - sm.x; sd[i][1] = s[i].y - sm.y; sd[i][2] = s[i].z - sm.z;
        dd[i][0] = d[i].x - dm.x; dd[i][1] = d[i].y - dm.y; dd[i][2] = d[i].z - dm.z;
    }
    m3 sigma;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sigma.m[r][c] = inv_n * ((dd[0][r] * sd[0][c] + dd[1][r] * sd[1][c]) + dd[2][r] * sd[2][c]);
    m3 U, V;
    float sv[3];
    jacobi_svd3(sigma, U, sv, V);
    float S2 = 1.f;
    if (det3_e(U) * det3_e(V) < 0.f) S2 = -1.f;
    m3 R;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float a0 = (U.m[r][0] * 1.f) * V.m[c][0];
            float a1 = (U.m[r][1] * 1.f) * V.m[c][1];
            float a2 = (U.m[r][2] * S2) * V.m[c][2];
            R.m[r][c] = a0 + (a1 + a2);
        }
    return R;
}

// pcl::getEulerAngles (pcl-1.8.1/common/include/pcl/common/impl/eigen.hpp:664-669): the double C
// functions applied to float arguments, narrowed to float on store.
HD void euler_zyx(const m3 &R, float &roll, float &pitch, float &yaw) {
    roll = (float)atan2((double)R.m[2][1], (double)R.m[2][2]);
    pitch = (float)asin((double)-R.m[2][0]);
    yaw = (float)atan2((double)R.m[1][0], (double)R.m[0][0]);
}

// ComputeNearstTwoPointsOfTwo3DLine (code/PLADE/util.cpp:1167-1229) -- closest points of two lines
// whose direction vectors are ALREADY normalised by the caller.  Exact closed form in fp64 (the
// reference solves a 9x9 fp32 SVD system; deviation documented in DESIGN.md).  Returns false when
// the two direction vectors are bitwise equal (reference returns -1 => length -1).
HD bool closest_points(f3 u1, f3 p1, f3 u2, f3 p2, f3 &q1, f3 &q2, double &len) {
    if (u1.x == u2.x && u1.y == u2.y && u1.z == u2.z) return false;
    double ax = u1.x, ay = u1.y, az = u1.z, bx = u2.x, by = u2.y, bz = u2.z;
    double wx = (double)p1.x - p2.x, wy = (double)p1.y - p2.y, wz = (double)p1.z - p2.z;
    double a = ax * ax + ay * ay + az * az;
    double b = ax * bx + ay * by + az * bz;
    double c = bx * bx + by * by + bz * bz;
    double d = ax * wx + ay * wy + az * wz;
    double e = bx * wx + by * wy + bz * wz;
    double den = a * c - b * b;
    double t1 = (b * e - c * d) / den;
    double t2 = (a * e - b * d) / den;
    q1 = f3((float)(p1.x + t1 * ax), (float)(p1.y + t1 * ay), (float)(p1.z + t1 * az));
    q2 = f3((float)(p2.x + t2 * bx), (float)(p2.y + t2 * by), (float)(p2.z + t2 * bz));
    len = norm_e(q1 - q2);
    return true;
}

// ComputeIntersectionPointOf23DLine (code/PLADE/util.cpp:1461-1500): least-squares point of two
// lines = midpoint of their common perpendicular (closed form, fp64; the reference uses a 6x5 fp32
// SVD solve).  false when |v1.v2| > 0.9999.
HD bool lines_meet(f3 v1, f3 p1, f3 v2, f3 p2, f3 &out) {
    if (fabsf(dot_e(v1, v2)) > 0.9999) return false;
    double ax = v1.x, ay = v1.y, az = v1.z, bx = v2.x, by = v2.y, bz = v2.z;
    double wx = (double)p1.x - p2.x, wy = (double)p1.y - p2.y, wz = (double)p1.z - p2.z;
    double a = ax * ax + ay * ay + az * az, b = ax * bx + ay * by + az * bz, c = bx * bx + by * by + bz * bz;
    double d = ax * wx + ay * wy + az * wz, e = bx * wx + by * wy + bz * wz;
    double den = a * c - b * b;
    double t1 = (b * e - c * d) / den, t2 = (a * e - b * d) / den;
    double x1 = p1.x + t1 * ax, y1 = p1.y + t1 * ay, z1 = p1.z + t1 * az;
    double x2 = p2.x + t2 * bx, y2 = p2.y + t2 * by, z2 = p2.z + t2 * bz;
    out = f3((float)(0.5 * (x1 + x2)), (float)(0.5 * (y1 + y2)), (float)(0.5 * (z1 + z2)));
    return true;
}

// ComputeIntersectionLineOfTwoPlanes (code/PLADE/util.cpp:626-676): direction = normalised cross of
// the normalised normals; point from a 2x2 fp64 solve (cv::Mat::inv 2x2 path, opencv lapack.cpp:1036-1073).
HD bool plane_plane_line(const float *pl1, const float *pl2, f3 &vec, f3 &pt) {
    f3 p1 = normalized_e(f3(pl1[0], pl1[1], pl1[2]));
    f3 p2 = normalized_e(f3(pl2[0], pl2[1], pl2[2]));
    if (fabsf(dot_e(p1, p2)) > 0.95) return false;
    vec = normalized_e(cross(p1, p2));
    const double b0 = -pl1[3], b1 = -pl2[3];
    float a00, a01, a10, a11;
    int which;
    if (fabsf(pl1[0] * pl2[1] - pl2[0] * pl1[1]) > 1e-6) { a00 = pl1[0]; a01 = pl1[1]; a10 = pl2[0]; a11 = pl2[1]; which = 0; }
    else if (fabsf(pl1[0] * pl2[2] - pl2[0] * pl1[2]) > 1e-6) { a00 = pl1[0]; a01 = pl1[2]; a10 = pl2[0]; a11 = pl2[2]; which = 1; }
    else if (fabsf(pl1[1] * pl2[2] - pl2[1] * pl1[2]) > 1e-6) { a00 = pl1[1]; a01 = pl1[2]; a10 = pl2[1]; a11 = pl2[2]; which = 2; }
    else return false;
    double A00 = a00, A01 = a01, A10 = a10, A11 = a11;
    double det = A00 * A11 - A01 * A10;
    double r0 = 0, r1 = 0;
    if (det != 0.) {
        det = 1. / det;
        double i00 = A11 * det, i11 = A00 * det, i01 = -A01 * det, i10 = -A10 * det;
        r0 = i00 * b0 + i01 * b1;
        r1 = i10 * b0 + i11 * b1;
    }
    if (which == 0) pt = f3((float)r0, (float)r1, 0.f);
    else if (which == 1) pt = f3((float)r0, 0.f, (float)r1);
    else pt = f3(0.f, (float)r0, (float)r1);
    return true;
}

// ComputeDescriptorVectorForPairLines, method22 (code/PLADE/util.cpp:533-577).  d[0] is the caller's.
HD void descriptor22(f3 l1vec, f3 l2vec, f3 l1sp1, f3 l1sp2, f3 l2sp1, f3 l2sp2, float *d, f3 &newLine1, f3 &newLine2) {
    float angle1 = fabsf(dot_e(l1vec, l2sp1)), angle2 = fabsf(dot_e(l1vec, l2sp2));
    f3 n2a, n2b, n1a, n1b;
    if (angle1 <= angle2) { n2a = l2sp1; n2b = l2sp2; } else { n2a = l2sp2; n2b = l2sp1; }
    newLine2 = cross(n2a, n2b);
    angle1 = fabsf(dot_e(l2vec, l1sp1));
    angle2 = fabsf(dot_e(l2vec, l1sp2));
    if (angle1 <= angle2) { n1a = l1sp1; n1b = l1sp2; } else { n1a = l1sp2; n1b = l1sp1; }
    newLine1 = cross(n1a, n1b);
    d[1] = dot_e(newLine1, newLine2);
    d[2] = dot_e(n1a, n1b);
    d[3] = dot_e(n2a, n2b);
    d[4] = dot_e(newLine1, n2a);
    d[5] = dot_e(newLine1, n2b);
    d[6] = dot_e(newLine2, n1a);
    d[7] = dot_e(newLine2, n1b);
}

// ProjectPoints2Plane (code/PLADE/util.h:292-340), finite branch
HD f3 project_to_plane(f3 p, const float *pl) {
    float A = pl[0], B = pl[1], C = pl[2], D = pl[3];
    float k = -(A * p.x + B * p.y + C * p.z + D) / (A * A + B * B + C * C);
    return f3(p.x + k * A, p.y + k * B, p.z + k * C);
}

}  // namespace plade
