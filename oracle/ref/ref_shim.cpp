// oracle/ref/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern-"C" shim over the pieces of the reference that compile in this
// container straight from /root/reference (no reference source is copied into
// this repo; the sources are compiled where they lie by oracle/ref/Makefile and
// the only outputs go to oracle/_ref/).  Everything in this file is OUR adaptor
// code: it drives
//   * Schnabel Efficient-RANSAC   code/3rd_party/ransac            (A2-A5)
//   * ANN 1.1.2 KdTreeSearchNDim  code/3rd_party/ann_1.1.2         (A7)
//   * FLANN KDTreeSingleIndex     code/3rd_party/flann             (A9/A12/A13)
//   * Eigen 3.4.0                 code/3rd_party/eigen-3.4.0       (A8, OBB)
// exactly the way PLADE's own (un-buildable here: PCL needs Boost) glue does,
// citing the glue's file:line next to each entry point.
//
// Used (a) to validate oracle/plade_oracle.cpp, (b) to generate tests/golden/*,
// (c) as the "reference" CPU baseline for plane extraction in bench.py.
// Never linked into, imported by or called from the product (plade_amd/).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <ctime>
#include <vector>
#include <list>
#include <algorithm>
#include <limits>

// ---- Schnabel RANSAC ------------------------------------------------------
#include "RansacShapeDetector.h"
#include "PlanePrimitiveShapeConstructor.h"
#include "PlanePrimitiveShape.h"
#include "ScorePrimitiveShapeVisitor.h"
#include "FlatNormalThreshPointCompatibilityFunc.h"
#include "Octree.h"
#include "Candidate.h"
typedef ::PointCloud PointCloud_Ransac;

// ---- ANN -------------------------------------------------------------------
#include <ANN/ANN.h>

// ---- FLANN -----------------------------------------------------------------
#include <flann/flann.hpp>

// ---- Eigen -----------------------------------------------------------------
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/Eigenvalues>
#include <Eigen/LU>

// Deterministic RANSAC: the reference seeds from time(NULL)
// (ransac/RansacShapeDetector.cpp:463-464).  This library is linked with
// -Bsymbolic so the reference objects inside it bind to THIS time().
static time_t g_fake_time = 0;
static bool g_use_fake_time = false;
extern "C" time_t time(time_t *t) {
    time_t v;
    if (g_use_fake_time) v = g_fake_time;
    else { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); v = ts.tv_sec; }
    if (t) *t = v;
    return v;
}

namespace {

struct FVec8 { float v[8]; float operator[](int i) const { return v[i]; } };

// mirrors code/PLADE/plane_extraction.cpp:173-200 (copy into Schnabel's cloud)
void fill_cloud(PointCloud_Ransac &pc, const float *pn, int n) {
    pc.resize(n);
    for (int i = 0; i < n; ++i) {
        pc[i] = Point(Vec3f(pn[6 * i], pn[6 * i + 1], pn[6 * i + 2]),
                      Vec3f(pn[6 * i + 3], pn[6 * i + 4], pn[6 * i + 5]));
        pc[i].index = i;
    }
}

// mirrors code/PLADE/plane_extraction.cpp:71-80 INCLUDING the Z bug
// (`minZ = std::max(z, maxZ)` so maxZ stays -FLT_MAX).
void set_bbox_like_plade(PointCloud_Ransac &pc, const float *pn, int n) {
    float minX(std::numeric_limits<float>::max()), minY(minX), minZ(minX);
    float maxX(-std::numeric_limits<float>::max()), maxY(maxX), maxZ(maxX);
    for (int i = 0; i < n; ++i) {
        float x = pn[6 * i], y = pn[6 * i + 1], z = pn[6 * i + 2];
        minX = std::min(x, minX); minY = std::min(y, minY); minZ = std::min(z, minZ);
        maxX = std::max(x, maxX); maxY = std::max(y, maxY); minZ = std::max(z, maxZ);
    }
    pc.setBBox(Vec3f(minX, minY, minZ), Vec3f(maxX, maxY, maxZ));
}

}  // namespace

extern "C" {

// scale used for eps / bitmap eps: plane_extraction.cpp:93-96 + ransac/PointCloud.h:94-98
float ref_cloud_scale(const float *pos_nrm, int n) {
    PointCloud_Ransac pc;
    set_bbox_like_plade(pc, pos_nrm, n);
    return pc.getScale();
}

// PlaneExtraction::detect (plane_extraction.cpp:61-200) on the real libransac.
// planes_out: P x 4 = (nx,ny,nz,d) with d = -n.p (plane_extraction.cpp:146-148)
// offsets_out: P+1 prefix offsets into idx_out (original point indices, in the
// order plane_extraction.cpp:115-131 reads them: reverse iteration of the tail).
// fake_time >= 0 pins time(NULL); < 0 leaves the wall clock (true reference behaviour).
int ref_ransac_detect(const float *pos_nrm, int n, unsigned min_support, float dist_rel,
                      float bitmap_rel, float normal_thresh, float overlook, long fake_time,
                      float *planes_out, int *offsets_out, int *idx_out, int max_planes,
                      int *remaining_out) {
    if (n < 3) return 0;
    g_use_fake_time = fake_time >= 0;
    g_fake_time = (time_t)fake_time;
    PointCloud_Ransac pc;
    fill_cloud(pc, pos_nrm, n);
    set_bbox_like_plade(pc, pos_nrm, n);

    RansacShapeDetector::Options opt;
    opt.m_minSupport = min_support;
    opt.m_epsilon = dist_rel * pc.getScale();
    opt.m_bitmapEpsilon = bitmap_rel * pc.getScale();
    opt.m_normalThresh = normal_thresh;
    opt.m_probability = overlook;
    RansacShapeDetector detector(opt);
    detector.Add(new PlanePrimitiveShapeConstructor());
    MiscLib::Vector<std::pair<MiscLib::RefCountPtr<PrimitiveShape>, size_t> > shapes;
    size_t remaining = detector.Detect(pc, 0, pc.size(), &shapes);
    g_use_fake_time = false;
    if (remaining_out) *remaining_out = (int)remaining;

    PointCloud_Ransac::reverse_iterator start = pc.rbegin();
    int np = 0, off = 0;
    offsets_out[0] = 0;
    for (auto it = shapes.begin(); it != shapes.end(); ++it) {
        const PrimitiveShape *prim = it->first;
        size_t num = it->second;
        PointCloud_Ransac::reverse_iterator pit = start;
        std::vector<int> vts;
        vts.reserve(num);
        for (size_t c = 0; c < num; ++c) { vts.push_back((int)pit->index); ++pit; }
        start = pit;
        if (num < min_support) continue;
        if (prim->Identifier() != 0) continue;
        if (np >= max_planes) break;
        const Plane &pl = dynamic_cast<const PlanePrimitiveShape *>(prim)->Internal();
        const Vec3f &p = pl.getPosition();
        Vec3f nn = pl.getNormal();
        nn.normalize();
        planes_out[4 * np + 0] = nn[0];
        planes_out[4 * np + 1] = nn[1];
        planes_out[4 * np + 2] = nn[2];
        planes_out[4 * np + 3] = -(nn[0] * p[0] + nn[1] * p[1] + nn[2] * p[2]);
        for (size_t c = 0; c < num; ++c) idx_out[off + c] = vts[c];
        off += (int)num;
        ++np;
        offsets_out[np] = off;
    }
    return np;
}

// ---------------------------------------------------------------------------
// G1: plane-score KATs through the real octree + ScorePrimitiveShapeVisitor
// (ransac/ScorePrimitiveShapeVisitor.h:39-46, ScoreAACubeTreeStrategy.h:41-108,
//  FlatNormalThreshPointCompatibilityFunc.h:14-23), subset-octree flavour
// (ImmediateOctreeType: Build() physically reorders the cloud).
//  in : pos_nrm N x 6, shape_index_by_orig N (-1 = unassigned), H planes given as
//       3 sample points each (9 floats) -> Plane::Init(p1,p2,p3) (ransac/Plane.cpp:29-38)
//  out: reordered cloud (N x 6) + orig index per position, plane (n, dist) per
//       hypothesis (H x 4, valid flag in ok_out), counts, and the m_indices list
//       (positions in the REORDERED cloud, in visitor push order) per hypothesis.
int ref_score_kat(const float *pos_nrm, int n, const int *shape_index_by_orig,
                  const float *tri, int h, float eps, float normal_thresh,
                  float *reordered_out, int *orig_index_out, float *planes_out, int *ok_out,
                  int *counts_out, int *lists_out /* h x n, row-major */) {
    PointCloud_Ransac pc;
    fill_cloud(pc, pos_nrm, n);
    GfxTL::AACube<GfxTL::Vector3Df> bcube;
    bcube.Bound(pc.begin(), pc.end());
    ImmediateOctreeType oct;
    oct.ContainedData(&pc);
    oct.DataRange(0, n);
    oct.MaxBucketSize() = 20;
    oct.MaxSubdivisionLevel() = 10;
    oct.Build(bcube);
    MiscLib::Vector<int> shapeIndex(n, -1);
    for (int i = 0; i < n; ++i) {
        orig_index_out[i] = (int)pc[i].index;
        shapeIndex[i] = shape_index_by_orig[pc[i].index];
        for (int k = 0; k < 3; ++k) {
            reordered_out[6 * i + k] = pc[i].pos[k];
            reordered_out[6 * i + 3 + k] = pc[i].normal[k];
        }
    }
    ScorePrimitiveShapeVisitor<FlatNormalThreshPointCompatibilityFunc, ImmediateOctreeType>
        visitor(eps, normal_thresh);
    visitor.SetShapeIndex(shapeIndex);
    visitor.SetOctree(oct);
    for (int j = 0; j < h; ++j) {
        Plane pl;
        const float *t = tri + 9 * j;
        bool ok = pl.Init(Vec3f(t[0], t[1], t[2]), Vec3f(t[3], t[4], t[5]), Vec3f(t[6], t[7], t[8]));
        ok_out[j] = ok ? 1 : 0;
        counts_out[j] = 0;
        if (!ok) continue;
        planes_out[4 * j + 0] = pl.getNormal()[0];
        planes_out[4 * j + 1] = pl.getNormal()[1];
        planes_out[4 * j + 2] = pl.getNormal()[2];
        planes_out[4 * j + 3] = pl.SignedDistToOrigin();
        PlanePrimitiveShape shape(pl);
        auto *ind = new MiscLib::RefCounted<MiscLib::Vector<size_t> >;
        visitor.SetIndices(ind);
        shape.Visit(&visitor);
        counts_out[j] = (int)ind->size();
        for (size_t k = 0; k < ind->size(); ++k) lists_out[(size_t)j * n + k] = (int)(*ind)[k];
        ind->Release();
    }
    return 0;
}

// G3 + A5: ConnectedComponent (ransac/BitmapPrimitiveShape.cpp:97-265 via
// Candidate.cpp:89-94) and LSFit (PlanePrimitiveShape.cpp:98-111 -> Plane.cpp:169-176)
// on a caller-given plane (normal n, point p) and inlier index list.
// kept_out receives the surviving indices in their post-CC order; returns count.
int ref_connected_component(const float *pos_nrm, int n, const float *normal, const float *point,
                            const int *indices, int m, float bitmap_eps, int do_filtering,
                            int *kept_out) {
    PointCloud_Ransac pc;
    fill_cloud(pc, pos_nrm, n);
    Plane pl(Vec3f(point[0], point[1], point[2]), Vec3f(normal[0], normal[1], normal[2]));
    PlanePrimitiveShape shape(pl);
    MiscLib::Vector<size_t> ind(m);
    for (int i = 0; i < m; ++i) ind[i] = indices[i];
    size_t kept = shape.ConnectedComponent(pc, bitmap_eps, &ind, do_filtering != 0);
    for (size_t i = 0; i < kept; ++i) kept_out[i] = (int)ind[i];
    return (int)kept;
}

// out7 = normal(3), position(3) (= mean), dist
int ref_ls_fit(const float *pos_nrm, int n, const int *indices, int m, float *out7) {
    PointCloud_Ransac pc;
    fill_cloud(pc, pos_nrm, n);
    MiscLib::Vector<size_t> ind(m);
    for (int i = 0; i < m; ++i) ind[i] = indices[i];
    Plane pl;
    pl.LeastSquaresFit(pc, ind.begin(), ind.end());
    for (int k = 0; k < 3; ++k) { out7[k] = pl.getNormal()[k]; out7[3 + k] = pl.getPosition()[k]; }
    out7[6] = pl.SignedDistToOrigin();
    return 0;
}

// Candidate::WeightedScore (ransac/Candidate.cpp:77-87, ScoreComputer.h:10-16)
float ref_weighted_score(const float *pos_nrm, int n, const float *normal, const float *point,
                         const int *indices, int m, float eps) {
    Plane pl(Vec3f(point[0], point[1], point[2]), Vec3f(normal[0], normal[1], normal[2]));
    float score = 0;
    for (int i = 0; i < m; ++i) {
        const float *p = pos_nrm + 6 * (size_t)indices[i];
        score += weigh(pl.Distance(Vec3f(p[0], p[1], p[2])), eps);
    }
    return score;
}

// ---------------------------------------------------------------------------
// G5: the descriptor index exactly as util.cpp builds/queries it:
// KdTreeSearchNDim<vec,8>::end() + find_neighbors(p, 0, radius, ...)
// (ann_1.1.2/include/ANN/ANN.h:914-927, 978-1029; call site util.cpp:163).
// Returns total matches; per-query offsets (Dq+1), neighbour indices and the
// (float-truncated, as the wrapper stores them) squared distances.
long ref_ann_radius_match(const float *tgt, int dt, const float *qry, int dq, float radius,
                          long *offsets_out, int *nbr_out, float *dist_out, long cap) {
    std::vector<FVec8> pts(dt);
    for (int i = 0; i < dt; ++i) memcpy(pts[i].v, tgt + 8 * (size_t)i, 32);
    KdTreeSearchNDim<FVec8, 8> tree;
    tree.begin();
    tree.add_vertex_set(&pts);
    tree.end();
    long total = 0;
    offsets_out[0] = 0;
    std::vector<int> nb;
    std::vector<float> nd;
    for (int q = 0; q < dq; ++q) {
        FVec8 p;
        memcpy(p.v, qry + 8 * (size_t)q, 32);
        tree.find_neighbors(p, 0u, radius, nb, nd);
        for (size_t k = 0; k < nb.size(); ++k) {
            if (total < cap) { nbr_out[total] = nb[k]; dist_out[total] = nd[k]; }
            ++total;
        }
        offsets_out[q + 1] = total;
    }
    return total;
}

// ---------------------------------------------------------------------------
// FLANN pieces, composed the way pcl::KdTreeFLANN drives them
// (pcl-1.8.1/kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:90-210):
// KDTreeSingleIndexParams(15), L2_Simple<float>, SearchParams(-1, 0, sorted=true).
struct RefTree {
    std::vector<float> data;
    flann::Index<flann::L2_Simple<float> > *index;
    int n;
};

void *ref_flann_build(const float *xyz, int n) {
    RefTree *t = new RefTree;
    t->data.assign(xyz, xyz + 3 * (size_t)n);
    t->n = n;
    t->index = new flann::Index<flann::L2_Simple<float> >(
        flann::Matrix<float>(t->data.data(), n, 3), flann::KDTreeSingleIndexParams(15));
    t->index->buildIndex();
    return t;
}
void ref_flann_free(void *h) {
    RefTree *t = (RefTree *)h;
    delete t->index;
    delete t;
}
// pcl::KdTreeFLANN::radiusSearch (kdtree_flann.hpp:169-210)
int ref_flann_radius(void *h, const float *q, double radius, unsigned max_nn, int *idx_out,
                     float *d_out, int cap) {
    RefTree *t = (RefTree *)h;
    if (max_nn == 0 || max_nn > (unsigned)t->n) max_nn = t->n;
    std::vector<std::vector<int> > indices(1);
    std::vector<std::vector<float> > dists(1);
    flann::SearchParams params(-1, 0.0f, true);
    params.max_neighbors = (max_nn == (unsigned)t->n) ? -1 : (int)max_nn;
    float qq[3] = {q[0], q[1], q[2]};
    int k = t->index->radiusSearch(flann::Matrix<float>(qq, 1, 3), indices, dists,
                                   static_cast<float>(radius * radius), params);
    for (int i = 0; i < k && i < cap; ++i) { idx_out[i] = indices[0][i]; if (d_out) d_out[i] = dists[0][i]; }
    return k;
}
// pcl::KdTreeFLANN::nearestKSearch (kdtree_flann.hpp:133-166)
int ref_flann_knn(void *h, const float *q, int k, int *idx_out, float *d_out) {
    RefTree *t = (RefTree *)h;
    if (k > t->n) k = t->n;
    float qq[3] = {q[0], q[1], q[2]};
    flann::Matrix<int> im(idx_out, 1, k);
    flann::Matrix<float> dm(d_out, 1, k);
    t->index->knnSearch(flann::Matrix<float>(qq, 1, 3), im, dm, k, flann::SearchParams(-1, 0.0f));
    return k;
}

// G7: ComputeOverlap<PointXYZ> (code/PLADE/util.h:611-647) driven as
// plade.cpp:553-559: query cloud = transformed source_ds (transform done by the
// caller per pcl transforms.hpp:69-71), dest tree over target_ds.
// Returns count (the integer the ratio is formed from), -1 if the coarse sphere is empty.
int ref_overlap_count(const float *query_xyz, int nq, const float *dest_xyz, int nd,
                      const float *center, float query_radius, float inlier_dist) {
    RefTree *dest = (RefTree *)ref_flann_build(dest_xyz, nd);
    std::vector<int> nb(nd);
    int k = ref_flann_radius(dest, center, query_radius, 0, nb.data(), nullptr, nd);
    if (k <= 0) { ref_flann_free(dest); return -1; }
    std::vector<float> sub(3 * (size_t)k);
    for (int i = 0; i < k; ++i) memcpy(&sub[3 * (size_t)i], dest_xyz + 3 * (size_t)nb[i], 12);
    RefTree *nt = (RefTree *)ref_flann_build(sub.data(), k);
    int count = 0, one;
    float d1;
    for (int i = 0; i < nq; ++i)
        if (ref_flann_radius(nt, query_xyz + 3 * (size_t)i, inlier_dist, 1, &one, &d1, 1) > 0) ++count;
    ref_flann_free(nt);
    ref_flann_free(dest);
    return count;
}

// G10: ClusterTransformation (code/PLADE/util.cpp:1245-1277) = pcl::ConditionalEuclideanClustering<PointXYZINormal>::segment
// (pcl-1.8.1/segmentation/include/pcl/segmentation/impl/conditional_euclidean_clustering.hpp:42-138) with the condition
// EnforceSimilarity (util.cpp:1232-1243), composed over FLANN the way its pcl::search::KdTree searcher is
// (KdTreeFLANN::radiusSearch, kdtree_flann.hpp:169-210: sorted results, float(radius * radius), max_nn = 0 -> all).
// t_xyz = the translations (PointXYZINormal::x/y/z), euler = (normal_x, normal_y, normal_z) = pcl::getEulerAngles of the
// rotations (computed by the caller: PCL itself does not compile here).  cluster_of[i] = index of i's cluster in the
// order segment() creates them (min_cluster_size 1, max = all: every cluster is kept, util.cpp:1271-1272).
int ref_cluster_transforms(const float *t_xyz, const float *euler, int m, float distance_threshold, float g_angle,
                           int *cluster_of) {
    if (m <= 0) return 0;
    RefTree *tree = (RefTree *)ref_flann_build(t_xyz, m);
    const double cluster_tolerance = distance_threshold;   // setClusterTolerance(float) -> double member (hpp / .h:170)
    std::vector<bool> processed(m, false);
    std::vector<int> nn_indices(m);
    std::vector<float> nn_distances(m);
    int n_clusters = 0;
    for (int iii = 0; iii < m; ++iii) {                                             // hpp:76
        if (processed[iii]) continue;                                               // hpp:79
        std::vector<int> current_cluster;
        int cii = 0;
        current_cluster.push_back(iii);                                             // hpp:87
        processed[iii] = true;
        while (cii < (int)current_cluster.size()) {                                 // hpp:91
            const int a = current_cluster[cii];
            const int k = ref_flann_radius(tree, t_xyz + 3 * (size_t)a, cluster_tolerance, 0, nn_indices.data(), nn_distances.data(), m);
            if (k < 1) { cii++; continue; }                                         // hpp:94-98
            for (int nii = 1; nii < k; ++nii) {                                     // hpp:101: the first neighbour is skipped
                const int b = nn_indices[nii];
                if (processed[b]) continue;                                         // hpp:104
                // EnforceSimilarity(point_a = seed, point_b = neighbour, squared_distance), util.cpp:1232-1243
                Eigen::VectorXf temp(3);
                temp[0] = euler[3 * (size_t)a] - euler[3 * (size_t)b];
                temp[1] = euler[3 * (size_t)a + 1] - euler[3 * (size_t)b + 1];
                temp[2] = euler[3 * (size_t)a + 2] - euler[3 * (size_t)b + 2];
                if (temp.squaredNorm() < g_angle) {                                 // util.cpp:1239
                    current_cluster.push_back(b);                                   // hpp:111
                    processed[b] = true;
                }
            }
            cii++;
        }
        for (size_t q = 0; q < current_cluster.size(); ++q) cluster_of[current_cluster[q]] = n_clusters;
        ++n_clusters;
    }
    ref_flann_free(tree);
    return n_clusters;
}

// G11: one of the two walks of AreTwoPlanesPenetrable (code/PLADE/util.cpp:1379-1405 / :1416-1442) along the common
// segment of two plane rectangles, with KdTree1 / KdTree2 = pcl::search::KdTree<PointXYZ> composed over FLANN as above:
// A = the cloud whose points are classified against plane B (util.cpp:1393-1402), B = the gate cloud (a step is skipped
// unless B has two points within searchRadius / 2, radiusSearch(..., max_nn = 2) < 2, util.cpp:1384-1389).  The geometry in
// front of the walks (rectangle / line intersections through cv::solve) is not part of this: OpenCV does not compile here.
int ref_pen_walk(const float *pts_a, int na, const float *pts_b, int nb, const float *plane_b4, const float *start3,
                 const float *direc3, float length, float searchRadius, float minDistance, int *positive, int *negative,
                 int *skipped) {
    RefTree *ta = na > 0 ? (RefTree *)ref_flann_build(pts_a, na) : nullptr;
    RefTree *tb = nb > 0 ? (RefTree *)ref_flann_build(pts_b, nb) : nullptr;
    const Eigen::Vector3f startPoint(start3[0], start3[1], start3[2]), direc(direc3[0], direc3[1], direc3[2]);
    Eigen::Vector3f searchPoint;
    std::vector<int> neighbor(std::max(na, 2));
    std::vector<float> neighborLength(std::max(na, 2));
    int negativeNum = 0, positiveNum = 0, count = 0;
    std::vector<bool> checkIndex1(na, true);
    for (float dist = 0; dist < length; dist += searchRadius) {                     // util.cpp:1381
        searchPoint = startPoint + dist * direc;
        const float sp[3] = {searchPoint(0), searchPoint(1), searchPoint(2)};
        int two[2];
        float twod[2];
        const int kb = tb ? ref_flann_radius(tb, sp, searchRadius / 2, 2, two, twod, 2) : 0;
        if (kb < 2) { count++; continue; }                                          // util.cpp:1384-1389
        const int ka = ta ? ref_flann_radius(ta, sp, searchRadius, 0, neighbor.data(), neighborLength.data(), na) : 0;
        for (int i = 0; i < ka; i++) {
            if (checkIndex1[neighbor[i]]) {
                checkIndex1[neighbor[i]] = false;
                const float *p = pts_a + 3 * (size_t)neighbor[i];
                float tempDistance = plane_b4[0] * p[0] + plane_b4[1] * p[1] + plane_b4[2] * p[2] + plane_b4[3];
                if (std::fabs(tempDistance) > minDistance) {
                    if (tempDistance >= 0) positiveNum++;
                    else negativeNum++;
                }
            }
        }
    }
    if (ta) ref_flann_free(ta);
    if (tb) ref_flann_free(tb);
    *positive = positiveNum; *negative = negativeNum; *skipped = count;
    return 0;
}

// ---------------------------------------------------------------------------
// Eigen pieces.
// G6: pcl::TransformationEstimationSVD -> pcl::umeyama -> Eigen::umeyama(src,dst,false)
// (transformation_estimation_svd.hpp:118-148; util.cpp:604-624). src/dst: 3 points each,
// given row-major as [p0 p1 p2] (9 floats).  Out: 4x4 row-major.
void ref_umeyama3(const float *src, const float *dst, float *T16) {
    Eigen::Matrix<float, 3, Eigen::Dynamic> s(3, 3), d(3, 3);
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) { s(k, i) = src[3 * i + k]; d(k, i) = dst[3 * i + k]; }
    Eigen::Matrix4f T = Eigen::umeyama(s, d, false);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) T16[4 * r + c] = T(r, c);
}
// ComputeBoundingBox's eigen step (util.h:196-201): SelfAdjointEigenSolver<Matrix3f>
// (iterative compute()), eigenvectors column-major in, row-major out; col(2) = col0 x col1.
void ref_selfadjoint_eig3(const float *cov9, float *evals3, float *evecs9) {
    Eigen::Matrix3f c;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) c(r, k) = cov9[3 * r + k];
    Eigen::SelfAdjointEigenSolver<Eigen::Matrix3f> es(c, Eigen::ComputeEigenvectors);
    Eigen::Matrix3f v = es.eigenvectors();
    v.col(2) = v.col(0).cross(v.col(1));
    for (int r = 0; r < 3; ++r) {
        evals3[r] = es.eigenvalues()(r);
        for (int k = 0; k < 3; ++k) evecs9[3 * r + k] = v(r, k);
    }
}
void ref_inverse4(const float *m16, float *out16) {
    Eigen::Matrix4f m;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m(r, c) = m16[4 * r + c];
    Eigen::Matrix4f inv = m.inverse();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out16[4 * r + c] = inv(r, c);
}
// Eigen fixed-size Matrix3f * Vector3f + Vector3f (e.g. plade.cpp:555, util.cpp:359,621)
void ref_affine3(const float *R9, const float *v3, const float *t3, float *out3) {
    Eigen::Matrix3f R;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R(r, c) = R9[3 * r + c];
    Eigen::Vector3f v(v3[0], v3[1], v3[2]), t(t3[0], t3[1], t3[2]);
    Eigen::Vector3f o = R * v + t;
    out3[0] = o[0]; out3[1] = o[1]; out3[2] = o[2];
}
// Eigen's operator<< for Matrix<float,4,4> (main.cpp:86): default IOFormat.
int ref_format_matrix4(const float *m16, char *buf, int cap) {
    Eigen::Matrix4f m;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) m(r, c) = m16[4 * r + c];
    std::ostringstream os;
    os << m;
    std::string s = os.str();
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // extern "C"
