/* oracle/plade_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C entry points of the CPU restatement of PLADE's registration hot path
 * (oracle/plade_oracle.cpp).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (plade_amd/,
 * libplade_hip.so) never links, imports or calls it.
 */
#ifndef PLADE_ORACLE_H
#define PLADE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* A3: one plane hypothesis (n, dist = n.p0) against N points (pos_nrm N x 6). */
int orc_score_plane(const float *pos_nrm, const int32_t *shape_index, int n, const float *plane4,
                    float eps, float cos_thresh, int32_t *idx_out, int32_t *count_out);
/* Plane::Init(p1,p2,p3) (ransac/Plane.cpp:29-38); returns 0 if degenerate. */
int orc_plane_from_points(const float *tri9, float *plane4);
/* A4: largest connected component of the inlier bitmap; returns kept count. */
int orc_connected_component(const float *pos_nrm, int n, const float *normal3, const float *point3,
                            const int32_t *indices, int m, float bitmap_eps, int do_filtering,
                            int32_t *kept_out);
/* A5: LS refit -> normal(3), mean(3), dist. */
int orc_ls_fit(const float *pos_nrm, int n, const int32_t *indices, int m, float *out7);
float orc_weighted_score(const float *pos_nrm, int n, const float *normal3, const float *point3,
                         const int32_t *indices, int m, float eps);
float orc_cloud_scale(const float *pos_nrm, int n);

/* A13 */
float orc_average_spacing(const float *xyz, int n, int stride_floats, int k, int samples);
/* sort_mode 0: std::sort (PCL-faithful, unstable); 1: stable (ascending point index).  In orc_registration the same
 * switch also selects the summation order of the bounding-box centroids / covariances (orc_bounding_box_mode): 0 = every
 * order as PCL has it, 1 = the deterministic orders of the GPU path. */
int orc_voxel_downsample(const float *xyz, int n, int stride_floats, float leaf, int sort_mode,
                         float *out_xyz, int32_t *n_out);
/* util.h:186-248: center(3), whd (width,height,depth doubles), corners (8x3, may be NULL) */
int orc_bounding_box(const float *xyz, int n, float *center3, double *whd3, float *corners24);
/* sum_mode 0: PCL's point-after-point fp32 sums (= orc_bounding_box); 1: lane t of 1024 adds points t, t + 1024, ... one
 * after the other, the lane sums are added in lane order (the GPU path's order, a re-association of the same additions). */
int orc_bounding_box_mode(const float *xyz, int n, int sum_mode, float *center3, double *whd3, float *corners24);

/* A6 */
int orc_intersection_line(const float *plane_a4, const float *plane_b4, float *vec3, float *point3);
int orc_closest_points(const float *u1, const float *p1, const float *u2, const float *p2,
                       float *q1, float *q2, double *len);
/* A7: all target descriptors within `radius` of each query; sorted (query, dist2, index). */
int64_t orc_match_descriptors(const float *qry, int dq, const float *tgt, int dt, float radius,
                              int64_t *offsets_out, int32_t *nbr_out, double *dist2_out,
                              int64_t cap);
/* A8 */
void orc_umeyama3(const float *src9, const float *dst9, float *R9);
void orc_selfadjoint_eig3(const float *cov9, float *evals3, float *evecs9);
/* A12: u32 count for one candidate (T 4x4 row-major, centre = R c_s + T). */
int orc_overlap_count(const float *src_ds, int ns, const float *tgt_ds, int nt, const float *T16,
                      const float *center3, float src_radius, float inlier_dist);

/* Whole deterministic stage: registration(T, target, source, target_planes, source_planes)
 * (code/PLADE/plade.cpp:31-580).  Planes: P x 4 (n,d); offsets P+1; point indices.
 * voxel_sort_mode as orc_voxel_downsample.  max_candidates = 200 (plade.cpp:54).
 * Returns 1 on success, 0 on "registration failed".  The handle keeps named
 * intermediate arrays for stage-level parity tests (orc_dump_get). */
typedef struct orc_reg orc_reg;
orc_reg *orc_reg_create(void);
/* ClusterTransformation (util.cpp:1245-1277): cluster index (creation order) per candidate; returns the number of clusters.
 * euler = pcl::getEulerAngles of the rotations (orc_euler_angles). */
int orc_cluster_transforms(const float *t_xyz, const float *euler, int m, float distance_threshold, float g_angle, int *cluster_of);
void orc_euler_angles(const float *R9, float *rpy3);
/* One walk of AreTwoPlanesPenetrable (util.cpp:1379-1405): points of A on either side of plane B along the segment. */
int orc_pen_walk(const float *pts_a, int na, const float *pts_b, int nb, const float *plane_b4, const float *start3,
                 const float *direc3, float length, float searchRadius, float minDistance, int *positive, int *negative,
                 int *skipped);
/* closest points of two lines / least-squares point of two lines (util.cpp:1167-1229, 1461-1500): 0 = exact closed form in fp64
 * (default; what the HIP path computes), 1 = "svd_fp32": the reference's cv::solve(DECOMP_SVD) in float, restated from OpenCV 2.4's
 * JacobiSVDImpl_ / SVBkSbImpl_ (lapack.cpp:533-710, 751-812) */
void orc_set_closest_point_mode(int mode);
int orc_intersection_point(const float *v1, const float *p1, const float *v2, const float *p2, float *out);
/* cv::solve(A (m x n, row-major), B (m), X (n), DECOMP_SVD) for CV_32F, the restated solver itself */
void orc_solve_svd_f32(const float *A, const float *B, float *X, int m, int n);
/* the hypot() of that solver (orc_math.h: the explicit formula kernel and oracle share) and this machine's libm hypot, for the test
 * that pins the one against the other */
double orc_hypot(double x, double y);
double orc_libm_hypot(double x, double y);
void orc_reg_destroy(orc_reg *);
int orc_registration(orc_reg *h, const float *tgt_pos_nrm, int nt, const float *src_pos_nrm, int ns,
                     const float *tgt_planes, const int32_t *tgt_offsets, const int32_t *tgt_idx,
                     int pt, const float *src_planes, const int32_t *src_offsets,
                     const int32_t *src_idx, int ps, int voxel_sort_mode, int max_candidates,
                     float *T16_out);
/* Stress configurations (10^4 candidates): the same run, but only every pen_stride-th candidate of the penetration
 * filter's list is evaluated (pen_flags = -1 for the others) and the run ends after that stage (returns 0) with
 * every intermediate up to pen_tested / pen_flags dumped.  pen_stride = 1 is orc_registration. */
int orc_registration_sampled(orc_reg *h, const float *tgt_pos_nrm, int nt, const float *src_pos_nrm, int ns,
                             const float *tgt_planes, const int32_t *tgt_offsets, const int32_t *tgt_idx,
                             int pt, const float *src_planes, const int32_t *src_offsets,
                             const int32_t *src_idx, int ps, int voxel_sort_mode, int max_candidates,
                             int pen_stride, float *T16_out);
/* name -> (ptr, nbytes); returns 0 if found */
int orc_dump_get(orc_reg *h, const char *name, const void **ptr, int64_t *nbytes);
/* seconds spent per stage in the last orc_registration: name list via orc_dump "timing_names" */

#ifdef __cplusplus
}
#endif
#endif
