/* include/plade_hip.h -- C ABI of libplade_hip.so, the MI355X (gfx950) implementation of
 * PLADE's registration hot path.  Plain pointers and sizes only; every entry point
 * returns 0 on success and a negative PLADE_E* code on failure (never throws).
 *
 * The reference (chsl/PLADE) has no FFI layer; these symbols are cut at the internal
 * seams listed in SURVEY.md section 8b.  Each declaration cites the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Host pointers unless a parameter is documented as a plade_cloud handle.
 * One plade_ctx per host thread / device stream; a ctx is not re-entrant.
 */
#ifndef PLADE_HIP_H
#define PLADE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PLADE_OK 0
#define PLADE_EINVAL (-1)   /* bad argument */
#define PLADE_EDEVICE (-2)  /* HIP runtime error / no gfx950 device */
#define PLADE_ECAP (-3)     /* caller-provided capacity too small */
#define PLADE_EFAIL (-4)    /* registration failed (reference returns false) */
#define PLADE_ELIMIT (-5)   /* internal limit exceeded */

#define PLADE_GROUP_MAX 8   /* pairs per group of plade_registration_pairs */

typedef struct plade_ctx plade_ctx;
typedef struct plade_cloud plade_cloud; /* device-resident oriented point cloud */
typedef struct plade_comm plade_comm;   /* RCCL communicator of this rank (multi-GPU), see below */

/* Context: owns the HIP stream, scratch pools and the debug dump. */
int plade_ctx_create(int device, plade_ctx **out);
void plade_ctx_destroy(plade_ctx *ctx);
const char *plade_last_error(const plade_ctx *ctx);
const char *plade_version(void);
/* hipDeviceSynchronize on `device`: everything this process has queued there has finished (benchmark brackets). */
int plade_device_synchronize(int device);
/* GPUs this process sees (hipGetDeviceCount; 0 without a GPU or a HIP runtime): the CLI's batch mode spreads the pairs of the
 * list (the loop code/PLADE/main.cpp:122-148) over all of them by default. */
int plade_device_count(void);

/* Tunables that are hard-coded literals in the reference (defaults = reference values):
 *   max_planes      40     code/PLADE/plade.cpp:604   (extract(): top-40 cap)
 *   min_planes      10     code/PLADE/plade.cpp:603
 *   max_candidates  200    code/PLADE/plade.cpp:54    (maxCandidateResultNum)
 *   init_min_support 10000 code/PLADE/plade.cpp:602
 *   orient_normals  0      0 = reference behaviour (plane_extraction.cpp:43-58: correct_normal divides by a
 *                          counter that stays 0, so its average normal is NaN and the test never flips:
 *                          the plane normal keeps the sign of the LS-fit eigenvector); 1 = the evident
 *                          intent: flip (n, d) so that n agrees with the mean normal of the plane's inliers.
 *                          The C ABI never looks at the environment for anything that selects a result or an algorithm
 *                          (only PLADE_TRACE_* / PLADE_DEBUG_* printing and A/B timing hooks); the C++
 *                          API / CLI (plade_host.cpp) turn it on with env PLADE_ORIENT_NORMALS=1.
 *   unoriented_normals 0   1 = the "unoriented normals" mode README.md:109-110 describes: every plane takes part
 *                          with both orientations (2x planes, ~4x line pairs / descriptors), so a pair registers
 *                          whatever the signs of the extracted plane normals are.  C++ API / CLI: env PLADE_UNORIENTED_NORMALS=1.
 *   ransac_seed     fixed  the reference seeds from time(NULL) (RansacShapeDetector.cpp:463-464)
 *   dump            0      bit 0: keep named intermediates for plade_dump_get (tests); bit 1 (2): time the scan kernels
 *                          (HIP events on the launch stream + the kernel's own clock; the extraction loop is then
 *                          launched kernel by kernel); bits 1 + 2 (6): the kernel's own clock only, inside the captured
 *                          graph of the iteration as the unprofiled path launches it (bench.py's roofline leg)
 *   ransac_topup    1      schedule of the plane extraction's hypothesis rounds.  1 = every iteration draws a round and
 *                          what the previous batch left of the candidate pool competes with the new draws (the
 *                          reference's loop generates candidates in every pass, RansacShapeDetector.cpp:548-617);
 *                          0 = a round is drawn only when the pool is empty: strict best-first acceptance, two more
 *                          launches per iteration and more iterations.  Same acceptance semantics; which of two touching
 *                          faces takes contested points can differ (tests/test_gpu_golden.py).
 *   match_window    0      enumeration of the descriptor match (seam S2): 0 = brute force up to 2e10 descriptor pairs,
 *                          length-windowed above; 1 = always windowed; -1 = never.  The lists are identical either way.
 *   match_cell_budget 0    (query, chunk) cells per slab of the windowed enumeration; 0 = 2^26.  Test hook.
 *   prepare_sides   0      what the two clouds of a pair go through between plane extraction and descriptor match (downsampling,
 *                          boxes, grids, line pairs): 1 = side by side (source on an auxiliary stream + host thread: the shortest
 *                          single registration), 2 = one after the other on the context's stream (fewer streams and threads: the
 *                          higher batch throughput), 0 = 2 inside a group of several pairs or with host_wait != 0, else 1.
 *                          Results do not depend on it.
 *   group_max_points 48e6  points of all clouds that go through ONE extraction sequence of plade_registration_pairs*: a group
 *                          that holds more is registered in consecutive parts within this budget (the extraction's work area
 *                          takes ~0.9 KB of HBM per point it serves at once); results do not depend on it.  0 = default. */
typedef struct plade_params {
    int32_t max_planes;
    int32_t min_planes;
    int32_t max_candidates;
    int32_t init_min_support;
    int32_t orient_normals;
    int32_t dump;
    uint64_t ransac_seed;
    int32_t host_wait;   /* how the calling thread waits for the GPU: 0 = spin (lowest latency; the HIP runtime keeps
                          * one CPU busy per waiting thread), 1 = poll + 50-100 us sleeps (throughput mode: many
                          * contexts in flight per CPU), 2 = as 1 and the next plane-extraction iteration is
                          * queued before the current one has reported (a few contexts per GPU whose stream would
                          * otherwise idle while the host sleeps; with >= 8 in flight the empty speculative
                          * iteration costs more than it hides).  C++ API / CLI: env PLADE_HOST_WAIT=spin|sleep. */
    int32_t unoriented_normals;
    int32_t ransac_topup;
    int32_t match_window;
    uint32_t match_cell_budget;
    uint32_t group_max_points;
    int32_t prepare_sides;
    int32_t closest_point_mode;   /* arithmetic of ComputeNearstTwoPointsOfTwo3DLine (code/PLADE/util.cpp:1167-1229) and
                          * ComputeIntersectionPointOf23DLine (util.cpp:1461-1500): 1 = "svd_fp32" (DEFAULT = the reference's own
                          * arithmetic): cv::solve(A, B, X, DECOMP_SVD) on the 9 x 9 / 6 x 5 float systems, rounding for rounding
                          * (opencv/modules/core/src/lapack.cpp:533-710, 751-812, 1335-1460; one system per lane,
                          * plade_amd/csrc/k_svd.h); 0 = exact closed form in fp64: the better-conditioned evaluation of the
                          * same inputs, an OPT-IN deviation (like orient_normals = 1) -- on axis-aligned scenes the
                          * reference's solves are ill-conditioned and the two modes part.
                          * C++ API / CLI: env PLADE_CLOSEST_POINT_MODE=closed_form. */
} plade_params;
void plade_default_params(plade_params *p);
int plade_set_params(plade_ctx *ctx, const plade_params *p);

/* ---- seam S1a: plane scoring ------------------------------------------------------------
 * Replaces m_shape->Visit(&scoreVisitor) (code/3rd_party/ransac/Candidate.h:174,290) =
 * ScorePrimitiveShapeVisitorImpl::operator() (ransac/ScorePrimitiveShapeVisitor.h:39-46) with
 * FlatNormalThreshPointCompatibilityFunc (ransac/FlatNormalThreshPointCompatibilityFunc.h:14-23):
 * inlier <=> shape_index[i] == -1 && |dist - n.p_i| < eps && |n.n_i| >= cos_thresh.
 * pos_nrm: N x 6 (x y z nx ny nz); shape_index may be NULL (all unassigned);
 * planes: H x 4 = (n, dist = n.p0); counts: H; idx_out (optional): H x cap, ascending point
 * index per hypothesis (rows are truncated at cap; counts stay exact). */
int plade_score_planes(plade_ctx *ctx, const float *pos_nrm, const int32_t *shape_index, uint32_t n,
                       const float *planes, uint32_t h, float eps, float cos_thresh,
                       uint32_t *counts, uint32_t *idx_out, uint32_t cap);
/* The same visitor over a SUBSET of the cloud -- the call shape of candidate generation and bound refinement:
 * Candidate::ImproveBounds(..., maxSubset = 1) scores H new hypotheses on subset 0 only
 * (ransac/RansacShapeDetector.cpp:163 -> Candidate.h:154-179, one nested random subset per call); the GPU loop
 * scores a sampling round on a stratified subset in one launch.  sub_index: m point indices (< n, any order,
 * repeats allowed); counts[j] = #{ s < m : shape_index[sub_index[s]] == -1 and point sub_index[s] is compatible
 * with hypothesis j }; *n_unassigned (optional) = #{ s : shape_index[sub_index[s]] == -1 } (the loop's estimate of
 * the support on the whole cloud scales the counts by remaining / this). */
int plade_score_planes_subset(plade_ctx *ctx, const float *pos_nrm, const int32_t *shape_index, uint32_t n,
                              const uint32_t *sub_index, uint32_t m, const float *planes, uint32_t h, float eps,
                              float cos_thresh, uint32_t *counts, uint32_t *n_unassigned);

/* ---- seam S1c: connected component + LS refit of one plane candidate ---------------------
 * Replaces PlanePrimitiveShape/BitmapPrimitiveShape::ConnectedComponent
 * (ransac/BitmapPrimitiveShape.cpp:97-265, PlanePrimitiveShape.cpp:164-207), followed by
 * PlanePrimitiveShape::LSFit (PlanePrimitiveShape.cpp:98-111 -> Plane::LeastSquaresFit, Plane.cpp:169-176)
 * and Candidate::WeightedScore (Candidate.cpp:77-87) on the kept points -- the per-slot body of the
 * acceptance loop RansacShapeDetector.cpp:618-656.
 * plane = Plane(point, normal); idx: m distinct point indices (the score list, any order); kept_out (cap m):
 * the indices of the largest 8-connected bitmap component in list order, n_kept their number;
 * fit_out[7] = LS plane of the kept points (unit normal, mean, dist = mean.normal);
 * wscore_out = sum over kept points of exp(-d^2 / (2/9 w_eps^2)) against the INPUT plane. */
int plade_plane_component(plade_ctx *ctx, const float *pos_nrm, uint32_t n, const float normal[3],
                          const float point[3], const int32_t *idx, uint32_t m, float bitmap_eps,
                          int closing_filter, float w_eps, int32_t *kept_out, uint32_t *n_kept,
                          float *fit_out, double *wscore_out);

/* ---- seam S1b: whole plane-extraction stage --------------------------------------------
 * Replaces PlaneExtraction::detect (code/PLADE/plane_extraction.cpp:173-200 -> :61-168 ->
 * RansacShapeDetector::Detect, ransac/RansacShapeDetector.cpp:455-907).
 * dist_rel/bitmap_rel are relative to max(dx,dy) of the bbox (the reference's Z-ignoring scale,
 * plane_extraction.cpp:71-80).  Output: planes_out P x 4 = (unit n, d = -n.p); offsets_out P+1;
 * idx_out = original point indices of each plane's support (capacity n). */
int plade_extract_planes(plade_ctx *ctx, const float *pos_nrm, uint32_t n, uint32_t min_support,
                         float dist_rel, float bitmap_rel, float cos_thresh, float overlook_p,
                         float *planes_out, int32_t *offsets_out, int32_t *idx_out,
                         uint32_t max_planes, uint32_t *n_planes_out);

/* ---- seam S2: descriptor match ------------------------------------------------------------
 * Replaces the kdtree22.find_neighbors loop (code/PLADE/util.cpp:133-293, call at :163) =
 * KdTreeSearchNDim<VectorXf,8>::find_neighbors(p, 0, radius, ...) (ann_1.1.2/include/ANN/ANN.h:
 * 978-1029): all target descriptors with sum_d (double(q_d)-double(t_d))^2 <= double(float(r*r)).
 * Output sorted by (query, dist2, target index); offsets: Dq+1; t_idx/dist2 capacity `cap`
 * (may be NULL with cap 0 to size the result); *n_pairs receives the exact total. */
int plade_match_descriptors(plade_ctx *ctx, const float *src, uint32_t ds, const float *tgt,
                            uint32_t dt, float radius, int64_t *offsets, uint32_t *t_idx,
                            double *dist2, uint64_t cap, uint64_t *n_pairs);

/* ---- seam S3: candidate verification ------------------------------------------------------
 * Replaces the loop code/PLADE/plade.cpp:547-564: per candidate k,
 *   count_k = #{ p in src_ds : exists t in tgt_ds with |t - c_k|^2 < float(R^2) and
 *                              |t - T_k p|^2 < float(leaf^2) }
 * (ComputeOverlap, code/PLADE/util.h:611-647; pcl transformPointCloud, FLANN strict <).
 * T: K x 16 row-major; centers: K x 3 (= R c_s + T, formed by the caller as plade.cpp:555);
 * counts[k] = -1 when the coarse sphere holds no target point (overlap ratio 0). */
int plade_overlap_counts(plade_ctx *ctx, const float *src_ds, uint32_t n_s, const float *tgt_ds,
                         uint32_t n_t, const float *T, uint32_t k, const float *centers,
                         float src_radius, float inlier_dist, int32_t *counts);

/* ---- seam of the clustering stage (A9) ------------------------------------------------------
 * Replaces ClusterTransformation (code/PLADE/util.cpp:1245-1277) = pcl::ConditionalEuclideanClustering::segment
 * (pcl-1.8.1/segmentation/include/pcl/segmentation/impl/conditional_euclidean_clustering.hpp:42-138) with the condition
 * EnforceSimilarity (util.cpp:1232-1243): candidates a, b are joined when |t_a - t_b|^2 < float(dist_threshold^2) and
 * |euler_a - euler_b|^2 < angle_gate; clusters = connected components.  t_xyz, euler: m x 3 (translation; roll, pitch,
 * yaw as pcl::getEulerAngles gives them); cluster_of[i] = index of i's cluster, clusters numbered by their smallest
 * member (the order PCL creates them in). */
int plade_cluster_transforms(plade_ctx *ctx, const float *t_xyz, const float *euler, uint32_t m, float dist_threshold,
                             float angle_gate, int32_t *cluster_of, uint32_t *n_clusters);

/* ---- seams of the line geometry (A6 / A11) ------------------------------------------------------
 * plade_closest_points replaces ComputeNearstTwoPointsOfTwo3DLine (code/PLADE/util.cpp:1167-1229) for n line pairs:
 * u1, u2 (n x 3, directions: normalised first, as the reference does in place), p1, p2 (n x 3, a point of each line) ->
 * q1, q2 (n x 3, the closest points), len (n doubles: (q1 - q2).norm() in float, widened), ok (n: 0 where the two
 * normalised directions are bitwise equal -- the reference returns -1 there -- else 1).
 * plade_lines_meet replaces ComputeIntersectionPointOf23DLine (util.cpp:1461-1500): v1, v2 are used as given;
 * ok = 0 where |v1 . v2| > 0.9999.
 * mode: 0 = closed form (fp64), 1 = the reference's cv::solve(DECOMP_SVD) in float (plade_params.closest_point_mode). */
int plade_closest_points(plade_ctx *ctx, int32_t mode, const float *u1, const float *p1, const float *u2, const float *p2,
                         uint32_t n, float *q1, float *q2, double *len, int32_t *ok);
int plade_lines_meet(plade_ctx *ctx, int32_t mode, const float *v1, const float *p1, const float *v2, const float *p2,
                     uint32_t n, float *out, int32_t *ok);

/* ---- supporting stage entry points (A13) ------------------------------------------------- */
/* average_spacing(cloud, k) (code/PLADE/util.cpp:1619-1648); xyz read with `stride` floats. */
int plade_average_spacing(plade_ctx *ctx, const float *xyz, uint32_t n, uint32_t stride,
                          uint32_t k, uint32_t samples, float *spacing_out);
/* DownSamplePointCloud -> pcl::VoxelGrid (code/PLADE/util.h:161-184); out capacity n. */
int plade_voxel_downsample(plade_ctx *ctx, const float *xyz, uint32_t n, uint32_t stride, float leaf,
                           float *out_xyz, uint32_t *n_out);

/* ---- registration() overloads (code/PLADE/plade.h) ---------------------------------------- */
/* plade.h:74-79  registration(T, target, source, target_planes, source_planes) -- the
 * deterministic parity boundary.  planes: P x 4 (n, d), offsets P+1, idx. T16: row-major 4x4. */
int plade_registration_planes(plade_ctx *ctx, const float *tgt_pos_nrm, uint32_t n_t,
                              const float *src_pos_nrm, uint32_t n_s, const float *tgt_planes,
                              const int32_t *tgt_offsets, const int32_t *tgt_idx, uint32_t p_t,
                              const float *src_planes, const int32_t *src_offsets,
                              const int32_t *src_idx, uint32_t p_s, float *T16);
/* plade.h:58-61  registration(T, target, source): auto-tuned plane extraction (plade.cpp:602-662) */
int plade_registration(plade_ctx *ctx, const float *tgt_pos_nrm, uint32_t n_t,
                       const float *src_pos_nrm, uint32_t n_s, float *T16);
/* The same call in BATCH mode (code/PLADE/main.cpp:97-158: a plain loop of registration() over the pairs of
 * file_pairs.txt).  Registers (tgt, src) exactly like plade_registration and, before it starts computing, queues the
 * upload of the pair the NEXT call on this ctx will be handed (next_*; NULL / 0 = none) on a stream of its own, so that
 * the PCIe transfer of pair i+1 runs under the kernels of pair i.  The next call recognises its clouds by pointer and size
 * and skips its upload; any other pair is uploaded as usual.  The caller must leave the next_* buffers untouched until
 * that call (page-lock them with plade_host_pin for a truly asynchronous copy).  Results are identical to
 * plade_registration's. */
int plade_registration_next(plade_ctx *ctx, const float *tgt_pos_nrm, uint32_t n_t,
                            const float *src_pos_nrm, uint32_t n_s, const float *next_tgt_pos_nrm,
                            uint32_t next_n_t, const float *next_src_pos_nrm, uint32_t next_n_s, float *T16);
/* Batch mode, `count` (1..PLADE_GROUP_MAX) consecutive pairs of the list per call (the loop of code/PLADE/main.cpp:122-148
 * taken up to PLADE_GROUP_MAX pairs at a time).  Every pair is registered exactly like plade_registration -- results are bit-identical to
 * registering it alone -- but the plane extraction of all clouds of the group (PlaneExtraction::detect x 2 count) is ONE launch
 * sequence: its ~150 kernels per cloud pair are short and mostly latency-bound, and carrying the clouds of up to 8 pairs per
 * launch divides the commands, host waits and much of the GPU time per registration (~0.9 GB of HBM per cloud of the group); behind the extraction the pairs proceed
 * concurrently, pairs 1.. on internal peer contexts (plade_pair_ctx).  tgt_pos_nrm / src_pos_nrm: count pointers to N x 6 arrays, n_t / n_s their point
 * counts; next_*: the clouds the NEXT call on this ctx will be handed (next_count = 0: none), prefetched as
 * plade_registration_next does.  T16: count x 16 (identity where a pair fails); status[i]: PLADE_OK, PLADE_EFAIL (the
 * reference returns false) or another PLADE_E* code for pair i.  The return value reports errors that concern the whole
 * call (bad arguments, upload, plane extraction); plade_last_error(plade_pair_ctx(ctx, i)) has pair i's message. */
int plade_registration_pairs(plade_ctx *ctx, uint32_t count, const float *const *tgt_pos_nrm, const uint32_t *n_t,
                             const float *const *src_pos_nrm, const uint32_t *n_s, uint32_t next_count,
                             const float *const *next_tgt_pos_nrm, const uint32_t *next_n_t,
                             const float *const *next_src_pos_nrm, const uint32_t *next_n_s, float *T16, int32_t *status);
/* The context that carried pair `index` of the last plade_registration_pairs* call on ctx (0: ctx itself; 1..: its peers, NULL
 * before the first call with that many pairs): stats, dump and last error of that pair are read from it with the entry points below.
 * Borrowed -- it is destroyed with ctx; do not register on it. */
plade_ctx *plade_pair_ctx(plade_ctx *ctx, uint32_t index);
/* plade.h:91-96  registration(T, target, source, min_support_target, min_support_source) */
int plade_registration_minsupport(plade_ctx *ctx, const float *tgt_pos_nrm, uint32_t n_t,
                                  const float *src_pos_nrm, uint32_t n_s, int32_t min_support_t,
                                  int32_t min_support_s, float *T16);

/* ---- second sharding axis: the candidates of ONE pair over several GPUs -------------------------------------
 * The K candidate transforms of the verification loop (code/PLADE/plade.cpp:547-564) are independent.  With a shard set, a
 * registration whose verification holds at least `min_candidates` candidates scores only candidates k with k % world == rank
 * on this context's GPU (every rank runs the same registration on its own copy of the pair, so all ranks hold the same
 * candidate list) and then calls `exchange`: values holds `count` int32 words of which this rank has filled those at indices
 * i with (i % (count / 2)) % world == rank (first half: counts, second half: sphere flags); on return ALL words must be
 * valid on every rank (an all-gather: RCCL / MPI / torch.distributed, the caller's choice -- the library itself links no
 * communication library).  exchange returns 0 on success.  world <= 1 or exchange == NULL switches the axis off. */
typedef int (*plade_exchange_fn)(void *user, int32_t *values, uint32_t count, uint32_t rank, uint32_t world);
int plade_set_candidate_shard(plade_ctx *ctx, uint32_t rank, uint32_t world, uint32_t min_candidates, plade_exchange_fn exchange,
                              void *user);
/* The shard applies to calls that register ONE pair (plade_registration, _next, _dev, _planes, _minsupport and groups of
 * one); inside a group of several pairs (plade_registration_pairs* with count > 1) it is switched off for the call: the pairs'
 * tails run on concurrent threads, and collectives entered from several threads in an order that may differ between the
 * ranks would mismatch -- a batch shards whole pairs over the ranks instead. */

/* ---- RCCL communicator (multi-GPU): one process per GPU, ranks of one node over xGMI ---------------------------
 * The two exchange steps of the path -- the 68 bytes of result per pair that batch mode gathers once per batch (the loop
 * code/PLADE/main.cpp:122-148 sharded pair i -> rank i % world) and the 8 bytes per candidate of the candidate shard above --
 * are each ONE ncclAllGather.  librccl is opened with dlopen() on first use: a single-GPU host never loads it and the library
 * keeps no link dependency on it.
 *   plade_comm_unique_id   rank 0: 128 bytes (ncclGetUniqueId) that the host hands to every rank of the job by its own means
 *                          (bench.py / plade_amd.rccl_comm: the ranks' rendezvous file; MPI_Bcast; a pipe)
 *   plade_comm_create      every rank, collectively: ncclCommInitRank on `device`
 *   plade_comm_all_gather  send: `bytes` bytes of this rank (host memory), recv: world x bytes in rank order (host memory);
 *                          staged through device memory, one ncclAllGather, blocking.  A barrier is an all-gather of one word.
 *   plade_set_candidate_shard_comm   the candidate shard with the library doing the exchange itself: the counts the
 *                          verification kernel wrote stay in device memory, one ncclAllGather on the context's stream behind that
 *                          kernel, one read-back of all ranks' counts -- no host callback.  comm == NULL switches the axis off (a
 *                          communicator of one rank still runs the exchange: that is how a one-GPU box tests it).  The communicator must outlive its use by the context and serve one call at a time.
 * Errors: PLADE_EDEVICE (no librccl, RCCL error; plade_comm_last_error(comm or NULL) has the text). */
#define PLADE_COMM_ID_BYTES 128
int plade_comm_unique_id(void *id128);
int plade_comm_create(int device, uint32_t rank, uint32_t world, const void *id128, plade_comm **out);
int plade_comm_all_gather(plade_comm *comm, const void *send, void *recv, uint64_t bytes);
void plade_comm_destroy(plade_comm *comm);
const char *plade_comm_last_error(const plade_comm *comm);
int plade_set_candidate_shard_comm(plade_ctx *ctx, plade_comm *comm, uint32_t min_candidates);

/* Optional page-locking of caller-owned cloud buffers (hipHostRegister / hipHostUnregister): the host-pointer overloads
 * above then upload by asynchronous DMA instead of through the runtime's bounce buffer.  The reference has no counterpart
 * (its clouds never leave host memory); a host that keeps its PLY staging buffers alive pins them once. */
int plade_host_pin(plade_ctx *ctx, const void *ptr, size_t bytes);
int plade_host_unpin(plade_ctx *ctx, const void *ptr);

/* Device-resident clouds: upload once, register many times (bench: inputs resident in HBM). */
int plade_cloud_upload(plade_ctx *ctx, const float *pos_nrm, uint32_t n, plade_cloud **out);
void plade_cloud_free(plade_ctx *ctx, plade_cloud *c);
int plade_registration_dev(plade_ctx *ctx, plade_cloud *tgt, plade_cloud *src, float *T16);
/* plade_registration_pairs on resident clouds */
int plade_registration_pairs_dev(plade_ctx *ctx, uint32_t count, plade_cloud *const *tgt, plade_cloud *const *src, float *T16,
                                 int32_t *status);

/* ---- instrumentation ---------------------------------------------------------------------- */
/* Named intermediates of the last registration (when params.dump != 0). Returns 0 if found;
 * the pointer stays valid until the next call on this ctx. */
int plade_dump_get(plade_ctx *ctx, const char *name, const void **ptr, int64_t *nbytes);
/* Per-stage GPU/host seconds and byte counts of the last registration:
 * names is a ';'-separated list, values has one double per name. */
int plade_stats_get(plade_ctx *ctx, const char **names, const double **values, int32_t *count);
/* ---- diagnostic: the device-wide stable radix sort every grid of the path is built with ----------------
 * (radix_sort.hip; no reference counterpart -- it stands where PCL sorts voxel indices with std::sort,
 * voxel_grid.hpp:325, and where the kd-trees are built.)  keys: n keys of key_bytes (4 or 8) each, of which the low
 * `bits` bits are significant; vals: n u32 payloads.  Outputs are host arrays of the same shapes; equal keys keep
 * their input order. */
int plade_sort_pairs(plade_ctx *ctx, const void *keys, const uint32_t *vals, uint32_t n, int key_bytes, int bits,
                     void *keys_out, uint32_t *vals_out);
/* The same sort over up to 16 independent arrays in ONE launch sequence (how the Morton order of the clouds of a group is
 * built): segment s = items [seg_off[s], seg_off[s + 1]) of the u32 keys / values, seg_off ascending with seg_off[0] = 0 and
 * seg_off[nseg] = n; every segment is sorted on its own (stable) and left in its own range. */
int plade_sort_segments(plade_ctx *ctx, const uint32_t *keys, const uint32_t *vals, const uint32_t *seg_off, uint32_t nseg, int bits,
                        uint32_t *keys_out, uint32_t *vals_out);

/* Test seam: the device -> host hand-over every readback of the library goes through (n_ranges arrays of `words` 32-bit
 * words read back through one wait; no reference counterpart).  *mismatches = words that arrived wrong (0 expected). */
int plade_selftest_readback(plade_ctx *ctx, uint32_t n_ranges, uint32_t words, uint32_t *mismatches);

/* Diagnostic: `count` launches, on this context's stream, of a kernel of `blocks` workgroups that returns at once (mbytes = 0) or
 * streams `mbytes` MB of scratch memory; returns when they have finished.  tools/exp_interference.py runs it beside the
 * registrations to measure what foreign kernel boundaries, workgroup dispatches and memory traffic cost them. */
int plade_diag_launches(plade_ctx *ctx, uint32_t count, uint32_t blocks, uint32_t mbytes);
/* PLY ingest of the CLI: replaces load_ply_cloud (code/PLADE/util.cpp:1505-1546) over PlyReader::read
 * (code/PLADE/ply_reader.cpp:46-152, collect_elements :277-386) and rply (code/3rd_party/rply/rply.c).  Reads the `vertex`
 * element's float / double properties x y z (or X Y Z) and nx ny nz of an ascii, binary_little_endian or binary_big_endian
 * file into a malloc'ed n x 6 float array (x y z nx ny nz per point; free with plade_ply_free).  A binary file in the host's
 * byte order whose vertex element is exactly `float x y z nx ny nz` is read with one bulk read.  Returns PLADE_OK, or
 * PLADE_EINVAL with a message in `err` (NUL-terminated, truncated to err_cap) wherever the reference's function returns
 * false: unreadable / malformed file, no vertex points, "the number of points does not equal to the number of normals in the
 * file" (util.cpp:1533-1536), an empty cloud.  Host code only: no context, no GPU. */
int plade_ply_read(const char *path, float **pos_nrm, uint64_t *n, char *err, size_t err_cap);
void plade_ply_free(float *pos_nrm);
/* Seam of the one host-side stage whose tie-breaking shapes the result: the order in which
 * std::sort(sortVec.begin(), sortVec.end(), myCompareGreater) (code/PLADE/util.cpp:335-345, util.h:347-365) leaves clusters
 * of the given sizes -- order[i] = index of the cluster at sorted position i.  mode 0: the library's implementation
 * (exact_sort.h: libstdc++'s introsort with a block-wise partition), 1: std::sort itself, 2 / 3: the block-wise / the
 * sequential partition with the recursion depth limited to `depth_limit` (< 0: the library's 2 lg n).  Host code only: no
 * context, no GPU. */
int plade_diag_cluster_order(const float *sizes, uint32_t n, int32_t mode, int32_t depth_limit, int32_t *order);
/* Host seam of the register form of the reference's least-squares solver (plade_amd/csrc/k_svd.h, RegSolver: cv::solve(...,
 * DECOMP_SVD) of opencv/modules/core/src/lapack.cpp:533-812, 1335-1460 with every loop unrolled over compile-time bounds): the
 * functions the kernels inline, instantiated for the host, so that the arithmetic can be compared with the oracle's
 * restatement where there is no GPU.  kind 0: n systems of ComputeNearstTwoPointsOfTwo3DLine (code/PLADE/util.cpp:1167-1229;
 * a = u1, b = p1, c = u2, d = p2, each n x 3; o1 = point1, o2 = point2); kind 1: of ComputeIntersectionPointOf23DLine
 * (util.cpp:1461-1500; a = v1, b = p1, c = v2, d = p2; o1 = the point, o2 unused).  ok[i]: 1 solved, 0 a column of the system
 * vanished (the kernels hand such a system to the general form, which completes it as lapack.cpp:650-699 does), -1 the
 * reference's guard fired (identical directions / |v1.v2| > 0.9999).  Host code only: no context, no GPU. */
int plade_diag_line_solver_host(int32_t kind, const float *a, const float *b, const float *c, const float *d, uint32_t n,
                                float *o1, float *o2, int32_t *ok);
/* Times `iters` launches of one hot kernel on resident synthetic-shaped data with HIP events on
 * the ctx stream (used by bench.py for the roofline figure): which = "score" | "overlap" | "match". */
int plade_kernel_time(plade_ctx *ctx, const char *which, int iters, double *avg_seconds,
                      double *algorithmic_bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* PLADE_HIP_H */
