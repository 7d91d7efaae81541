# DESIGN — PLADE registration hot path on MI355X (gfx950)

Rounds 1-2. Everything below refers to `/root/repo`; reference citations are relative to the reference
tree (`chsl/PLADE`). `SURVEY.md` §8 is the scope contract; this file says how each row is built.

## 1. The path and its boundary

The accelerated path is `registration()` of `code/PLADE/plade.h:44-96`:

```
PLY -> registration(T, file, file)                         plade.cpp:665-706   (host, plade_amd/csrc/plade_host.cpp)
        -> registration(T, target, source)                  plade.cpp:638-662
             extract(): RANSAC plane extraction x2          plade.cpp:602-635  -> S1b  plade_extract_planes
                 per-hypothesis inlier scoring               Candidate.h:174,290 -> S1a plade_score_planes
                 connected component + LS refit + weighted score of one candidate
                                                            RansacShapeDetector.cpp:618-656 -> S1c plade_plane_component
        -> registration(T, target, source, planes, planes)  plade.cpp:31-580   -> plade_registration_planes
             average_spacing, voxel grids, OBBs, lines      (K9 + host)
             line-pair descriptors                          (K4)
             descriptor radius match                        (K5)  -> S2  plade_match_descriptors
             one SE(3) per match                            (K6)
             clustering                                     (A9 kernel)
             plane-consistency count                        (K7)
             penetration filter                             (A11 kernels)
             overlap verification of <= 201 candidates      (K8)  -> S3  plade_overlap_counts
```

The reference has no FFI layer, so the **drop-in boundary is a C ABI cut at the internal seams of
SURVEY §8b**: `include/plade_hip.h` (25 `extern "C"` symbols, plain pointers and sizes, negative error
codes, one opaque `plade_ctx` per host thread/stream). Every declaration cites the reference interface
it replaces. The C++ host above it (`plade_amd/csrc/plade.h`, `plade_host.cpp`, `main.cpp`,
`ply_reader.cpp`) keeps the four `registration()` overloads and the CLI (`PLADE tgt.ply src.ply out.txt`,
`PLADE pairs.txt out.txt`) with the reference's argument order, return values, console/`std::cerr` messages,
result-file grammar (Eigen's default `operator<<` format is reproduced and tested against real Eigen,
`tests/test_oracle_golden.py::test_eigen_stream_format...`) and exit codes. `plade_amd/__init__.py` is the
same ABI through ctypes for the tests and `bench.py`. **No CPU fallback exists**: `plade_ctx_create`
returns `PLADE_EDEVICE` without a GPU and `plade_amd.Context()` raises (`tests/test_abi.py`).

Toolchains: the reference is C++ and so is the host side here (g++ / hipcc); PyTorch is only used by
`bench.py` for `torch.distributed` (RCCL) and device synchronisation.

## 2. Parity: what is pinned against what

`oracle/` is **test infrastructure** (headers say so): `oracle/plade_oracle.cpp` + `orc_math.h` restate
the reference's algorithm on the CPU, function by function, each citing file:line.  It is pinned against
the **real reference** wherever the reference compiles in this image:

| piece of the reference | builds here? | used as |
|---|---|---|
| Schnabel RANSAC (`code/3rd_party/ransac`, 26 files) | yes, plain g++ (`oracle/ref/Makefile`) | G1 score lists, G3 connected component, A5 LS fit, weighted score; the *reference* plane extraction of the CPU baseline |
| ANN 1.1.2 (`KdTreeSearchNDim`) | yes | G5 radius match |
| FLANN (header only) composed like `pcl::KdTreeFLANN` | yes | G7 overlap counts, kNN / radius semantics, average spacing |
| Eigen 3.4.0 (header only) | yes | umeyama, `SelfAdjointEigenSolver`, `Matrix3f*v+t`, `Matrix4f::inverse`, `operator<<` |
| PCL 1.8.1 and everything in `code/PLADE` that includes it | **no** (needs Boost: absent) | restated from source reading only |
| OpenCV 2.4 core (`cv::solve`, `Mat::inv`) | **no** (needs cmake-generated `cvconfig.h`, `opencv_modules.hpp`) | restated / replaced, see deviations |

`oracle/ref/ref_shim.cpp` is *our* adaptor over those libraries, compiled **in place** from
`/root/reference` into `oracle/_ref/libplade_ref.so` (git-ignored, not gpurun-ignored). Golden vectors
generated from it are committed under `tests/golden/` together with the generator
(`tools/make_golden.py`); `tests/test_oracle_golden.py` checks the oracle against them everywhere,
`tests/test_oracle_vs_ref.py` repeats it live where `_ref` exists, and
`test_g8_polyhedron_pair_reproduces_recorded_result` runs the oracle on the reference's own
`sample_data/polyhedron_*` pair: the result agrees with `polyhedron_source_groundtruth.txt` **and** with the
authors' recorded output (`sample_data/file_pairs_results.txt:3-7`) to 5e-5 per entry.

Pinned results (all asserted in tests):

* A3 plane score: ordered inlier lists bit-identical to libransac's octree visitor (24 hypotheses).
* A4 bitmap connected component: kept index lists identical to libransac (12 multi-component cases, with/without closing).
* A5 LS fit: normal within 5e-5, mean within 1e-5 of libransac (the reference accumulates in fp32 sequentially; see below).
* A7 radius match: membership and distances identical to libann incl. distances straddling `float(r*r)`;
  order = (dist, index) — ANN's order among **value-identical** distances is kd-tree dependent (only exact duplicates are affected).
* A8 umeyama rotation, OBB eigen-decomposition: **bit-identical** to Eigen 3.4.0 (the Eigen evaluation orders
  `x0+(x1+x2)` for fixed-size and sequential for dynamic-size reductions were determined experimentally and are pinned by fixtures).
* A12 overlap counts and A13 average spacing: identical to the FLANN composition (integers / bit-equal float).

**Parity unpinned** (PCL glue cannot be compiled here; restated from source reading, documented in the
oracle header): VoxelGrid ordering, `ConditionalEuclideanClustering`, plane-consistency count, penetration
filter, the driver `plade.cpp:31-580`. The end-to-end G8 check above covers them jointly.

GPU-vs-reference fixtures (`-m gpu`, through the C ABI, no oracle involved, `tests/test_gpu_golden.py`):
K1 lists = libransac's (G1), K2/K3 kept lists = libransac's `ConnectedComponent` and its LS fit / weighted
score within the tolerances above (G3, seam S1c), K5 = libann (G5), K8 = the FLANN composition (G7), and
**G8 end to end**: on the reference's own sample pair (`sample_data/polyhedron_*.ply`, committed as data in
`tests/golden/g8_polyhedron.npz` together with the planes the reference's RANSAC extracts from it) the
planes-given overload on the GPU reproduces the authors' recorded transform
(`sample_data/file_pairs_results.txt:3-7`) and the shipped ground truth to 5e-5 per entry, bit-identical to the
oracle; the full overload with the GPU's own plane extraction lands within 1e-2 (Frobenius) of it. **G2 plane
sets**: on the same clouds the GPU extraction (S1b, min_support 625 = where the reference's `extract()` ends)
finds every one of the 27 + 26 planes the reference's RANSAC found, with the same coefficients (|Δd| < 2e-3, sign
aside) and the same supports (13748, 12232, 9013 … identical point counts, Jaccard > 0.95 for all); being
exhaustive rather than probabilistic it reports 11 further small faces of ≈ 660 points the reference run missed.
**G9 real scan**: the reference's indoor scan `sample_data/room_target.ply` (94 052 points, noisy, cluttered) with
a *surrogate* source cut from it (the matching source scan is not shipped; crop 75 %, thin to 80 %, move by the
inverse of `room_source_groundtruth.txt` — `tools/make_golden.py`) and two independent draws of libransac's planes
(14–16 planes each): the planes-given registration on the GPU equals the oracle in every dumped intermediate
(1 600–2 600 matches, ≈ 180 verified candidates) and in the final transform, bit for bit, and the full overload
with the GPU's own plane extraction lands 2.3e-3 (Frobenius) from the ground truth. The plane-set parity (G2)
holds on this scan too: every plane of the four libransac draws (14–16 per cloud) is found by the GPU extraction,
Jaccard ≥ 0.97 for all but one (0.89) and 1.00 for most (`test_g2_plane_sets_on_the_real_room_scan`), and on a
20 000-point synthetic scene with 18 libransac planes down to 485 points (`g2_synth20k.npz`).

**`extract()` level** (round 2, `tests/test_gpu_faithful.py::test_extract_auto_tuning_against_libransac`,
fixture `g2_extract.npz` = the halving loop `plade.cpp:602-635` over libransac for eight pinned `time()` seeds on the
reference's four sample clouds). libransac's plane count at a given min_support depends on the seed, and so does the
level its loop ends at (625 or 1250 on two of the four clouds). It is also not consistent with itself: on the polyhedron
target its run at 1250 reports 7–8 planes in 8 of 8 seeds although its own run at 625 shows 14 planes with ≥ 1327
points — its lazily scored search stops on the overlook-probability bound first. The GPU search scores every
hypothesis of a round exactly and finds the planes above min_support, so its count lies inside libransac's range at
10000 and 5000 on all four clouds, is never below libransac's best draw minus one, and its loop ends at a level
libransac's ends at for some seed or ONE halving step earlier (14 planes ≥ 1250 instead of 27 planes ≥ 625); every
plane libransac reports is among the GPU's at the same min_support (G2 tests).

**BASELINE configs under `-m gpu`** (round 2, `tests/test_gpu_configs.py`): configs[3] — a 64-line `file_pairs.txt` of
1M-point PLYs through the CLI on one GPU, every block equal to the library result, input order kept, and a batch killed
half-way leaves a valid prefix of blocks; configs[4] — the 10M-point hall (`max_planes=100`, `max_candidates=10000`):
overlap counts, plane-consistency counts and the penetration flags of every 16th candidate against the oracle at the
planes-given boundary (`orc_registration_sampled`), plus the overlap seam on the full clouds.

GPU-vs-oracle (`-m gpu`, through the C ABI): seam kernels bit-exact (`tests/test_gpu_seams.py`), **every
intermediate of the planes-given registration bit-exact** — spacing, voxel clouds, boxes, lines, descriptors,
match lists (fp64 distances), all initial (R,T), clusters, plane counts, penetration flags, candidate list,
integer overlap counts, scores and the final 4x4 (`tests/test_gpu_registration.py`, 34 named arrays) — and
the full `registration()` equals the oracle run on the planes the GPU extracted
(`tests/test_gpu_ransac.py`), within 1e-4 Frobenius of the reference-faithful oracle mode. At the bench size
(1M-point pairs) the same 34 arrays + the final transform are bit-identical to the oracle
(`tests/test_gpu_properties.py::test_registration_full_size_every_intermediate_equals_oracle`), K1 lists equal a
vectorised numpy fp32 restatement, and the size-independent properties (partition / nesting of score lists,
voxel-grid checksum-of-checksums and idempotence, match symmetry, overlap identities, SE(3) equivariance,
determinism) hold. `tests/test_gpu_edge.py`: empty and ragged inputs, masked-out points, exact duplicates,
capacity overflow, invalid arguments, failure reporting, four contexts in flight on one GPU.

### Deliberate deviations from the reference (all behind documented switches or unavoidable)

1. **Closest points of two lines** (`util.cpp:1167-1229`) and line/line intersection (`:1461-1500`): exact closed
   form in fp64 instead of OpenCV's fp32 9x9 / 6x5 SVD solves (OpenCV unbuildable here). Survey probe:
   |Δ| ≤ 8.8e-5 at 10 m scale — this is the reference's own solver noise. Same formula in oracle and GPU.
2. **RANSAC** is time-seeded and irreproducible in the reference (even with a pinned `time()` two in-process
   runs differ). The GPU driver is a different, deterministic search with the *same acceptance semantics*
   (§4); parity is kernel-level + plane-set level + the planes-given boundary.
3. **Plane normal orientation** (`params.orient_normals`, default **0 = the reference's behaviour** since round 2):
   the reference's `correct_normal` (`plane_extraction.cpp:43-58`) divides by a counter that is never incremented, so
   it is a NaN no-op and plane normals keep the LS-fit eigenvector sign (checked against libransac:
   `tests/test_gpu_faithful.py::test_ls_fit_sign_is_libransac_s`). In a Manhattan scene that leaves only 3 independent
   sign bits and the reference registers a pair with probability 1/8 (measured: all x-planes flipped ⇒ zero correct
   matches). `orient_normals = 1` (env `PLADE_ORIENT_NORMALS=1`) applies the evident intent — agree with the mean inlier
   normal — and is what `bench.py`, `smoke()` and the synthetic-scene tests ask for explicitly (`tests/conftest.py`
   sets the env default for the suite; the faithful-mode tests create their contexts with `orient_normals=0`);
   the CPU baseline applies the same rule to libransac's planes. The library default, the faithful mode end to end
   against the oracle, and the sign behaviour are tested in `tests/test_gpu_faithful.py`.
   **Unoriented-normals mode** (`params.unoriented_normals`, env `PLADE_UNORIENTED_NORMALS=1`; `README.md:109-110` of
   the reference: "treat each plane as two planes with opposite orientations"): every TARGET plane takes part twice,
   as (n, d) and (−n, −d) with the same support, so that a source plane whose normal came out flipped still finds its
   partner: 2× target planes ⇒ ≈ 4× target descriptors. `MirroredPlanes` (`pipeline.h`), `oracle.mirror_planes`;
   bit-exact against the oracle on mirrored planes, and registers pairs whose source normals are flipped at random,
   which the default mode cannot (`tests/test_gpu_unoriented.py`).
4. **Summation orders.** PCL sorts (voxel, point) pairs with an unstable `std::sort`, so the fp32 sum order inside a
   voxel is implementation-defined. The oracle implements both (`sort_mode=0` PCL-faithful, `1` stable); the GPU sums
   in ascending input position (= mode 1). Same voxels, coordinates ≤ 1e-5 apart; final transforms agree within 1e-4
   (tested at ≤ 100k and, since round 2, at 1M points: `tests/test_gpu_properties.py`). The **OBB sums** of round 2's
   GPU stage (`k_obb.hip`, centroid and covariance of up to 150 000 points per box) follow the same pattern: PCL adds
   the points one after the other in fp32; the GPU adds them in a lane-strided order (lane t of 1024 takes points
   t, t + 1024, …, then the 1024 lane sums in lane order), the oracle's `bounding_box(…, sum_mode=1)` does exactly the
   same additions and the two are bit-identical; `sum_mode=0` is PCL's order, ≤ 1e-6 relative apart.
5. LS plane fit accumulates in fp64 with a fixed reduction tree (deterministic); the reference's fp32
   sequential sums carry ~1e-4 relative noise on `d` at 1e5 points. Tolerance-level parity only for this row.
6. `extract()`'s >40-plane cap sorts with a `>=` comparator (UB, `plade.cpp:612-615`); a stable descending sort is used.
7. The 40-plane, 10-plane, 200-candidate, 10000-support literals are `plade_params` fields (defaults = reference).
8. Inputs the reference's kd-trees / voxel grids cannot digest are refused instead of looped on: non-finite
   coordinates → `PLADE_EINVAL` at upload (NaN / zero *normals* are fine: they never pass the normal test);
   degenerate clouds (collinear, coplanar, one point 1e12 m away, unrelated clouds) return `false` in
   milliseconds (`tests/test_gpu_edge.py`). Fixed limits fail with `PLADE_ELIMIT` and a message: 1 Mi-pixel
   plane bitmaps, 1024 steps per penetration segment, 2^31 (query, chunk) cells in K5 (inside the length windows
   for large tables), 48 M grid cells.

Not a deviation but worth knowing when reading accuracy numbers: every threshold of the reference is a multiple
of the *average point spacing* (leaf 4s, length threshold 5s, cluster radius 2.5s …) while the synthetic
generator's sensor noise is a fixed 5 mm. At 1M points/cloud the error vs ground truth is 4e-5–4e-3, at 4M it is
3e-2–8e-2 and at 16M no candidate survives near the truth — the CPU oracle gives bit-identical transforms at
1M and 4M (`tools/check_1m_parity.py`), i.e. this is the reference algorithm's density sensitivity.

## 3. Data layout in HBM

* Cloud (`plade_cloud`): the PLY layout `N x 6` fp32 (AoS, 24 B/point) **and** an SoA copy
  `x|y|z|nx|ny|nz` (each plane padded to 16 B). Scan kernels read SoA with 16 B/lane loads (4 points per
  lane, 1 KiB per wave instruction); gather-style stages (voxel grid, kNN) read the AoS copy.
  1M points = 24 MB + 24 MB; the RANSAC work area adds per cloud ("slot") a Morton-ordered SoA copy (24 MB),
  `assigned` (4 MB), a 16 k-point stratified subset, the loop state (`RState`, 100 KB) and per acceptance chain
  (8 chains) two slabs: a FIXED one (header with the four refit slots, 1 Mi-pixel bitmap, labels; same layout for
  every cloud size, zeroed once — every labelling pass leaves the bitmap clean) and a VARIABLE one at offsets that
  depend on the cloud size only (`ChainLayout`: mask bytes, per-tile counts and (u,v) boxes, four N-sized index
  lists, (u,v) parameters, pixel indices, moment rows): ≈ 33 MB per chain. Measured: 2.2 GB of HBM per `plade_ctx`
  once it has registered 1M-point pairs (work areas are grow-only and reused: no growth over 600 registrations
  of mixed sizes), so the eight contexts `bench.py` keeps in flight hold ≈ 18 GB of the 288 GB.
* Planes: `P x 4` coefficients + offsets + index lists (host; the source list also stays on the device).
* Downsampled clouds: AoS `float3` + SoA, plus the target grid (`float4` points in cell order, `cell_start/end`).
* Candidates: `float4[4]` per match = rows of `[R|T]` + Euler angles (64 B, one line).

## 4. Kernels (all hand-written HIP for gfx950, wave64; `-ffp-contract=off`)

fp32 expressions are evaluated in the reference's operation order (`plade_amd/csrc/common.h`, `geom.h`), no
FMA contraction; `+ - * / sqrt` are correctly rounded on CDNA4, so integer decisions agree with x86. 75 `__global__`
kernels; the only library device code left is rocPRIM's single-launch block sort for lists of ≤ 16 Ki items
(`prims.hip`; `hipcub::DeviceScan` was replaced in round 2 by `k_scan_u32`, a single-launch decoupled look-back scan
whose look-back words carry a generation tag, so nothing is reset between uses).

| kernel (file) | reference rows | bound | algorithmic bytes / unit | mapping |
|---|---|---|---|---|
| **K1** `k_r_mark`, `k_r_rescore`, `k_r_score_sub` (`ransac.hip`); seam S1a `k_score_multi`, `k_score_mark` + `k_compact` (`k_score.hip`) | A3 | **HBM** | 28 B/point/launch and cloud (12 pos + 12 normal + 4 shapeIndex), shared by all hypotheses of the launch; + 1 mask byte per 4 points per chain for the marking form | 1024 points per workgroup, 4 points/lane via 16 B loads, plane coefficients in LDS, `__ballot`+`s_bcnt` reduction, one LDS write per wave per hypothesis; `k_r_mark` tests the tile against the ≤ 8 chains of the cloud's batch and writes 4-bit masks + per-tile counts + per-tile (u,v) boxes; `k_r_rescore` counts the ≤ 48 pool candidates; a launch covers the tiles of both clouds of the pair |
| K2 `k_r_compact_raster`, `k_r_label`, `k_r_select_cc` (`ransac.hip`) | A4 | gather / LDS latency | — | ordered compaction of the masks (offsets from the per-tile counts, no scan launch) fused with the (u,v) parametrisation and the bitmap rasterisation; closing + lock-free union-find labelling in LDS (≤ 8192 pixels, one 1024-lane workgroup per chain; smaller pixel index wins ⇒ root = raster-first pixel), component = most pixels, raster-first on ties. Compaction / selection / removal run on **looped grids** (≤ 1024 workgroups striding over the tiles) instead of (977 tiles × 16 chains) mostly empty workgroups |
| K3 `k_r_select_cc` (moments) + `k_r_fit` | A5 | HBM gather | 24 B/inlier | the component-selection pass accumulates the 12 fp64 moments, the Gaussian weighted score and the kept count (one row of 14 doubles per 1024 list positions), `k_r_fit` reduces the rows in a fixed order and runs the 3×3 Jacobi on one lane; a refit that reproduces the previous slot's plane bit for bit marks the rest of the chain converged |
| K4 `k_pair_table` (`k_lines.hip`) | A6 | trivial (L² ≤ 6e5 pairs) | — | one lane per ordered pair; the reference's in-place re-normalisation sequence is reproduced from a per-line iterate table |
| **K5** `k_match<count/fill>`, `k_rank_lists`, windowed `k_match_win` (`k_match.hip`) | A7 | fp64 ALU | 24 fp64 flop/pair | lane owns a query (8 doubles in VGPRs), target tiles broadcast from LDS, count → scan → fill (query-major, ascending target) → per-query rank by (dist², position). Above 2e10 pairs both tables are sorted by the first descriptor component (|Δ₀| ≤ r is necessary for a match) and every group of 64 neighbouring queries enumerates one contiguous window of targets with the same exact test (identical lists to brute force, `tests/test_gpu_seams.py`). **At scale** (`tools/k5_scale.py`, `profiles/k5scale_r2_*`: 100 + 60 mutually non-parallel planes, D_t = 1.53e7, D_s = 8.9e5, 1.26e6 matches): count + fill = 2 × 108 ms, i.e. 1.5e15 "brute-force-equivalent" flop/s — ten times the fp32 vector peak — because the windows skip > 99 % of the D_s·D_t pairs; an MFMA prefilter over all pairs (F_match = 3.3e14 flop ⇒ ≥ 130 ms at bf16 dense peak *before* the exact fp64 re-test and its margin logic) cannot beat that, so K5 stays exact fp64 on the vector ALUs. At BASELINE configs[2] K5 is 0.1 ms (D_t ≈ 9e3), at configs[4] 0.8 ms (D_t = 7.7e4) |
| K6 `k_transforms` (`k_cluster.hip`) | A8 | ALU | — | lane per match, fp32 Jacobi SVD in registers, bit-exact with Eigen; also reduces the translations' bbox for A9 |
| A9 `k_cell_spans`, `k_cluster_edges` | A9 | L2 gather / atomics | — | sorted-cell grid on T, nine row spans per cell precomputed; lock-free union-find (smaller index wins ⇒ root = PCL's seed). Round 2: a lane walks the first 16 entries of its span itself, the rest of the long spans (the dense cells around the true transformation) is shared out over the wavefront, 64 consecutive entries per step: 420 → 160 µs at 169k candidates. The 1.26e6-match stress scene above spends 2.2 s here: nearly all candidates fall into a few cells and the stage is O(M²) — as PCL's radius searches would be |
| K7 `k_plane_consistency` | A10 | LDS | — | lane per cluster, plane tables in LDS |
| A11 `k_pen_setup`, `k_pen_walk` (`k_penetration.hip`) | A11 | latency | — | all (candidate, i, j) triples gated in parallel into per-plane-pair slots; one wavefront walks each surviving intersection segment over in-plane cell grids. Exact reference arithmetic on the visited points; early exit once the candidate is rejected |
| **K8** `k_overlap` (`k_overlap.hip`) | A12 | HBM stream + L2 gather | K·n_s·12 B + n_t·12 B | lane = source point, candidates' T in LDS, 27-cell probe of a compact occupancy index (bitmap + rank + starts), one atomic per workgroup per candidate |
| **K9** `k_voxel_keys/runs/centroids`, `k_knn_grid` (`k_voxel.hip`) | A13 | sort-bound / L2 | 12 B/point | 64-bit key (group, k, j, i) + stable radix sort; `k_voxel_runs` = head flags + single-launch scan + run heads + group offsets + **coalesced gather of xyz into sorted order** in one kernel; `k_voxel_centroids` then sums contiguous memory sequentially per voxel (ascending input position, bit-exact with the oracle's mode 1). Exact kNN by ring search on a uniform grid, one wave per query |
| **OBB** `k_obb_units` (`k_obb.hip`, round 2) | A13 / f3 | latency | 24 B/point (two passes) | `ComputeBoundingBox` (`util.h:186-248`) of the downsampled cloud and of every per-plane cloud + the projection of the box corners onto the plane (`plade.cpp:106-117`): one 1024-lane workgroup per box, lane-strided sums (§2 deviation 4), Eigen's 3×3 self-adjoint solver and PCL's min/max / corner arithmetic operation for operation on one lane (`hostgeom.h`); a few dozen floats per box go back to the host. Replaces ≈ 1 ms of host code and a D2H of both downsampled clouds per registration |

XCDs and caches. The MI355X's eight XCDs have private L2s that are **not coherent with each other**: every kernel
boundary writes the L2s back and invalidates them, and a device-scope release/acquire inside a kernel does the same.
Consequences measured this round: (1) a pass over a 1M-point cloud is always served by the 256 MB Infinity Cache
(memory side, survives kernel boundaries), never by L2 — FETCH_SIZE equals the full 28 MB per launch; (2) fusing two
kernels of the acceptance chain with a "last workgroup done" ticket (`__threadfence()` + atomic in ≈ 2000 workgroups)
cost 40 % of the whole throughput (415 → 238 reg/s; round 1 had measured the same idea 20–100 % slower in isolation),
so cross-workgroup hand-overs are only done through generation-tagged 64-bit words updated with relaxed atomics (scan,
radix-sort look-back, voxel runs), never through fences; (3) per-launch timing atomics on ONE address from 8000
wavefronts took 160 µs — they are spread over 64 addresses. Index structures probed at random (verification grid,
penetration grids, cluster grid) are kept to ≤ 1–2 MB.

Sorting (`radix_sort.hip`): every grid of the path is built by a stable LSD radix sort of packed cell keys whose
significant bit count is known (Morton order 24–25 bits, voxel grids ≈ 30 bits, verification / penetration / cluster
grids). 1 + ⌈bits/8⌉ launches (⌈bits/9⌉ with 9-bit digits where that saves a pass) and no memset: `k_rs_histogram`
(per-workgroup partial histograms of all digit places; also clears the look-back states) and `k_rs_pass` (512 lanes ×
16 keys per tile; atomic ticket ⇒ every predecessor tile is already running; stable rank by a wave-level match of the
digit; decoupled look-back with 8 predecessor states per round; the tile is staged in LDS in digit order and written
as runs). PMC: no over-fetch. 32 passes per registration, 0.9 ms of its 8.8 ms of GPU time — latency-bound at these
sizes (a pass moves 16 MB in 22 µs alone).

Why no MFMA anywhere: every hot loop is HBM/L2-bound integer-output work (SURVEY §8d); the one dense contraction
(K5) is handled by windowing, see its row.

### GPU RANSAC driver (`ransac.hip`, rewritten in round 2: control on the device)

Morton-sort the clouds once (both clouds of a pair in ONE sort, cloud bit on top of the 24-bit key) ⇒ "octree cell at
level l" = contiguous key range, which gives Schnabel's stratified sampling with two binary searches. Per round: 4096
hypotheses sampled and verified on the device → scored on a 16 k stratified subset in one launch → ≤ 48 *distinct*
leaders (single-barrier greedy on 32-bit keys) re-scored on all unassigned points in one HBM pass → accepted while ≥
min_support.

**Device-driven loop.** Round 1 decided everything on the host behind ≈ 30 read-backs per cloud. Now the whole loop
state of a cloud (`RState`: parameters, remaining points, drawn candidates, candidate pool, batch, accepted planes,
statistics) lives in HBM and one *iteration* is a fixed sequence of 27 launches

    sample → score on the subset → leaders → re-score the pool → select batch →
    4 × { mark, compact + rasterise, label, select + moments, fit } → decide → remove points

whose kernels read what to do from that state: a cloud that has an empty batch or has finished makes its workgroups
return at once. The sequence is captured once per cloud-size pair as a hipGraph and replayed. The host reads nothing
back in between: `k_r_decide` replays the reference's `newScore > oldScore && newSize > minSupport` logic over the four
slots of every chain, does the removal bookkeeping and the `(1−|S|/n)³` update; `k_r_select` applies the stopping rule
(`CandidateFailureProbability ≤ p`) when a round yields nothing above min_support. The iteration count — and, when the
call ends, the planes — go to the host through a **host-mapped result block** (`hipHostMallocMapped | Coherent`) whose
flag word carries (iterations, a 7-bit generation of the call, done). The host polls that word; with sleeping waits it
queues the next iteration *before* the current one has reported (a stale speculative iteration finds `done` and
returns; the generation tag keeps its report from being mistaken for the next call's), with spinning waits it queues an
iteration only when needed.

**A round per iteration ("top-up").** The first device-driven version drew a new round of hypotheses only when the pool
was empty: after a round's first batch (8 planes) the rest of the pool gave one or two planes per iteration — the
survivors are mostly faces that conflict with each other — and a 1M-point pair needed 7 iterations. The reference's
loop generates new candidates in EVERY pass before it takes the best one (`RansacShapeDetector.cpp:548-617`); now every
iteration draws a round, and what the previous batch left of the pool takes the first hypothesis slots and competes
with the new draws on the subset (versions of planes accepted since score next to nothing there and drop out): 5
iterations (3–4 with batches + the confirming round), 27 instead of 29 launches each, 427 → 455 reg/s and 6.0 → 4.8 ms
for one registration alone. Registration results on 24 synthetic seeds: 24/24 correct in both schemes, same error
range. Plane sets: touching faces may exchange a few contested points (which face is accepted first), e.g. one of the
53 planes of the polyhedron fixture gets 777 instead of libransac's 771 points, the other 52 stay exact
(`PLADE_RANSAC_TOPUP=0` = the first scheme, all 53 exact).

**Batched acceptance.** The reference accepts one candidate at a time (score(3ε) → connected component → weighted
score → ≤ 3 LS refits, `RansacShapeDetector.cpp:618-656`). Here the best candidate and every further pool candidate
that is conflict-free (normals more than 2·acos(0.8)+5° apart, or near-parallel planes more than 8ε apart everywhere
in the bounding box) **with every better pool candidate** — not only with those already in the batch, so that a
candidate never overtakes a better one it competes with — are accepted together, ≤ 8 chains per cloud. A pool rule
that also struck tilted versions of the pick ("diverse pool") cut the iterations from 7 to 5 but registered 1 of 16
synthetic pairs to a symmetric alignment: removed. **Both clouds of a pair ("slots") go through the same launches**
(blockIdx selects the cloud), `extract()`'s halving loops of the two clouds are merged (`extract_pair`: a cloud that
needs another detect call takes part with its own min_support, the other keeps its result).

Seam S1c (`plade_plane_component`) runs the same chain kernels on a caller-given list (list mode of the mark kernel).

### Experiments measured and not kept (details and numbers: `profiles/r2_experiments.md`)

* **Shared per-device extractor** (one thread per device drives ONE launch sequence over up to 32 slots, the clouds of
  all registrations in flight joining and leaving iteration by iteration; bit-identical results). The summed kernel
  time per registration fell from 9.5 to 6.9 ms, but the extractor's stream is one dependent chain of 30 launches per
  iteration (1.4 ms at 3.3 pairs, 1.9 ms at 5.2 pairs — the per-point passes of a batch are bandwidth-bound, 4.4 TB/s),
  and the mean kernel concurrency fell from 3.7 to 2.5: 377–413 reg/s against 414 for eight independent per-context
  loops interleaving on the four hardware queues. 2 or 4 chains, a high-priority stream, 12–24 registrations in flight:
  no better. Reverted (commits 027067d, 04eb36b in the history).
* Fit fused into the selection kernel via a last-workgroup ticket: −40 % throughput (see XCDs above).
* Two-stage hashed leaders kernel (169 vs 65 µs), 16 chains per cloud (identical batches), OBB sums in contiguous
  chunks per lane (220 vs 31 µs per pass), `GPU_MAX_HW_QUEUES=8` (355 reg/s), two processes × 4 in flight (no gain).

## 5. Measurement (`bench.py`)

Step = full `registration(T, target, source)` of one synthetic 1M-point pair (BASELINE configs[2]; `orient_normals=1`,
§2 deviation 3), clouds resident in HBM (`plade_registration_dev`). One registration alone is latency-bound (≈ 290
dependent commands), so `bench.py` keeps **8 independent registrations in flight per GPU** (one `plade_ctx` + host
thread each, `--inflight`), which is how a batch of pairs (configs[3]) is processed. Round-2 numbers (MI355X,
`profiles/r2_bench_n1.json`, `profiles/r2_kernel_stats.csv`, `profiles/r2_pmc_hbm.csv`):

| | value |
|---|---|
| registrations/s, 1 GPU, clouds in HBM (`value`) | **@VALUE@** (`python bench.py`: 512 steps, @MS@ ms/step at 8 in flight, sleeping host waits, @BUSY@ busy host threads; 512/512 ok, all registrations of a pair bit-identical across contexts; max ‖T−T_gt‖_F = 1.5e-3 = what the CPU oracle gives on the same planes). Round 1: 406; this round before the top-up rule: 427. Short runs read lower because the timed region is bracketed by synchronisations, i.e. it contains the fill and the drain of the 8-deep pipeline (≈ 17 ms latency per registration at full load): `--steps 20` 398–431, `--steps 64` 455, `--steps 512` 467. 30 000-step soak of the final build: 470.5 reg/s, 30 000/30 000 ok and bit-identical per pair, 2.6 busy host threads, no cgroup throttling (host-buffer leg of the same run, 2000 steps: 436.7 reg/s, identical results) |
| the same with the clouds in page-locked HOST memory (`host_buffers_rank0`: `plade_registration`, H2D + SoA conversion + bounding box inside the timed region) | **@HOSTVALUE@** reg/s (384 steps, 8 in flight) = 48 MB per registration over PCIe while the other contexts compute (the 24 MB copies run on the SDMA engines at 43 GB/s, 0.56 ms each: `tools/trace_host.sh`, `tools/h2d_rate.py`); results identical to the resident ones. Each context stalls ≈ 1.3 ms per registration for its own upload, so more contexts hide more of it: 437 / 437 / 445 reg/s at 8 / 10 / 12 in flight over 600–2000 steps (resident: 470 / 453 / 450). The task's contract keeps `value` = resident; this is the PCIe-inclusive figure. `plade_host_pin` page-locks caller buffers (from pageable memory the runtime stages through its own bounce buffer and the call blocks) |
| one registration alone (spinning waits) | **@LAT@ ms** (before the top-up rule 6.0–6.6 ms, round 1: 6.9 ms) |
| commands per registration (rocprofv3, `r2_kernel_stats.csv`) | **@KERNELS@ kernels + @COPIES@ copies / fills** (round 1: 445 + 91 = 536; this round before the top-up rule: 308 + 57), @GPUMS@ ms of summed GPU time; the extraction loop reads nothing back (round 1: ≈ 30 read-backs + syncs per cloud); one helper thread at a time per registration (source spacing during the extraction, source side of the preparation afterwards; round 1: two to three) |
| CPU baseline (same box, 1 core of an EPYC 9575F, 256 logical CPUs, cgroup budget 16) | 0.51 reg/s — libransac (reference, `oracle/_ref`) 5.5 s + oracle port 6.2 s per 6 registrations |
| `roofline` kernel `k_r_mark` | @WORK@ working launches per registration (each scans the clouds of the pair that still have a batch: @ALGOL@ MB algorithmic on average = 28 B per point of those clouds + the mask bytes) + ≈ @IDLE@ launches of the fixed sequence that find nothing to do. **@MARKUS@ µs per working launch ⇒ @MARKTB@ TB/s = @MARKFRAC@ of 8 TB/s**, measured *inside the kernel* on the device wall clock (min start / max end over its wavefronts) under the load of the timed region. rocprofv3 of the same command (`r2_kernel_stats.csv`): @RPUS@ µs averaged over all @RPLAUNCH@ launches per registration — i.e. @RPTOTAL@ µs per registration = @WORK@ working launches × @MARKUS@ µs + @IDLE@ idle ones × @IDLEUS@ µs: the same durations (per launch of that mixed population: (@WORK@ × @ALGOL@ MB / @RPLAUNCH@) / @RPUS@ µs = @RPFRAC@ of 8 TB/s). HIP events around the launch read @EVUS@ µs under load: the hardware queue is shared with other streams' kernels, which run between the two events (alone on the GPU: 11.4 µs by rocprofv3 ⇒ 4.6 TB/s). PMC (`r2_pmc_hbm.csv`): @TRAFFIC@ GB fetched + written per registration by this kernel for @ALGO@ GB algorithmic — no over-fetch |
| the same kernel on large clouds | configs[4] (10M-point clouds, `profiles/config5_r2_*`): 100 cloud passes of 280 MB in 72 launches × 68.6 µs ⇒ **5.7 TB/s = 71 % of 8 TB/s** |
| why this kernel | it moves the most HBM bytes of the step (@ALGO@ of @STEPB@ GB). By GPU time under load it is 5th (5.0 %) behind `k_r_label` (7.1 %, LDS union-find, no HBM figure), `k_r_compact_raster` (6.7 %), `k_r_select_cc` (5.4 %, index-driven gathers) and a sort pass; K1 together (`k_r_mark` + `k_r_rescore` + `k_r_score_sub`) is @K1SHARE@ % |
| whole step vs SURVEY §8d `B_total` | @STEPB@ GB algorithmic per registration (bytes counted per launch and cloud actually scanned) / @MS@ ms = @STEPFRAC@ % of the 8 TB/s peak: the step is bound by its dependent commands, not by bandwidth |
| remaining over-fetch (PMC) | index-driven gathers from randomly ordered input: `k_gather_cloud` 275 + 67 MB for 48 + 48 MB, `k_voxel_runs` 134 MB for 24 MB (its writes and `k_voxel_centroids`' 12.8 MB reads are now exactly algorithmic: the sorted copy is written coalesced and summed from contiguous memory). The synthetic clouds are in random point order — the worst case; scanner order is spatially coherent |

**What limits the step.** A HIP process drives the GPU through 4 hardware queues; with 8 registrations in flight all
four are busy 99.9 % of the time with 3.7 kernels running on average (`tools/trace_load.sh`, `tools/concurrency.py`), so
throughput ≈ 3.7 / Σ(kernel durations per registration): @GPUMS@ ms under load ⇒ @MS@ ms per step. `GPU_MAX_HW_QUEUES` =
2 / 3 / 4 / 5 / 6 / 8 gives 294 / 362 / 420 / 343 / 323 / 355 reg/s (measured before the top-up rule): four concurrently
running kernels is what the part sustains for this mix; more processes or more registrations in flight do not help
either (§4 experiments). Every kernel of a 1M-point registration is a single wave of workgroups — its duration is a
chain of dependent memory accesses, not bandwidth (`k_r_mark`: 11 µs alone for 56 MB, 18 µs under load; 69 µs for
560 MB at 10M points) — and every kernel boundary is an L2 write-back + invalidate on this 8-XCD part, so what moves the
number is fewer launches and fewer passes: the top-up rule took 2 of 7 iterations out (427 → 467 reg/s), while halving
the duration of single kernels (`k_cluster_edges`, the OBB kernel) stays inside the run-to-run noise. Doubling the points
(2M per cloud) costs +1.0 ms per step. RANSAC is 47 % of the GPU time (≈ 5 iterations × 27 launches; the per-iteration
sampling trio sample / score-on-subset / leaders is 11 %), the sorts 17 % (32 passes), clustering + penetration +
verification + spacing 14 %.

Host side: every wait of the HIP runtime spins; `plade_params.host_wait = 1` polls with 15–40 µs sleeps instead, small
device-to-host readbacks go to a pinned arena as asynchronous copies, and the extraction's loop reports through
host-mapped memory: @BUSY@ busy host threads at 8 in flight (round 1: 3.7), budget 16.

CLI end to end (`tools/time_cli.py`, binary 1M-point PLY pairs = 24 MB per cloud in the page cache, one GPU): a single
`PLADE target.ply source.ply result.txt` process takes 0.32–0.39 s, ≈ 0.3 s of it HIP start-up; BASELINE configs[3]'s 64
pairs through `PLADE pairs.txt out.txt` finish in 0.87 s with the default 4 workers (74 pairs/s), 256 pairs in 1.55 s
(165 pairs/s; 280 pairs/s marginal: a worker reads 48 MB of PLY per pair, ≈ 10 ms, before it registers). Every block
equals the library result, blocks are in input order and **streamed** as pairs finish (`OrderedWriter`, flushed per
block: a batch killed half-way leaves a valid prefix, `tests/test_gpu_configs.py`). Device allocations are not the
start-up cost (207 `hipMalloc`s of a context's first registration take 3 ms; with 4 workers starting at once they
contend: 270 ms summed over the threads, `tools/cli_alloc.sh`).

`cpu_baseline.kind` is "port": plane extraction is the *reference's own* RANSAC library, the rest the oracle. It is
single-threaded because the reference is (`ransac/CMakeLists.txt:222-235`).

## 6. Multi-GPU

Scan pairs are independent (`main.cpp:97-158` is a plain loop): pair `i` → rank `i % world` (`plade_amd/batch.py`), one
process per GPU, no data-path collective; the 4x4 results are gathered to rank 0 with one `dist.all_gather` over RCCL
(gloo in `tests/test_distributed_gloo.py`, world size 2; the whole N = 2 control flow of `bench.py` runs on the one-GPU
box with two ranks sharing the GPU over gloo, `tests/test_gpu_bench_world2.py`). `bench.py` reports `"scaling": "weak"`.
Every C-ABI entry point binds the calling thread to its context's GPU first. The C++ CLI shards batch mode over
`PLADE_GPUS=N` × `PLADE_INFLIGHT=M` worker threads (one `plade_ctx` each, destroyed by the worker that made it). Within
one pair the RANSAC accept/remove loop is sequential ⇒ replicas only. The second axis of SURVEY §8e — the K candidate
transforms of one pair are independent in the verification step — is available at the seam
(`plade_amd.batch.sharded_overlap_counts`, gloo + GPU tests); the default pipeline does not use it: with K ≤ 201 the
verification kernel is 0.2 ms of a 6 ms registration. Host budget per rank: 8 worker threads + 8 helper threads, 2.6 of
them busy on average (1.3 + 0.17 per registration in flight): 8 ranks at 8 in flight need ≈ 21 busy cores. `bench.py`
therefore picks the in-flight count from the CPU quota the ranks of a node share (`inflight_for_budget`: 8 when there
are ≥ 2.7 CPUs per rank, 4 for 8 ranks on the 16-CPU containers of this pool — ≈ 415 instead of 467 reg/s per GPU, but
no throttling, which cost a third of the throughput in round 1's experiments); `--inflight N` overrides it. No 8-GPU
run exists (the driver's to launch).

## 7. Out of scope / next

Out of scope (SURVEY §2): ResultViewer, OpenCV imgproc / boundary-line branch (dead code), the unused 6-D/4-D
descriptor kinds and their kd-trees, `ComputeMeanDistanceOfLine2Plane` (results unused; only its re-normalisation side
effect is reproduced), CI.

Lessons kept: a counter updated atomically by thousands of workgroups serialises at the memory side (≈ 20 ns per
update on one address); device-scope fences cost an L2 write-back + invalidate on every XCD; HIP events do not time a
kernel when the hardware queue is shared.

Next (in order): (1) fewer, fatter commands in the acceptance chain without fences: the four refit slots re-scan the
whole cloud (4 × 56 MB per iteration) although slots 1–3 can only keep points within 3ε of a plane that moved by a
fraction of ε — per-tile (u, v, distance) boxes from slot 0 would let later slots skip ≈ 90 % of the tiles *and*
shrink the compaction / selection grids; (2) `k_r_label` (24 µs) and `k_r_fit` (12 µs) are latency chains of dependent
loads: a two-level labelling (per-row runs in registers, unions in LDS) and a fit in the last *wavefront* of the
selection kernel via relaxed tagged words instead of a ticket + fence; (3) sorts: the per-plane voxel grids could reuse
the whole-cloud voxel order with one extra pass instead of four; (4) the Morton-ordered copy as the source of the voxel
and kNN stages (kills the gather over-fetch; needs the original index as a tie-break in the voxel keys to keep the
summation order); (5) clustering at > 1e6 candidates: skip pairs already in the same component with a per-cell
representative; (6) PLY ingest is a bulk read for the native layout, ascii stays `strtod`-bound.
