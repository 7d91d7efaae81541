"""The HIP path against the committed golden vectors (tests/golden/, produced by the REAL reference
pieces -- libransac, libann, FLANN -- with tools/make_golden.py), called through the C ABI.
Bit-exact integer outputs; no oracle involved."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_g1_score_lists_from_libransac(ctx):
    """K1 vs ScorePrimitiveShapeVisitor of libransac: counts and ordered index lists."""
    g = load("g1_score.npz")
    ok = g["ok"].astype(bool)
    planes = g["planes"][ok]
    counts, lists = ctx.score_planes(g["cloud"], g["shape_index"], planes, float(g["eps"]), float(g["cos_t"]),
                                     want_indices=True)
    want_counts = g["counts"][ok]
    assert np.array_equal(counts.astype(np.int32), want_counts)
    pos = k = 0
    for j, c in enumerate(g["counts"]):
        if not ok[j]:
            continue
        assert np.array_equal(lists[k], g["lists"][pos:pos + c]), f"hypothesis {j}"
        pos += c
        k += 1


def test_g5_descriptor_match_from_libann(ctx):
    """K5 vs ANN's fixed-radius search: membership, (dist, index) order, fp64 distances."""
    g = load("g5_ann.npz")
    off, nbr, d2 = ctx.match_descriptors(g["qry"], g["tgt"], float(g["radius"]))
    assert np.array_equal(off, g["offsets"])
    assert np.array_equal(nbr, g["nbr"])
    assert np.array_equal(d2.astype(np.float32), g["dist"])
    for q in range(len(off) - 1):  # ANN's own tie order differs only inside exact ties
        assert sorted(g["nbr_ann_order"][off[q]:off[q + 1]]) == sorted(nbr[off[q]:off[q + 1]])


def test_g10_clusters_from_the_flann_composition(ctx):
    """The clustering stage's kernels (k_t_keys / k_cell_spans / k_cluster_edges / k_flatten behind the seam
    plade_cluster_transforms) vs pcl::ConditionalEuclideanClustering::segment composed over FLANN."""
    g = load("g10_cluster.npz")
    for name in str(g["names"]).split(";"):
        lab, n = ctx.cluster_transforms(g[f"{name}_t"], g[f"{name}_euler"], float(g[f"{name}_dist"]), float(g[f"{name}_angle"]))
        assert n == int(g[f"{name}_n"]) and np.array_equal(lab, g[f"{name}_cluster_of"]), name
    lab, n = ctx.cluster_transforms(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 0.1, 0.01)
    assert n == 0 and len(lab) == 0


def test_g7_overlap_counts_from_flann(ctx):
    """K8 vs FLANN radius searches composed as util.h:611-647."""
    g = load("g7_overlap.npz")
    got = ctx.overlap_counts(g["src"], g["tgt"], g["T"], g["centers"], float(g["radius"]), float(g["leaf"]))
    assert np.array_equal(got, g["counts"])


def test_g_radius_sets_via_overlap_kernel(ctx):
    """FLANN radius-search membership (g_radius.npz) seen through K8: with the identity transform and
    the query as the only source point, count = 1 iff the query has a target point within the radius."""
    g = load("g_radius.npz")
    r = float(g["radius"])
    T = np.eye(4, dtype=np.float32)[None]
    for q, size in zip(g["queries"][:20], g["sizes"][:20]):
        got = ctx.overlap_counts(q[None, :], g["cloud"], T, q[None, :], np.float32(1e6), np.float32(r))
        assert got[0] == (1 if size > 0 else 0)


def test_g3_connected_component_lsfit_wscore_from_libransac(ctx):
    """K2/K3 (seam S1c) vs libransac's ConnectedComponent, LSFit and WeightedScore."""
    g = load("g3_cc.npz")
    multi = 0
    for i in range(int(g["n"])):
        kept, fit, ws = ctx.plane_component(g[f"pts_{i}"], g[f"normal_{i}"], g[f"point_{i}"], g[f"idx_{i}"],
                                            float(g[f"beps_{i}"]), bool(g[f"filt_{i}"]), 0.15)
        assert np.array_equal(kept, g[f"kept_{i}"]), f"case {i}"       # integer output: bit-exact
        multi += len(kept) < len(g[f"idx_{i}"])
        if f"fit_{i}" in g.files:
            ref = g[f"fit_{i}"]
            sgn = np.sign(fit[:3] @ ref[:3])
            # the reference accumulates mean/covariance sequentially in fp32 (GfxTL/Mean.h:31-46), the
            # kernel in fp64 with a fixed tree: tolerance = the reference's own rounding noise
            assert np.abs(sgn * fit[:3] - ref[:3]).max() < 5e-5
            assert np.abs(fit[3:6] - ref[3:6]).max() < 1e-5
            assert abs(ws - float(g[f"wscore_{i}"])) <= 1e-4 * max(1.0, float(g[f"wscore_{i}"]))
    assert multi >= 6


def test_g8_reference_sample_pair_on_the_gpu(ctx, oracle):
    """End to end on the reference's own sample pair.  With the planes the reference's RANSAC extracted
    (planes-given overload, plade.h:74) the GPU reproduces the authors' recorded result and the shipped ground
    truth, bit-identical to the oracle; with its own plane extraction (plade.h:58) it lands on the same pose."""
    g = load("g8_polyhedron.npz")
    tp, sp = (g["t_coef"], g["t_off"], g["t_idx"]), (g["s_coef"], g["s_off"], g["s_idx"])
    ok, T = ctx.registration_planes(g["target"], g["source"], tp, sp)
    assert ok
    assert np.abs(T - g["recorded"]).max() < 5e-5      # sample_data/file_pairs_results.txt:3-7
    assert np.abs(T - g["groundtruth"]).max() < 5e-5   # sample_data/polyhedron_source_groundtruth.txt
    ok_o, T_o, _ = oracle.registration(g["target"], g["source"], tp, sp, voxel_sort_mode=1)
    assert ok_o and np.array_equal(T, T_o)
    ok2, T2 = ctx.registration(g["target"], g["source"])
    assert ok2 and np.linalg.norm(T2.astype(np.float64) - g["groundtruth"]) < 1e-2


def test_g9_real_room_scan_every_intermediate_equals_oracle(oracle):
    """A real indoor scan (the reference's sample_data/room_target.ply; surrogate source, tools/make_golden.py) with
    the planes libransac extracted (two independent draws): all dumped intermediates and the final transform equal
    the oracle's bit for bit."""
    import plade_amd
    g = load("g9_room.npz")
    ctx = plade_amd.Context(0, dump=1)
    for pre in ("", "b"):
        tp = (g[f"t{pre}_coef"], g[f"t{pre}_off"], g[f"t{pre}_idx"])
        sp = (g[f"s{pre}_coef"], g[f"s{pre}_off"], g[f"s{pre}_idx"])
        ok, T = ctx.registration_planes(g["target"], g["source"], tp, sp)
        d = ctx.dump()
        ok_o, T_o, do = oracle.registration(g["target"], g["source"], tp, sp, voxel_sort_mode=1)
        assert ok and ok_o and np.array_equal(T, T_o)
        common = [k for k in do if k in d and not k.startswith("timing")]
        assert len(common) >= 30 and "initial_RT" in common
        for k in common:
            assert np.asarray(d[k]).shape == np.asarray(do[k]).shape and np.array_equal(d[k], do[k]), (pre, k)
        assert np.linalg.norm(T - g["groundtruth"]) < 0.1
    ctx.close()


def test_g9_real_room_scan_own_plane_extraction(ctx):
    """Same pair through the full path (plade.h:58): the GPU plane extraction + registration land on the ground truth."""
    g = load("g9_room.npz")
    ok, T = ctx.registration(g["target"], g["source"])
    assert ok and np.linalg.norm(T.astype(np.float64) - g["groundtruth"]) < 1e-2
    ok2, T2 = ctx.registration(g["target"], g["source"])
    assert ok2 and np.array_equal(T, T2)


def _g2_compare(planes_by_cloud, g):
    """Per libransac plane: the best GPU plane under the ROUND-1 tolerances (|cos| > 0.9999, |delta d| < 2e-3, sign aside);
    returns (number of libransac planes, list of (cloud, plane, Jaccard, |S_gpu|, |S_ref|) of the matched planes, list of
    libransac planes without a match under those tolerances)."""
    total, matched, unmatched = 0, [], []
    for c, (rc, ro, ri) in enumerate(((g["t_coef"], g["t_off"], g["t_idx"]), (g["s_coef"], g["s_off"], g["s_idx"]))):
        coef, off, idx = planes_by_cloud[c]
        assert len(rc) <= len(coef) <= 2 * len(rc)
        sets = [set(idx[off[p]:off[p + 1]].tolist()) for p in range(len(coef))]
        for p in range(len(rc)):
            total += 1
            ref_set = set(ri[ro[p]:ro[p + 1]].tolist())
            cos = coef[:, :3] @ rc[p, :3]
            best, best_q = 0.0, -1
            for q in np.nonzero(np.abs(cos) > 0.9999)[0]:
                if abs(coef[q, 3] - rc[p, 3] * np.sign(cos[q])) >= 2e-3:
                    continue
                j = len(sets[q] & ref_set) / len(sets[q] | ref_set)
                if j > best:
                    best, best_q = j, q
            if best_q < 0:
                # which GPU plane is it, under any tolerance?
                j_any = max((len(sets[q] & ref_set) / len(sets[q] | ref_set), q) for q in range(len(coef)))
                unmatched.append((c, p, len(ref_set), j_any[0], len(sets[j_any[1]]), float(abs(cos[j_any[1]]))))
            else:
                matched.append((c, p, best, len(sets[best_q]), len(ref_set)))
    return total, matched, unmatched


def test_g2_plane_sets_against_the_reference_ransac(ctx):
    """Plane-set level parity of the GPU extraction (seam S1b, default schedule) with the reference's Schnabel RANSAC on
    the reference's sample cloud, under the tolerances this test has had since round 1 (|cos| > 0.9999, |delta d| < 2e-3,
    sign aside): EVERY plane libransac found is found with the same coefficients and the same points, with ONE named
    exception that is asserted as measured -- a 771-point face gets six points that libransac gave to a neighbouring face
    it accepted earlier (777 points, Jaccard 0.992, normal 0.8 degrees off).  Which of two touching faces takes the
    contested points depends on the order of acceptance, in libransac on its time() seed; the best-first order of the
    first schedule (plade_params.ransac_topup = 0, next test) reproduces all 53.  The GPU search is more exhaustive (the reference
    stops on a probability bound) and may report further small faces above min_support."""
    g = load("g8_polyhedron.npz")
    # extract() of plade.cpp:602-635 ends at 10000 / 16 here
    total, matched, unmatched = _g2_compare([ctx.extract_planes(g["target"], 625), ctx.extract_planes(g["source"], 625)], g)
    assert total == 53
    assert len(unmatched) <= 1, unmatched
    for (c, p, j, n_gpu, n_ref) in matched:
        assert j > 0.95, (c, p, j)
    assert sum(n_gpu == n_ref and j == 1.0 for (_, _, j, n_gpu, n_ref) in matched) >= total - 1 - len(unmatched), matched
    for (c, p, n_ref, j, n_gpu, cosv) in unmatched:   # the named exception, as measured
        assert n_ref == 771 and j >= 0.99 and abs(n_gpu - n_ref) <= 6 and cosv > 0.9998, unmatched


def test_g2_plane_sets_best_first_schedule_reproduces_every_plane():
    """The same comparison with plade_params.ransac_topup = 0 (hypotheses drawn only when the pool is empty: strict
    best-first acceptance): all 53 planes under the round-1 tolerances, all with libransac's supports exactly.  Both
    schedules run in THIS process, one context each (the switch is a parameter of the context, not of the environment):
    the default schedule must still show its one named exception next to it."""
    import plade_amd
    g = load("g8_polyhedron.npz")
    res = {}
    for mode in (0, 1):
        c = plade_amd.Context(0, orient_normals=1, ransac_topup=mode)
        planes = [c.extract_planes(cloud, 625) for cloud in (g["target"], g["source"])]
        c.close()
        res[mode] = _g2_compare(planes, g)
    total, matched, unmatched = res[0]
    assert total == 53 and not unmatched, unmatched
    assert all(j == 1.0 and n_gpu == n_ref for (_, _, j, n_gpu, n_ref) in matched), matched
    total1, matched1, unmatched1 = res[1]
    assert total1 == 53 and len(unmatched1) <= 1
    assert sum(n_gpu == n_ref and j == 1.0 for (_, _, j, n_gpu, n_ref) in matched1) >= 52 - len(unmatched1)


def test_g2_plane_sets_on_the_real_room_scan(ctx):
    """The same plane-set parity on a real, noisy scan (G9: sample_data/room_target.ply and the surrogate source cut
    from it): every plane of both libransac draws is found by the GPU extraction with (nearly) the same support."""
    g = load("g9_room.npz")
    for cloud, pre in ((g["target"], "t"), (g["source"], "s")):
        coef, off, idx = ctx.extract_planes(cloud, 625)   # below the smallest support the reference draws ended at
        sets = [set(idx[off[p]:off[p + 1]].tolist()) for p in range(len(coef))]
        for draw in ("", "b"):
            rc, ro, ri = g[f"{pre}{draw}_coef"], g[f"{pre}{draw}_off"], g[f"{pre}{draw}_idx"]
            assert int(np.diff(ro).min()) > 625
            best = []
            for p in range(len(rc)):
                ref = set(ri[ro[p]:ro[p + 1]].tolist())
                cand = [q for q in range(len(coef)) if abs(coef[q, :3] @ rc[p, :3]) > 0.99]
                best.append(max([len(sets[q] & ref) / len(sets[q] | ref) for q in cand], default=0.0))
            assert min(best) > 0.85, (pre, draw, best)
            assert sum(b >= 0.97 for b in best) >= len(rc) - 1, (pre, draw, best)


def test_g2_plane_sets_on_a_20k_synthetic_scene(ctx):
    """G2, synthetic part: 20 000 points, libransac at min_support 250 -- every plane it found is found by
    the GPU extraction with (nearly) the same support."""
    g = load("g2_synth20k.npz")
    rc, ro, ri = g["coef"], g["off"], g["idx"]
    coef, off, idx = ctx.extract_planes(g["cloud"], int(g["min_support"]))
    sets = [set(idx[off[p]:off[p + 1]].tolist()) for p in range(len(coef))]
    best = []
    for p in range(len(rc)):
        ref = set(ri[ro[p]:ro[p + 1]].tolist())
        cand = [q for q in range(len(coef)) if abs(coef[q, :3] @ rc[p, :3]) > 0.99]
        best.append(max([len(sets[q] & ref) / len(sets[q] | ref) for q in cand], default=0.0))
    assert min(best) > 0.85, best
    assert sum(b >= 0.97 for b in best) >= len(rc) - 2, best
    assert len(rc) <= len(coef) <= len(rc) + 6
