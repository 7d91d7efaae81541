"""GPU plane extraction (seam S1b) and the full registration() overload (plade.h:58).

The reference's RANSAC is time-seeded and not reproducible even by itself, so the
stage is pinned by (a) kernel parity (test_gpu_seams.py, test_gpu_ransac_kernels below), (b) plane-set
level checks against the generator's ground truth, (c) the planes-given boundary: the oracle run on
the planes the GPU extracted must give the same transform."""
import numpy as np
import pytest

import plade_amd
from plade_amd.synth import make_pair, sample_scene, planes_from_labels
from conftest import GT_TOL, CLOSED_FORM

pytestmark = pytest.mark.gpu


def _match_planes(coef, gt_coef):
    """for every GT plane the best extracted plane (same orientation)"""
    out = []
    for g in gt_coef:
        c = coef[:, :3] @ g[:3]
        dd = np.abs(coef[:, 3] - g[3])
        ok = np.nonzero((c > 0.999) & (dd < 0.03))[0]
        out.append(ok)
    return out


def test_extract_planes_recovers_scene(ctx, oracle):
    n = 120000
    cloud, labels = sample_scene(n, scene_seed=11, sample_seed=12, n_boxes=6, return_labels=True)
    gt_coef, gt_off, gt_idx = planes_from_labels(cloud, labels)
    min_support = 1200
    coef, off, idx = ctx.extract_planes(cloud, min_support)
    P = len(coef)
    assert P >= 10
    sup = np.diff(off)
    assert (sup >= min_support).all()
    assert len(np.unique(idx)) == len(idx), "a point belongs to at most one plane"
    assert idx.min() >= 0 and idx.max() < n
    # unit normals oriented like the inlier normals, d = -n.p
    assert np.allclose(np.linalg.norm(coef[:, :3], axis=1), 1.0, atol=1e-5)
    scale = oracle.cloud_scale(cloud)
    eps3 = 3 * 0.005 * scale
    for p in range(P):
        ids = idx[off[p]:off[p + 1]]
        assert cloud[ids, 3:].mean(0) @ coef[p, :3] > 0
        dist = np.abs(cloud[ids, :3] @ coef[p, :3] + coef[p, 3])
        assert (dist < eps3 * 1.001).mean() > 0.999
        assert (np.abs(cloud[ids, 3:] @ coef[p, :3]) >= 0.8 - 1e-4).mean() > 0.999
    # every ground-truth face that is clearly above min_support is found exactly once, and most of its points
    m = _match_planes(coef, gt_coef)
    for g in range(len(gt_coef)):
        face = gt_idx[gt_off[g]:gt_off[g + 1]]
        if len(face) < 1.5 * min_support:
            continue
        assert len(m[g]) >= 1, f"face {g} ({len(face)} pts) not extracted"
        best = max(m[g], key=lambda p: sup[p])
        got = set(idx[off[best]:off[best + 1]].tolist())
        cover = len(got.intersection(face.tolist())) / len(face)
        assert cover > 0.9, (g, cover)


def test_extract_planes_deterministic_and_small_inputs(ctx):
    cloud = sample_scene(50000, scene_seed=2, sample_seed=3, n_boxes=3)
    a = ctx.extract_planes(cloud, 800)
    b = ctx.extract_planes(cloud, 800)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # fewer than 3 points: nothing (plane_extraction.cpp:181-184)
    c = ctx.extract_planes(cloud[:2], 1)
    assert len(c[0]) == 0
    # min_support larger than the cloud: nothing
    d = ctx.extract_planes(cloud[:5000], 6000)
    assert len(d[0]) == 0


@pytest.mark.parametrize("seed", [0, 2])
def test_full_registration_matches_oracle_on_same_planes(oracle, seed):
    import plade_amd
    n = 100000
    tg, sr, Tgt = make_pair(n, seed=seed)
    ctx = plade_amd.Context(0, dump=1, orient_normals=1)
    ok, T = ctx.registration(tg, sr)
    assert ok
    d = ctx.dump()
    assert np.linalg.norm(T.astype(np.float64) - Tgt) < GT_TOL, "registration should recover the generator's SE(3)"
    tp = (d["tgt_planes"].reshape(-1, 4), d["tgt_plane_offsets"], d["tgt_plane_idx"])
    sp = (d["src_planes"].reshape(-1, 4), d["src_plane_offsets"], d["src_plane_idx"])
    assert 10 <= len(tp[0]) <= 40 and 10 <= len(sp[0]) <= 40
    ok_o, T_o, do = oracle.registration(tg, sr, tp, sp, voxel_sort_mode=1)
    assert ok_o
    assert np.array_equal(T, T_o)
    assert np.array_equal(d["overlap_counts"], do["overlap_counts"])
    ok_f, T_f, _ = oracle.registration(tg, sr, tp, sp, voxel_sort_mode=0)
    assert np.linalg.norm(T.astype(np.float64) - T_f.astype(np.float64)) <= 1e-4
    ctx.close()


def test_registration_dev_equals_host_pointer_path():
    import plade_amd
    tg, sr, Tgt = make_pair(60000, seed=5, n_boxes=6)
    ctx = plade_amd.Context(0, orient_normals=1)
    ok1, T1 = ctx.registration(tg, sr)
    ct, cs = ctx.upload(tg), ctx.upload(sr)
    ok2, T2 = ctx.registration_dev(ct, cs)
    ok3, T3 = ctx.registration_dev(ct, cs)
    assert ok1 and ok2 and ok3
    assert np.array_equal(T1, T2) and np.array_equal(T2, T3)
    ct.free(); cs.free()
    ctx.close()


def test_both_hypothesis_schedules_register_the_same_pairs():
    """The extraction draws a round of hypotheses in every iteration (what is left of the candidate pool competes with
    the new draws, as in the reference's loop, RansacShapeDetector.cpp:548-617); plade_params.ransac_topup = 0 draws only
    when the pool is empty (two more launches per iteration, more iterations).  Both are the same search with the same
    acceptance semantics: the same large planes with (nearly) the same supports, and the same registration up to the
    noise of which small faces were found.  The switch is a parameter of the context: both schedules run in this process."""
    res = {}
    for mode in (1, 0):
        ctx = plade_amd.Context(0, orient_normals=1, ransac_topup=mode)
        out = {}
        for seed in (3, 7):
            tg, sr, Tgt = make_pair(200000, seed=seed)
            co, off, idx = ctx.extract_planes(tg, 2000)
            ok, T = ctx.registration(tg, sr)
            st = ctx.stats()
            out[seed] = {"supports": sorted((int(b - a) for a, b in zip(off[:-1], off[1:])), reverse=True), "ok": bool(ok),
                         "err": float(np.linalg.norm(T - Tgt)), "iterations": int(st["ransac_iterations"])}
        ctx.close()
        res[mode] = out
    for seed in res[1]:
        a, b = res[1][seed], res[0][seed]
        assert a["ok"] and b["ok"] and a["err"] < GT_TOL and b["err"] < GT_TOL, (seed, a["err"], b["err"])
        # the six largest planes (walls, floor, ceiling) are the same surfaces with the same supports to a fraction of a percent
        for x, y in zip(a["supports"][:6], b["supports"][:6]):
            assert abs(x - y) <= 0.01 * max(x, y), (seed, a["supports"][:8], b["supports"][:8])
        assert abs(len(a["supports"]) - len(b["supports"])) <= 4
        assert a["iterations"] <= b["iterations"] + 3   # usually fewer (1M-point pairs: 5 instead of 7), never many more


def test_batch_mode_prefetch_returns_the_bits_of_the_plain_call():
    """plade_registration_next (plade.h:58 in batch mode: the next pair's upload runs under the current pair's kernels):
    same transforms as plade_registration, the prefetched copy is the one used, and a call that is handed another pair
    than the one announced falls back to its own upload."""
    import plade_amd
    pairs = [make_pair(80000, seed=s)[:2] for s in (3, 4, 5)]
    pairs = [(np.ascontiguousarray(a), np.ascontiguousarray(b)) for a, b in pairs]
    ctx = plade_amd.Context(0, orient_normals=1)
    want = [ctx.registration(tg, sr) for tg, sr in pairs]
    for hw in (0, 1):
        ctx.set_params(host_wait=hw)
        got, used = [], []
        for k, (tg, sr) in enumerate(pairs):
            nxt = pairs[k + 1] if k + 1 < len(pairs) else (None, None)
            got.append(ctx.registration_next(tg, sr, nxt[0], nxt[1]))
            used.append(ctx.stats().get("upload_prefetched", 0.0))
        assert used == [0.0, 1.0, 1.0]
        for (ok, T), (ok2, T2) in zip(want, got):
            assert ok and ok2 and np.array_equal(T, T2)
    # announced pair 1, handed pair 2: own upload, right answer; the stale prefetch is dropped
    ctx.registration_next(pairs[0][0], pairs[0][1], pairs[1][0], pairs[1][1])
    ok, T = ctx.registration_next(pairs[2][0], pairs[2][1])
    assert ctx.stats().get("upload_prefetched", 0.0) == 0.0 and ok and np.array_equal(T, want[2][1])
    ok, T = ctx.registration(pairs[1][0], pairs[1][1])
    assert ok and np.array_equal(T, want[1][1])
    # a NaN in an announced cloud is refused when that cloud's turn comes, as the plain call refuses it
    bad = pairs[1][0].copy()
    bad[7, 1] = np.nan
    ctx.registration_next(pairs[0][0], pairs[0][1], bad, pairs[1][1])
    with pytest.raises(plade_amd.PladeError):
        ctx.registration_next(bad, pairs[1][1])
    ok, T = ctx.registration(pairs[0][0], pairs[0][1])
    assert ok and np.array_equal(T, want[0][1])
    ctx.close()


def test_tall_scene_takes_the_global_memory_labelling_path():
    """The bitmap cell is 2 % of max(dx, dy) -- Z is ignored (plane_extraction.cpp:71-80) -- so the walls of a tall, narrow
    shaft rasterise to far more than the 4096 pixels the labelling kernel keeps in LDS: it labels in global memory instead,
    and the walls come out whole."""
    import plade_amd
    rng = np.random.default_rng(4)
    n_wall, h = 60000, 40.0
    walls = []
    for ax, c, sgn in ((0, 0.0, 1), (0, 2.0, -1), (1, 0.0, 1), (1, 2.0, -1)):
        p = np.zeros((n_wall, 6), np.float32)
        p[:, ax] = c + rng.normal(0, 0.002, n_wall)
        p[:, 1 - ax] = rng.random(n_wall) * 2.0
        p[:, 2] = rng.random(n_wall) * h
        p[:, 3 + ax] = sgn
        walls.append(p)
    cloud = np.ascontiguousarray(np.concatenate(walls)[rng.permutation(4 * n_wall)])
    ctx = plade_amd.Context(0, orient_normals=1)
    coef, off, idx = ctx.extract_planes(cloud, 5000)
    assert len(coef) == 4
    sizes = np.diff(off)
    assert sizes.min() > 0.97 * n_wall          # 2 m / 0.04 m x 40 m / 0.04 m = 50 000 pixels per wall, one component each
    for p in range(4):
        ax = int(np.argmax(np.abs(coef[p, :3])))
        assert ax in (0, 1) and abs(abs(coef[p, ax]) - 1) < 1e-3
        assert np.all(np.abs(cloud[idx[off[p]:off[p + 1]], 3 + ax]) == 1)
    # the same cloud squeezed to 4 m: the walls fit the LDS labelling (50 x 100 pixels); same points per wall
    flat = cloud.copy()
    flat[:, 2] *= 0.1
    coef2, off2, idx2 = ctx.extract_planes(flat, 5000)
    assert len(coef2) == 4 and sorted(np.diff(off2)) == sorted(sizes)
    ctx.close()
