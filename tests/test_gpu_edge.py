"""Edge cases of every C-ABI entry point on the GPU: empty and ragged inputs, exact duplicates and
collisions, masked-out (already assigned) points, capacity overflow, invalid arguments, failure
reporting, and independence of contexts that share a GPU."""
import threading

import numpy as np
import pytest

import plade_amd
from plade_amd.synth import make_pair, planes_from_labels, sample_scene
from conftest import GT_TOL, CLOSED_FORM

pytestmark = pytest.mark.gpu


def test_empty_inputs_every_seam(ctx):
    e6, e3, e8 = np.zeros((0, 6), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 8), np.float32)
    assert ctx.score_planes(e6, None, np.array([[0, 0, 1, 0]], np.float32), 0.1, 0.8)[0] == 0
    off, nbr, d2 = ctx.match_descriptors(e8, e8)
    assert list(off) == [0] and len(nbr) == 0
    off, nbr, d2 = ctx.match_descriptors(np.zeros((3, 8), np.float32), e8)
    assert list(off) == [0, 0, 0, 0]
    assert len(ctx.voxel_downsample(e3, 0.1)) == 0
    T = np.eye(4, dtype=np.float32)[None]
    c = np.zeros((1, 3), np.float32)
    pts = np.random.default_rng(0).random((100, 3)).astype(np.float32)
    assert ctx.overlap_counts(e3, pts, T, c, 10.0, 0.1)[0] in (0, -1)     # no source points
    assert ctx.overlap_counts(pts, e3, T, c, 10.0, 0.1)[0] == -1          # no target points: empty sphere
    assert len(ctx.overlap_counts(pts, pts, np.zeros((0, 4, 4), np.float32), e3, 10.0, 0.1)) == 0
    kept, fit, ws = ctx.plane_component(sample_scene(256, 1, 1), [0, 0, 1], [0, 0, 0], np.zeros(0, np.int32), 0.1, True, 0.1)
    assert len(kept) == 0 and ws == 0.0


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 65, 255, 257, 1023, 1025, 4097])
def test_ragged_sizes_score_and_compaction(ctx, oracle, n):
    """Every tail shape of the 4-points-per-lane / 1024-points-per-block tiling."""
    cloud = sample_scene(max(n, 64), scene_seed=11, sample_seed=n)[:n]
    planes = np.array([[0, 0, 1, 0.0], [1, 0, 0, 0.0], [0, 1, 0, 8.0]], np.float32)
    counts, lists = ctx.score_planes(cloud, None, planes, 0.2, 0.5, want_indices=True)
    for j in range(3):
        ref = oracle.score_plane(cloud, None, planes[j], 0.2, 0.5)
        assert counts[j] == len(ref) and np.array_equal(lists[j], ref)


def test_all_points_masked_out(ctx):
    cloud = sample_scene(5000, 3, 3)
    si = np.zeros(len(cloud), np.int32)          # everything already belongs to shape 0
    counts = ctx.score_planes(cloud, si, np.array([[0, 0, 1, 0]], np.float32), 10.0, 0.0)
    assert counts[0] == 0
    si[::7] = -1
    counts = ctx.score_planes(cloud, si, np.array([[0, 0, 1, 0]], np.float32), 1e6, 0.0)
    assert counts[0] == len(si[::7])


def test_score_capacity_truncates_lists_not_counts(ctx):
    cloud = sample_scene(3000, 3, 5)
    L = ctx.L
    import ctypes as C
    pl = np.array([[0, 0, 1, 0]], np.float32)
    counts = np.zeros(1, np.uint32)
    idx = np.full(10, 0xFFFFFFFF, np.uint32)
    rc = L.plade_score_planes(ctx.h, cloud.ctypes.data_as(C.c_void_p), None, len(cloud), pl.ctypes.data_as(C.c_void_p), 1,
                              C.c_float(1e6), C.c_float(0.0), counts.ctypes.data_as(C.c_void_p),
                              idx.ctypes.data_as(C.c_void_p), 10)
    assert rc == 0 and counts[0] == len(cloud) and list(idx) == list(range(10))


def test_duplicate_points_and_descriptors(ctx, oracle):
    rng = np.random.default_rng(5)
    base = sample_scene(2000, 4, 4)
    cloud = np.concatenate([base, base, base[:500]])       # exact duplicates: voxel sums, spacing zeros
    got = ctx.voxel_downsample(cloud, 0.2)
    assert np.array_equal(got, oracle.voxel_downsample(cloud, 0.2, 1))
    assert ctx.average_spacing(cloud) == np.float32(oracle.average_spacing(cloud))
    t = (rng.random((300, 8)) * 0.1).astype(np.float32)
    t = np.concatenate([t, t])                               # every target twice: full distance ties
    q = t[:50].copy()
    o1, n1, d1 = oracle.match_descriptors(q, t, 0.04)
    o2, n2, d2 = ctx.match_descriptors(q, t, 0.04)
    assert np.array_equal(o1, o2) and np.array_equal(n1, n2) and np.array_equal(d1, d2)
    assert (d2 == 0).sum() >= 100                             # the exact-duplicate hits are there


def test_match_capacity_overflow_is_reported(ctx):
    import ctypes as C
    t = np.zeros((64, 8), np.float32)
    q = np.zeros((4, 8), np.float32)
    off = np.zeros(5, np.int64)
    nbr = np.zeros(8, np.uint32)
    d2 = np.zeros(8, np.float64)
    total = C.c_uint64()
    rc = ctx.L.plade_match_descriptors(ctx.h, q.ctypes.data_as(C.c_void_p), 4, t.ctypes.data_as(C.c_void_p), 64,
                                       C.c_float(0.04), off.ctypes.data_as(C.c_void_p), nbr.ctypes.data_as(C.c_void_p),
                                       d2.ctypes.data_as(C.c_void_p), 8, C.byref(total))
    assert rc == plade_amd.PLADE_ECAP and total.value == 256 and list(off) == [0, 64, 128, 192, 256]


def test_invalid_arguments_return_codes_not_crashes(ctx):
    with pytest.raises(plade_amd.PladeError) as e:
        ctx.registration_planes(np.zeros((10, 6), np.float32), np.zeros((10, 6), np.float32),
                                (np.zeros((1, 4), np.float32), np.array([0, 20], np.int32), np.arange(20, dtype=np.int32)),
                                (np.zeros((1, 4), np.float32), np.array([0, 5], np.int32), np.arange(5, dtype=np.int32)))
    assert e.value.code == plade_amd.PLADE_EINVAL and "plane" in str(e.value).lower()
    with pytest.raises(plade_amd.PladeError):
        ctx.plane_component(sample_scene(100, 1, 1), [0, 0, 1], [0, 0, 0], np.array([5, 5], np.int32), 0.1, True, 0.1)
    # a degenerate cloud (all points identical) has no bounding box: error, not a hang
    same = np.tile(np.array([[1, 2, 3, 0, 0, 1]], np.float32), (5000, 1))
    with pytest.raises(plade_amd.PladeError):
        ctx.extract_planes(same, 100)


def test_registration_failure_paths(ctx):
    rng = np.random.default_rng(0)
    noise = np.concatenate([rng.random((20000, 3)) * 5, rng.normal(size=(20000, 3))], 1).astype(np.float32)
    noise[:, 3:] /= np.linalg.norm(noise[:, 3:], axis=1, keepdims=True)
    ok, T = ctx.registration(noise, noise)                  # no planes at all -> false, identity
    assert not ok and np.array_equal(T, np.eye(4, dtype=np.float32))
    ok, T = ctx.registration_minsupport(noise, noise, 5000, 5000)
    assert not ok and np.array_equal(T, np.eye(4, dtype=np.float32))


def test_contexts_sharing_a_gpu_are_independent():
    """Registrations in flight on one GPU (bench.py --inflight, the CLI's PLADE_INFLIGHT): same results
    as one at a time."""
    pairs = [make_pair(60000, seed=s) for s in (3, 4)]
    ref_ctx = plade_amd.Context(0, orient_normals=1)
    want = [ref_ctx.registration(tg, sr) for (tg, sr, _) in pairs]
    ref_ctx.set_params(**CLOSED_FORM)
    accurate = [ref_ctx.registration(tg, sr) for (tg, sr, _) in pairs]
    ref_ctx.close()
    got = {}

    def work(w):
        c = plade_amd.Context(0, orient_normals=1)
        for rep in range(10):   # enough repetitions to expose a missing cross-stream dependency (one was found this way)
            for i, (tg, sr, _) in enumerate(pairs):
                got[(w, rep, i)] = c.registration(tg, sr)
        c.close()

    ths = [threading.Thread(target=work, args=(w,)) for w in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert len(got) == 80
    for (w, rep, i), (ok, T) in got.items():
        assert ok == want[i][0] and np.array_equal(T, want[i][1]), (w, rep, i)
    for i, (_, _, Tgt) in enumerate(pairs):
        assert want[i][0] and accurate[i][0] and np.linalg.norm(accurate[i][1] - Tgt) < 1e-2   # 60k points: coarser than the 1M pairs (closed form: conftest.GT_TOL)


@pytest.mark.timeout(120)
def test_non_finite_inputs_neither_hang_nor_poison_the_context(ctx):
    """NaN / infinite coordinates are refused (every grid and threshold derives from them; PCL's kd-trees and
    voxel grids cannot take them either); NaN or zero normals just never pass the normal test."""
    tg, sr, Tgt = make_pair(100000, seed=2)
    for poison in ("nan", "inf"):
        a, b = tg.copy(), sr.copy()
        a[::997, :3] = np.nan if poison == "nan" else np.inf
        with pytest.raises(plade_amd.PladeError) as e:
            ctx.registration(a, b)
        assert e.value.code == plade_amd.PLADE_EINVAL and "non-finite" in str(e.value)
        with pytest.raises(plade_amd.PladeError):
            ctx.voxel_downsample(a[:, :3].copy(), 0.1)
    a, b = tg.copy(), sr.copy()
    a[::991, 3:] = np.nan
    b[::3, 3:] = 0.0
    ok, T = ctx.registration(a, b)
    assert ok and np.linalg.norm(T - Tgt) < GT_TOL
    a[:, 3:] = np.nan                                   # no usable normal at all: no planes, clean failure
    ok, T = ctx.registration(a, b)
    assert not ok and np.array_equal(T, np.eye(4, dtype=np.float32))
    ok, T = ctx.registration(tg, sr)                    # the context is still good
    assert ok and np.linalg.norm(T - Tgt) < GT_TOL


@pytest.mark.timeout(180)
@pytest.mark.parametrize("case", ["line", "coplanar", "huge", "tiny", "far_outlier", "disjoint"])
def test_degenerate_geometry_terminates(ctx, oracle, case):
    """Inputs no plane-based registration can solve must come back quickly with `false` (or a clean error), not
    hang: collinear and coplanar clouds, absurd scales, one point a billion metres away, unrelated clouds."""
    tg, sr, _ = make_pair(60000, seed=2)
    a, b = tg.copy(), sr.copy()
    if case == "line":
        for c in (a, b):
            c[:, 1:3] = 0.0
            c[:, 3:] = [0, 0, 1]
    if case == "coplanar":
        for c in (a, b):
            c[:, 2] = 0.0
            c[:, 3:] = [0, 0, 1]
    if case == "huge":
        a[:, :3] *= 1e18
        b[:, :3] *= 1e18
    if case == "tiny":
        a[:, :3] *= 1e-18
        b[:, :3] *= 1e-18
    if case == "far_outlier":
        a[0, :3] = [1e12, 0, 0]
    if case == "disjoint":
        b[:, :3] = np.random.default_rng(0).random((len(b), 3)) * 3
    try:
        ok, T = ctx.registration(a, b)
        assert not ok and np.array_equal(T, np.eye(4, dtype=np.float32))
    except plade_amd.PladeError as e:
        assert e.code in (plade_amd.PLADE_EINVAL, plade_amd.PLADE_ELIMIT, plade_amd.PLADE_EFAIL)
    if case == "line":   # the spacing of a collinear cloud is still the exact kNN value
        assert ctx.average_spacing(a) == np.float32(oracle.average_spacing(a))


def test_match_lists_longer_than_the_in_lds_ranking(ctx, oracle):
    """Lists above 4096 entries take the two-radix-sort path of k_match.hip instead of the per-query ranking."""
    rng = np.random.default_rng(9)
    t = (rng.random((6000, 8)) * 0.01).astype(np.float32)     # everything within the radius of everything
    t[1000:1500] = t[0:500]                                     # exact distance ties inside the long lists
    q = np.concatenate([t[:3], (rng.random((2, 8)) * 0.01).astype(np.float32)])
    o1, n1, d1 = oracle.match_descriptors(q, t, 0.04)
    o2, n2, d2 = ctx.match_descriptors(q, t, 0.04)
    assert (np.diff(o2) > 4096).all()
    assert np.array_equal(o1, o2) and np.array_equal(n1, n2) and np.array_equal(d1, d2)


def test_host_wait_sleep_mode_returns_the_same_bits(ctx):
    """plade_params.host_wait = 1 (poll + sleep instead of the HIP runtime's spinning waits) changes how the host
    waits, not what the GPU computes."""
    import plade_amd
    tg, sr, _ = make_pair(120000, seed=5)
    ok, T = ctx.registration(tg, sr)
    c2 = plade_amd.Context(0, host_wait=1, orient_normals=1)
    ok2, T2 = c2.registration(tg, sr)
    c2.close()
    assert ok and ok2 and np.array_equal(T, T2)


def test_candidate_shards_of_the_verification_seam_add_up(ctx):
    """the candidates of one pair split over ranks (round robin) give, shard by shard, the counts of the
    unsharded call (plade_amd.batch.sharded_overlap_counts does this with an all_gather; gloo test on CPU)."""
    from plade_amd.batch import shard
    rng = np.random.default_rng(3)
    tg = ctx.voxel_downsample(make_pair(200000, seed=2)[0], 0.05)
    K = 37
    T = np.tile(np.eye(4, dtype=np.float32), (K, 1, 1))
    T[:, :3, 3] = rng.normal(0, 0.05, (K, 3)).astype(np.float32)
    centers = (tg.mean(0)[None, :] + T[:, :3, 3]).astype(np.float32)
    full = ctx.overlap_counts(tg, tg, T, centers, np.float32(4.0), np.float32(0.05))
    for world in (2, 3):
        out = np.zeros(K, np.int32)
        for r in range(world):
            idx = shard(K, r, world)
            out[idx] = ctx.overlap_counts(tg, tg, T[idx], centers[idx], np.float32(4.0), np.float32(0.05))
        assert np.array_equal(out, full)


@pytest.mark.gpu
@pytest.mark.parametrize("host_wait", [0, 1])
def test_device_to_host_hand_over_many_and_large_ranges(host_wait):
    """Every readback of the library is noted by d2h() and handed over by one kernel per wait into a host-mapped arena, with a
    flag word the host polls (ctx.h, prims.hip k_copy_out).  The registration never has more than a handful pending; this
    drives the seam with more ranges than one launch takes (8), with ranges of many workgroups, ragged sizes, and both kinds
    of host wait, repeatedly on one context (the sequence word must keep advancing)."""
    import plade_amd
    c = plade_amd.Context(0, host_wait=host_wait)
    for n_ranges, words in ((1, 1), (3, 7), (8, 1000), (9, 1000), (23, 4099), (5, 200000), (64, 17), (12, 65536)):
        for _ in range(3):
            assert c.selftest_readback(n_ranges, words) == 0, (n_ranges, words)
    c.close()
