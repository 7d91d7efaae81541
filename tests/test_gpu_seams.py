"""Parity of the three seam kernels (S1a score, S2 match, S3 overlap) against the oracle,
called through the C ABI of libplade_hip.so.  Bit-exact integer outputs."""
import numpy as np
import pytest

from plade_amd.synth import sample_scene

pytestmark = pytest.mark.gpu


def _hyps(rng, cloud, h):
    tri = cloud[rng.integers(0, len(cloud), (h, 3)), :3].reshape(h, 9)
    return tri


@pytest.mark.parametrize("n,h", [(1, 3), (1023, 5), (1024, 7), (1025, 130), (50000, 40)])
def test_score_planes_bit_exact(ctx, oracle, n, h):
    rng = np.random.default_rng(n + h)
    cloud = sample_scene(max(n, 64), scene_seed=3, sample_seed=n)[:n]
    si = np.full(n, -1, np.int32)
    si[rng.random(n) < 0.2] = 2
    planes = []
    base = sample_scene(4096, scene_seed=3, sample_seed=99)
    for t in _hyps(rng, base, h):
        ok, pl = oracle.plane_from_points(t)
        planes.append(pl if ok else np.array([0, 0, 1, 0.3], np.float32))
    planes = np.array(planes, np.float32)
    eps, cos_t = 0.05, 0.8
    counts, lists = ctx.score_planes(cloud, si, planes, eps, cos_t, want_indices=True)
    for j in range(h):
        ref = oracle.score_plane(cloud, si, planes[j], eps, cos_t)
        assert counts[j] == len(ref)
        assert np.array_equal(lists[j], ref)
    # no shape index
    counts2 = ctx.score_planes(cloud, None, planes, eps, cos_t)
    for j in range(min(h, 8)):
        assert counts2[j] == len(oracle.score_plane(cloud, None, planes[j], eps, cos_t))


@pytest.mark.parametrize("n,m,h", [(5000, 1, 3), (50000, 1023, 70), (50000, 16384, 200), (20000, 20000, 4100)])
def test_score_planes_subset_bit_exact(ctx, oracle, n, m, h):
    """The loop's subset-scoring kernel (k_r_score_sub: a round of hypotheses on the stratified subset, the call shape of
    Candidate::ImproveBounds on subset 0, RansacShapeDetector.cpp:163) against the oracle's visitor on the gathered subset."""
    rng = np.random.default_rng(n + m + h)
    cloud = sample_scene(n, scene_seed=5, sample_seed=n)
    si = np.full(n, -1, np.int32)
    si[rng.random(n) < 0.25] = 0
    sub = (np.arange(m, dtype=np.uint32) * max(1, n // m)) if m > 1 else np.array([n // 2], np.uint32)
    if m == 1023:
        sub = rng.integers(0, n, m).astype(np.uint32)   # any order, repeats allowed
    base = sample_scene(4096, scene_seed=5, sample_seed=98)
    planes = []
    for t in _hyps(rng, base, h):
        ok, pl = oracle.plane_from_points(t)
        planes.append(pl if ok else np.array([0, 0, 1, 0.3], np.float32))
    planes = np.array(planes, np.float32)
    eps, cos_t = 0.05, 0.8
    counts, un = ctx.score_planes_subset(cloud, si, sub, planes, eps, cos_t)
    assert un == int((si[sub] == -1).sum())
    sc, ss = np.ascontiguousarray(cloud[sub]), np.ascontiguousarray(si[sub])
    for j in list(range(min(h, 40))) + list(range(max(0, h - 10), h)):
        assert counts[j] == len(oracle.score_plane(sc, ss, planes[j], eps, cos_t)), j
    # vectorised fp32 restatement for all hypotheses (FlatNormalThreshPointCompatibilityFunc.h:14-23, left-to-right dots)
    x, q = sc[:, :3], sc[:, 3:]
    for j in range(h):
        pl = planes[j]
        d = (pl[0] * x[:, 0] + pl[1] * x[:, 1]) + pl[2] * x[:, 2]
        nd = (pl[0] * q[:, 0] + pl[1] * q[:, 1]) + pl[2] * q[:, 2]
        ref = int(((np.abs(pl[3] - d) < np.float32(eps)) & (np.abs(nd) >= np.float32(cos_t)) & (ss == -1)).sum())
        assert counts[j] == ref, j
    c0, un0 = ctx.score_planes_subset(cloud, None, sub, planes[:3], eps, cos_t)
    assert un0 == m


def test_score_planes_empty(ctx):
    counts = ctx.score_planes(np.zeros((0, 6), np.float32), None, np.array([[0, 0, 1, 0]], np.float32), 0.1, 0.8)
    assert counts[0] == 0


@pytest.mark.parametrize("dq,dt", [(0, 10), (7, 0), (300, 5000), (1000, 9000)])
def test_match_descriptors_exact(ctx, oracle, dq, dt):
    rng = np.random.default_rng(dq * 7 + dt)
    t = (rng.random((dt, 8)) * 0.25).astype(np.float32)
    if dt >= 200:
        t[100:200] = t[0:100]  # exact duplicates -> distance ties
    q = (rng.random((dq, 8)) * 0.25).astype(np.float32)
    if dq and dt:
        k = min(dq, dt) // 2
        q[:k] = t[:k] + rng.normal(0, 0.01, (k, 8)).astype(np.float32)
        q[-1] = t[0]
    o1, n1, d1 = oracle.match_descriptors(q, t, 0.04)
    o2, n2, d2 = ctx.match_descriptors(q, t, 0.04)
    assert np.array_equal(o1, o2)
    assert np.array_equal(n1, n2)
    assert np.array_equal(d1, d2)  # fp64 distances bit-identical


def test_match_radius_boundary(ctx, oracle):
    # distances straddling r^2 = float(0.04f*0.04f): membership uses <= in double
    r2 = float(np.float32(0.04) * np.float32(0.04))
    base = np.zeros((1, 8), np.float32)
    t = np.zeros((64, 8), np.float32)
    for i in range(64):
        t[i, 0] = np.float32(np.sqrt(r2)) + np.float32((i - 32) * 1e-9)
    o1, n1, d1 = oracle.match_descriptors(base, t, 0.04)
    o2, n2, d2 = ctx.match_descriptors(base, t, 0.04)
    assert np.array_equal(n1, n2) and np.array_equal(d1, d2)


@pytest.mark.parametrize("seed", [0, 1])
def test_overlap_counts_bit_exact(ctx, oracle, seed):
    rng = np.random.default_rng(seed)
    cloud = sample_scene(60000, scene_seed=5, sample_seed=seed)
    leaf = np.float32(0.12)
    tg = oracle.voxel_downsample(cloud, leaf, 1)
    sr = (tg + rng.normal(0, 0.03, tg.shape)).astype(np.float32)[:: 2]
    K = 19
    Ts, cs = [], []
    for k in range(K):
        ang = rng.normal(0, 0.03 if k % 3 else 0.5)
        c, s = np.cos(ang), np.sin(ang)
        T = np.eye(4, dtype=np.float32)
        T[:2, :2] = [[c, -s], [s, c]]
        T[:3, 3] = rng.normal(0, 0.1 if k % 2 else 2.0, 3)
        Ts.append(T)
        cs.append((T[:3, :3] @ np.array([0.1, 0.2, 0.0], np.float32) + T[:3, 3]).astype(np.float32))
    Ts = np.array(Ts, np.float32)
    cs = np.array(cs, np.float32)
    # one candidate whose coarse sphere is empty
    Ts[-1, :3, 3] = 500.0
    cs[-1] = 500.0
    radius = np.float32(4.0)
    got = ctx.overlap_counts(sr, tg, Ts, cs, radius, leaf)
    for k in range(K):
        ref = oracle.overlap_count(sr, tg, Ts[k], cs[k], radius, leaf)
        assert got[k] == ref, (k, got[k], ref)
    assert got[-1] == -1


@pytest.mark.parametrize("n", [5, 700, 20000, 150000])
def test_average_spacing_bit_exact(ctx, oracle, n):
    cloud = sample_scene(max(n, 64), scene_seed=6, sample_seed=n)[:n]
    assert ctx.average_spacing(cloud) == np.float32(oracle.average_spacing(cloud))
    # xyz-only input with a different stride
    assert ctx.average_spacing(cloud[:, :3].copy()) == np.float32(oracle.average_spacing(cloud[:, :3].copy()))


def test_average_spacing_golden(ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_spacing.npz"))
    assert ctx.average_spacing(g["cloud"]) == g["spacing"]  # value composed from the reference's FLANN kNN


@pytest.mark.parametrize("n,leaf", [(1, 0.1), (3000, 0.05), (3000, 50.0), (80000, 0.12)])
def test_voxel_downsample_bit_exact(ctx, oracle, n, leaf):
    cloud = sample_scene(max(n, 64), scene_seed=8, sample_seed=n)[:n]
    got = ctx.voxel_downsample(cloud, leaf)
    want = oracle.voxel_downsample(cloud, leaf, 1)   # stable in-voxel order
    assert got.shape == want.shape and np.array_equal(got, want)
    faithful = oracle.voxel_downsample(cloud, leaf, 0)  # PCL's std::sort order: same voxels, <= few ulp apart
    assert faithful.shape == got.shape and np.abs(faithful - got).max() <= 1e-5


def test_match_windowed_equals_brute_force(oracle):
    """The length-windowed enumeration used for large descriptor tables (k_match.hip) returns exactly what the
    brute-force kernel returns: same lists, same order, same fp64 distances."""
    import plade_amd
    rng = np.random.default_rng(11)
    t = np.concatenate([(rng.random((30000, 1)) * 60).astype(np.float32), (rng.random((30000, 7)) * 0.2).astype(np.float32)], 1)
    q = t[rng.integers(0, len(t), 4000)] + rng.normal(0, 0.008, (4000, 8)).astype(np.float32)
    q[:50] = t[:50]                        # exact hits
    t[100:150] = t[0:50]                   # duplicated targets: distance ties broken by target index
    out = {}
    for mode in (-1, 1):
        c = plade_amd.Context(0, match_window=mode)
        out[mode] = c.match_descriptors(q, t, 0.04)
        c.close()
    for a, b in zip(out[-1], out[1]):
        assert np.array_equal(a, b)
    assert len(out[1][1]) > 4000
    # the same in slabs of a few hundred sorted queries (large tables: the (query, chunk) cells are laid out slab by slab)
    c = plade_amd.Context(0, match_window=1, match_cell_budget=4096)
    slabbed = c.match_descriptors(q, t, 0.04)
    c.close()
    for a, b in zip(out[-1], slabbed):
        assert np.array_equal(a, b)
    o, n, d = oracle.match_descriptors(q[:300], t, 0.04)
    assert np.array_equal(n, out[1][1][: len(n)]) and np.array_equal(d, out[1][2][: len(d)])


@pytest.mark.parametrize("dtype,bits", [(np.uint32, 24), (np.uint32, 32), (np.uint32, 9), (np.uint64, 30), (np.uint64, 47),
                                        (np.uint64, 64),
                                        # bit counts where 9-bit digits save a pass (the voxel / cell grids)
                                        (np.uint32, 18), (np.uint32, 27), (np.uint64, 26), (np.uint64, 35), (np.uint64, 63)])
@pytest.mark.parametrize("n", [16385, 100003, 1 << 20, 3000001])
def test_radix_sort_is_numpy_stable_argsort(ctx, dtype, bits, n):
    """The hand-written onesweep sort behind every grid (radix_sort.hip) = numpy's stable sort: sorted keys and, through
    the payload, the input order of equal keys -- for ragged tile counts, few distinct digits, 32 and 64-bit keys."""
    rng = np.random.default_rng(n + bits)
    hi = (1 << bits) - 1
    keys = rng.integers(0, hi, n, dtype=np.uint64, endpoint=True)
    keys[: n // 3] &= 0xFF            # many duplicates: stability matters, upper digits all equal
    keys[n // 3: n // 2] = hi         # one value repeated: a single digit bin takes whole tiles
    keys = keys.astype(dtype)
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = ctx.sort_pairs(keys, vals, bits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ko, keys[order]) and np.array_equal(vo, order.astype(np.uint32))


def test_radix_sort_sorted_reversed_and_constant_inputs(ctx):
    n = 250000
    vals = np.arange(n, dtype=np.uint32)
    for keys in (np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32)[::-1].copy(), np.full(n, 7, np.uint32),
                 np.zeros(n, np.uint32)):
        ko, vo = ctx.sort_pairs(keys, vals, 18)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(ko, keys[order]) and np.array_equal(vo, order.astype(np.uint32))


@pytest.mark.parametrize("sizes,bits", [((1000000, 1000000, 300000), 24), ((5000, 0, 70000, 1, 4097, 4096, 123457, 9), 24),
                                        (tuple([60000] * 16), 27), ((3000001, 17), 20), ((0, 0, 5), 9)])
def test_radix_sort_over_segments_sorts_every_array_on_its_own(ctx, sizes, bits):
    """Up to 16 independent arrays in one launch sequence (the Morton order of the clouds of a group, ransac.hip): every
    segment must come out as numpy's stable argsort of that segment alone -- own histogram, own prefix chain, empty and
    single-item segments, segments that end inside a tile, the 16-key-per-lane tiles above 3M keys; and an ordinary sort
    right behind it must find the histogram words of all sixteen segments zeroed."""
    rng = np.random.default_rng(len(sizes) * 1000 + bits)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(off[-1])
    keys = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
    keys[: n // 3] &= np.uint32(0xff)          # many ties in front: stability matters
    vals = rng.permutation(n).astype(np.uint32)
    ko, vo = ctx.sort_segments(keys, vals, off, bits)
    for s in range(len(sizes)):
        b, e = int(off[s]), int(off[s + 1])
        order = np.argsort(keys[b:e], kind="stable")
        assert np.array_equal(ko[b:e], keys[b:e][order]) and np.array_equal(vo[b:e], vals[b:e][order]), s
    k2 = rng.integers(0, 1 << 20, 200000, dtype=np.uint64).astype(np.uint32)
    v2 = np.arange(len(k2), dtype=np.uint32)
    for _ in range(2):                          # both histogram buffers
        ko2, vo2 = ctx.sort_pairs(k2, v2, 20)
        order = np.argsort(k2, kind="stable")
        assert np.array_equal(ko2, k2[order]) and np.array_equal(vo2, order.astype(np.uint32))


@pytest.mark.parametrize("filt", [True, False])
def test_plane_component_bitmap_larger_than_the_lds_labelling(ctx, oracle, filt):
    """Seam S1c on a bitmap of ~360 x 300 pixels (k_r_label keeps bitmaps of up to 4096 pixels in LDS; beyond that it labels
    in global memory): kept list = the oracle's (BitmapPrimitiveShape.cpp:97-265), LS fit and weighted score as for the
    small case."""
    rng = np.random.default_rng(12 + filt)
    m = 60000
    uv = rng.random((m, 2)) * [36.0, 30.0]
    # three islands separated by empty bands (wider than a closing step), the largest one in the middle
    keep = ((uv[:, 0] < 9) | ((uv[:, 0] > 11) & (uv[:, 0] < 29)) | (uv[:, 0] > 31.5)) & ~((uv[:, 1] > 14) & (uv[:, 1] < 15.5) & (uv[:, 0] > 31.5))
    uv = uv[keep]
    n = np.array([0.2, -0.3, 0.93], np.float64); n /= np.linalg.norm(n)
    a = np.cross(n, [0, 0, 1.0]); a /= np.linalg.norm(a)
    b = np.cross(n, a)
    pts = (uv[:, :1] * a + uv[:, 1:] * b + rng.normal(0, 0.01, (len(uv), 1)) * n + [1.0, 2.0, 3.0]).astype(np.float32)
    cloud = np.concatenate([pts, np.tile(n.astype(np.float32), (len(pts), 1))], 1)
    extra = sample_scene(5000, 3, 3)
    cloud = np.ascontiguousarray(np.concatenate([cloud, extra]))
    idx = rng.permutation(len(pts)).astype(np.int32)
    nrm, point = n.astype(np.float32), np.array([1.0, 2.0, 3.0], np.float32)
    kept, fit, ws = ctx.plane_component(cloud, nrm, point, idx, 0.1, filt, 0.15)
    ref = oracle.connected_component(cloud, nrm, point, idx, 0.1, filt)
    assert 0.3 * len(idx) < len(ref) < 0.8 * len(idx)          # one island of three
    assert np.array_equal(kept, ref)
    rf = oracle.ls_fit(cloud, ref)                        # unit normal, mean, dist
    sgn = np.sign(fit[:3] @ rf[:3])
    # the oracle (like the reference) adds 35 000 coordinates of magnitude 20 sequentially in fp32: its own noise is ~5e-5
    assert np.abs(sgn * fit[:3] - rf[:3]).max() < 5e-5 and np.abs(fit[3:6] - rf[3:6]).max() < 2e-4
    wref = oracle.weighted_score(cloud, nrm, point, ref, 0.15)
    assert abs(ws - wref) <= 1e-4 * max(1.0, wref)


def _overlap_case(seed=3):
    rng = np.random.default_rng(seed)
    cloud = sample_scene(40000, scene_seed=5, sample_seed=seed)
    leaf = np.float32(0.12)
    K = 37
    Ts, cs = [], []
    for k in range(K):
        ang = rng.normal(0, 0.03 if k % 3 else 0.7)
        c, s = np.cos(ang), np.sin(ang)
        T = np.eye(4, dtype=np.float32)
        T[:2, :2] = [[c, -s], [s, c]]
        T[:3, 3] = rng.normal(0, 0.1 if k % 2 else 3.0, 3)
        Ts.append(T)
        cs.append((T[:3, :3] @ np.array([0.1, 0.2, 0.0], np.float32) + T[:3, 3]).astype(np.float32))
    Ts = np.array(Ts, np.float32)
    cs = np.array(cs, np.float32)
    # candidates that throw the source far outside the target's grid, exactly one cell outside, and onto NaN
    Ts[-1, :3, 3] = 1.0e6
    Ts[-2, 0, 3] = np.nan
    Ts[-3, :3, 3] = [float(cloud[:, 0].max() - cloud[:, 0].min()) + float(leaf), 0.0, 0.0]
    return cloud, leaf, Ts, cs


def test_overlap_dense_rows_equal_bitmap_index_and_oracle(ctx, oracle, tmp_path):
    """The verification kernel's two target indices -- the dense row table (r5, the default) and the bitmap + rank index of
    rounds 3-5 (PLADE_OVERLAP_INDEX_COMPACT=1, looked up once per process: a child process) -- count the same pairs, candidate by
    candidate, incl. candidates that land outside the grid; and both equal the oracle (util.h:611-647)."""
    import subprocess, sys, os
    cloud, leaf, Ts, cs = _overlap_case()
    tg = oracle.voxel_downsample(cloud, leaf, 1)
    sr = np.ascontiguousarray(tg[::2] + np.float32(0.01))
    radius = np.float32(6.0)
    got = ctx.overlap_counts(sr, tg, Ts, cs, radius, leaf)
    np.savez(tmp_path / "in.npz", sr=sr, tg=tg, Ts=Ts, cs=cs, radius=radius, leaf=leaf)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import plade_amd; d = np.load(%r); c = plade_amd.Context(0); "
            "np.save(%r, c.overlap_counts(d['sr'], d['tg'], d['Ts'], d['cs'], d['radius'], d['leaf']))"
            % (root, str(tmp_path / "in.npz"), str(tmp_path / "out.npy")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PLADE_OVERLAP_INDEX_COMPACT="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    old = np.load(tmp_path / "out.npy")
    assert np.array_equal(np.asarray(got), old)
    for k in range(0, len(Ts), 3):
        if np.isnan(Ts[k]).any():
            continue
        assert got[k] == oracle.overlap_count(sr, tg, Ts[k], cs[k], radius, leaf), k
