"""Live comparison of the oracle with the compiled reference pieces (oracle/_ref).  Skipped where the
library has not been built (it needs /root/reference); the committed golden vectors cover the same
ground everywhere else."""
import os
import numpy as np
import pytest

from oracle.oracle import have_reference
from plade_amd.synth import sample_scene

pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref/libplade_ref.so not built")

SAMPLE = "/root/reference/sample_data"


@pytest.fixture(scope="module")
def ref():
    from oracle.oracle import Reference
    return Reference()


def test_randomised_score_cc_overlap_knn(oracle, ref):
    rng = np.random.default_rng(5)
    cloud = sample_scene(8000, scene_seed=9, sample_seed=10, n_boxes=3)
    si = np.full(len(cloud), -1, np.int32)
    si[rng.random(len(cloud)) < 0.3] = 0
    tri = cloud[rng.integers(0, len(cloud), (12, 3)), :3].reshape(12, 9)
    re, orig, planes, ok, counts, lists = ref.score_kat(cloud, si, tri, 0.04, 0.8)
    for j in range(12):
        if ok[j]:
            assert np.array_equal(oracle.score_plane(re, si[orig], planes[j], 0.04, 0.8), lists[j])
    assert oracle.cloud_scale(cloud) == ref.cloud_scale(cloud)
    q = cloud[::40, :3]
    idx, d = ref.knn_d2(cloud[:, :3], q, 6)
    assert d.shape == (len(q), 6)


@pytest.mark.skipif(not os.path.exists(os.path.join(SAMPLE, "polyhedron_target.ply")), reason="reference sample data absent")
def test_g8_polyhedron_pair_reproduces_recorded_result(oracle, ref):
    """sample_data/file_pairs_results.txt:3-7 and polyhedron_source_groundtruth.txt agree to 1e-5; the
    oracle on libransac's planes must land on the same transform."""
    from plade_amd.plyio import read_ply
    tg = read_ply(os.path.join(SAMPLE, "polyhedron_target.ply"))
    sr = read_ply(os.path.join(SAMPLE, "polyhedron_source.ply"))
    gt = np.loadtxt(os.path.join(SAMPLE, "polyhedron_source_groundtruth.txt"))

    def extract(c, seed):
        ms, trials = 10000, 1
        pl = ref.ransac_detect(c, ms, fake_time=seed)
        ms //= 2
        while len(pl[0]) < 10 and trials < 10 and ms >= 200:
            pl = ref.ransac_detect(c, ms, fake_time=seed)
            ms //= 2
            trials += 1
        return pl
    ok, T, d = oracle.registration(tg, sr, extract(tg, 1), extract(sr, 2))
    assert ok
    assert np.abs(T - gt).max() < 5e-5
    recorded = np.array([[-0.506082, 0.860669, 0.0559446, -0.252576], [0.821345, 0.500721, -0.273261, 0.863337],
                         [-0.2632, -0.0923425, -0.960312, 0.154749], [0, 0, 0, 1]])
    assert np.abs(T - recorded).max() < 5e-5


def test_cluster_and_penetration_walk_live_against_flann(oracle, ref):
    """G10 / G11 live (fresh random inputs, not the committed vectors)."""
    rng = np.random.default_rng(77)
    for trial in range(3):
        m = 2000
        t = (rng.uniform(-1, 1, (30, 3))[rng.integers(0, 30, m)] + rng.normal(0, 0.03, (m, 3))).astype(np.float32)
        e = (rng.integers(0, 2, (m, 1)) * 0.2 + rng.normal(0, 0.03, (m, 3))).astype(np.float32)
        t[50:80] = t[:30]
        a, na = oracle.cluster_transforms(t, e, 0.06, 0.004)
        b, nb = ref.cluster_transforms(t, e, 0.06, 0.004)
        assert na == nb and np.array_equal(a, b)
    for trial in range(6):
        pa = np.concatenate([rng.uniform(-1, 1, (1500, 2)), rng.normal(0, 0.004, (1500, 1))], 1).astype(np.float32)
        pb = np.concatenate([rng.uniform(-1, 1, (1200, 1)), rng.normal(0, 0.004, (1200, 1)), rng.uniform(-1, 1, (1200, 1))], 1).astype(np.float32)
        args = (np.array([0, 1, 0, 0], np.float32), np.array([-0.8, 0, 0], np.float32), np.array([1, 0, 0], np.float32),
                float(rng.uniform(0.4, 1.6)), float(rng.uniform(0.06, 0.2)), 0.01)
        assert oracle.pen_walk(pa, pb, *args) == ref.pen_walk(pa, pb, *args)
