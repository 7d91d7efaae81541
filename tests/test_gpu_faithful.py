"""The reference-faithful defaults of the library, pinned against the real reference pieces.

* params.orient_normals = 0 (the library default): plane normals keep the sign the acceptance chain ends with -- the
  reference never flips them (code/PLADE/plane_extraction.cpp:43-58: correct_normal's average normal is NaN).
* extract() (code/PLADE/plade.cpp:602-635): the auto-tuning loop over min_support, compared with the same loop
  driven over libransac (tests/golden/g2_extract.npz, tools/make_golden_extract.py).
"""
import os

import numpy as np
import pytest

import plade_amd
from plade_amd.synth import make_pair

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.fixture(scope="module")
def faithful():
    c = plade_amd.Context(0, orient_normals=0, dump=1)
    yield c
    c.close()


def test_library_default_is_the_reference_behaviour(monkeypatch):
    """plade_default_params is pure: the reference's literals, whatever the environment says (the opt-in switches are read
    by the C++ API / CLI only, plade_host.cpp)."""
    monkeypatch.setenv("PLADE_ORIENT_NORMALS", "1")
    monkeypatch.setenv("PLADE_UNORIENTED_NORMALS", "1")
    monkeypatch.setenv("PLADE_HOST_WAIT", "sleep")
    p = plade_amd.Params()
    plade_amd.load_library().plade_default_params(p)
    assert p.orient_normals == 0 and p.unoriented_normals == 0 and p.host_wait == 0
    assert (p.max_planes, p.min_planes, p.max_candidates, p.init_min_support) == (40, 10, 200, 10000)


def test_ls_fit_sign_is_libransac_s(ctx):
    """Plane::LeastSquaresFit (ransac/Plane.h:65-74 -> GfxTL Jacobi from the identity): the eigenvector sign the GPU
    fit (fp64 Jacobi from the identity, k_fit_final) returns is the one libransac returns, on every G3 case."""
    g = load("g3_cc.npz")
    seen = 0
    for i in range(int(g["n"])):
        if f"fit_{i}" not in g.files:
            continue
        _, fit, _ = ctx.plane_component(g[f"pts_{i}"], g[f"normal_{i}"], g[f"point_{i}"], g[f"idx_{i}"],
                                        float(g[f"beps_{i}"]), bool(g[f"filt_{i}"]), 0.15)
        assert fit[:3] @ g[f"fit_{i}"][:3] > 0.9999, i
        seen += 1
    assert seen >= 6


def _match(coef, rc):
    """index of the GPU plane that is libransac's plane p (same plane up to sign), -1 if none"""
    out = []
    for p in range(len(rc)):
        cos = coef[:, :3] @ rc[p, :3]
        ok = np.nonzero((np.abs(cos) > 0.9995) & (np.abs(coef[:, 3] - rc[p, 3] * np.sign(cos)) < 5e-3))[0]
        out.append(int(ok[0]) if len(ok) else -1)
    return out


def test_faithful_mode_leaves_the_chain_s_sign(faithful, ctx):
    """orient_normals = 0 vs 1 on the reference's sample clouds: the same planes and the same supports, only the sign of
    (n, d) may differ; with 0 the sign is the acceptance chain's (LS-fit eigenvector, or the three-sample hypothesis when
    no refit improved it: RansacShapeDetector.cpp:618-656), which is what libransac reports for the same plane in the
    large majority of cases (both run a Jacobi from the identity; the hypothesis' sign is arbitrary in both)."""
    g8, g9 = load("g8_polyhedron.npz"), load("g9_room.npz")
    agree = total = flipped = 0
    for cloud, rc in ((g8["target"], g8["t_coef"]), (g8["source"], g8["s_coef"]), (g9["target"], g9["t_coef"]),
                      (g9["source"], g9["s_coef"])):
        c0, o0, i0 = faithful.extract_planes(cloud, 625)
        ctx.set_params(orient_normals=1)
        c1, o1, i1 = ctx.extract_planes(cloud, 625)
        assert np.array_equal(o0, o1) and np.array_equal(i0, i1)
        sgn = np.sign(np.einsum("ij,ij->i", c0[:, :3], c1[:, :3]))
        assert np.array_equal(c0, c1 * sgn[:, None])          # bitwise: a flip negates n and d, nothing else
        flipped += int((sgn < 0).sum())
        # the oriented mode really is oriented, the faithful mode is not (inlier normals of these clouds are oriented)
        for p in range(len(c1)):
            ids = i1[o1[p]:o1[p + 1]]
            assert cloud[ids, 3:].astype(np.float64).mean(0) @ c1[p, :3] > 0
        for p, q in enumerate(_match(c0, rc)):
            if q >= 0:
                total += 1
                agree += (c0[q, :3] @ rc[p, :3]) > 0
    assert flipped >= 3, "the faithful mode must differ from the oriented one somewhere"
    assert total >= 60 and agree >= 0.7 * total, (agree, total)


@pytest.mark.parametrize("which", ["polyhedron", "synthetic"])
def test_faithful_registration_equals_oracle_on_the_same_planes(faithful, oracle, which):
    """End to end with orient_normals = 0: whatever the unoriented planes lead to (the reference registers such a pair
    only when the signs happen to be consistent), the oracle run on the planes the GPU extracted gives the same
    verdict, the same intermediates and the same transform bit for bit."""
    if which == "polyhedron":
        g = load("g8_polyhedron.npz")
        tg, sr = g["target"], g["source"]
    else:
        tg, sr, _ = make_pair(100000, seed=2)
    ok, T = faithful.registration(tg, sr)
    d = faithful.dump()
    tp = (d["tgt_planes"].reshape(-1, 4), d["tgt_plane_offsets"], d["tgt_plane_idx"])
    sp = (d["src_planes"].reshape(-1, 4), d["src_plane_offsets"], d["src_plane_idx"])
    assert len(tp[0]) >= 10 and len(sp[0]) >= 10
    ok_o, T_o, do = oracle.registration(tg, sr, tp, sp, voxel_sort_mode=1)
    assert ok == ok_o
    assert np.array_equal(T, T_o)
    for k in [k for k in do if k in d and not k.startswith("timing")]:
        assert np.array_equal(d[k], do[k]), k


def test_extract_auto_tuning_against_libransac(faithful):
    """extract() level (plade.cpp:602-635): the halving loop over min_support on the reference's sample clouds.  The
    fixture holds the loop's trace over libransac for eight pinned time() seeds.

    What the data says: libransac's plane count at a given min_support depends on its time() seed, and so does the
    level its loop ends at (625 or 1250 on two of the four clouds).  It is also not consistent with itself: on the
    polyhedron target its run at min_support 1250 reports 7-8 planes in 8 of 8 seeds, although its own run at 625 (g8
    fixture) shows 14 planes with 1327 points or more in that cloud -- its lazily scored search stops on the
    overlook-probability bound before the faces of 1327-1542 points have been found.  The GPU search scores every
    hypothesis of a round exactly and finds the planes above min_support (all 14 there), so
      * at every level it finds at least as many planes as libransac's best draw minus one and at most a handful more
        (at 10000 and 5000 its count lies inside libransac's range on all four clouds),
      * its loop ends at a level libransac's loop ends at for some seed, or ONE halving step earlier (polyhedron target:
        14 planes >= 1250 instead of 27 planes >= 625; room target: 10 planes >= 2500 where libransac finds 7-9);
        every plane libransac reports is among the GPU's at the same min_support (test_g2_* check the sets)."""
    g = load("g2_extract.npz")
    g8, g9 = load("g8_polyhedron.npz"), load("g9_room.npz")
    pairs = {"poly": (g8["target"], g8["source"]), "room": (g9["target"], g9["source"])}
    exact_levels = 0
    for name, (tg, sr) in pairs.items():
        faithful.registration(tg, sr)   # runs extract() on both clouds; the verdict is not the point here
        st = faithful.stats()
        for side, tag in (("t", "_tgt"), ("s", "_src")):
            trace = g[f"{name}_{side}_trace"]            # seeds x 10 x (min_support, planes)
            finals = set(int(v) for v in g[f"{name}_{side}_final_min_support"])
            got_final = int(st["extract_final_min_support" + tag])
            k = 1
            while f"extract_planes_trial{k + 1}{tag}" in st:
                k += 1
            got = [int(st[f"extract_planes_trial{j}{tag}"]) for j in range(1, k + 1)]
            print(name, side, "gpu trace", got, "final", got_final, "libransac finals", sorted(finals))
            assert got_final in finals or got_final // 2 in finals, (name, side, got_final, finals)
            assert got_final == 10000 // 2 ** (len(got) - 1)
            for j, P in enumerate(got):
                ref = trace[:, j, 1]
                ref = ref[ref >= 0]
                assert len(ref) > 0
                assert ref.max() - 1 <= P <= ref.max() + 8, (name, side, j, P, ref)
                if 10000 // 2 ** j >= 5000:
                    assert ref.min() <= P <= ref.max(), (name, side, j, P, ref)
                    exact_levels += ref.min() == ref.max()
            assert got[-1] >= 10
    assert exact_levels >= 6
