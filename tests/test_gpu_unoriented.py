"""The "unoriented normals" mode (README.md:109-110; plade_params.unoriented_normals): every target plane takes part with
both orientations, so a pair registers whatever the signs of the extracted plane normals are.  The oracle restates the
mode as a transformation of the target plane set (oracle.mirror_planes); the GPU path must equal it bit for bit."""
import numpy as np
import pytest

import plade_amd
from plade_amd.synth import make_pair, planes_from_labels
from conftest import GT_TOL, CLOSED_FORM

pytestmark = pytest.mark.gpu


def _flip_source_normals_per_plane(sr, labels, seed, fraction):
    """The source's point normals turned around on a random part of its planes (scanners / normal estimation without a
    consistent viewpoint give exactly this): the extracted planes, oriented like their inliers' normals, inherit it."""
    rng = np.random.default_rng(seed)
    out = sr.copy()
    faces = np.unique(labels[labels >= 0])
    flipped = faces[rng.random(len(faces)) < fraction]
    sel = np.isin(labels, flipped)
    out[sel, 3:] = -out[sel, 3:]
    return out, len(flipped), len(faces)


def test_planes_given_boundary_equals_oracle_with_mirrored_target(oracle):
    tg, sr, Tgt, tl, sl = make_pair(60000, seed=1, n_boxes=5, return_labels=True)
    tp, sp = planes_from_labels(tg, tl), planes_from_labels(sr, sl)
    rng = np.random.default_rng(3)
    sgn = np.where(rng.random(len(sp[0])) < 0.5, -1.0, 1.0).astype(np.float32)
    sp_flipped = (sp[0] * sgn[:, None], sp[1], sp[2])            # source plane signs at random
    ctx = plade_amd.Context(0, dump=1, unoriented_normals=1)
    ok, T = ctx.registration_planes(tg, sr, tp, sp_flipped)
    d = ctx.dump()
    ok_o, T_o, do = oracle.registration(tg, sr, tp, sp_flipped, voxel_sort_mode=1, unoriented_normals=True)
    assert ok and ok_o and np.array_equal(T, T_o)
    common = [k for k in do if k in d and not k.startswith("timing")]
    assert len(common) >= 30
    for k in common:
        assert np.asarray(d[k]).shape == np.asarray(do[k]).shape and np.array_equal(d[k], do[k]), k
    assert np.linalg.norm(T.astype(np.float64) - Tgt) < GT_TOL
    # the mirrored target really is twice the planes, and its descriptor table holds every sign pattern
    assert len(d["tgt_plane_center_radius"]) == 2 * 4 * len(tp[0])
    ctx.set_params(unoriented_normals=0)
    ok0, T0 = ctx.registration_planes(tg, sr, tp, sp)            # consistent signs: same answer without the mode
    assert ok0 and np.linalg.norm(T0.astype(np.float64) - T.astype(np.float64)) < GT_TOL / 4   # (another descriptor table: another draw of the ill-conditioned solves)
    ctx.close()


@pytest.mark.parametrize("seed,fraction", [(0, 1.0), (4, 0.5), (2, 0.7)])
def test_pair_with_flipped_source_normals_registers_in_this_mode(oracle, seed, fraction):
    tg, sr, Tgt, tl, sl = make_pair(100000, seed=seed, return_labels=True)
    sr_f, n_flip, n_faces = _flip_source_normals_per_plane(sr, sl, seed, fraction)
    assert 3 <= n_flip <= n_faces
    if fraction == 1.0:
        # every source plane comes out with the opposite orientation: nothing the oriented pipeline can match
        # (with only part of the planes flipped the consistent remainder may still carry a registration)
        plain = plade_amd.Context(0, orient_normals=1)
        ok_p, T_p = plain.registration(tg, sr_f)
        plain.close()
        assert (not ok_p) or np.linalg.norm(T_p.astype(np.float64) - Tgt) > 0.1
    ctx = plade_amd.Context(0, orient_normals=1, unoriented_normals=1, dump=1)
    ok, T = ctx.registration(tg, sr_f)
    d = ctx.dump()
    assert ok and np.linalg.norm(T.astype(np.float64) - Tgt) < GT_TOL
    tp = (d["tgt_planes"].reshape(-1, 4), d["tgt_plane_offsets"], d["tgt_plane_idx"])
    sp = (d["src_planes"].reshape(-1, 4), d["src_plane_offsets"], d["src_plane_idx"])
    ok_o, T_o, do = oracle.registration(tg, sr_f, tp, sp, voxel_sort_mode=1, unoriented_normals=True)
    assert ok_o and np.array_equal(T, T_o)
    for k in ("overlap_counts", "pen_flags", "plane_match_counts", "match_nbr", "tgt_desc", "src_desc"):
        assert np.array_equal(d[k], do[k]), k
    ctx.close()
