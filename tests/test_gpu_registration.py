"""Stage-by-stage parity of registration(T, target, source, target_planes, source_planes)
(plade.h:74) between the HIP pipeline and the oracle on identical planes.  Integer stages are
bit-exact; the final transform is within the north-star tolerance (1e-4 Frobenius)."""
import numpy as np
import pytest

from plade_amd.synth import make_pair, planes_from_labels
from conftest import GT_TOL, CLOSED_FORM

pytestmark = pytest.mark.gpu

EXACT = ["average_spacing", "scale", "tgt_ds", "src_ds", "tgt_bcenter", "src_bcenter", "tgt_radius", "src_radius",
         "tgt_plane_ds_offsets", "src_plane_ds_offsets", "tgt_plane_ds", "src_plane_ds",
         "tgt_plane_center_radius", "src_plane_center_radius", "tgt_plane_four", "src_plane_four",
         "tgt_lines", "src_lines", "tgt_desc", "src_desc", "match_offsets", "match_nbr", "match_dist2",
         "initial_RT", "cluster_sizes", "cluster_seeds", "plane_match_counts", "pen_tested", "pen_flags",
         "candidates", "candidate_centers", "overlap_counts", "scores", "best_index"]


def _pair(n, seed, n_boxes):
    tg, sr, Tgt, tl, sl = make_pair(n, seed=seed, n_boxes=n_boxes, return_labels=True)
    return tg, sr, Tgt, planes_from_labels(tg, tl), planes_from_labels(sr, sl)


@pytest.mark.parametrize("n,seed,n_boxes", [(30000, 0, 4), (40000, 3, 5)])
def test_registration_planes_stagewise_parity(oracle, n, seed, n_boxes):
    import plade_amd
    tg, sr, Tgt, tp, sp = _pair(n, seed, n_boxes)
    ctx = plade_amd.Context(0, dump=1)
    ok_g, T_g = ctx.registration_planes(tg, sr, tp, sp)
    dg = ctx.dump()
    ok_o, T_o, do = oracle.registration(tg, sr, tp, sp, voxel_sort_mode=1)
    assert ok_g == ok_o
    for name in EXACT:
        assert name in dg, f"GPU dump misses {name}"
        a, b = dg[name], do[name]
        assert a.shape == b.shape, (name, a.shape, b.shape)
        assert np.array_equal(a, b), (name, int((a != b).sum()), a[:4], b[:4])
    assert np.array_equal(T_g, T_o)
    # the reference-faithful (std::sort voxel order) oracle agrees within the north-star tolerance
    ok_f, T_f, df = oracle.registration(tg, sr, tp, sp, voxel_sort_mode=0)
    assert ok_f
    assert np.linalg.norm(T_g.astype(np.float64) - T_f.astype(np.float64)) <= 1e-4
    # (the NUMBER of verified candidates may differ by a few between the two voxel orders: with the reference's fp32 solves a
    #  last-bit change of a centroid moves descriptors across the match radius; the winner and the transform above do not move)
    assert abs(len(dg["overlap_counts"]) - len(df["overlap_counts"])) <= 8
    # and the registration is right: close to the generator's ground truth
    assert np.linalg.norm(T_g.astype(np.float64) - Tgt) < GT_TOL
    ctx.close()


def test_registration_planes_failure_is_reported(oracle):
    """Too few consistent planes: the reference returns false ("no matched result found")."""
    import plade_amd
    tg, sr, Tgt, tp, sp = _pair(20000, 1, 3)
    # keep two parallel planes only -> no intersection lines -> no candidates
    keep = [0, 1]
    def sub(pl):
        co, off, idx = pl
        parts = [idx[off[i]:off[i + 1]] for i in keep]
        return co[keep], np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int32), np.concatenate(parts)
    ctx = plade_amd.Context(0)
    ok_g, T_g = ctx.registration_planes(tg, sr, sub(tp), sub(sp))
    ok_o, T_o, _ = oracle.registration(tg, sr, sub(tp), sub(sp), voxel_sort_mode=1)
    assert not ok_g and not ok_o
    assert np.array_equal(T_g, np.eye(4, dtype=np.float32))
    ctx.close()


def test_obb_stage_against_pcl_summation_order(oracle):
    """k_obb_units adds the points of a box in a lane-strided order, PCL (compute3DCentroid / computeCovarianceMatrixNormalized,
    centroid.hpp:79-121,250-259) one after the other in fp32; the every-intermediate tests compare the GPU with the oracle
    run in the GPU's order (sum_mode 1).  This test bounds the deviation from PCL's own order per unit: the GPU's box centre,
    radius and projected corners against the oracle in sum_mode 0 on the same downsampled clouds (whole cloud and every
    plane), relative to the box size."""
    import plade_amd
    tg, sr, Tgt, tp, sp = _pair(60000, 2, 5)
    ctx = plade_amd.Context(0, dump=1)
    ok, T = ctx.registration_planes(tg, sr, tp, sp)
    d = ctx.dump()
    ctx.close()
    assert ok
    worst = 0.0
    for side, planes in (("tgt", tp), ("src", sp)):
        ds = d[f"{side}_ds"].reshape(-1, 3)
        _, c0, whd0, _ = oracle.bounding_box(ds, sum_mode=0)
        size = float(max(whd0))
        assert np.abs(d[f"{side}_bcenter"] - c0).max() <= 1e-5 * size
        assert abs(float(d[f"{side}_radius"][0]) - size / 2) <= 1e-5 * size
        worst = max(worst, float(np.abs(d[f"{side}_bcenter"] - c0).max()) / size)
        off = d[f"{side}_plane_ds_offsets"]
        pds = d[f"{side}_plane_ds"].reshape(-1, 3)
        four = d[f"{side}_plane_four"].reshape(-1, 4, 3)
        pcr = d[f"{side}_plane_center_radius"].reshape(-1, 4)
        for i in range(len(off) - 1):
            pts = pds[off[i]:off[i + 1]]
            if len(pts) < 3:
                continue
            _, c, whd, corners = oracle.bounding_box(pts, sum_mode=0)
            n4 = planes[0][i].astype(np.float64)
            cor = corners.reshape(8, 3)[:4].astype(np.float64)
            proj = cor - (cor @ n4[:3] + n4[3])[:, None] * n4[:3]          # ProjectPoints2Plane (util.h:292-340)
            sz = float(max(whd))
            # the principal axes of a near-square face are ill-conditioned (two nearly equal eigenvalues): compare the
            # rectangle as a whole -- centre and half diagonal -- tightly, the corners only where the axes are well separated
            assert np.abs(pcr[i, :3] - (proj[0] + proj[2]) / 2).max() <= 2e-5 * sz + 1e-6, (side, i)
            assert abs(pcr[i, 3] - np.linalg.norm(proj[0] - proj[2]) / 2) <= 2e-5 * sz + 1e-6, (side, i)
            ev = np.sort(np.asarray(whd))
            if ev[2] - ev[1] > 0.05 * ev[2] and ev[1] - ev[0] > 0.05 * ev[2]:
                assert np.abs(four[i] - proj).max() <= 1e-4 * sz, (side, i)
            worst = max(worst, float(np.abs(pcr[i, :3] - (proj[0] + proj[2]) / 2).max()) / sz)
    print("largest relative deviation of a box centre from PCL's summation order:", worst)
