"""Groups of up to eight registrations per launch sequence (plade_registration_pairs / _dev, batch mode of
code/PLADE/main.cpp:122-148 taken several pairs at a time): every pair of a group must come out bit for bit as the same
pair registered alone -- extracted planes, every dumped intermediate of the registration, the final transform -- whatever
its partner is, in either position of the group, through host pointers (with and without the prefetch of the next group)
and on resident clouds."""
import numpy as np
import pytest

import plade_amd
from plade_amd.synth import make_pair

pytestmark = pytest.mark.gpu

PLANE_KEYS = ["tgt_planes", "tgt_plane_offsets", "tgt_plane_idx", "src_planes", "src_plane_offsets", "src_plane_idx"]
STAGE_KEYS = ["average_spacing", "tgt_ds", "src_ds", "tgt_plane_ds", "src_plane_ds", "tgt_desc", "src_desc", "match_nbr", "match_dist2",
              "initial_RT", "cluster_sizes", "cluster_seeds", "plane_match_counts", "pen_tested", "pen_flags", "candidates",
              "overlap_counts", "scores", "best_index"]


@pytest.fixture(scope="module")
def scenes():
    out = []
    for n, seed in ((200000, 3), (120000, 7), (200000, 11)):
        tg, sr, Tgt = make_pair(n, seed=seed)
        out.append((np.ascontiguousarray(tg, np.float32), np.ascontiguousarray(sr, np.float32), Tgt))
    return out


@pytest.fixture(scope="module")
def alone(scenes):
    """every pair registered alone, with all intermediates"""
    c = plade_amd.Context(0, orient_normals=1, dump=1)
    out = []
    for tg, sr, _ in scenes:
        ok, T = c.registration(tg, sr)
        out.append((ok, T, c.dump(), c.stats()))
    c.close()
    return out


def _same(d, ref, keys):
    for k in keys:
        assert k in d and k in ref, k
        assert d[k].shape == ref[k].shape and np.array_equal(d[k], ref[k]), k


@pytest.mark.parametrize("a,b", [(0, 1), (1, 0), (2, 2), (0, 2)])
def test_group_of_two_equals_the_pairs_alone(scenes, alone, a, b):
    c = plade_amd.Context(0, orient_normals=1, dump=1)
    res = c.registration_pairs([(scenes[a][0], scenes[a][1]), (scenes[b][0], scenes[b][1])])
    for pos, i in enumerate((a, b)):
        ok, T = res[pos]
        assert ok == alone[i][0] and ok
        assert np.array_equal(T, alone[i][1])
        d = c.dump(pair=pos)
        _same(d, alone[i][2], PLANE_KEYS + STAGE_KEYS)
        assert np.linalg.norm(T.astype(np.float64) - scenes[i][2]) < 5e-2
        st = c.stats(pair=pos)
        for k in ("n_planes_tgt", "n_planes_src", "n_matches", "n_clusters", "n_candidates_verified", "extract_planes_trial1_tgt",
                  "extract_planes_trial1_src", "extract_final_min_support_tgt", "extract_final_min_support_src"):
            assert st[k] == alone[i][3][k], (k, st[k], alone[i][3][k])
    c.close()


@pytest.mark.parametrize("members", [(0, 1, 2), (2, 1, 0, 1), (1, 1, 1, 1), (0, 1, 2, 1, 0), (2, 0, 1, 1, 2, 0, 1), (0, 1, 2, 2, 1, 0, 0, 1)])
def test_groups_of_three_to_eight(scenes, alone, members):
    """PLADE_GROUP_MAX = 8 pairs (sixteen clouds) per extraction sequence: same bits as the pairs alone, the planes included."""
    c = plade_amd.Context(0, orient_normals=1, dump=1)
    res = c.registration_pairs([(scenes[i][0], scenes[i][1]) for i in members])
    for pos, i in enumerate(members):
        ok, T = res[pos]
        assert ok and np.array_equal(T, alone[i][1])
        _same(c.dump(pair=pos), alone[i][2], PLANE_KEYS + ["overlap_counts", "scores", "match_nbr"])
    c.close()


def test_group_of_one_is_the_plain_call(scenes, alone):
    c = plade_amd.Context(0, orient_normals=1)
    (ok, T), = c.registration_pairs([(scenes[1][0], scenes[1][1])])
    assert ok and np.array_equal(T, alone[1][1])
    ct, cs = c.upload(scenes[1][0]), c.upload(scenes[1][1])
    (ok2, T2), = c.registration_pairs_dev([(ct, cs)])
    assert ok2 and np.array_equal(T2, alone[1][1])
    ct.free(); cs.free()
    c.close()


def test_groups_in_batch_mode_with_prefetch_and_on_resident_clouds(scenes, alone):
    """A list of five pairs taken two at a time, the next group announced to every call (its upload runs under the current
    group's kernels), the last group a single pair; then the same groups on resident clouds."""
    order = [0, 1, 2, 1, 0]
    c = plade_amd.Context(0, orient_normals=1, host_wait=1)
    for tg, sr, _ in scenes:
        c.pin(tg); c.pin(sr)
    groups = [order[i:i + 2] for i in range(0, len(order), 2)]
    got = []
    for gi, g in enumerate(groups):
        nxt = [(scenes[i][0], scenes[i][1]) for i in groups[gi + 1]] if gi + 1 < len(groups) else None
        got += c.registration_pairs([(scenes[i][0], scenes[i][1]) for i in g], nxt)
        if gi > 0:
            assert c.stats().get("upload_prefetched", 0) == 1
    for (ok, T), i in zip(got, order):
        assert ok and np.array_equal(T, alone[i][1])
    # a call that is handed other clouds than the announced ones uploads them itself
    c.registration_pairs([(scenes[0][0], scenes[0][1])], [(scenes[1][0], scenes[1][1]), (scenes[2][0], scenes[2][1])])
    (ok, T), = c.registration_pairs([(scenes[2][0], scenes[2][1])])
    assert ok and np.array_equal(T, alone[2][1]) and c.stats().get("upload_prefetched", 0) == 0
    for tg, sr, _ in scenes:
        c.unpin(tg); c.unpin(sr)
    res = [(c.upload(tg), c.upload(sr)) for tg, sr, _ in scenes]
    for g in groups:
        for (ok, T), i in zip(c.registration_pairs_dev([res[i] for i in g]), g):
            assert ok and np.array_equal(T, alone[i][1])
    for ct, cs in res:
        ct.free(); cs.free()
    c.close()


def test_a_failing_pair_does_not_touch_its_partner(scenes, alone):
    """One pair of the group cannot be registered (a cloud without planes: the reference returns false after its halving
    loop, plade.cpp:646-657): its status says so, its transform is the identity, its partner's result is untouched --
    in both positions."""
    rng = np.random.default_rng(5)
    blob = np.concatenate([rng.normal(0, 1, (50000, 3)), rng.normal(0, 1, (50000, 3))], 1).astype(np.float32)
    blob[:, 3:] /= np.linalg.norm(blob[:, 3:], axis=1, keepdims=True)
    c = plade_amd.Context(0, orient_normals=1)
    for pos in (0, 1):
        prs = [(scenes[1][0], scenes[1][1]), (scenes[1][0], scenes[1][1])]
        prs[pos] = (blob, blob)
        res = c.registration_pairs(prs)
        assert not res[pos][0] and np.array_equal(res[pos][1], np.eye(4, dtype=np.float32))
        assert "too few planes" in c.pair_error(pos)
        assert res[1 - pos][0] and np.array_equal(res[1 - pos][1], alone[1][1])
    c.close()


def test_bad_group_arguments_are_refused(scenes):
    c = plade_amd.Context(0, orient_normals=1)
    pr = (scenes[1][0], scenes[1][1])
    with pytest.raises((plade_amd.PladeError, ValueError)):
        c.registration_pairs([pr] * 9)
    with pytest.raises((plade_amd.PladeError, ValueError)):
        c.registration_pairs([])
    # the library reads N x 6 floats behind every pointer: anything else is refused before the call (python -O strips asserts)
    for bad in (pr[0][:, :3].copy(), pr[0].reshape(-1), pr[0].astype(np.float64), pr[0][::2]):
        with pytest.raises(ValueError):
            c.registration_pairs([(bad, pr[1])])
        with pytest.raises(ValueError):
            c.registration_pairs([pr], [(pr[0], bad)])
    c.close()


@pytest.mark.parametrize("kind", ["nan", "inf", "no_extent"])
@pytest.mark.parametrize("pos", [0, 2, 3])
def test_a_malformed_cloud_fails_its_pair_only(scenes, alone, kind, pos):
    """A cloud the path refuses -- a non-finite coordinate (the upload's validation) or a bounding box without extent (the
    extraction's) -- raises inside the launch sequence the whole group shares.  The reference's loop (main.cpp:122-148) fails
    that pair only; so does the group call: the pair's status carries the error code and its context the message, its
    transform is the identity, and the three healthy neighbours return the bits of the pairs alone (advisor r4) -- through
    host pointers with a prefetch announced, and on resident clouds."""
    bad = scenes[1][1].copy()
    if kind == "nan":
        bad[1234, 1] = np.nan
    elif kind == "inf":
        bad[77, 0] = np.inf
    else:
        bad[:, :3] = bad[0, :3]          # every point in one place
    order = [0, 1, 2, 1]
    prs = [(scenes[k][0], scenes[k][1]) for k in order]
    prs[pos] = (prs[pos][0], bad)
    c = plade_amd.Context(0, orient_normals=1)
    for rep in range(2):                 # the second call takes the prefetched clouds of the first
        res = c.registration_pairs(prs, prs if rep == 0 else None, raise_on_error=False)
        for q, (st, T) in enumerate(res):
            if q == pos:
                assert st not in (plade_amd.PLADE_OK,), (kind, st)
                assert np.array_equal(T, np.eye(4, dtype=np.float32))
                if kind != "no_extent":
                    assert st == plade_amd.PLADE_EINVAL and "finite" in c.pair_error(q), c.pair_error(q)
            else:
                assert st == plade_amd.PLADE_OK and np.array_equal(T, alone[order[q]][1]), (kind, pos, q)
        assert c.stats().get("group_fallback_pair_by_pair", 0) == 1
    if kind == "no_extent":              # resident clouds: the upload accepts it, the extraction refuses it
        cl = [(c.upload(a), c.upload(b)) for a, b in prs]
        res = c.registration_pairs_dev(cl, raise_on_error=False)
        for q, (st, T) in enumerate(res):
            if q == pos:
                assert st != plade_amd.PLADE_OK
            else:
                assert st == plade_amd.PLADE_OK and np.array_equal(T, alone[order[q]][1])
        for a, b in cl:
            a.free(); b.free()
    # the context is as good as new
    ok, T = c.registration_pairs(prs[:1] if pos else prs[1:2])[0]
    assert ok
    c.close()


@pytest.mark.parametrize("mode", [2, 6])
def test_profiled_modes_time_the_scan_kernels_and_change_nothing(scenes, alone, mode):
    """params.dump & 2 times the scan kernels of the extraction loop (bench.py's roofline leg): mode 2 launches the loop
    kernel by kernel with HIP events around the scans, mode 6 keeps the iteration's captured graph and has the kernels
    stamp the device clock themselves.  Both report one record per working launch with the same algorithmic bytes
    (28 B per point and launch + the mask bytes), and the registration is the unprofiled one bit for bit."""
    c = plade_amd.Context(0, orient_normals=1, dump=mode)
    res = c.registration_pairs([(scenes[0][0], scenes[0][1]), (scenes[1][0], scenes[1][1])])
    st = c.stats()
    for pos, i in enumerate((0, 1)):
        assert res[pos][0] and np.array_equal(res[pos][1], alone[i][1])
    launches, secs, byts = st["k_score_mark_clock_launches"], st["k_score_mark_clock_seconds"], st["k_score_mark_clock_bytes"]
    assert launches >= 4 and launches == st["k_score_mark_launches"] and byts == st["k_score_mark_bytes"]
    # every working launch reads at least one cloud and at most the four of the group
    n_min, n_max = 120000, 2 * 200000 + 2 * 120000
    assert 28.0 * n_min * launches <= byts <= 28.5 * n_max * launches
    assert 1e-6 * launches < secs < 5e-3 * launches
    c.close()
    # the same byte count in both modes: it is a property of the extraction, not of how it was launched
    key = "_profiled_bytes"
    seen = globals().setdefault(key, {})
    seen[mode] = (launches, byts)
    if len(seen) == 2:
        assert seen[2] == seen[6]


def test_deferred_readbacks_keep_their_contract(scenes, alone, monkeypatch):
    """ctx.d2h() only notes a range; the next wait delivers it, so the range must not change in between.  Under
    PLADE_DEBUG_READS=1 every noted range is also copied at once and compared with what the wait delivers: a group of
    registrations (every stage's read-backs, both host paths) must pass the check and return the usual bits."""
    monkeypatch.setenv("PLADE_DEBUG_READS", "1")
    c = plade_amd.Context(0, orient_normals=1, dump=1)
    res = c.registration_pairs([(scenes[0][0], scenes[0][1]), (scenes[2][0], scenes[2][1])])
    for pos, i in enumerate((0, 2)):
        assert res[pos][0] and np.array_equal(res[pos][1], alone[i][1])
    ok, T = c.registration(scenes[1][0], scenes[1][1])
    assert ok and np.array_equal(T, alone[1][1])
    c.close()


@pytest.mark.parametrize("budget,parts", [(900000, 3), (500000, 5), (0, 1)])
def test_a_group_over_the_point_budget_is_registered_in_parts(scenes, alone, budget, parts):
    """plade_params.group_max_points bounds what one extraction sequence serves at once (its work area takes ~0.9 KB of HBM per
    point): a group that holds more is registered in consecutive parts -- pair j still on peer context j -- and every pair
    comes out as the pair alone, bit for bit, whatever the partition."""
    members = (0, 1, 2, 1, 0)           # 400k / 240k / 400k / 240k / 400k points per pair
    c = plade_amd.Context(0, orient_normals=1, dump=1, group_max_points=budget)
    res = c.registration_pairs([(scenes[i][0], scenes[i][1]) for i in members])
    assert c.stats()["group_parts"] == parts
    for pos, i in enumerate(members):
        ok, T = res[pos]
        assert ok and np.array_equal(T, alone[i][1])
        _same(c.dump(pair=pos), alone[i][2], PLANE_KEYS + ["overlap_counts", "scores", "match_nbr"])
    c.close()


@pytest.mark.parametrize("sides", [1, 2])
def test_sides_of_a_pair_side_by_side_or_one_after_the_other(scenes, alone, sides):
    """plade_params.prepare_sides: the two clouds of a pair are prepared side by side (source on the auxiliary stream and a
    helper thread) or one after the other on the context's stream; alone and inside a group, every intermediate and the
    transform are the same bits (`alone` was registered with the default: side by side for a single pair)."""
    c = plade_amd.Context(0, orient_normals=1, dump=1, prepare_sides=sides)
    ok, T = c.registration(scenes[0][0], scenes[0][1])
    assert ok and np.array_equal(T, alone[0][1])
    _same(c.dump(), alone[0][2], PLANE_KEYS + STAGE_KEYS)
    res = c.registration_pairs([(scenes[1][0], scenes[1][1]), (scenes[2][0], scenes[2][1])])
    for pos, i in enumerate((1, 2)):
        assert res[pos][0] and np.array_equal(res[pos][1], alone[i][1])
        _same(c.dump(pair=pos), alone[i][2], PLANE_KEYS + STAGE_KEYS)
    c.close()


def test_lock_step_merges_the_pairs_launches_and_changes_nothing(scenes, alone, tmp_path):
    """Behind the plane extraction the pairs of a group run in LOCK STEP (plade_amd/csrc/launch.h, combiner.hip): what the eight
    host threads launch is collected per pair and issued merged -- one launch per kernel for all pairs, one copy kernel for
    their uploads, one hand-over kernel and one wait for their read-backs.  The statistics of the call say how much was merged;
    PLADE_NO_LOCKSTEP=1 (a process-wide A/B switch, hence the subprocess) runs every pair's tail on its own stream as round 4
    did: both return the bits of the pairs alone."""
    import os
    import subprocess
    import sys
    order = [0, 1, 2, 1, 0, 2, 1, 0]
    prs = [(scenes[k][0], scenes[k][1]) for k in order]
    c = plade_amd.Context(0, orient_normals=1, dump=1)
    res = c.registration_pairs(prs)
    st = c.stats()
    for q, (ok, T) in enumerate(res):
        assert ok == alone[order[q]][0] and np.array_equal(T, alone[order[q]][1]), q
        _same(c.dump(pair=q), alone[order[q]][2], PLANE_KEYS + STAGE_KEYS)
    c.close()
    asked, issued, waits = st["lockstep_operations_asked"], st["lockstep_commands_issued"], st["lockstep_group_waits"]
    print(f"lock step, 8 pairs: {asked:.0f} launches / copies / fills asked for by the pairs -> {issued:.0f} commands on the stream, {waits:.0f} group waits")
    assert asked > 300 and issued < 0.45 * asked and waits <= 40
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import plade_amd\n"
        "from plade_amd.synth import make_pair\n"
        "sc = [tuple(np.ascontiguousarray(a, np.float32) for a in make_pair(n, seed=s)[:2]) for n, s in ((200000, 3), (120000, 7), (200000, 11))]\n"
        "c = plade_amd.Context(0, orient_normals=1)\n"
        "res = c.registration_pairs([sc[k] for k in %r])\n"
        "assert 'lockstep_operations_asked' not in c.stats()\n"
        "np.save(%r, np.stack([T for ok, T in res]))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), order, str(tmp_path / "T.npy"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PLADE_NO_LOCKSTEP="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    Tn = np.load(tmp_path / "T.npy")
    for q in range(len(order)):
        assert np.array_equal(Tn[q], alone[order[q]][1]), q
