"""Host-side logic that needs no GPU: the synthetic generator, PLY I/O, the CLI's argument / failure
behaviour (code/PLADE/main.cpp), and oracle properties on small inputs."""
import os
import subprocess

import numpy as np
import pytest

from plade_amd.synth import make_pair, sample_scene, planes_from_labels
from plade_amd.plyio import read_ply, write_ply

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "plade_amd", "PLADE")


def test_generator_is_deterministic_and_oriented():
    a, la = sample_scene(20000, scene_seed=3, sample_seed=4, return_labels=True)
    b, lb = sample_scene(20000, scene_seed=3, sample_seed=4, return_labels=True)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    assert np.allclose(np.linalg.norm(a[:, 3:], axis=1), 1, atol=1e-5)
    assert abs((la < 0).mean() - 0.03) < 0.005
    tg, sr, T = make_pair(20000, seed=1)
    assert abs(len(sr) / len(tg) - 1) < 0.05 and len(sr) < 1.2 * len(tg)   # no swap (plade.cpp:690)
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-9)
    co, off, idx = planes_from_labels(a, la)
    assert len(co) >= 15 and off[-1] == len(idx) and len(np.unique(idx)) == len(idx)


def test_ply_round_trip(tmp_path):
    a = sample_scene(1000, scene_seed=1, sample_seed=2)
    p = tmp_path / "c.ply"
    write_ply(str(p), a)
    assert np.array_equal(read_ply(str(p)), a)


@pytest.mark.skipif(not os.path.exists(CLI), reason="CLI not built")
def test_cli_usage_and_failure_paths(tmp_path):
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage 1" in r.stderr and "Usage 2" in r.stderr
    out = tmp_path / "res.txt"
    # wrong extension: registration() refuses before touching the GPU (plade.cpp:671-674)
    r = subprocess.run([CLI, "a.xyz", "b.xyz", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "only PLY format is accepted" in r.stderr
    assert out.read_text() == ("registration failed, an identity matrix is recorded:\n"
                               "1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n")
    # unreadable pair list / result file
    r = subprocess.run([CLI, str(tmp_path / "nope.txt"), str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "failed opening the file containing pairs" in r.stderr
    # batch mode with missing files: every pair skipped -> "registration all failed (0 pairs)"
    lst = tmp_path / "pairs.txt"
    lst.write_text("/no/such/a.ply\n/no/such/b.ply\n")
    r = subprocess.run([CLI, str(lst), str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "file doesn't exist" in r.stderr and "registration all failed" in r.stderr


def test_oracle_edge_cases(oracle):
    # empty / tiny inputs
    assert len(oracle.score_plane(np.zeros((0, 6), np.float32), None, np.array([0, 0, 1, 0], np.float32), 0.1, 0.8)) == 0
    off, nbr, d2 = oracle.match_descriptors(np.zeros((0, 8), np.float32), np.zeros((5, 8), np.float32))
    assert len(nbr) == 0 and off.tolist() == [0]
    off, nbr, d2 = oracle.match_descriptors(np.zeros((3, 8), np.float32), np.zeros((0, 8), np.float32))
    assert len(nbr) == 0 and off.tolist() == [0, 0, 0, 0]
    # parallel planes have no intersection line (|n1.n2| > 0.95, util.cpp:634)
    rc, v, p = oracle.intersection_line([0, 0, 1, -1], [0, 0.05, 0.9987, 2])
    assert rc != 0
    rc, v, p = oracle.intersection_line([0, 0, 1, -1], [1, 0, 0, -2])
    assert rc == 0 and abs(abs(v[1]) - 1) < 1e-6 and np.allclose(p, [2, 0, 1])
    # voxel grid: idempotent centroid for one point per voxel, ordered by (k, j, i)
    pts = np.array([[0.9, 0.1, 0.1], [0.1, 0.1, 0.1], [0.1, 0.9, 0.1], [0.1, 0.1, 0.9]], np.float32)
    ds = oracle.voxel_downsample(pts, 0.5, 0)
    assert np.array_equal(ds, pts[[1, 0, 2, 3]])
    # stable and std::sort orders agree to a few ulp on a real cloud
    c = sample_scene(20000, scene_seed=7, sample_seed=8)
    a, b = oracle.voxel_downsample(c, 0.2, 0), oracle.voxel_downsample(c, 0.2, 1)
    assert a.shape == b.shape and np.abs(a - b).max() < 1e-5


def test_oracle_registration_small_pair_recovers_ground_truth(oracle):
    tg, sr, Tgt, tl, sl = make_pair(30000, seed=0, n_boxes=4, return_labels=True)
    ok, T, d = oracle.registration(tg, sr, planes_from_labels(tg, tl), planes_from_labels(sr, sl))
    assert ok and np.linalg.norm(T - Tgt) < 0.01
    assert d["overlap_counts"].max() > 0.5 * len(d["src_ds"]) / 3


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not os.path.exists(CLI), reason="CLI not built")
def test_cli_malformed_ply_files_fail_cleanly(tmp_path):
    """Untrusted headers: negative / huge vertex counts, truncated data, negative list counts and very long
    ascii lines end in the reference's "loading ... failed" path (exit code 1), never in a crash or an allocation of
    the size the header claims.  None of these reaches the GPU."""
    good = tmp_path / "good.ply"
    write_ply(str(good), sample_scene(500, scene_seed=1, sample_seed=2))
    props = "".join(f"property float {p}\n" for p in ("x", "y", "z", "nx", "ny", "nz"))
    cases = {
        "negative.ply": b"ply\nformat binary_little_endian 1.0\nelement vertex -1\n" + props.encode() + b"end_header\n",
        "huge.ply": b"ply\nformat binary_little_endian 1.0\nelement vertex 4000000000\n" + props.encode() + b"end_header\n" + b"\0" * 240,
        "beyond32.ply": b"ply\nformat binary_little_endian 1.0\nelement vertex 99999999999999\n" + props.encode() + b"end_header\n",
        "truncated.ply": b"ply\nformat binary_little_endian 1.0\nelement vertex 100\n" + props.encode() + b"end_header\n" + b"\0" * 100,
        "nonsense.ply": b"ply\nformat ascii 1.0\nelement vertex 2\n" + props.encode() + b"end_header\n1 2 3 0 0 1\nfoo bar\n",
    }
    for name, blob in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        out = tmp_path / "res.txt"
        r = subprocess.run([CLI, str(p), str(good), str(out)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1, (name, r.returncode, r.stderr)
        assert "loading target point cloud failed" in r.stderr, (name, r.stderr)
        assert out.read_text().startswith("registration failed, an identity matrix is recorded:")
    # a NEGATIVE list count is not an error for the reference (rply.c:833-846: the value loop simply does not run; golden case
    # negative_list_length of tests/test_ply_reader.py): the face is empty, the vertex behind it is read
    neglist = tmp_path / "neglist.ply"
    neglist.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement face 1\nproperty list int int vertex_indices\n"
                        b"element vertex 1\n" + props.encode() + b"end_header\n" + (-5).to_bytes(4, "little", signed=True) + b"\0" * 20 + b"\0\0\x80\x3f")
    import plade_amd
    assert np.array_equal(plade_amd.read_ply(str(neglist)), np.array([[0, 0, 0, 0, 0, 1]], np.float32))
    # an ascii vertex line far longer than any fixed buffer parses (trailing blanks), and extra properties are ignored
    long_ascii = tmp_path / "long.ply"
    with open(long_ascii, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2\n" + props + "end_header\n")
        f.write("1 2 3 0 0 1" + " " * 10000 + "\n4 5 6 0 1 0\n")
    from plade_amd.plyio import read_ply as rp
    assert rp(str(long_ascii)).shape == (2, 6)


@pytest.mark.skipif(not os.path.exists(CLI), reason="CLI not built")
def test_cli_batch_blocks_are_streamed_in_input_order_without_a_gpu(tmp_path):
    """Batch mode on a box without a GPU: every registration fails (no CPU fallback), and the result file still holds
    one block per pair, in input order, with the reference's grammar (main.cpp:134-143) -- written by the ordered
    writer while four workers run.  Names that cannot be opened are reported and skipped (main.cpp:127)."""
    if not _no_gpu():
        pytest.skip("this variant is for boxes without a GPU; tests/test_gpu_configs.py covers the GPU box")
    names = []
    for k in range(6):
        p = tmp_path / f"c{k}.ply"
        write_ply(str(p), sample_scene(300, scene_seed=k, sample_seed=k + 1))
        names.append(str(p))
    lst = tmp_path / "pairs.txt"
    lst.write_text(f"{names[0]}\n{names[1]}\n\n/no/such/file.ply\n{names[2]}\n{names[3]}\n{names[4]}\n{names[5]}\n{names[0]}\n")
    out = tmp_path / "res.txt"
    r = subprocess.run([CLI, str(lst), str(out)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, PLADE_INFLIGHT="4"))
    assert r.returncode == 1 and "registration all failed (3 pairs)" in r.stderr
    assert "file doesn't exist: /no/such/file.ply" in r.stderr
    ident = "1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n"
    want = "".join(f"target: {names[2 * k]}\nsource: {names[2 * k + 1]}\nregistration failed, an identity matrix is recorded:\n{ident}\n"
                   for k in range(3))
    assert out.read_text() == want
    # the per-pair console output comes out in input order too
    tf = [l for l in r.stdout.split("\n") if l.startswith("target file: ")]
    assert tf == [f"target file: {names[2 * k]}" for k in range(3)]


def test_bench_lowers_the_in_flight_count_to_fit_a_shared_cpu_quota():
    """bench.py keeps 4 groups (of 8 pairs) in flight per GPU unless the ranks of the node share a CPU quota that their host
    threads would exceed (a throttled container loses far more than the last percent of GPU throughput): the largest count
    whose measured host load x ranks fits 90 % of the quota."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.inflight_for_budget(16, 1) == 4 and b.inflight_for_budget(256, 8) == 4 and b.inflight_for_budget(24, 8) == 4
    # r5 (lock step: 1.02 / 1.50 / 1.53 / 1.71 busy threads for 1 / 2 / 3 / 4 groups in flight): the pool's 16-CPU boxes carry 4 groups on each
    # of 8 ranks (8 x 1.71 = 13.7 <= 14.4); a CPU less and it is 3
    assert b.inflight_for_budget(16, 8) == 4
    assert b.inflight_for_budget(15, 8) == 3 and b.inflight_for_budget(17, 8) == 4 and b.inflight_for_budget(12, 8) == 1 and b.inflight_for_budget(1, 8) == 1
    busy = b.BUSY_THREADS_BY_GROUPS
    assert all(busy[a] <= busy[c] for a, c in zip(sorted(busy), sorted(busy)[1:]))


def test_host_sources_compile_against_the_real_eigen():
    """plade_host.cpp / main.cpp are written against Eigen::Matrix<float,4,4> and Eigen::Vector3f as the reference's
    plade.h / plane_extraction.h use them; this image has no Boost (hence no PCL), but Eigen is header-only and vendored by
    the reference: with -DPLADE_USE_REAL_EIGEN the host sources must compile against it (only the PCL types stay
    stand-ins).  Runs where /root/reference is mounted (the build container)."""
    import shutil
    import subprocess
    eigen = "/root/reference/code/3rd_party/eigen-3.4.0"
    if not os.path.isdir(os.path.join(eigen, "Eigen")) or not shutil.which("g++"):
        pytest.skip("no vendored Eigen here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "plade_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DPLADE_USE_REAL_EIGEN", "-I", eigen, "-I", os.path.join(root, "include"),
                        "-I", csrc, os.path.join(csrc, "plade_host.cpp"), os.path.join(csrc, "main.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def _cluster_size_cases():
    rng = np.random.default_rng(11)
    for rep in range(400):
        n = int(rng.integers(0, 700)) if rep < 250 else int(rng.integers(700, 70000))
        mode = rep % 7
        if mode == 0:     # what a registration produces: mostly clusters of one or two candidates, a few large ones
            u = rng.random(n)
            s = np.where(u < 0.6, 1, np.where(u < 0.8, 2, np.where(u < 0.9, 3, rng.integers(1, 41, n))))
        elif mode == 1:
            s = rng.integers(0, 3, n)
        elif mode == 2:
            s = rng.integers(0, 100000, n)
        elif mode == 3:
            s = np.arange(n, 0, -1)            # already in order
        elif mode == 4:
            s = np.arange(n)                   # reversed
        elif mode == 5:
            s = np.ones(n)                     # all tied
        else:                                  # organ pipe
            s = np.minimum(np.arange(n), np.arange(n)[::-1])
        yield np.asarray(s, np.float32)


def test_cluster_order_is_std_sorts_permutation_ties_included():
    """util.cpp:335-345 orders the clusters of candidate transforms with an unstable std::sort on their sizes; most sizes
    are tied, so the permutation is a property of libstdc++'s introsort.  The library's block-wise restatement
    (exact_sort.h) must reproduce it cell for cell."""
    import plade_amd
    for s in _cluster_size_cases():
        ref = plade_amd.cluster_order(s, mode=1)
        got = plade_amd.cluster_order(s, mode=0)
        assert np.array_equal(got, ref), (len(s), s[:8])
        assert np.all(np.diff(s[got]) <= 0)


def test_cluster_order_heap_sort_branch():
    """Below a recursion depth of 2 lg n introsort finishes a range with heap sort; real inputs never get there, so the
    branch is forced with small depth limits: the block-wise partition must hand the heap sort the same cells in the same
    state as the sequential one (which IS the library's loop: at the library's own limit it equals std::sort above)."""
    import plade_amd
    for k, s in enumerate(_cluster_size_cases()):
        if k % 4:
            continue
        for depth in (0, 1, 3, 6):
            a = plade_amd.cluster_order(s, mode=2, depth_limit=depth)
            b = plade_amd.cluster_order(s, mode=3, depth_limit=depth)
            assert np.array_equal(a, b), (len(s), depth)
        assert np.array_equal(plade_amd.cluster_order(s, mode=3), plade_amd.cluster_order(s, mode=1))


def test_hypot_formula_is_glibc_s(oracle):
    """closest_point_mode = svd_fp32: OpenCV's Jacobi rotation calls the C library's hypot() (lapack.cpp:579).  The GPU cannot,
    so kernel (plade_amd/csrc/k_svd.h: hypot_corrected) and oracle (orc_math.h: hypot_glibc235) evaluate one explicit formula --
    the kernel of glibc 2.35's hypot without FMA.  Here: that formula against THIS machine's libm, bit for bit, on arguments
    spread over 40 binades (the solver's are squared norms and inner products of columns of magnitude 1 ... 1e13)."""
    rng = np.random.default_rng(7)
    n = 200000
    x = (rng.random(n) - 0.5) * np.exp2(rng.uniform(-20, 45, n))
    y = (rng.random(n) - 0.5) * np.exp2(rng.uniform(-20, 45, n))
    x[:100] = 0.0
    y[100:200] = 0.0
    y[200:300] = x[200:300]
    bad = 0
    for a, b in zip(x.tolist(), y.tolist()):
        bad += oracle.L.orc_hypot(a, b) != oracle.L.orc_libm_hypot(a, b)
    assert bad == 0
