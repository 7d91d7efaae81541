"""The reference's user-facing surface on the GPU: the PLADE command line (code/PLADE/main.cpp) and the
four C++ registration() overloads of plade.h, built from plade_amd/csrc with g++ and run as a user would."""
import os
import subprocess

import numpy as np
import pytest

import plade_amd
from plade_amd.plyio import write_ply
from plade_amd.synth import make_pair
from conftest import GT_TOL, CLOSED_FORM

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "plade_amd", "PLADE")
ORIENTED_ENV = dict(os.environ, PLADE_ORIENT_NORMALS="1")   # synthetic Manhattan scenes (tests/conftest.py)


def parse_results(path):
    """result-file grammar of main.cpp:84-86,136-140"""
    blocks, cur = [], None
    for line in open(path).read().split("\n"):
        if line.startswith("target: "):
            cur = {"target": line[8:], "rows": [], "failed": False}
            blocks.append(cur)
        elif line.startswith("source: "):
            cur["source"] = line[8:]
        elif line.startswith("registration failed"):
            cur["failed"] = True
        elif line.startswith("transformation:") or not line.strip():
            continue
        else:
            cur["rows"].append([float(x) for x in line.split()])
    for b in blocks:
        b["T"] = np.array(b["rows"], np.float64)
        assert b["T"].shape == (4, 4)
    return blocks


@pytest.fixture(scope="module")
def ply_pairs(tmp_path_factory):
    d = tmp_path_factory.mktemp("ply")
    out = []
    for s in (3, 4, 5):
        tg, sr, Tgt = make_pair(80000, seed=s)
        pt, ps = str(d / f"t{s}.ply"), str(d / f"s{s}.ply")
        write_ply(pt, tg)
        write_ply(ps, sr)
        out.append((pt, ps, tg, sr, Tgt))
    return d, out


def test_cli_single_pair_matches_the_library(ply_pairs, ctx):
    d, pairs = ply_pairs
    pt, ps, tg, sr, Tgt = pairs[0]
    res = str(d / "one.txt")
    r = subprocess.run([CLI, pt, ps, res], capture_output=True, text=True, timeout=300, env=ORIENTED_ENV)
    assert r.returncode == 0, r.stderr
    (b,) = parse_results(res)
    assert b["target"] == pt and b["source"] == ps and not b["failed"]
    ok, T = ctx.registration(tg, sr)
    assert ok
    # Eigen's default stream format prints 6 significant digits
    assert np.allclose(b["T"], T, rtol=2e-5, atol=2e-6)
    assert np.linalg.norm(b["T"] - Tgt) < GT_TOL


def test_cli_batch_mode_in_flight_keeps_input_order(ply_pairs, ctx):
    d, pairs = ply_pairs
    lst = d / "file_pairs.txt"
    lst.write_text("".join(f"{pt}\n{ps}\n" for (pt, ps, *_r) in pairs))
    res = str(d / "batch.txt")
    env = dict(ORIENTED_ENV, PLADE_INFLIGHT="3", PLADE_GPUS="1")
    r = subprocess.run([CLI, str(lst), res], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    blocks = parse_results(res)
    assert [b["target"] for b in blocks] == [p[0] for p in pairs]
    for b, (pt, ps, tg, sr, Tgt) in zip(blocks, pairs):
        ok, T = ctx.registration(tg, sr)
        assert ok and not b["failed"]
        assert np.allclose(b["T"], T, rtol=2e-5, atol=2e-6)


def test_cli_batch_in_groups_and_on_two_logical_gpus(ply_pairs, ctx):
    """Batch mode takes PLADE_GROUP consecutive pairs of the list per call (one group: registration_group, plade.h); every
    block must be what the single-pair path writes, whatever the grouping -- groups of 1, 2, 3 (a ragged last group), 4 and the default 8 (here one group of all seven pairs) --
    and with PLADE_GPUS=2 worker sets, here both mapped onto the one GPU of the box (PLADE_GPU_MAP=0,0: the per-device
    branch of main.cpp runs with two device numbers)."""
    d, pairs = ply_pairs
    order = [0, 1, 2, 1, 0, 2, 2]
    lst = d / "file_pairs_groups.txt"
    lst.write_text("".join(f"{pairs[i][0]}\n{pairs[i][1]}\n" for i in order))
    want = {}
    for i in set(order):
        ok, T = ctx.registration(pairs[i][2], pairs[i][3])
        assert ok
        want[i] = T
    texts = {}
    for tag, extra in (("g1", {"PLADE_GROUP": "1", "PLADE_INFLIGHT": "2"}), ("g2", {"PLADE_GROUP": "2", "PLADE_INFLIGHT": "2"}),
                       ("g3", {"PLADE_GROUP": "3", "PLADE_INFLIGHT": "1"}), ("g4", {"PLADE_GROUP": "4", "PLADE_INFLIGHT": "2"}),
                       ("g8", {"PLADE_INFLIGHT": "2"}),
                       ("two_gpus", {"PLADE_GPUS": "2", "PLADE_GPU_MAP": "0,0", "PLADE_INFLIGHT": "1", "PLADE_GROUP": "2"})):
        res = str(d / f"batch_{tag}.txt")
        r = subprocess.run([CLI, str(lst), res], capture_output=True, text=True, timeout=600, env=dict(ORIENTED_ENV, **extra))
        assert r.returncode == 0, (tag, r.stderr)
        blocks = parse_results(res)
        assert [b["target"] for b in blocks] == [pairs[i][0] for i in order], tag
        for b, i in zip(blocks, order):
            assert not b["failed"] and np.allclose(b["T"], want[i], rtol=2e-5, atol=2e-6), (tag, i)
        texts[tag] = open(res).read()
        # the console reads like the sequential run: every pair's messages, in input order
        tf = [l for l in r.stdout.split("\n") if l.startswith("target file: ")]
        assert tf == [f"target file: {pairs[i][0]}" for i in order], tag
        assert r.stdout.count("done. time: ") == len(order)
    assert len(set(texts.values())) == 1          # the same file, byte for byte


def test_cli_group_with_a_broken_pair(ply_pairs, tmp_path):
    """A pair of a group that cannot be loaded (not a PLY) or registered fails alone: its block records the identity, its
    partners' blocks are untouched (main.cpp:141-146 per pair)."""
    d, pairs = ply_pairs
    bad = tmp_path / "broken.ply"
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nend_header\n0\n1\n2\n")
    lst = tmp_path / "pairs.txt"
    lst.write_text(f"{pairs[0][0]}\n{pairs[0][1]}\n{bad}\n{pairs[1][1]}\n{pairs[1][0]}\n{pairs[1][1]}\n")
    res = str(tmp_path / "r.txt")
    r = subprocess.run([CLI, str(lst), res], capture_output=True, text=True, timeout=600, env=dict(ORIENTED_ENV, PLADE_GROUP="3"))
    assert r.returncode == 0, r.stderr
    blocks = parse_results(res)
    assert [b["failed"] for b in blocks] == [False, True, False]
    assert np.array_equal(blocks[1]["T"], np.eye(4))
    assert "registration of 1 (out of 3) pairs failed" in r.stderr


def test_cli_ascii_ply_and_swap_of_a_larger_source(tmp_path, ctx):
    """ascii PLY input (ply_reader.cpp) and the |source| >= 1.2 |target| swap + inverse of plade.cpp:690-703."""
    tg, sr, Tgt = make_pair(60000, seed=6)
    tg2 = make_pair(60000, seed=6)[0].copy()    # the full scene twice (second sample shifted by < 1 mm): 2x the points
    tg2[:, :3] += np.float32(5e-4)
    big_src, small_tgt = np.concatenate([tg, tg2]), sr   # register the full scene onto the cropped one
    T_expected = np.linalg.inv(Tgt)
    pt, ps = str(tmp_path / "t.ply"), str(tmp_path / "s.ply")
    with open(pt, "w") as f:                     # ascii target
        f.write(f"ply\nformat ascii 1.0\nelement vertex {len(small_tgt)}\n" +
                "".join(f"property float {p}\n" for p in ("x", "y", "z", "nx", "ny", "nz")) + "end_header\n")
        np.savetxt(f, small_tgt, fmt="%.9g")
    write_ply(ps, big_src)
    assert len(big_src) >= 1.2 * len(small_tgt)
    res = str(tmp_path / "r.txt")
    r = subprocess.run([CLI, pt, ps, res], capture_output=True, text=True, timeout=300, env=ORIENTED_ENV)
    assert r.returncode == 0, r.stderr
    (b,) = parse_results(res)
    assert not b["failed"] and np.linalg.norm(b["T"] - T_expected) < GT_TOL


def test_cxx_api_four_overloads(ply_pairs, tmp_path, ctx):
    d, pairs = ply_pairs
    pt, ps, tg, sr, Tgt = pairs[1]
    exe = str(tmp_path / "api_harness")
    csrc = os.path.join(ROOT, "plade_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                           os.path.join(ROOT, "tests", "cxx", "api_harness.cpp"), os.path.join(csrc, "plade_host.cpp"),
                           os.path.join(csrc, "ply_reader.cpp"), "-o", exe, "-L", os.path.join(ROOT, "plade_amd"),
                           "-lplade_hip", "-Wl,-rpath," + os.path.join(ROOT, "plade_amd")])
    r = subprocess.run([exe, pt, ps], capture_output=True, text=True, timeout=600, env=ORIENTED_ENV)
    assert r.returncode == 0, r.stderr
    out, cur = {}, None
    for l in r.stdout.split("\n"):
        w = l.split()
        if w and w[0][1:] in ("files", "clouds", "minsupport", "planes", "noplanes") and w[0][0] == "@":
            cur = w[0][1:]
            out[cur] = [int(w[1]), []]
        elif w and w[0] == "@row":
            out[cur][1].append([float(x) for x in w[1:]])
    out = {k: (v[0], np.array(v[1])) for k, v in out.items()}
    ok, T = ctx.registration(tg, sr)
    assert ok
    assert out["files"][0] == 1 and np.array_equal(out["files"][1].astype(np.float32), T)
    assert out["clouds"][0] == 1 and np.array_equal(out["clouds"][1].astype(np.float32), T)
    ok2, T2 = ctx.registration_minsupport(tg, sr, 1500, 1500)
    assert out["minsupport"][0] == int(ok2) and np.array_equal(out["minsupport"][1].astype(np.float32), T2)
    assert out["planes"][0] == 1 and np.linalg.norm(out["planes"][1] - Tgt) < GT_TOL
    assert out["noplanes"][0] == 0 and np.array_equal(out["noplanes"][1], np.eye(4))


def test_cli_shipped_default_is_the_reference_faithful_mode(tmp_path):
    """The CLI without any PLADE_* switch runs the library's shipped default (orient_normals = 0: plane normals keep the
    LS-fit eigenvector's sign, plane_extraction.cpp:43-58) -- on the reference's own sample pair (polyhedron, committed as
    data in g8_polyhedron.npz) the printed block equals the default-mode library result."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_polyhedron.npz"), allow_pickle=False)
    pt, ps, res = str(tmp_path / "t.ply"), str(tmp_path / "s.ply"), str(tmp_path / "r.txt")
    write_ply(pt, g["target"])
    write_ply(ps, g["source"])
    env = {k: v for k, v in os.environ.items() if not k.startswith("PLADE_")}
    r = subprocess.run([CLI, pt, ps, res], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    (b,) = parse_results(res)
    c = plade_amd.Context(0)                      # the C ABI's defaults: no parameter set, no environment read
    assert c.params.orient_normals == 0
    ok, T = c.registration(g["target"], g["source"])
    c.close()
    assert ok == (not b["failed"])
    assert np.allclose(b["T"], T, rtol=2e-5, atol=2e-6)
    # (whether the unoriented planes lead to the right alignment depends on their signs, in the reference as much as
    # here: tests/test_gpu_faithful.py pins that mode against the oracle)
