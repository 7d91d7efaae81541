"""the CLI's PLY ingest (plade_ply_read, plade_amd/csrc/ply_reader.cpp) against the REFERENCE's ingest.

tests/golden/ply_cases.npz holds what load_ply_cloud (code/PLADE/util.cpp:1505-1546) over PlyReader (code/PLADE/ply_reader.cpp:46-152,
277-386) and rply (code/3rd_party/rply/rply.c) -- compiled from the reference's own sources into oracle/_ref, recipe
oracle/ref/Makefile, generator tools/make_golden_ply.py -- makes of ~65 small files: ascii, both binary byte orders, float and
double coordinates, extra and shuffled properties, X Y Z, list elements before and behind the vertices, CRLF headers, values
that straddle lines, out-of-range / non-finite / malformed numbers, integer-typed coordinates, missing normals, empty and
truncated files.  Our reader must return the same floats, bit for bit, and fail on the same files.  Where the reference tree
and oracle/_ref are present (this container), the reference's three sample PLYs are read by both, live."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import plade_amd

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ply_cases.npz"))
NAMES = [str(n) for n in GOLD["names"]]
SAMPLE_DIR = "/root/reference/sample_data"
REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libplade_ref.so")


def _same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))


@pytest.mark.parametrize("name", NAMES)
def test_reader_equals_the_reference_on_the_case(tmp_path, name):
    path = tmp_path / "case.ply"
    path.write_bytes(GOLD["file_" + name].tobytes())
    want_ok, want = bool(GOLD["ok_" + name]), GOLD["cloud_" + name]
    if want_ok:
        got = plade_amd.read_ply(str(path))
        assert _same_bits(got, want), name
    else:
        with pytest.raises(plade_amd.PladeError) as e:
            plade_amd.read_ply(str(path))
        assert str(e.value)          # a message, not an empty error


def test_the_goldens_cover_both_outcomes_and_both_paths():
    oks = [bool(GOLD["ok_" + n]) for n in NAMES]
    assert sum(oks) >= 30 and len(oks) - sum(oks) >= 25
    assert not bool(GOLD["ok_truncated_binary_fails"]) and bool(GOLD["ok_bin_le_plain6"])      # the bulk-read path, good and bad


def test_missing_file_fails_like_the_reference(tmp_path):
    with pytest.raises(plade_amd.PladeError):
        plade_amd.read_ply(str(tmp_path / "does_not_exist.ply"))      # ply_reader.cpp:48-51


def test_overlong_comment_line_is_refused(tmp_path):
    # the reference overflows a fixed buffer here (rply.c:571-583) and dies; nothing to pin, but we must not
    p = tmp_path / "c.ply"
    p.write_bytes(b"ply\nformat ascii 1.0\ncomment " + b"x" * 5000 + b"\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\n"
                  b"property float nx\nproperty float ny\nproperty float nz\nend_header\n1 2 3 0 0 1\n")
    with pytest.raises(plade_amd.PladeError):
        plade_amd.read_ply(str(p))


def test_large_plain_file_takes_the_bulk_path_and_round_trips(tmp_path):
    from plade_amd.plyio import write_ply
    rng = np.random.default_rng(3)
    a = rng.normal(size=(200000, 6)).astype(np.float32)
    p = tmp_path / "big.ply"
    write_ply(str(p), a)
    assert np.array_equal(plade_amd.read_ply(str(p)).view(np.uint32), a.view(np.uint32))


@pytest.mark.skipif(not (os.path.isdir(SAMPLE_DIR) and os.path.exists(REF_SO)), reason="reference tree / oracle/_ref not present")
def test_sample_data_live_against_the_reference_reader():
    L = C.CDLL(REF_SO)
    L.ref_ply_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_long)]
    L.ref_ply_free.argtypes = [C.POINTER(C.c_float)]
    seen = 0
    for fn, ok, n, digest in zip(GOLD["sample_names"], GOLD["sample_ok"], GOLD["sample_n"], GOLD["sample_sha256"]):
        path = os.path.join(SAMPLE_DIR, str(fn))
        ptr, cnt = C.POINTER(C.c_float)(), C.c_long(0)
        assert bool(L.ref_ply_read(os.fsencode(path), C.byref(ptr), C.byref(cnt))) == bool(ok)
        ref = np.ctypeslib.as_array(ptr, shape=(cnt.value, 6)).copy()
        L.ref_ply_free(ptr)
        got = plade_amd.read_ply(path)
        assert len(got) == int(n) == len(ref) and _same_bits(got, ref)
        assert hashlib.sha256(got.tobytes()).hexdigest() == str(digest)       # the committed golden of the same read
        seen += 1
    assert seen == 3
