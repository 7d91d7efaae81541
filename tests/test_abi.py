"""The C-ABI library loads, exports every symbol include/plade_hip.h declares, and refuses to run
without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import plade_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "plade_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(plade_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = plade_amd.load_library()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(declared) == sorted(plade_amd.ABI_SYMBOLS), "python binding and header disagree"
    assert b"gfx950" in L.plade_version()


def test_params_defaults_match_reference_literals():
    L = plade_amd.load_library()
    p = plade_amd.Params()
    L.plade_default_params(ctypes.byref(p))
    assert (p.max_planes, p.min_planes, p.max_candidates, p.init_min_support) == (40, 10, 200, 10000)
    # the arithmetic that selects results is the reference's by default: unflipped plane normals (plane_extraction.cpp:43-58)
    # and the fp32 cv::solve(DECOMP_SVD) closest points (util.cpp:1183-1226); the better-conditioned variants are opt-ins
    assert (p.orient_normals, p.unoriented_normals, p.closest_point_mode) == (0, 0, 1)
    q = plade_amd.default_params()
    assert q.closest_point_mode == 1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    L = plade_amd.load_library()
    h = ctypes.c_void_p()
    rc = L.plade_ctx_create(0, ctypes.byref(h))
    assert rc == plade_amd.PLADE_EDEVICE and not h.value
    try:
        plade_amd.Context(0)
    except plade_amd.PladeError as e:
        assert e.code == plade_amd.PLADE_EDEVICE
    else:
        raise AssertionError("Context() must fail loudly without a GPU")


def test_product_never_touches_the_oracle():
    """plade_amd/ (python + C++ sources) may not import, include or link anything under oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "plade_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<][^\">]*oracle|plade_oracle|libplade_oracle|orc_[a-z_]+\(", txt, re.M):
                    bad.append(f)
    assert not bad, bad
    mk = open(os.path.join(ROOT, "Makefile")).read()
    lib_rule = mk.split("plade_amd/libplade_hip.so:")[1].split("\n\n")[0]
    assert "oracle" not in lib_rule


def test_the_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/plade_hip.h must compile as C99 on its own (a cgo / ctypes / JNI binding includes
    nothing else)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "plade_hip.h"\nint main(void) { plade_params p; plade_default_params(&p); return (int)sizeof(p) & 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
