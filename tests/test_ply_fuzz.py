"""SURVEY 8f1, second net: 800 RANDOM PLY files -- random formats, element orders, property sets / types / spellings / duplicates,
list properties, CRLF, comments, odd version strings, awkward numbers (nan, inf, 1e39, hex floats, fractions and overflows in integer
columns), truncations and trailing garbage -- read by plade_ply_read and by the REFERENCE's reader compiled from its own sources
(oracle/_ref; in a child process, tests/ply_ref_helper.py, because rply has fixed buffers and may die on a hostile file: such files
are skipped, the reference has no behaviour there).  Same verdict on every file, same floats bit for bit (NaNs matching NaNs).
Runs where the reference tree's build exists (this container); tests/test_ply_reader.py holds the committed goldens for everywhere."""
import hashlib
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import plade_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libplade_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs the reference tree)")

TYPES = ["char", "uchar", "short", "ushort", "int", "uint", "float", "double", "int8", "uint8", "int16", "uint16", "int32", "uint32", "float32", "float64"]
FMT = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d"}
for a, b in (("int8", "char"), ("uint8", "uchar"), ("int16", "short"), ("uint16", "ushort"), ("int32", "int"), ("uint32", "uint"), ("float32", "float"), ("float64", "double")):
    FMT[a] = FMT[b]
RANGE = {"b": (-128, 127), "B": (0, 255), "h": (-32768, 32767), "H": (0, 65535), "i": (-2**31, 2**31 - 1), "I": (0, 2**32 - 1)}


def random_file(rng):
    mode = rng.choice(["ascii", "binary_little_endian", "binary_big_endian"], p=[0.5, 0.3, 0.2])
    nl = "\r\n" if rng.random() < 0.15 else "\n"
    version = rng.choice(["1.0", "1.0", "1.0", "1.0", "1.0", "1.0", "1.0textureless", "1.1", "2.0", "1"])
    names = ["x", "y", "z", "nx", "ny", "nz"]
    extra = ["red", "green", "blue", "r", "g", "b", "intensity", "X", "Y", "Z", "alpha", "x", "nz", "label"]
    props = list(names)
    if rng.random() < 0.12:
        props.remove(rng.choice(props))                           # an incomplete triple
    for _ in range(rng.integers(0, 4)):
        props.insert(rng.integers(0, len(props) + 1), rng.choice(extra))
    if rng.random() < 0.3:
        rng.shuffle(props)
    ptype = []
    for p in props:
        if p in ("red", "green", "blue", "alpha", "label") and rng.random() < 0.8:
            ptype.append(rng.choice(["uchar", "int", "uint8", "short"]))
        elif rng.random() < 0.05:
            ptype.append(rng.choice(TYPES))
        else:
            ptype.append(rng.choice(["float", "float", "double", "float32", "float64"]))
    vlist = rng.random() < 0.15                                   # a list property inside the vertex element
    nv = int(rng.choice([0, 1, 2, 3, 5, 9, 4, 7, 2, 3]))
    elems = [("vertex", nv)]
    if rng.random() < 0.4:
        elems.insert(rng.integers(0, 2), ("face", int(rng.integers(0, 4))))
    if rng.random() < 0.1:
        elems.append(("edge", int(rng.integers(1, 3))))
    head = "ply" + nl + f"format {mode} {version}" + nl
    if rng.random() < 0.3:
        head += "comment fuzz " + "c" * int(rng.integers(0, 40)) + nl
    body = bytearray()
    be = mode == "binary_big_endian"

    def put(t, v, out_words):
        f = FMT[t]
        if mode == "ascii":
            if f in "fd":
                r = rng.random()
                w = ("nan" if r < 0.01 else "inf" if r < 0.02 else "1e39" if r < 0.03 else "0x1.8p1" if r < 0.04 else "%.9g" % v if f == "f" else "%.17g" % v)
            else:
                lo, hi = RANGE[f]
                r = rng.random()
                iv = int(min(max(int(v), lo), hi))
                w = ("7.5" if r < 0.01 else str(hi + 1) if r < 0.02 else str(iv))
            out_words.append(w)
        else:
            if f in "fd":
                body.extend(struct.pack((">" if be else "<") + f, v))
            else:
                lo, hi = RANGE[f]
                body.extend(struct.pack((">" if be else "<") + f, int(min(max(int(v), lo), hi))))

    for name, cnt in elems:
        head += f"element {name} {cnt}" + nl
        if name == "vertex":
            for p, t in zip(props, ptype):
                head += f"property {t} {p}" + nl
            if vlist:
                head += "property list uchar float w" + nl
            if rng.random() < 0.1:
                head += "comment inside" + nl
        elif name == "face":
            head += "property list uchar int vertex_indices" + nl
        else:
            head += "property int vertex1" + nl + "property int vertex2" + nl
    head += "end_header" + nl
    for name, cnt in elems:
        for i in range(cnt):
            words = []
            if name == "vertex":
                for p, t in zip(props, ptype):
                    put(t, float(rng.normal() * 10), words)
                if vlist:
                    k = int(rng.integers(0, 3))
                    put("uchar", k, words)
                    for _ in range(k):
                        put("float", float(rng.normal()), words)
            elif name == "face":
                k = int(rng.integers(0, 5))
                put("uchar", k, words)
                for _ in range(k):
                    put("int", int(rng.integers(0, 9)), words)
            else:
                put("int", 0, words); put("int", 1, words)
            if mode == "ascii":
                sep = "\n" if rng.random() < 0.05 else " "
                body.extend((sep.join(words) + nl).encode())
    data = head.encode() + bytes(body)
    r = rng.random()
    if r < 0.08 and len(data) > 8:
        data = data[:int(rng.integers(4, len(data)))]            # truncated anywhere, header included
    elif r < 0.14:
        data += b" trailing garbage \x00\x01"
    return data


def reference_verdicts(paths):
    out, start = [None] * len(paths), 0
    while start < len(paths):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ply_ref_helper.py")] + paths[start:], capture_output=True, text=True, timeout=600)
        lines = r.stdout.splitlines()
        for k, l in enumerate(lines):
            ok, n, digest = l.split()
            out[start + k] = (ok == "1", int(n), digest)
        if len(lines) == len(paths) - start:
            break
        out[start + len(lines)] = "died"                           # the reference's process ended inside this file
        start += len(lines) + 1
    return out


def test_random_ply_files_read_like_the_reference(tmp_path):
    rng = np.random.default_rng(int(os.environ.get("PLADE_FUZZ_SEED", "20260929")))
    paths = []
    for k in range(800):
        p = tmp_path / f"f{k}.ply"
        p.write_bytes(random_file(rng))
        paths.append(str(p))
    ref = reference_verdicts(paths)
    accepted = refused = died = 0
    for p, want in zip(paths, ref):
        if want == "died":
            died += 1
            try:
                plade_amd.read_ply(p)                              # whatever the verdict, no crash
            except plade_amd.PladeError:
                pass
            continue
        try:
            got = plade_amd.read_ply(p)
            ok = True
        except plade_amd.PladeError:
            ok = False
        assert ok == want[0], (p, open(p, "rb").read()[:600], want)
        if ok:
            nan = np.isnan(got)
            g = got.copy(); g[nan] = 0.0
            assert len(got) == want[1] and hashlib.sha256(g.tobytes() + nan.tobytes()).hexdigest() == want[2], (p, open(p, "rb").read()[:600], got[:3])
            accepted += 1
        else:
            refused += 1
    print(f"random PLY files: {accepted} accepted, {refused} refused by both readers, {died} ended the reference's process")
    assert accepted >= 150 and refused >= 150 and died <= 30
