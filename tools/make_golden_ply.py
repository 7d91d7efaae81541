#!/usr/bin/env python3
"""tests/golden/ply_cases.npz -- what the REFERENCE's PLY ingest makes of small PLY files of every kind, and of its own sample data.

The ingest is the reference's own code compiled from its sources (oracle/ref/Makefile: code/3rd_party/rply/rply.c +
code/PLADE/ply_reader.cpp; the PCL-typed caller load_ply_cloud, code/PLADE/util.cpp:1505-1546, restated in
oracle/ref/ref_ply_shim.cpp).  Every case is data: the bytes of a file, whether the reference accepts it, and the N x 6 float32
array it returns.  For the three sample PLYs under /root/reference/sample_data only the point count and a SHA-256 of the
returned bytes are kept (tests/test_ply_reader.py re-reads them live where the reference tree is present).
Run in the build container: python tools/make_golden_ply.py"""
import ctypes as C
import hashlib
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ref_read(L, path):
    ptr, n = C.POINTER(C.c_float)(), C.c_long(0)
    ok = L.ref_ply_read(os.fsencode(path), C.byref(ptr), C.byref(n))
    if not ok:
        return False, np.zeros((0, 6), np.float32)
    a = np.ctypeslib.as_array(ptr, shape=(n.value, 6)).copy()
    L.ref_ply_free(ptr)
    return True, a


def cases():
    rng = np.random.default_rng(7)
    n = 23
    P = rng.normal(size=(n, 3)).astype(np.float32) * 3
    N = rng.normal(size=(n, 3)).astype(np.float32)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    Pd = rng.normal(size=(n, 3)) * 1e3 + 1e-7          # doubles that do not fit a float
    Nd = rng.normal(size=(n, 3))
    PN = np.concatenate([P, N], 1)
    out = {}

    def hdr(fmt, elems, nl="\n", magic="ply", version="1.0", extra=()):
        s = magic + nl + f"format {fmt} {version}" + nl
        for e in extra:
            s += e + nl
        for name, count, props in elems:
            s += f"element {name} {count}" + nl
            for p in props:
                s += "property " + p + nl
        return (s + "end_header" + nl).encode()

    f6 = ["float x", "float y", "float z", "float nx", "float ny", "float nz"]
    out["bin_le_plain6"] = hdr("binary_little_endian", [("vertex", n, f6)]) + PN.astype("<f4").tobytes()
    out["bin_be_plain6"] = hdr("binary_big_endian", [("vertex", n, f6)]) + PN.astype(">f4").tobytes()
    out["ascii_plain6"] = hdr("ascii", [("vertex", n, f6)]) + "".join(" ".join(repr(float(v)) for v in r) + "\n" for r in PN).encode()
    d6 = [p.replace("float", "double") for p in f6]
    PNd = np.concatenate([Pd, Nd], 1)
    out["bin_le_double"] = hdr("binary_little_endian", [("vertex", n, d6)]) + PNd.astype("<f8").tobytes()
    out["bin_be_double_typenames"] = hdr("binary_big_endian", [("vertex", n, [p.replace("float", "float64") for p in f6])]) + PNd.astype(">f8").tobytes()
    out["ascii_double_17_digits"] = hdr("ascii", [("vertex", n, d6)]) + "".join(" ".join("%.17g" % v for v in r) + "\n" for r in PNd).encode()
    # extra properties around and between, a face element with lists behind
    props = ["float x", "float y", "float z", "uchar red", "uchar green", "uchar blue", "float nx", "float ny", "float nz", "float intensity", "int label"]
    rows = b""
    for i in range(n):
        rows += struct.pack("<3f3B3ffi", *P[i], i % 256, 7, 255, *N[i], 0.5 * i, -i)
    faces = b"".join(struct.pack("<B3i", 3, i, (i + 1) % n, (i + 2) % n) for i in range(5))
    out["bin_le_extra_props_faces"] = hdr("binary_little_endian", [("vertex", n, props), ("face", 5, ["list uchar int vertex_indices"])],
                                          extra=["comment made by hand", "obj_info something else"]) + rows + faces
    rows = b""
    for i in range(n):
        rows += struct.pack(">3f3B3ffi", *P[i], i % 256, 7, 255, *N[i], 0.5 * i, -i)
    faces_be = b"".join(struct.pack(">B3i", 3, i, (i + 1) % n, (i + 2) % n) for i in range(5))
    out["bin_be_extra_props_faces"] = hdr("binary_big_endian", [("vertex", n, props), ("face", 5, ["list uchar int vertex_indices"])]) + rows + faces_be
    arows = "".join(f"{float(P[i,0])!r} {float(P[i,1])!r} {float(P[i,2])!r} {i % 256} 7 255 {float(N[i,0])!r} {float(N[i,1])!r} {float(N[i,2])!r} {0.5 * i} {-i}\n" for i in range(n))
    afaces = "".join(f"3 {i} {(i + 1) % n} {(i + 2) % n}\n" for i in range(5))
    out["ascii_extra_props_faces_comments"] = hdr("ascii", [("vertex", n, props), ("face", 5, ["list uchar int vertex_indices"])],
                                                  extra=["comment a", "comment   b  c", "obj_info d"]) + (arows + afaces).encode()
    out["ascii_face_element_first"] = hdr("ascii", [("face", 5, ["list uchar int vertex_indices"]), ("vertex", n, f6)]) + \
        (afaces + "".join(" ".join(repr(float(v)) for v in r) + "\n" for r in PN)).encode()
    out["bin_le_face_element_first"] = hdr("binary_little_endian", [("face", 5, ["list uchar int vertex_indices"]), ("vertex", n, f6)]) + faces + PN.astype("<f4").tobytes()
    out["ascii_uppercase_XYZ"] = hdr("ascii", [("vertex", n, ["float X", "float Y", "float Z", "float nx", "float ny", "float nz"])]) + \
        "".join(" ".join(repr(float(v)) for v in r) + "\n" for r in PN).encode()
    out["ascii_lower_xy_only_then_XYZ"] = hdr("ascii", [("vertex", 3, ["float x", "float y", "float X", "float Y", "float Z", "float nx", "float ny", "float nz"])]) + \
        b"9 9 1 2 3 0 0 1\n9 9 4 5 6 0 1 0\n9 9 7 8 9 1 0 0\n"
    out["ascii_shuffled_columns"] = hdr("ascii", [("vertex", n, ["float nz", "float y", "float nx", "float x", "float ny", "float z"])]) + \
        "".join(f"{r[5]!r} {r[1]!r} {r[3]!r} {r[0]!r} {r[4]!r} {r[2]!r}\n" for r in PN.tolist()).encode()
    out["ascii_duplicate_x"] = hdr("ascii", [("vertex", 2, ["float x", "float x", "float y", "float z", "float nx", "float ny", "float nz"])]) + b"1 2 3 4 0 0 1\n5 6 7 8 0 1 0\n"
    out["ascii_duplicate_int_x_then_float_x"] = hdr("ascii", [("vertex", 2, ["int x", "float x", "float y", "float z", "float nx", "float ny", "float nz"])]) + b"1 2.5 3 4 0 0 1\n5 6.5 7 8 0 1 0\n"
    out["ascii_duplicate_float_x_then_int_x"] = hdr("ascii", [("vertex", 2, ["float x", "int x", "float y", "float z", "float nx", "float ny", "float nz"])]) + b"1.5 2 3 4 0 0 1\n5.5 6 7 8 0 1 0\n"
    out["ascii_triplicate_nz"] = hdr("ascii", [("vertex", 2, ["float x", "float y", "float z", "float nx", "float ny", "float nz", "double nz", "float nz"])]) + b"1 2 3 0 0 1 7 8\n4 5 6 0 1 0 7 8\n"
    out["ascii_values_across_lines"] = hdr("ascii", [("vertex", 3, f6)]) + b"1 2\n3 0 0\n1 4 5 6\n\n\n0 1 0 7\t8 9 1 0 0"
    out["ascii_no_final_newline_crlf"] = hdr("ascii", [("vertex", 2, f6)], nl="\r\n") + b"1 2 3 0 0 1\r\n4 5 6 0 1 0"
    out["bin_le_crlf_header"] = hdr("binary_little_endian", [("vertex", n, f6)], nl="\r\n") + PN.astype("<f4").tobytes()
    out["bin_le_crlf_magic_only"] = b"ply\r\n" + hdr("binary_little_endian", [("vertex", n, f6)])[4:] + b"\n" + PN.astype("<f4").tobytes()
    out["ascii_version_1.0textureless"] = hdr("ascii", [("vertex", 2, f6)], version="1.0textureless") + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["ascii_version_2.0_fails"] = hdr("ascii", [("vertex", 2, f6)], version="2.0") + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["ascii_version_1.5"] = hdr("ascii", [("vertex", 2, f6)], version="1.5") + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["bad_magic_fails"] = hdr("ascii", [("vertex", 2, f6)], magic="plx") + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["magic_without_blank_fails"] = b"plyformat ascii 1.0\nelement vertex 1\n" + b"".join(("property " + p + "\n").encode() for p in f6) + b"end_header\n1 2 3 0 0 1\n"
    out["unknown_format_fails"] = hdr("binary", [("vertex", 2, f6)]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["unknown_keyword_fails"] = hdr("ascii", [("vertex", 2, f6)], extra=["texture foo.png"]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["unknown_type_fails"] = hdr("ascii", [("vertex", 2, ["float x", "float y", "half z", "float nx", "float ny", "float nz"])]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["property_before_element_fails"] = b"ply\nformat ascii 1.0\nproperty float x\nelement vertex 1\n" + b"".join(("property " + p + "\n").encode() for p in f6) + b"end_header\n1 2 3 0 0 1\n"
    out["int_xyz_fails"] = hdr("ascii", [("vertex", 2, ["int x", "int y", "int z", "float nx", "float ny", "float nz"])]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["int_everything_fails"] = hdr("ascii", [("vertex", 2, [p.replace("float", "int") for p in f6])]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["no_normals_fails"] = hdr("ascii", [("vertex", 2, f6[:3])]) + b"1 2 3\n4 5 6\n"
    out["no_points_fails"] = hdr("ascii", [("vertex", 2, f6[3:])]) + b"0 0 1\n0 1 0\n"
    out["normals_incomplete_fails"] = hdr("ascii", [("vertex", 2, f6[:5])]) + b"1 2 3 0 0\n4 5 6 0 1\n"
    out["zero_vertices_fails"] = hdr("ascii", [("vertex", 0, f6)])
    out["zero_vertices_with_faces_fails"] = hdr("ascii", [("vertex", 0, f6), ("face", 1, ["list uchar int vertex_indices"])]) + b"3 0 1 2\n"
    out["no_vertex_element_fails"] = hdr("ascii", [("vertices", 2, f6)]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["truncated_binary_fails"] = out["bin_le_plain6"][:-5]
    out["truncated_binary_generic_fails"] = out["bin_le_extra_props_faces"][:-3]
    out["truncated_ascii_fails"] = hdr("ascii", [("vertex", 2, f6)]) + b"1 2 3 0 0 1\n4 5 6 0 1\n"
    out["trailing_bytes_ignored"] = out["bin_le_plain6"] + b"garbage behind the data"
    out["ascii_float_out_of_range_fails"] = hdr("ascii", [("vertex", 2, f6)]) + b"1 2 3 0 0 1\n4e39 5 6 0 1 0\n"
    out["ascii_double_beyond_float_is_inf"] = hdr("ascii", [("vertex", 2, d6)]) + b"1 2 3 0 0 1\n4e39 -5e300 6 0 1 0\n"
    out["ascii_nan_accepted"] = hdr("ascii", [("vertex", 2, f6)]) + b"nan 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["ascii_inf_fails"] = hdr("ascii", [("vertex", 2, f6)]) + b"inf 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["ascii_garbage_number_fails"] = hdr("ascii", [("vertex", 2, f6)]) + b"1 2 3 0 0 1\n4 5x 6 0 1 0\n"
    out["ascii_hex_float_accepted"] = hdr("ascii", [("vertex", 2, f6)]) + b"0x1.8p1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["ascii_uchar_out_of_range_fails"] = hdr("ascii", [("vertex", 2, f6 + ["uchar red"])]) + b"1 2 3 0 0 1 255\n4 5 6 0 1 0 256\n"
    out["ascii_int_with_fraction_fails"] = hdr("ascii", [("vertex", 2, f6 + ["int label"])]) + b"1 2 3 0 0 1 7\n4 5 6 0 1 0 7.5\n"
    out["element_count_with_suffix"] = hdr("ascii", [("vertex", "2abc", f6)]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["element_count_not_a_number_fails"] = hdr("ascii", [("vertex", "two", f6)]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["negative_element_count_fails"] = hdr("ascii", [("vertex", -2, f6)]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["list_in_vertex_element"] = hdr("ascii", [("vertex", 2, ["float x", "list uchar float w", "float y", "float z", "float nx", "float ny", "float nz"])]) + \
        b"1 2 0.5 0.25 2 3 0 0 1\n4 0 5 6 0 1 0\n"
    out["negative_list_length"] = hdr("ascii", [("vertex", 2, ["float x", "list char float w", "float y", "float z", "float nx", "float ny", "float nz"])]) + \
        b"1 -1 2 3 0 0 1\n4 -3 5 6 0 1 0\n"
    out["face_list_truncated_fails"] = hdr("ascii", [("vertex", 2, f6), ("face", 2, ["list uchar int vertex_indices"])]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n3 0 1 2\n3 0 1\n"
    out["face_index_garbage_fails"] = hdr("ascii", [("vertex", 2, f6), ("face", 1, ["list uchar int vertex_indices"])]) + b"1 2 3 0 0 1\n4 5 6 0 1 0\n3 0 1 x\n"
    out["float_colors_warn_only"] = hdr("ascii", [("vertex", 2, f6 + ["float r", "float g", "float b"])]) + b"1 2 3 0 0 1 .1 .2 .3\n4 5 6 0 1 0 .4 .5 .6\n"
    out["mixed_float_points_double_normals"] = hdr("binary_little_endian", [("vertex", n, f6[:3] + d6[3:])]) + \
        b"".join(struct.pack("<3f3d", *P[i], *Nd[i]) for i in range(n))
    out["comment_inside_element"] = hdr("ascii", [("vertex", 2, f6[:3] + ["float nx"]), ]).replace(b"end_header\n", b"") + \
        b"comment between properties\nproperty float ny\nobj_info x\nproperty float nz\nend_header\n1 2 3 0 0 1\n4 5 6 0 1 0\n"
    out["empty_file_fails"] = b""
    out["header_only_fails"] = hdr("ascii", [("vertex", 2, f6)])
    out["header_not_terminated_fails"] = hdr("ascii", [("vertex", 2, f6)]).replace(b"end_header\n", b"")
    # (a comment line of 1024 characters or more overflows a fixed buffer inside rply -- rply.c:571-583 -- and ends the reference
    #  process: nothing to pin; our reader refuses such a header)
    out["big_header_many_comments"] = hdr("binary_little_endian", [("vertex", n, f6)], extra=["comment " + "y" * 900] * 90) + PN.astype("<f4").tobytes()
    return out


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libplade_ref.so"))
    L.ref_ply_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_long)]
    L.ref_ply_free.argtypes = [C.POINTER(C.c_float)]
    blob = {}
    names = []
    with tempfile.TemporaryDirectory() as d:
        for name, data in cases().items():
            path = os.path.join(d, "case.ply")
            with open(path, "wb") as f:
                f.write(data)
            ok, a = ref_read(L, path)
            assert ok == (not name.endswith("_fails")), (name, ok)     # the names say what the reference does
            names.append(name)
            blob["file_" + name] = np.frombuffer(data, np.uint8)
            blob["ok_" + name] = np.array(ok)
            blob["cloud_" + name] = a
            print(f"{name:40s} {'ok  ' if ok else 'FAIL'} n = {len(a)}")
    # the reference's own sample data: count + digest (the files themselves stay where they are)
    sample = {}
    sd = "/root/reference/sample_data"
    for fn in sorted(os.listdir(sd)):
        if fn.endswith(".ply"):
            ok, a = ref_read(L, os.path.join(sd, fn))
            sample[fn] = (ok, len(a), hashlib.sha256(a.tobytes()).hexdigest())
            print(fn, sample[fn])
    blob["names"] = np.array(names)
    blob["sample_names"] = np.array(list(sample))
    blob["sample_ok"] = np.array([v[0] for v in sample.values()])
    blob["sample_n"] = np.array([v[1] for v in sample.values()], np.int64)
    blob["sample_sha256"] = np.array([v[2] for v in sample.values()])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ply_cases.npz"), **blob)


if __name__ == "__main__":
    main()
