"""Minimal PLY reader/writer for tests, tools and bench (binary-LE / ascii,
`float x y z nx ny nz` vertex layout: the layout PLADE reads, code/PLADE/util.cpp:1521-1540)."""
import numpy as np


def read_ply(path):
    with open(path, "rb") as f:
        header = b""
        while not header.endswith(b"end_header\n"):
            line = f.readline()
            if not line:
                raise ValueError("bad PLY header")
            header += line
        lines = header.decode("ascii", "replace").split("\n")
        fmt = [l for l in lines if l.startswith("format")][0].split()[1]
        n = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        props = []
        in_vertex = False
        for l in lines:
            if l.startswith("element"):
                in_vertex = l.startswith("element vertex")
            elif l.startswith("property") and in_vertex:
                props.append(l.split()[-1])
        want = ["x", "y", "z", "nx", "ny", "nz"]
        cols = [props.index(w) for w in want]
        if fmt == "ascii":
            a = np.loadtxt(f, dtype=np.float32, max_rows=n).reshape(n, -1)
        else:
            dt = "<f4" if fmt == "binary_little_endian" else ">f4"
            a = np.fromfile(f, dtype=dt, count=n * len(props)).reshape(n, len(props)).astype(np.float32)
    return np.ascontiguousarray(a[:, cols])


def write_ply(path, pos_nrm):
    a = np.ascontiguousarray(pos_nrm, dtype="<f4")
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(f"element vertex {len(a)}\n".encode())
        for p in ("x", "y", "z", "nx", "ny", "nz"):
            f.write(f"property float {p}\n".encode())
        f.write(b"end_header\n")
        a.tofile(f)
