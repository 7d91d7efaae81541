"""Child process of tests/test_ply_fuzz.py: reads the PLY files named on the command line with the REFERENCE's reader
(oracle/_ref: rply.c + code/PLADE/ply_reader.cpp compiled from the reference's sources, load_ply_cloud restated in
oracle/ref/ref_ply_shim.cpp) and prints one line per file -- `ok n sha256-of-the-floats` -- flushed at once, so that the parent
knows which file ended the process if the reference's parser dies on it (it has fixed buffers)."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libplade_ref.so"))
L.ref_ply_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_long)]
L.ref_ply_free.argtypes = [C.POINTER(C.c_float)]
devnull = os.open(os.devnull, os.O_WRONLY)
out = os.dup(1)
os.dup2(devnull, 1); os.dup2(devnull, 2)          # the reference prints warnings: keep the protocol channel clean
for path in sys.argv[1:]:
    ptr, n = C.POINTER(C.c_float)(), C.c_long(0)
    ok = L.ref_ply_read(os.fsencode(path), C.byref(ptr), C.byref(n))
    if ok:
        a = np.ctypeslib.as_array(ptr, shape=(n.value, 6)).copy()
        L.ref_ply_free(ptr)
        nan = np.isnan(a)
        a[nan] = 0.0
        digest = hashlib.sha256(a.tobytes() + nan.tobytes()).hexdigest()
        os.write(out, f"1 {n.value} {digest}\n".encode())
    else:
        os.write(out, b"0 0 -\n")
