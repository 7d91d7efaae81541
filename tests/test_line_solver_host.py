"""The register form of the reference's SVD solver (plade_amd/csrc/k_svd.h: RegSolver, what k_closest_svd and k_pen_setup_svd inline)
instantiated for the HOST (plade_diag_line_solver_host) against the oracle's restatement of cv::solve(DECOMP_SVD)
(oracle/plade_oracle.cpp; opencv/modules/core/src/lapack.cpp:533-812, 1335-1460): the 9 x 9 systems of
ComputeNearstTwoPointsOfTwo3DLine (code/PLADE/util.cpp:1183-1226) and the 6 x 5 systems of ComputeIntersectionPointOf23DLine
(util.cpp:1467-1497), bit for bit, on every kind of line pair the registration meets -- without a GPU.  The same functions on
the GPU: tests/test_gpu_svd_mode.py."""
import numpy as np
import pytest

import plade_amd
from test_gpu_svd_mode import _line_cases, _same_bits


@pytest.mark.parametrize("seed", [11, 12, 21])
def test_register_solver_equals_the_oracle_solver(oracle, seed):
    U1, P1, U2, P2 = _line_cases(seed)
    try:
        oracle.set_closest_point_mode("svd_fp32")
        q1, q2, ok = plade_amd.line_solver_host(0, U1, P1, U2, P2)
        guard = solved = 0
        for i in range(len(U1)):
            rc, o1, o2, _ = oracle.closest_points(U1[i], P1[i], U2[i], P2[i])
            if ok[i] == -1:
                guard += 1
                assert rc != 0, i                          # util.cpp:1173
                continue
            assert rc == 0 and ok[i] == 1, i               # (a 9 x 9 system of two non-identical directions has full rank)
            assert _same_bits(q1[i], o1) and _same_bits(q2[i], o2), (i, q1[i], o1, q2[i], o2)
            solved += 1
        assert guard >= 21 and solved >= 950
        V1 = U1 / np.maximum(np.linalg.norm(U1, axis=1, keepdims=True), 1e-30).astype(np.float32)
        V2 = U2 / np.maximum(np.linalg.norm(U2, axis=1, keepdims=True), 1e-30).astype(np.float32)
        o, ok = plade_amd.line_solver_host(1, V1, P1, V2, P2)
        guard = solved = deficient = 0
        for i in range(len(V1)):
            rc, oo = oracle.intersection_point(V1[i], P1[i], V2[i], P2[i])
            if ok[i] == -1:
                guard += 1
                assert rc != 0, i                          # util.cpp:1463
                continue
            assert rc == 0, i
            if ok[i] == 0:                                 # a vanished column (the zero direction): the kernels hand it to LaneSolver
                deficient += 1
                continue
            assert _same_bits(o[i], oo), (i, o[i], oo)
            solved += 1
        assert guard >= 20 and solved >= 700 and deficient <= 2
    finally:
        oracle.reset_closest_point_mode()
