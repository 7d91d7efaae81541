import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The library's default is the reference's behaviour: plane normals keep the sign of the LS-fit eigenvector
# (plane_extraction.cpp:43-58 never flips them).  The synthetic scenes of this suite are Manhattan rooms with
# ORIENTED point normals, where that behaviour registers a pair only when three independent sign bits happen to agree
# (1 in 8, in the reference as much as here), so the tests that register synthetic scenes ask for
# params.orient_normals = 1 explicitly (ORIENTED below; the session context has it), CLI subprocesses get
# PLADE_ORIENT_NORMALS=1 in their env (ORIENTED_ENV).  The shipped default (orient_normals = 0) is exercised by
# tests/test_gpu_faithful.py (library, end to end against the oracle) and tests/test_gpu_cli.py (the CLI).
ORIENTED = {"orient_normals": 1}
# Accuracy against the GENERATOR's ground truth.  The library's default arithmetic for the closest points of two lines is the
# reference's (plade_params.closest_point_mode = 1: cv::solve(DECOMP_SVD) on fp32 9 x 9 systems, bit-identical to the oracle's
# restatement); on these axis-aligned synthetic rooms those solves are ill-conditioned (intersection lines with base points
# kilometres away) and the registration lands 1e-2 ... 1.4e-1 (Frobenius) from the ground truth -- the
# reference would, too.  Tests that ask "did it register the pair" use GT_TOL; tests that pin an accuracy (equivariance, the
# 1e-3 of the bench scene) ask for the better-conditioned opt-in, CLOSED_FORM, like ORIENTED above.
GT_TOL = 0.2
CLOSED_FORM = {"closest_point_mode": 0}
ORIENTED_ENV = {"PLADE_ORIENT_NORMALS": "1"}
os.environ.pop("PLADE_ORIENT_NORMALS", None)     # nothing is inherited from the caller's shell
os.environ.pop("PLADE_UNORIENTED_NORMALS", None)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    """No GPU test may sit on the device for ever: 10 minutes each unless the test sets its own limit."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ctx():
    import plade_amd
    c = plade_amd.Context(0, **ORIENTED)
    yield c
    c.close()
