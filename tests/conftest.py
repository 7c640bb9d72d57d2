import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never run implicitly on a box without a GPU
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# Every GPU parity test of these modules runs twice: on a precision = 0 handle (fp32-input MFMA, the headline
# configuration) and on a precision = 2 one (the same fp32 GEMMs as three-term bf16 operands on the bf16 matrix cores,
# DESIGN.md section 12) -- same oracle, same fp32 tolerances.  A handle created with an explicit `precision=` keeps it.
BOTH_PRECISIONS = {"test_gpu_parity", "test_gpu_backward", "test_gpu_edge_shapes", "test_data_and_metrics", "test_gpu_properties"}


@pytest.fixture(autouse=True)
def stattn_precision(request, monkeypatch):
    p = getattr(request, "param", None)
    if p is not None:
        monkeypatch.setenv("STATTN_PRECISION", p)
    return p


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if mod in BOTH_PRECISIONS and metafunc.definition.get_closest_marker("gpu") and "STATTN_PRECISION" not in os.environ:
        metafunc.parametrize("stattn_precision", ["fp32", "split"], indirect=True)


def pytest_sessionfinish(session, exitstatus):
    # red-zone runs (tests/test_gpu_z1_redzone.py starts pytest in a child with STATTN_DBG_REDZONE=1): how many scans the session made
    if os.environ.get("STATTN_DBG_REDZONE"):
        try:
            from stattn import _native
            print("\nREDZONE_SCANS=%d" % _native.REDZONE_CHECKS[0])
        except Exception:
            pass
