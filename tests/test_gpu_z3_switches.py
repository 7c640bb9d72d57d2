"""Every product switch of csrc/switches.h selects between two paths of the SAME arithmetic: the path behind the switch gives the
default path's results.  STATTN_NO_RIDER, STATTN_NO_UPDATE_RIDER, STATTN_NO_ROW_WG and STATTN_NO_PANELS have tests of their own
(test_gpu_bf16.py, test_gpu_parity.py), STATTN_COMM_NO_OVERLAP is the comm_set_overlap(0) mode of test_gpu_backward.py /
test_gpu_dp2.py, STATTN_DBG_REDZONE is test_gpu_z1_redzone.py; here the remaining four -- and NO_PANELS once more on a training
step -- each in a child process (the switches are read once per process) against the default process: one optimisation step
(24 rows, D = 1024: riders and row-panel kernels) and one batched beam search of 20 rows.
(Written in round 6 while the GPU pool was closed to the build: sorts last, the driver runs pytest -x.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _child(tmp_path, tag, **env):
    out = str(tmp_path / ("%s.npz" % tag))
    e = dict(os.environ, **env)
    e.pop("STATTN_PRECISION", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_switch_worker.py"), out], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return np.load(out)


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    return _child(tmp_path_factory.mktemp("sw"), "default")


def _same(a, b, grads_tol=1e-5, fwd_tol=5e-6):
    for k in ('alphal', 'alphag', 'alpham', 'alphalt'):
        assert np.abs(a[k] - b[k]).max() <= fwd_tol, k
    assert np.abs(a["logit"] - b["logit"]).max() <= 5e-5 and np.abs(a['cost'] / b['cost'] - 1).max() <= 1e-5
    for k in a.files:
        if k.startswith("g_"):
            assert np.abs(a[k] - b[k]).max() <= grads_tol * np.abs(a[k]).max() + 1e-8, k
    for v in range(5):
        assert list(a["tok_%d" % v]) == list(b["tok_%d" % v]), v
        np.testing.assert_allclose(a["score_%d" % v], b["score_%d" % v], rtol=1e-5, atol=1e-5)


def test_default_process_takes_the_fast_paths(default_run):
    d = default_run
    assert int(d["pc_fwd_rider"]) > 0 and int(d["pc_fwd_panel"]) > 0 and int(d["pc_bwd_rider"]) > 0 and int(d["pc_bwd_panel"]) > 0
    assert int(d["graph_replays"]) > 0 and int(d["stats_words"]) > 0        # captured word loop, statistics epilogue on the 20-row word


@pytest.mark.parametrize("name,value", [("STATTN_GEMM_NOGROUP", "1"), ("STATTN_READOUT_NOPAIR", "1"), ("STATTN_BEAM_NOGRAPH", "1"),
                                        ("STATTN_WIDE_STATS_FROM", "65"), ("STATTN_NO_PANELS", "1")])
def test_switched_path_gives_the_default_paths_results(default_run, tmp_path, name, value):
    s = _child(tmp_path, name, **{name: value})
    # the switch really switched
    if name == "STATTN_BEAM_NOGRAPH":
        assert int(s["graph_replays"]) == 0
    if name == "STATTN_WIDE_STATS_FROM":
        assert int(s["stats_words"]) == 0
    if name == "STATTN_NO_PANELS":
        assert int(s["pc_fwd_panel"]) == 0 and int(s["pc_bwd_panel"]) == 0
    # same arithmetic, other launch structure: summation orders differ inside GEMMs (grouped / paired / split-K launches), nothing else
    _same(default_run, s, grads_tol=2e-5 if name in ("STATTN_NO_PANELS", "STATTN_READOUT_NOPAIR") else 1e-5)
