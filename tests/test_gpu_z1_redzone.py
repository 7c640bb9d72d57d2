"""Bounds check of the library's own (csrc/handle.h DevBuf, csrc/redzone.hip).

HIP AddressSanitizer cannot run on this pool (no XNACK), and the round-4 memory access fault on the 17-64-row vocabulary statistics
path was never reproduced.  With STATTN_DBG_REDZONE=1 every device buffer of the library carries 4 KiB of canary bytes on both
sides and the binding scans all of them after EVERY library call.  Here: (1) the detector fires -- a poked canary byte is reported
with the buffer's name; (2) the beam-search / sampler parity tests, the shapes of the round-4 fault among them, and the random
sampler sweep run under red zones in a child process: every test passes AND no canary byte was touched (a touched byte raises
NativeError in the call that follows the damage, so a green run means no kernel wrote outside any buffer of the library).
The switch is read once per process, hence the child processes."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

POKE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import stattn
from stattn import _native
from oracle import stattn_oracle as O
opt = O.default_options(dim=128, dim_word=64, n_words=211, ctxg_dim=128, ctxl_dim=96, ctxm_dim=64, ctxglm_dim=128)
P = O.random_params(opt, seed=3, dtype=np.float32)
batch = O.synthetic_batch(opt, B=3, T=5, K=4, t=4, seed=5)
model = stattn.Attention()
f_init, f_next = model.build_sampler(model.init_tparams(P), opt, None, None)
dec = f_next.decoder
f_init(batch['ctxg'][0], batch['mask_ctxg'][0])
dec.set_batch(**batch); dec.forward_train(); dec.backward(); dec.sync()
n0 = dec.redzone_checks
assert dec.redzone_buffers() > 20, dec.redzone_buffers()
assert n0 > 5, n0
for name, off in ((sys.argv[1], int(sys.argv[2])),):
    dec.redzone_poke(name, off)
    try:
        dec.sync()
    except _native.NativeError as e:
        assert "red zone" in str(e) and "'%%s'" %% name in str(e), str(e)
        print("DETECTED", name, off, "|", e)
    else:
        raise SystemExit("poke of %%s at %%d went unnoticed" %% (name, off))
    break          # (the damaged byte stays damaged: one poke per process)
print("POKE_OK")
''' % ROOT


def _env():
    env = dict(os.environ, STATTN_DBG_REDZONE="1")
    env.pop("STATTN_PRECISION", None)
    return env


@pytest.mark.parametrize("name,off", [("cost", 0), ("cost", 4095), ("hs", -1), ("hs", -4096), ("CL", 17)])
def test_a_damaged_canary_byte_is_reported_with_the_buffers_name(name, off):
    r = subprocess.run([sys.executable, "-c", POKE, name, str(off)], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DETECTED %s %d" % (name, off) in r.stdout and "POKE_OK" in r.stdout, r.stdout[-2000:]


def test_without_the_switch_nothing_is_guarded():
    code = ("import sys; sys.path.insert(0, %r); from stattn import _native; lib = _native.load_library(); "
            "assert lib.stattn_dbg_redzone_enabled() == 0; print('OFF')" % ROOT)
    env = dict(os.environ); env.pop("STATTN_DBG_REDZONE", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OFF" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("select", [
    # the beam-search tests: the 17-64-row statistics path whose round-4 fault was never reproduced, ragged vocabularies, batched
    # multi-video search, the riding updates, the evaluation shape and the captured word loop
    ["tests/test_gpu_parity.py", "-k", "mid_size_beams or batched_beam_search or update_rides or row_workgroup or msvd_eval_shape or "
                                       "c5_long_context or reference_executed_gen_sample or stochastic_gen_sample or gen_sample_matches"],
    # random sampler configurations (beam widths 1-12, several videos, ragged V) and random training configurations
    ["tests/test_gpu_fuzz.py"],
    # training forward + backward on the shapes with edges
    ["tests/test_gpu_edge_shapes.py"],
])
def test_parity_suites_touch_no_red_zone(select):
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + select,
                       env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=2400)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    m = re.search(r"REDZONE_SCANS=(\d+)", r.stdout)
    assert m and int(m.group(1)) > 50, tail            # the scans really ran in the child
    assert re.search(r"\b\d+ passed", r.stdout) and not re.search(r"\b\d+ (failed|error)", r.stdout), tail
