"""A data-parallel step with TWO (and three) RCCL ranks (BASELINE configs[2] is 8; the GPU boxes this repo is tested on have one
GPU).  Two processes share cuda:0.  RCCL refuses two ranks of one communicator on the same device *of the same host*, so
each rank is given its own NCCL_HOSTID: RCCL then treats them as two single-GPU nodes and connects them through its
socket transport over the loopback interface.  That is slow and says nothing about xGMI, but everything in csrc/comm.cpp
that a multi-rank run needs is real here: the unique id handed over by stattn.dp's file rendezvous, ncclCommInitRank with
nranks = 2, the parameter broadcast, the four gradient regions reduced on the side stream while backward runs (and the
single all-reduce with overlap off), the scalar all-reduce of the loss, and the exactness rule of SURVEY section 8e
(model_attention.py:1129-1147: NLL mean over the GLOBAL batch, regulariser summed, decay and clip once)."""
import json
import os
import subprocess
import sys
from collections import OrderedDict

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks_once(tmp, mode, world, timeout):
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(NCCL_HOSTID="stattn-test-host-%d" % r, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                   NCCL_SHM_DISABLE="1", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   STATTN_RENDEZVOUS_ID="dp2-%d-%d" % (os.getpid(), mode))
        log = open(os.path.join(tmp, "rank%d.log" % r), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "_dp2_worker.py"), str(r), str(world), tmp, str(mode)],
                                       env=env, stdout=log, stderr=subprocess.STDOUT), log))
    rcs = []
    try:
        for p, _ in procs:
            rcs.append(p.wait(timeout=timeout))
    except subprocess.TimeoutExpired:
        for p, _ in procs:                                     # exactly the processes started here
            if p.poll() is None:
                p.kill()
        rcs = None
    for _, log in procs:
        log.close()
    logs = "\n".join(open(os.path.join(tmp, "rank%d.log" % r)).read()[-3000:] for r in range(world))
    return rcs, logs


def _run_ranks(tmp, mode, world=2, timeout=150):
    """Ranks that time-share ONE GPU bootstrap RCCL over loopback sockets; on a loaded box the first such run of a session has been seen
    to sit in the bootstrap past any reasonable limit (the same command passes seconds later and on its own).  One retry with fresh
    processes and a fresh rendezvous file; the first attempt's logs (each rank dumps its Python stack after 100 s) are kept in the
    failure message if the second attempt hangs as well."""
    rcs, logs = _run_ranks_once(tmp, mode, world, timeout)
    if rcs is None:
        first = logs
        retry = os.path.join(tmp, "retry")
        os.makedirs(retry, exist_ok=True)
        rcs, logs = _run_ranks_once(retry, mode, world, timeout)
        if rcs is None:
            logs = "first attempt:\n" + first + "\nsecond attempt:\n" + logs
        else:
            for f in os.listdir(retry):
                os.replace(os.path.join(retry, f), os.path.join(tmp, f))
            print("RCCL ranks: the first attempt timed out, the retry ran\n" + first[-1500:])
    return rcs, logs


# mode 1: regions overlapped with backward (the default), 0: one all-reduce after it; three ranks: shards of 1 / 2 / 2 rows
@pytest.mark.parametrize("world,mode", [(2, 1), (2, 0), (3, 1)])
def test_rccl_ranks_reproduce_the_single_process_step(tmp_path, world, mode):
    import stattn
    from oracle import stattn_oracle_grad as OG
    sys.path.insert(0, HERE)
    import _dp2_worker as W
    tmp = str(tmp_path)
    rcs, logs = _run_ranks(tmp, mode, world)
    if rcs is None:
        pytest.fail("two-rank RCCL run timed out\n" + logs)
    # A communicator that cannot be BUILT on this box (no loopback interface, an RCCL that insists on one rank per device
    # whatever the host id says, ...) is an environment limit, not a defect of the path under test -- but it FAILS unless
    # STATTN_ALLOW_DP2_SKIP=1 asks for the skip.  Anything that
    # goes wrong after ncclCommInitRank succeeded fails the test.
    # A SECOND HIP runtime in a rank ("no ROCm-capable device", two libamdhip64 files mapped) is a defect of the library's
    # loader logic and fails the test, and so does a generic init failure without one of the environment signatures.
    assert "no ROCm-capable device" not in logs and "two HIP runtimes" not in logs, logs
    # (a SILENT skip would hide a broken N > 1 path; the environment excuse must be asked for by name)
    if any(rcs) and ("Duplicate GPU" in logs or "no socket interface" in logs.lower() or "Bootstrap : no" in logs):
        msg = "this RCCL build / box cannot run %d ranks on one GPU over loopback:\n" % world + logs[-800:]
        if os.environ.get("STATTN_ALLOW_DP2_SKIP") == "1":
            pytest.skip(msg)
        pytest.fail(msg + "\n(set STATTN_ALLOW_DP2_SKIP=1 to accept this as an environment limit)")
    assert rcs == [0] * world, logs
    O, opt, P, batch = W.problem()
    r = [np.load(os.path.join(tmp, "rank%d.npz" % i)) for i in range(world)]
    meta = [json.load(open(os.path.join(tmp, "rank%d.json" % i))) for i in range(world)]
    for i in range(world):
        assert meta[i]["comm_info"] == [i, world]
        assert meta[i]["stats"]["ranks"] == world               # what RCCL itself reports (ncclCommCount)
        assert meta[i]["stats"]["overlap"] == mode
        assert meta[i]["stats"]["regions"] == (5 if mode else 0)      # five regions of the flat buffer per pass (round 5: the decoder region in two)
        assert "librccl" in meta[i]["library"]
    # the broadcast made rank 1 start from rank 0's parameters
    for k in P:
        for i in range(world):
            np.testing.assert_array_equal(r[i]["s_" + k], P[k])
    # summed gradient == the oracle's full-batch gradient; both ranks hold the same bits
    ref = OG.loss_and_grads(P, opt, batch, alpha_c=W.ALPHA_C, decay_c=W.DECAY_C)
    ref0 = OG.loss_and_grads(P, opt, batch, alpha_c=W.ALPHA_C, decay_c=0.0, want=('grads',))['grads']
    for k in P:
        for i in range(1, world):
            np.testing.assert_array_equal(r[0]["g_" + k], r[i]["g_" + k])
    bad = []
    for k in P:
        g, t = np.asarray(r[0]["g_" + k], np.float64), np.asarray(ref0[k], np.float64)
        if np.abs(g - t).max() > 1e-4 * np.abs(t).max() + 5e-6:
            bad.append((k, float(np.abs(g - t).max()), float(np.abs(t).max())))
    assert not bad, bad
    np.testing.assert_allclose(meta[0]["loss"], ref['loss'], rtol=2e-4)
    for i in range(1, world):
        np.testing.assert_allclose(meta[i]["loss"], meta[0]["loss"], rtol=1e-6)
    # two steps later the replicas are still bit-identical and equal to a single process that saw the whole batch
    plain = stattn.Decoder(opt, lt_mode=1)
    plain.set_params(P)
    for _ in range(2):
        plain.set_batch(**batch)
        plain.forward_train(); plain.backward(nll_scale=1.0 / W.B_GLOBAL, alpha_c=W.ALPHA_C)
        plain.update(decay_c=W.DECAY_C, clip_c=W.CLIP_C)
    p_ref = plain.get_params()
    moved = 0.0
    for k in P:
        for i in range(1, world):
            np.testing.assert_array_equal(r[0]["p_" + k], r[i]["p_" + k])
        np.testing.assert_allclose(r[0]["p_" + k], p_ref[k], rtol=2e-4, atol=2e-6)
        moved = max(moved, float(np.abs(p_ref[k] - P[k]).max()))
    assert moved > 1e-4                                         # the update did something
