#!/usr/bin/env python
"""Timeline of beam_update_kernel inside a configs[0] per-video decode (library built with `make -C .../csrc clean all PROBES=1`):
microseconds between the stamps  start | scalars read | log-sum-exp done | candidates listed | selected | bookkeeping done |
gathers done | end.  usage: beam_probe.py [k]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import stattn
from stattn import _native

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = _native.load_library()
c = bench.CONFIGS["c1"]
dec = stattn.Decoder(bench.make_options(c))
dec.set_params(bench.fast_params(dec.param_shapes(), 1234))
f = bench.fast_features(1, c["T"], c["K"], c["F"], c["D"], 4321)
dec.beam_stage(f["ctxg"], f["mask_ctxg"], f["ctxl"], f["ctxm"])
dec.beam_search(k=k, maxlen=30, suppress_eos=True, resident=True)
assert lib.stattn_probe_beam(None) == 0
rows = []
for _ in range(20):
    dec.beam_search(k=k, maxlen=30, suppress_eos=True, resident=True)
    out = (C.c_longlong * 8)()
    assert lib.stattn_probe_beam(out) == 0
    st = np.array(list(out), np.float64) / 100.0
    rows.append(np.diff(st))
m = np.median(np.array(rows), axis=0)
names = ("scalars", "lse", "candidates", "select", "bookkeeping", "gathers", "ticket")
print("beam_update k=%d: " % k + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, m)) + "  | total %.2f us" % m.sum())
