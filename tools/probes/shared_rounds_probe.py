#!/usr/bin/env python
"""configs[4]-shaped beam search with other video counts: attention-launch time per (video, frame) item against the number of rounds of
resident workgroups (3 per CU = 768) -- is the 3.33-round grid of 32 videos paying for its last, third-full round?
usage: shared_rounds_probe.py [nvid ...]   (F is reduced to 512: the projections are not what is measured)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench, stattn

c = dict(bench.CONFIGS["c5"], F=512)
c["T"] = int(os.environ.get("PROBE_T", c["T"])); c["K"] = int(os.environ.get("PROBE_K", c["K"]))      # other grids: STATTN_SHARED_MIN=1 / 100000 forces either kernel
opt = bench.make_options(c)
dec = stattn.Decoder(opt)
dec.set_params(bench.fast_params(dec.param_shapes(), 1234))
for nv in [int(x) for x in sys.argv[1:]] or [24, 26, 29, 32, 35, 38]:
    f = bench.fast_features(nv, c["T"], c["K"], c["F"], c["D"], 777)
    dec.beam_stage(f["ctxg"], f["mask_ctxg"], f["ctxl"], f["ctxm"])
    dec.beam_search(k=5, maxlen=8, suppress_eos=True, resident=True)
    dec.set_profiling(True)
    dec.beam_search(k=5, maxlen=8, suppress_eos=True, resident=True)
    kms = dec.kernel_ms()
    dec.set_profiling(False)
    items = nv * c["T"]
    sp = kms["spatial"][0] * 1e3
    print("videos %2d  items %4d = %.2f rounds of 768  attention %6.1f us = %5.1f ns per item   temporal %5.1f  hproj %5.1f  lstm %5.1f  readout %5.1f" %
          (nv, items, items / 768.0, sp, sp * 1e3 / items, kms["temporal"][0] * 1e3, kms["hproj"][0] * 1e3, kms["lstm"][0] * 1e3, kms["readout"][0] * 1e3), flush=True)
