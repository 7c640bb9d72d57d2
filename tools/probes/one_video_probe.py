import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, stattn
c = bench.CONFIGS["c1"]; opt = bench.make_options(c)
dec = stattn.Decoder(opt); dec.set_params(bench.fast_params(dec.param_shapes(), 1234))
f = bench.fast_features(4, c["T"], c["K"], c["F"], c["D"], 4321)
for i in range(6):
    dec.beam_search(f["ctxg"][:1], f["mask_ctxg"][:1], f["ctxl"][:1], f["ctxm"][:1], k=1, maxlen=30, suppress_eos=True)
dec.sync()
