#!/usr/bin/env python
"""One skinny-GEMM shape, product kernel only: the target of rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
           use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
M, N, K, nseg = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 2048, 1024, 4))]
print(dec.time_skinny(M, N, K, nseg=nseg, variant=0, iters=20) * 1e3, "us")
