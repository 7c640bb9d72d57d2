#!/usr/bin/env python
"""Isolated GEMM launches under the clock-probe build (make PROBES=1 / tools/probes/build_gemm_var.sh probe -DSTATTN_PROBES, STATTN_GEMM_CLK=1):
block-0 duration and shader clock per launch, next to the launch time by events.  usage: gemm_clk.py M N K tA tB [M N K tA tB ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
           use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
a = [int(x) for x in sys.argv[1:]]
for i in range(0, len(a), 5):
    M, N, K, ta, tb = a[i:i + 5]
    ms = dec.time_gemm(M, N, K, iters=6, transA=bool(ta), transB=bool(tb))
    sys.stderr.flush()
    print("%s%s %dx%dx%d: %.1f us by events, %.1f TF" % ("T" if ta else "N", "T" if tb else "N", M, N, K, 1e3 * ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    dec._lib.stattn_sync(dec._h)
