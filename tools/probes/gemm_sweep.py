#!/usr/bin/env python
"""Sweep M (rows) at fixed N, K for the LDS-tiled GEMM: shows how the rate depends on how many output tiles there are
per resident workgroup.  usage: gemm_sweep.py N K M1 M2 ...   (env STATTN_GEMM_TILE / _NOSK / _SK select variants)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
           use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
N, K = int(sys.argv[1]), int(sys.argv[2])
for M in [int(x) for x in sys.argv[3:]]:
    ms = dec.time_gemm(M, N, K, iters=20)
    print("M=%6d N=%5d K=%5d  tiles128/512=%.3f  %.3f ms %.1f TF" % (M, N, K, (M / 128) * (N / 128) / 512, ms, 2.0 * M * N * K / ms / 1e9))
