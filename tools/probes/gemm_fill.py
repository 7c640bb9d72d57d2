#!/usr/bin/env python
"""128 x 128 tiles per CU against the rate: 256 / 512 / 768 / 1024 / 1536 / 2048 tiles of K = 4096 (one, two, ... workgroups per CU and round).
usage: STATTN_GEMM_TILE=22 gemm_fill.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
           use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
for K in (4096, 1024):
    for M, N in ((1024, 2048), (2048, 2048), (3072, 2048), (4096, 2048), (5120, 2048), (6144, 2048), (8192, 2048), (4096, 6144), (8192, 4096)):
        ms = min(dec.time_gemm(M, N, K, iters=10) for _ in range(3))
        t = (M // 128) * (N // 128)
        print("K=%5d tiles %5d (%.2f per CU)  %8.1f us  %6.1f TF" % (K, t, t / 256.0, 1e3 * ms, 2.0 * M * N * K / ms / 1e9))
