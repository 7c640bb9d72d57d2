#!/usr/bin/env python
"""One GEMM shape on the LDS-tiled kernel: the target of rocprofv3 --pmc passes.  usage: gemm_one.py M N K [transA transB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
           use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
a = [int(x) for x in sys.argv[1:]]
M, N, K = a[:3]
ta, tb = (a[3:5] + [0, 0])[:2]
ms = dec.time_gemm(M, N, K, iters=5, transA=bool(ta), transB=bool(tb))
print("%d %d %d: %.3f ms %.1f TF" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
