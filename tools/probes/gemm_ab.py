#!/usr/bin/env python
"""Times a fixed list of GEMM shapes on the LDS-tiled fp32 kernel (one process per tile setting because the
STATTN_GEMM_TILE switch is read once).  usage: gemm_ab.py [iters]  -- driven by tools/probes/gemm_ab.sh over tools/_var/*.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
           use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tile = os.environ.get("STATTN_GEMM_TILE", "auto")
shapes = [(4096, 4096, 512, 0, 0), (4096, 4096, 1024, 0, 0), (4096, 4096, 2048, 0, 0), (4096, 4096, 4096, 0, 0), (4096, 4096, 8192, 0, 0),
          (13312, 1024, 4096, 0, 0), (13312, 1024, 1024, 0, 0), (16384, 1024, 4096, 0, 0), (1920, 12032, 512, 0, 0),
          (4096, 1024, 13312, 1, 0), (1024, 1024, 13312, 1, 0), (13312, 1024, 2048, 0, 1)]
out = []
for M, N, K, ta, tb in shapes:
    ms = min(dec.time_gemm(M, N, K, iters=it, transA=bool(ta), transB=bool(tb)) for _ in range(3))
    out.append("%s%s %dx%dx%d %.1f" % ("T" if ta else "N", "T" if tb else "N", M, N, K, 2.0 * M * N * K / ms / 1e9))
print("tile=%s | " % tile + " | ".join(out))
