#!/usr/bin/env python
"""Time the LDS-tiled fp32 MFMA GEMM on the shapes the decoder uses (C2 config) -- TFLOP/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn

SHAPES = [  # (name, M, N, K, transA, transB)
    ("ff_local      NN", 13312, 1024, 4096, 0, 0),
    ("PL / LW       NN", 13312, 1024, 1024, 0, 0),
    ("CL.Wclt       NN", 1664, 1024, 1024, 0, 0),
    ("xproj         NN", 1920, 4096, 512, 0, 0),
    ("logits        NN", 1920, 12032, 512, 0, 0),
    ("dW_local      TN", 4096, 1024, 13312, 1, 0),
    ("dWcl (splitK) TN", 1024, 1024, 13312, 1, 0),
    ("dU            TN", 1024, 4096, 1920, 1, 0),
    ("dWo           TN", 512, 12032, 1920, 1, 0),
    ("dL+=dPL.WclT  NT", 13312, 1024, 1024, 0, 1),
    ("da=dlogit.WoT NT", 1920, 512, 12032, 0, 1),
    ("square 4096   NN", 4096, 4096, 4096, 0, 0),
]

def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    dec = stattn.Decoder(opt)
    for name, M, N, K, ta, tb in SHAPES:
        ms = dec.time_gemm(M, N, K, iters=20, transA=bool(ta), transB=bool(tb))
        print("%-18s M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))

if __name__ == "__main__":
    main()
