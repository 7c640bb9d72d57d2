#!/usr/bin/env python
"""fp32-MFMA GEMM against the split (three-term bf16) GEMM on the decoder's C2 shapes: ms, TFLOP/s of fp32 work."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
from gemm_probe import SHAPES

def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    decs = [stattn.Decoder(opt), stattn.Decoder(opt, precision="split")]
    print("%-18s %6s %6s %6s | %9s %9s | %9s %9s" % ("shape", "M", "N", "K", "fp32 ms", "TF", "split ms", "TF"))
    for name, M, N, K, ta, tb in SHAPES:
        if N % 128:
            continue
        r = []
        for d in decs:
            ms = d.time_gemm(M, N, K, iters=20, transA=bool(ta), transB=bool(tb))
            r += [ms, 2.0 * M * N * K / ms / 1e9]
        print("%-18s %6d %6d %6d | %9.3f %9.1f | %9.3f %9.1f" % ((name, M, N, K) + tuple(r)))

if __name__ == "__main__":
    main()
