#!/usr/bin/env python
"""Ablation of the register-streaming skinny GEMM (per-step dense work of the decoder)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stattn

def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    dec = stattn.Decoder(opt)
    names = {0: "full", 32: "no A loads", 64: "no B loads", 128: "old kernel (no LDS share)", 16: "full, tile-packed B", 20: "packed, no reduce", 8: "full, prefetch 2", 12: "prefetch 2, no reduce", 1: "loads only", 2: "mfma only", 4: "no reduce", 5: "loads, no reduce", 6: "mfma, no reduce"}
    for (M, N, K, nseg, what) in [(64, 2048, 1024, 4, "h.[Wd*|U]  (4 x 2048 cols)"), (64, 4096, 1024, 1, "N=4096"),
                                  (64, 12032, 512, 1, "a.Wo"), (5, 2048, 1024, 4, "m=5 decode"), (160, 2048, 1024, 4, "M=160")]:
        print("== %s  M=%d N=%d K=%d nseg=%d" % (what, M, N, K, nseg))
        for v in (0, 128, 32, 64, 6):
            ms = dec.time_skinny(M, N, K, nseg=nseg, variant=v)
            fl = 2.0 * M * N * K * nseg
            by = 4.0 * K * N * nseg
            print("   %-18s %7.2f us  %6.1f TFLOP/s  %6.2f TB/s weights" % (names[v], ms * 1e3, fl / ms / 1e9, by / ms / 1e9))

if __name__ == "__main__":
    main()
