#!/usr/bin/env python
"""Launch time of the register-streaming 64-column skinny GEMM (the <= 16-row sampler path) per shape.
(The ablations of round 1 lived in the kernel; the row-panel kernels' are in tools/probes/panel_probe.hip.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn

def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    dec = stattn.Decoder(opt)
    for (M, N, K, nseg, what) in [(1, 2048, 1024, 4, "m=1 decode: h.[Wd*|U]"), (5, 2048, 1024, 4, "m=5 decode"),
                                  (16, 2048, 1024, 4, "m=16"), (5, 12032, 512, 1, "a.Wo, m=5")]:
        ms = dec.time_skinny(M, N, K, nseg=nseg, variant=0)
        print("%-24s M=%3d N=%5d K=%4d nseg=%d  %7.2f us  %6.2f TB/s of weights" % (what, M, N, K, nseg, ms * 1e3, 4.0 * K * N * nseg / ms / 1e9))

if __name__ == "__main__":
    main()
