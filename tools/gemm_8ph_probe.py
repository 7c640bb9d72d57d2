#!/usr/bin/env python
"""Time model of the 256 x 256 eight-phase bf16 GEMM: whole rounds of 256 tiles (M = 16384 r, N = 1024) at several K, so that
us per tile = a + b * (K / 64) can be read off; STATTN_8PH_VAR selects an ablation (4 no epilogue, 12 + no DMA, 20 + no
fragment reads, 28 MFMAs only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stattn


def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    dec = stattn.Decoder(opt)
    dec.time_gemm_bf16(4096, 4096, 4096, 0, iters=10)
    tile = int(os.environ.get("PROBE_TILE", "88"))
    print("variant %s tile %d" % (os.environ.get("STATTN_8PH_VAR", "0"), tile))
    for rounds in (1, 4):
        M = 16384 * rounds
        row = []
        for K in (512, 1024, 2048, 4096):
            ms = min(dec.time_gemm_bf16(M, 1024, K, tile, iters=20) for _ in range(3))
            row.append("K=%d: %6.1f us/tile %5.0f TF" % (K, ms * 1e3 / rounds, 2.0 * M * 1024 * K / ms / 1e9))
        print("rounds %d | %s" % (rounds, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
