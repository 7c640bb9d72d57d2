#!/usr/bin/env python
"""LDS tile size sweep of the bf16-MFMA GEMM (BASELINE.json configs[3]) on the MSR-VTT-shape projections, the configs[1]
projection and square reference problems: TFLOP/s per (shape, workgroup tile), uniform [-1, 1) operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stattn

SHAPES = [  # C4: B=64, T=40, K=16, F=2048, D=1024
    ("ff_local  C4", 40960, 1024, 2048),
    ("PL / LW   C4", 40960, 1024, 1024),
    ("PL | LW   C4", 40960, 2048, 1024),
    ("ff_motion C4", 2560, 1024, 2048),
    ("pctxg/m   C4", 2560, 1024, 1024),
    ("xproj     C4", 1920, 4096, 512),
    ("readout   C4", 1920, 512, 2048),
    ("logits    C4", 1920, 12032, 512),
    ("ff_local  C2", 13312, 1024, 4096),
    ("PL | LW   C2", 13312, 2048, 1024),
    ("square 4096", 4096, 4096, 4096),
    ("square 8192", 8192, 8192, 8192),
]
TILES = [(11, "64x64"), (22, "128x128"), (84, "256x128g"), (88, "256x256p"), (0, "auto")]


def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    dec = stattn.Decoder(opt)
    dec.time_gemm_bf16(4096, 4096, 4096, 0, iters=10)      # warm the clocks
    print("8ph variant %s" % os.environ.get("STATTN_8PH_VAR", "0"))
    print("%-14s %6s %6s %6s | %s" % ("shape", "M", "N", "K", "  ".join("%9s" % n for _, n in TILES)))
    for name, M, N, K in SHAPES:
        row = []
        for tile, _ in TILES:
            try:
                ms = min(dec.time_gemm_bf16(M, N, K, tile, iters=20) for _ in range(2))
                row.append("%6.0f TF" % (2.0 * M * N * K / ms / 1e9))
            except ValueError:
                row.append("      n/a")          # tile needs an edge-free shape
        print("%-14s %6d %6d %6d | %s" % (name, M, N, K, "  ".join(row)), flush=True)


if __name__ == "__main__":
    main()
