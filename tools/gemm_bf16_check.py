#!/usr/bin/env python
"""Correctness + race screen of one bf16 GEMM tile kernel (STATTN_BF16_TILE, default 88 = the 256 x 256 eight-phase kernel):
float64 on the bf16-rounded operands, asymmetric shapes (transpose / k-mapping detecting), M edges, odd / single K-tile
counts, fused epilogue, and bitwise run-to-run equality on a chip-filling shape (a DMA / fragment-read race shows as a
differing word)."""
import os, sys
os.environ.setdefault("STATTN_BF16_TILE", "88")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stattn


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


SHAPES = [(256, 256, 64, False), (256, 256, 128, True), (512, 512, 192, False), (300, 256, 128, False), (257, 512, 320, True),
          (1000, 256, 640, True), (1024, 1024, 512, True), (2000, 768, 1024, False), (4096, 1024, 2048, True),
          (8192, 2048, 1024, True)]


def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    dec = stattn.Decoder(opt)
    bad = 0
    for M, N, K, transB in SHAPES:
        rng = np.random.RandomState(M + N + K)
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K) if transB else (K, N)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        add = rng.standard_normal((M, N)).astype(np.float32)
        Bm = B.T if transB else B
        ref = bf16_round(A).astype(np.float64) @ bf16_round(Bm).astype(np.float64)
        got = dec.gemm(A, B, kind=2, transB=transB)
        e1 = np.abs(got - ref).max()
        got2 = dec.gemm(A, B, bias=bias, add=add, act=1, kind=2, transB=transB)
        e2 = np.abs(got2 - np.tanh(ref + bias + add)).max()
        ok = e1 < 2e-6 * K + 1e-5 and e2 < 2e-5 * max(1, K / 64)
        nrep = 6 if M * N >= 1 << 22 else 2
        same = all(np.array_equal(dec.gemm(A, B, kind=2, transB=transB), got) for _ in range(nrep))
        print("%5d x %5d x %5d transB=%d  err %.2e  epi err %.2e  repeat-equal %s  %s" % (M, N, K, transB, e1, e2, same, "ok" if ok and same else "FAIL"), flush=True)
        bad += not (ok and same)
    print("FAILED %d" % bad if bad else "ALL OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
