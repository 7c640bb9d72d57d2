#!/usr/bin/env python
"""bf16 GEMM with transposed operands (through the transposing conversion kernels) and with split-K: float64 on the
bf16-rounded operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stattn


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def main():
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True, use_dropout=True, prev2out=True, ctx2out=True)
    dec = stattn.Decoder(opt)
    bad = 0
    for M, N, K, tA, tB, kind in [(512, 256, 1920, True, False, 2), (1024, 1024, 4096, True, False, 7), (256, 512, 640, True, True, 2),
                                  (2048, 1024, 8192, True, False, 7), (1920, 512, 12032, False, True, 7), (520, 256, 384, True, False, 7),
                                  (1024, 2048, 40960, True, False, 7)]:
        rng = np.random.RandomState(M + N + K)
        A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
        B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        Am = A.T if tA else A
        Bm = B.T if tB else B
        ref = bf16_round(Am).astype(np.float64) @ bf16_round(Bm).astype(np.float64) + bias
        got = dec.gemm(A, B, bias=bias, kind=kind, transA=tA, transB=tB)
        e = np.abs(got - ref).max()
        ok = e < 2e-6 * K + 1e-5
        print("%5d x %5d x %5d tA=%d tB=%d kind=%d  err %.2e  %s" % (M, N, K, tA, tB, kind, e, "ok" if ok else "FAIL"), flush=True)
        bad += not ok
    print("FAILED %d" % bad if bad else "ALL OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
