import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, stattn
def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True, use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
for M, N, K in [(1872, 512, 256), (300, 256, 128), (2048, 2048, 1024), (40960, 2048, 1024), (513, 1024, 192)]:
    rng = np.random.RandomState(1)
    A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = bf16_round(A).astype(np.float64) @ bf16_round(B.T).astype(np.float64) + bias
    got = dec.gemm(A, B, bias=bias, kind=6, transB=True)
    e = np.abs(got - ref)
    print(M, N, K, "err", e.max(), "first half", e[:, :N//2].max(), "second", e[:, N//2:].max())
