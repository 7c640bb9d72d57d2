import sys; sys.path.insert(0, "/root/repo")
import stattn
opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True, use_dropout=True, prev2out=True, ctx2out=True)
dec = stattn.Decoder(opt)
dec.time_gemm(4096, 4096, 1024, iters=5)
for M, N, K in [(160, 12032, 512), (160, 20096, 512), (320, 20096, 512), (160, 8192, 1024), (320, 8192, 1024), (640, 20096, 512), (640, 8192, 1024)]:
    ms = min(dec.time_gemm(M, N, K, iters=30) for _ in range(3))
    print(M, N, K, "%.1f us  %.1f TF" % (ms * 1e3, 2.0 * M * N * K / ms / 1e9))
