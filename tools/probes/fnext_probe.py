import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, stattn, bench
c = bench.CONFIGS['c1']; opt = bench.make_options(c)
dec = stattn.Decoder(opt); P = bench.fast_params(dec.param_shapes(), 1); dec.set_params(P)
b = bench.synthetic_batch(c, 3)
g,l,m,gm = b['ctxg'][0], b['ctxl'][0], b['ctxm'][0], b['mask_ctxg'][0]
_, h0, c0 = dec.f_init(g, gm)
for mm in (1, 5):
    x = np.zeros(mm, np.int64) + 5; h = np.tile(h0, (mm,1)); cc = np.tile(c0, (mm,1))
    for _ in range(5): dec.f_next(x, g, gm, l, None, m, None, h, cc)
    t0 = time.perf_counter(); N=200
    for _ in range(N): dec.f_next(x, g, gm, l, None, m, None, h, cc)
    print("m=%d f_next: %.1f us/call" % (mm, (time.perf_counter()-t0)/N*1e6))
p = np.random.rand(5, 12000).astype(np.float32)
t0 = time.perf_counter()
for _ in range(200): (0 - np.log(p)).flatten().argsort()[:5]
print("host log+argsort(5x12000): %.1f us" % ((time.perf_counter()-t0)/200*1e6))
