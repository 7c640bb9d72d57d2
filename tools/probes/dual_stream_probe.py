#!/usr/bin/env python
"""Would running the two halves of a batch as independent scans on two streams overlap the latency chains of the
per-step kernels?  Emulation with two handles (32 rows each, own streams) against one handle with 64 rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import stattn, bench

c = dict(bench.CONFIGS['c2'])
opt = bench.make_options(c)


def make(B, seed):
    cc = dict(c, B=B)
    dec = stattn.Decoder(opt)
    dec.set_params(bench.fast_params(dec.param_shapes(), 1234))
    dec.set_batch(**bench.synthetic_batch(cc, seed))
    dec.set_use_noise(0.0)
    return dec


def timeit(decs, n=20, train=False):
    def step():
        for d in decs:
            d.forward_train()
        if train:
            for d in decs:
                d.backward(nll_scale=1.0 / 64, alpha_c=0.7)
    for _ in range(3):
        step()
    for d in decs:
        d.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    for d in decs:
        d.sync()
    return (time.perf_counter() - t0) / n * 1e3


one = make(64, 1)
print("1 x 64 rows: forward %.3f ms, forward+backward %.3f ms" % (timeit([one]), timeit([one], train=True)))
two = [make(32, 1), make(32, 2)]
print("2 x 32 rows, two streams: forward %.3f ms, forward+backward %.3f ms" % (timeit(two), timeit(two, train=True)))
half = make(32, 1)
print("1 x 32 rows alone: forward %.3f ms, forward+backward %.3f ms" % (timeit([half]), timeit([half], train=True)))
