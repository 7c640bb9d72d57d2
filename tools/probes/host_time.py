#!/usr/bin/env python
"""Host-side enqueue time of forward_train / backward / update at configs[1] (no synchronisation inside the calls):
how far ahead of the GPU the launching thread runs.  usage (GPU box): python tools/probes/host_time.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench as B
import stattn
c = B.CONFIGS['c2']; options = B.make_options(c)
dec = stattn.Decoder(options, device=0, lt_mode=1)
params = B.fast_params(dec.param_shapes(), 1234); dec.set_params(params)
batch = B.synthetic_batch(c, 1234); dec.set_batch(**batch); dec.set_use_noise(1.0)
for _ in range(3):
    dec.forward_train(); dec.backward(nll_scale=1/64., alpha_c=0.7); dec.update(decay_c=1e-4, clip_c=10.0)
dec.sync()
tf=tb=tu=0; N=10
t_all0=time.perf_counter()
for _ in range(N):
    t0=time.perf_counter(); dec.forward_train(); t1=time.perf_counter(); dec.backward(nll_scale=1/64., alpha_c=0.7); t2=time.perf_counter(); dec.update(decay_c=1e-4, clip_c=10.0); t3=time.perf_counter()
    tf+=t1-t0; tb+=t2-t1; tu+=t3-t2
dec.sync(); t_all=time.perf_counter()-t_all0
print("host enqueue per step: forward %.2f ms, backward %.2f ms, update %.2f ms; wall per step %.2f ms" % (tf/N*1e3, tb/N*1e3, tu/N*1e3, t_all/N*1e3))
# the same but syncing before each step so the host starts with an empty queue
tf=tb=0
for _ in range(N):
    dec.sync(); t0=time.perf_counter(); dec.forward_train(); t1=time.perf_counter(); dec.sync(); t1b=time.perf_counter(); dec.backward(nll_scale=1/64., alpha_c=0.7); t2=time.perf_counter(); dec.update(decay_c=1e-4, clip_c=10.0)
    tf+=t1-t0; tb+=t2-t1b
print("from an empty queue: forward enqueue %.2f ms, backward enqueue %.2f ms" % (tf/N*1e3, tb/N*1e3))
