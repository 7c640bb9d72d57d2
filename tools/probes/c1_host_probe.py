import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import bench, stattn
c = bench.CONFIGS["c1"]; opt = bench.make_options(c)
dec = stattn.Decoder(opt); P = bench.fast_params(dec.param_shapes(), 1234); dec.set_params(P)
f = bench.fast_features(4, c["T"], c["K"], c["F"], c["D"], 4321)
def t(fn, n=20):
    fn(); dec.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dec.sync(); return (time.perf_counter() - t0) / n * 1e3
one = lambda: dec.beam_search(f["ctxg"][:1], f["mask_ctxg"][:1], f["ctxl"][:1], f["ctxm"][:1], k=1, maxlen=30, suppress_eos=True)
print("host features per call: %.3f ms" % t(one))
dec.beam_stage(f["ctxg"][:1], f["mask_ctxg"][:1], f["ctxl"][:1], f["ctxm"][:1])
res = lambda: dec.beam_search(k=1, maxlen=30, suppress_eos=True, resident=True)
print("resident: %.3f ms" % t(res))
for L in (2, 10, 30):
    r = lambda: dec.beam_search(k=1, maxlen=L, suppress_eos=True, resident=True)
    print("resident maxlen %d: %.3f ms" % (L, t(r)))
stage = lambda: dec.beam_stage(f["ctxg"][:1], f["mask_ctxg"][:1], f["ctxl"][:1], f["ctxm"][:1])
print("beam_stage alone: %.3f ms" % t(stage))
pin = {k: dec.pinned_empty(v[:1].shape) for k, v in f.items()}
for k in pin: pin[k][...] = f[k][:1]
stagep = lambda: dec.beam_stage(pin["ctxg"], pin["mask_ctxg"], pin["ctxl"], pin["ctxm"])
print("beam_stage from pinned arrays: %.3f ms" % t(stagep))
