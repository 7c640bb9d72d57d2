#!/usr/bin/env python
"""Would two half-batches on two HIP streams beat one batch on one?  (probe, not product)

The train step is a chain of launches that each leave most of the chip idle in a different way: the attention kernels are
HBM-bound, the row-panel recurrent GEMMs are latency chains at 16 % MFMA utilisation, the LDS-tiled GEMMs are MFMA-bound.
Two independent decoders with half the rows each, driven from two host threads on their own streams, let the hardware overlap
them.  This prints row-steps/s of: one decoder with B rows; one decoder with B/2 rows alone; two decoders with B/2 rows
each running concurrently (aggregate).  usage: python tools/probes/two_lane_probe.py [--config c2] [--steps 20]
"""
import argparse, importlib, os, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

stattn = importlib.import_module("video-description-with-spatial-temporal-attention_amd")
dp = importlib.import_module("video-description-with-spatial-temporal-attention_amd.dp")


def make(c, seed, precision):
    options = bench.make_options(c)
    dec = stattn.Decoder(options, device=0, precision=precision)
    dec.set_params(bench.fast_params(dec.param_shapes(), 1234))
    dec.set_batch(**bench.synthetic_batch(c, seed))
    dec.set_use_noise(1.0)
    dp.init_comm(dec, 0, 1)
    return dec, dp.DataParallelStep(dec, global_batch=c["B"], alpha_c=0.70602, decay_c=1e-4, clip_c=10.0)


def timed(steps_fns, decs, steps, warmup):
    def run(fn, n):
        for _ in range(n):
            fn()
    for fn in steps_fns:
        run(fn, warmup)
    for d in decs:
        d.sync()
    go = threading.Barrier(len(steps_fns) + 1)

    def worker(fn, d):
        go.wait()
        run(fn, steps)
        d.sync()
    th = [threading.Thread(target=worker, args=(fn, d)) for fn, d in zip(steps_fns, decs)]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--lanes", type=int, default=2)
    a = ap.parse_args()
    c = dict(bench.CONFIGS[a.config])
    B, t = c["B"], c["t"]
    d1, s1 = make(c, 1, a.precision)
    dt = timed([s1], [d1], a.steps, a.warmup)
    print("one decoder, %d rows           : %.3f ms/step  %.1f k row-steps/s" % (B, dt / a.steps * 1e3, B * t * a.steps / dt / 1e3))
    del s1, d1
    ch = dict(c, B=B // a.lanes)
    lanes = [make(ch, 10 + i, a.precision) for i in range(a.lanes)]
    dt = timed([lanes[0][1]], [lanes[0][0]], a.steps, a.warmup)
    print("one decoder, %d rows alone     : %.3f ms/step  %.1f k row-steps/s" % (ch["B"], dt / a.steps * 1e3, ch["B"] * t * a.steps / dt / 1e3))
    dt = timed([l[1] for l in lanes], [l[0] for l in lanes], a.steps, a.warmup)
    print("%d decoders x %d rows, %d threads: %.3f ms/step  %.1f k row-steps/s (aggregate)" %
          (a.lanes, ch["B"], a.lanes, dt / a.steps * 1e3, B * t * a.steps / dt / 1e3))


if __name__ == "__main__":
    main()
