#!/usr/bin/env python
"""Print the per-kernel-class microseconds of bench.py's beam_c5 / decode_c1 legs (A/B switches via the environment).
usage: leg_probe.py [c5|c1]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
args = argparse.Namespace(lt_mode=None, no_cpu_baseline=True)
if which == "c5":
    import stattn
    c = bench.CONFIGS["c2"]
    dec = stattn.Decoder(bench.make_options(c))
    params = bench.fast_params(dec.param_shapes(), 1234)
    del dec
    r = bench.leg_beam_c5(args, 0, params)
    print("value %.0f row-steps/s  %.2f ms per call  %.1f us per word (events %.1f)  projections %.2f ms" %
          (r["value"], r["ms_per_call"], r["us_per_word"], r["word_us_by_events"], r["projections_ms_per_call"]))
    print("  spatial %.1f us (%.2f of HBM)" % (r["roofline_hbm"]["ms_per_launch"] * 1e3, r["roofline_hbm"]["frac"]))
    for k, v in r["kernels"].items():
        print("  %-24s %7.1f us  %s" % (k, v["ms_per_launch"] * 1e3, ("%.2f of %s" % (v["frac"], v["bound"])) if v.get("frac") else ""))
else:
    r = bench.leg_decode_c1(args, 0)
    for k in ("k1", "k5"):
        print(k, "%.0f row-steps/s  %.2f ms per video  %.1f us per word  roofline %.3f" %
              (r[k]["value"], r[k]["ms_per_video"], r[k]["us_per_word"], r[k]["roofline"]["frac"]))
    print("batched4_k1", r["batched4_k1"])
