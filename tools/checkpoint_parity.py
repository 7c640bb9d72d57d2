#!/usr/bin/env python
"""Parity of the HIP path on a checkpoint in the REFERENCE's format (GPU).

    tools/checkpoint_parity.py model_best_so_far.npz [model_options.pkl] [--videos 4] [--frames 28] [--regions 8] [--steps 6]

The reference saves `numpy.savez(path, history_errs=..., **params)` (model_attention.py:1488-1490) and reloads it with
`load_params` (:1109-1113); its pretrained weights are an external download (README.md:53) that exists on neither box.  A user who
has the file runs this: the archive goes through the product's own reload path (`Attention.load_params` -> `init_tparams`), synthetic
features of the checkpoint's dimensions are decoded with `f_init` / `f_next` and scored with the training graph, and every quantity
north_star names is compared with the float64 oracle on the same weights: attention weights and logits 1e-4 absolute, state 1e-4,
cost 1e-4 relative.  Also reports how peaked the checkpoint's attention is (the mean largest weight of each softmax).
Options come from model_options.pkl when given (the reference pickles them next to the checkpoint, :1084-1086), otherwise they are
read off the array shapes.  `run(path)` is what tests/test_gpu_z2_trained_like.py calls on a synthetic archive."""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def options_from_archive(npz, pkl=None):
    from oracle import stattn_oracle as O
    if pkl:
        with open(pkl, "rb") as f:
            o = pickle.load(f, encoding="latin1")
        o = o.get("attention", o) if isinstance(o, dict) else o
    else:
        o = {}
    V, E = npz["Wemb"].shape
    D = npz["decoder_U"].shape[0]
    dims = dict(dim=D, dim_word=E, n_words=V, ctxg_dim=npz["ff_state_W"].shape[0], ctxglm_dim=npz["decoder_Wc"].shape[0],
                ctxl_dim=npz["ff_local_W"].shape[0], ctxm_dim=npz["ff_motion_W"].shape[0],
                selector="decoder_W_sel" in npz.files, ctx2out="ff_logit_ctxglm_W" in npz.files, prev2out=bool(o.get("prev2out", True)))
    return O.default_options(**dims)


def run(path, pkl=None, videos=3, frames=9, regions=5, steps=5, seed=5, precision="fp32", verbose=True):
    import stattn
    from oracle import stattn_oracle as O
    npz = np.load(path)
    opt = options_from_archive(npz, pkl)
    model = stattn.Attention()
    stattn.common.reset_rngs(1234)
    params = model.load_params(path, model.init_params(opt))          # the reference's reload sequence (:1105-1113)
    P = {k: np.asarray(v, np.float32) for k, v in params.items()}
    P64 = O.cast_params(P, np.float64)
    batch = O.synthetic_batch(opt, B=videos, T=frames, K=regions, t=steps, seed=seed)
    ref = O.build_model_forward(P64, opt, **{k: (v if v.dtype == np.int64 else v.astype(np.float64)) for k, v in batch.items()})
    peak = {k: float(ref[k].max(axis=-1).mean()) for k in ("alphal", "alphag", "alpham", "alphalt")}
    tparams = model.init_tparams(params)
    f_init, f_next = model.build_sampler(tparams, opt, None, None)
    dec = f_next.decoder if precision == "fp32" else stattn.Decoder(opt, precision=precision)
    if precision != "fp32":
        dec.set_params(P)
    # (1) the training graph
    dec.set_batch(**batch)
    dec.forward_train()
    out = dec.get_forward(logits=True)
    err = {k: float(np.abs(out[k] - ref[k]).max()) for k in ("alphal", "alphag", "alpham", "alphalt")}
    err["logit"] = float(np.abs(out["logit"] - ref["logit"].reshape(out["logit"].shape)).max())
    err["cost_rel"] = float(np.abs(out["cost"] / ref["cost"] - 1.0).max())
    # (2) the sampler: f_init and a chain of f_next calls on video 0 with teacher-forced words
    if precision == "fp32":
        g, l, m, gm = batch["ctxg"][0], batch["ctxl"][0], batch["ctxm"][0], batch["mask_ctxg"][0]
        _, h, c = f_init(g, gm)
        _, hr, cr = O.f_init(P64, opt, g.astype(np.float64), gm.astype(np.float64))
        err["f_init"] = float(max(np.abs(h - hr).max(), np.abs(c - cr).max()))
        h, c, hr, cr = h[None], c[None], hr[None], cr[None]
        x = np.array([-1], np.int64)
        worst = 0.0
        for s in range(steps):
            (pr, _, h, c), ex = dec.f_next(x, g, gm, l, None, m, None, h, c, extras=True)
            (prr, _, hr, cr), rx = O.f_next(P64, opt, x, g.astype(np.float64), gm, l.astype(np.float64), None, m.astype(np.float64), None, hr, cr, extras=True)
            worst = max(worst, float(np.abs(pr - prr).max()), float(np.abs(h - hr).max()), float(np.abs(c - cr).max()),
                        float(np.abs(ex["logit"] - rx["logit"]).max()), *(float(np.abs(ex[k] - rx[k]).max()) for k in ("alphal", "alphag", "alpham", "alphalt")))
            x = np.array([int(batch["x"][s, 0])], np.int64)
        err["f_next_chain"] = worst
    if verbose:
        print("checkpoint %s: D=%d E=%d V=%d Fl=%d Fm=%d selector=%d ctx2out=%d" % (path, opt["dim"], opt["dim_word"], opt["n_words"], opt["ctxl_dim"],
              opt["ctxm_dim"], opt["selector"], opt["ctx2out"]))
        print("  attention peakedness in the float64 oracle (mean largest weight): " + "  ".join("%s %.2f" % kv for kv in peak.items()) +
              "   max |logit| %.1f" % float(np.abs(ref["logit"]).max()))
        print("  HIP (%s) against the oracle: " % precision + "  ".join("%s %.2e" % kv for kv in err.items()))
    return err, peak


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("npz"); ap.add_argument("pkl", nargs="?")
    ap.add_argument("--videos", type=int, default=4); ap.add_argument("--frames", type=int, default=28)
    ap.add_argument("--regions", type=int, default=8); ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    err, _ = run(a.npz, a.pkl, a.videos, a.frames, a.regions, a.steps)
    bad = {k: v for k, v in err.items() if v >= 1e-4}
    print("PASS: every quantity within 1e-4" if not bad else "FAIL: %s" % bad)
    sys.exit(1 if bad else 0)
