#!/usr/bin/env python
"""Randomised parity sweep (GPU): random decoder shapes / option flags / batch shapes, forward quantities and all
gradients of a handle against the float64 / autograd oracle at the fp32 bar (1e-4) for precision fp32 and split and
both lt_modes, and -- every third case -- a bf16 handle (lt_mode 1) at the bf16 bars of tests/test_gpu_bf16.py (attention
weights 3e-3 on random shapes, logits 3e-2 or 1 % of the largest, gradients 5 % of their scale).  usage: fuzz_parity.py [n_cases] [seed] | large [n] [seed] | trained [n] [seed] | beam [n] [seed].  Prints one line per case and the worst ratios; exit code 1 on
a violation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stattn
from oracle import stattn_oracle as O
from oracle import stattn_oracle_grad as OG


def trained_like_scales(opt, rng):
    """Weight scales of a decoder AFTER training (VERDICT r04 'missing' 4: no reference checkpoint exists to load, and attention
    after training is far peakier than under random_params): scorer vectors U*_att 4-12 x larger (softmax inputs of standard
    deviation 3-8: the largest weight of a frame / region softmax is typically 0.6-0.99), recurrent and gate weights 1.5-3 x
    (saturating gates), a sharper vocabulary projection, a selector that is not centred on 1/2."""
    D = opt['dim']
    s = {}
    for k in ('Ug_att', 'Um_att', 'Ult_att', 'Ul_att'):
        s['decoder_' + k] = float(rng.uniform(4.0, 12.0)) / np.sqrt(D)
    for k in ('U', 'Wc'):
        s['decoder_' + k] = float(rng.uniform(1.5, 3.0)) / np.sqrt(D)
    s['decoder_W'] = float(rng.uniform(1.5, 3.0)) / np.sqrt(opt['dim_word'])
    s['decoder_W_sel'] = float(rng.uniform(2.0, 6.0)) / np.sqrt(D)
    s['ff_logit_W'] = float(rng.uniform(6.0, 14.0)) / np.sqrt(opt['dim_word'])
    s['ff_logit_b'] = 1.0
    return s


def run(n, seed, large=False, trained=False):
    """large: production-sized dimensions (D up to 1024, vocabulary up to 12 000, up to 64 rows): the row-panel kernels,
    the big GEMM tiles and the split softmax paths that the small cases do not reach."""
    rng = np.random.RandomState(seed)
    worst_f, worst_g, bad = 0.0, 0.0, 0
    for case in range(n):
        if large:
            D = int(rng.choice([512, 768, 1024]))
            dims = dict(dim=D, ctxg_dim=D, ctxglm_dim=D, dim_word=int(rng.choice([256, 512])),
                        n_words=int(rng.randint(3000, 12001)), ctxl_dim=int(64 * rng.randint(4, 33)), ctxm_dim=int(64 * rng.randint(4, 33)),
                        selector=bool(rng.randint(2)), prev2out=bool(rng.randint(2)), ctx2out=bool(rng.randint(2)))
            B, T, K, t = int(rng.randint(17, 65)), int(rng.randint(4, 27)), int(rng.randint(2, 13)), int(rng.randint(3, 8))
            if case % 4 == 3:          # more than 64 rows: the forward recurrent GEMMs run on the wide row-panel kernels (panelw.hip)
                B, T = int(rng.randint(65, 161)), int(rng.randint(3, 9))
        else:
            D = int(rng.choice([64, 128, 192, 256, 320]))
            dims = dict(dim=D, ctxg_dim=D, ctxglm_dim=D, dim_word=int(rng.choice([64, 128, 192])),
                        n_words=int(rng.randint(20, 1500)), ctxl_dim=int(32 * rng.randint(1, 12)), ctxm_dim=int(32 * rng.randint(1, 12)),
                        selector=bool(rng.randint(2)), prev2out=bool(rng.randint(2)), ctx2out=bool(rng.randint(2)))
            B, T, K, t = int(rng.randint(1, 40)), int(rng.randint(1, 30)), int(rng.randint(1, 20)), int(rng.randint(2, 9))
        lt_mode = int(rng.randint(2))
        precision = ["fp32", "split", "bf16"][case % 3]
        if precision == "bf16":
            lt_mode = 1
            if large and case % 2:
                D = 1024; dims.update(dim=D, ctxg_dim=D, ctxglm_dim=D)     # the D % 1024 == 0 kernel with the rider
        # (bf16: tests/test_gpu_bf16.py holds the BASELINE shapes to 2e-3 on the attention weights; random shapes with two or three
        # regions have weights near 1/2, where the same relative error is 2.0e-3 absolute: 3e-3 here)
        bar_a, bar_l, bar_g = (3e-3, 3e-2, 5e-2) if precision == "bf16" else (1e-4, 1e-4, 1e-4)
        opt = O.default_options(**dims)
        only = os.environ.get("FUZZ_ONLY")          # replay of one case, every gradient's error listed (and an fp32 handle's beside a bf16 one)
        if only is not None and case != int(only):
            rng.randint(1 << 30); rng.randint(1 << 30); rng.choice([0.0, 0.70602])      # (the draws of a case that is not run)
            continue
        P = O.random_params(opt, seed=int(rng.randint(1 << 30)), dtype=np.float32, scale=trained_like_scales(opt, rng) if trained else None)
        batch = O.synthetic_batch(opt, B=B, T=T, K=K, t=t, seed=int(rng.randint(1 << 30)))
        dec = stattn.Decoder(opt, lt_mode=lt_mode, precision=precision)
        dec.set_params(P)
        dec.set_batch(**batch)
        dec.forward_train()
        out = dec.get_forward(logits=True)
        ref = O.build_model_forward(O.cast_params(P, np.float64), opt,
                                    **{k: (v if v.dtype == np.int64 else v.astype(np.float64)) for k, v in batch.items()})
        ef = max(np.abs(out[k] - ref[k]).max() for k in ('alphal', 'alphag', 'alpham', 'alphalt')) / bar_a
        # (bf16: the 3e-2 bar is for logits of order one; random weights of a random shape may give larger ones -- 1 % of the largest then)
        bar_le = max(bar_l, 0.01 * float(np.abs(ref['logit']).max())) if precision == "bf16" else bar_l
        ef = max(ef, np.abs(out['logit'] - ref['logit'].reshape(out['logit'].shape)).max() / bar_le) * 1e-4   # in units of the fp32 bar
        if trained:        # how peaked the case really is: the mean over (step, row[, frame]) of the largest weight of each softmax
            detail_peak = ' peak ' + '/'.join('%.2f' % ref[k].max(axis=-1).mean() for k in ('alphal', 'alphag', 'alpham', 'alphalt'))
        detail = 'alphas %s logit %.2e (max |logit| %.2f)' % (['%.2e' % np.abs(out[k] - ref[k]).max() for k in ('alphal', 'alphag', 'alpham', 'alphalt')],
                                                              np.abs(out['logit'] - ref['logit'].reshape(out['logit'].shape)).max(), np.abs(ref['logit']).max())
        alpha_c = float(rng.choice([0.0, 0.70602]))
        dec.backward(alpha_c=alpha_c)
        got = dec.get_grads()
        rg = OG.loss_and_grads(P, opt, batch, alpha_c=alpha_c)
        eg, which = 0.0, ""
        for k in got:
            scale = np.abs(np.asarray(rg['grads'][k])).max()
            if precision == "bf16" and k == 'decoder_b_sel':
                # a SCALAR: the signed sum of the gate's pre-activation gradients over every (row, step).  Its bf16 error is a fraction of
                # sum |terms|, not of |sum| -- and sum |terms| >= max |dW_sel| (dW_sel = sum term * h with |h| < 1).  Seed 8801 case 11:
                # |sum| 6.5e-4 beside max |dW_sel| 2.4e-2, error 9e-5 (the fp32 handle: 2e-8) -- 14 % of |sum|, 0.4 % of the terms' scale.
                scale = max(scale, np.abs(np.asarray(rg['grads']['decoder_W_sel'])).max())
            r = np.abs(got[k] - rg['grads'][k]).max() / (bar_g * scale + 5e-6)      # <= 1 passes (zero-gradient floor 5e-6)
            if r > eg:
                eg, which = r, k
        if only is not None:
            d32 = stattn.Decoder(opt, lt_mode=lt_mode, precision="fp32")
            d32.set_params(P); d32.set_batch(**batch); d32.forward_train(); d32.backward(alpha_c=alpha_c)
            g32 = d32.get_grads()
            for k in sorted(got):
                ref_k = np.asarray(rg['grads'][k]); scale = np.abs(ref_k).max()
                print("   %-28s scale %.3e  |got - ref| %.3e (%.4f of scale)  fp32 handle %.3e" %
                      (k, scale, np.abs(got[k] - ref_k).max(), np.abs(got[k] - ref_k).max() / (scale + 1e-30), np.abs(g32[k] - ref_k).max()))
        ok = ef < 1e-4 and eg <= 1.0
        bad += not ok
        worst_f, worst_g = max(worst_f, ef), max(worst_g, eg)
        print("%3d %-5s lt%d D=%3d E=%3d V=%4d Fl=%3d Fm=%3d sel=%d p2o=%d c2o=%d B=%2d T=%2d K=%2d t=%d  fwd %.2e  grad %.2f of the bar (%s)%s"
              % (case, precision, lt_mode, D, dims['dim_word'], dims['n_words'], dims['ctxl_dim'], dims['ctxm_dim'], dims['selector'],
                 dims['prev2out'], dims['ctx2out'], B, T, K, t, ef, eg, which, (detail_peak + " max |logit| %.1f" % np.abs(ref['logit']).max() if trained else "") + ("" if ok else "   <-- FAIL " + detail)), flush=True)
        del dec
    print("cases %d  failures %d  worst forward error %.2e (bar 1e-4)  worst gradient %.2f of its bar" % (n, bad, worst_f, worst_g))
    return bad


def run_beam(n, seed, only=None, verbose=False):
    """Random sampler configurations: the device-resident batched beam search against the host-driven gen_sample loop over
    the same handle (identical hypotheses) and against the float64 oracle driver (scores within 1e-4; the best hypothesis
    may only differ when the oracle's two best scores are closer than that)."""
    rng = np.random.RandomState(seed)
    bad = 0
    relaxed = dict(order_only=0, device_equals_oracle=0)
    for case in range(n):
        D = int(rng.choice([64, 128, 192]))
        dims = dict(dim=D, ctxg_dim=D, ctxglm_dim=D, dim_word=int(rng.choice([64, 128])), n_words=int(rng.randint(12, 400)),
                    ctxl_dim=int(32 * rng.randint(1, 6)), ctxm_dim=int(32 * rng.randint(1, 6)),
                    selector=bool(rng.randint(2)), prev2out=bool(rng.randint(2)), ctx2out=bool(rng.randint(2)))
        nvid, T, K = int(rng.randint(1, 6)), int(rng.randint(1, 12)), int(rng.randint(1, 10))
        k, maxlen = int(rng.randint(1, 7)), int(rng.randint(3, 10))
        if case % 5 == 4:              # more than 64 rows: wide row-panel kernels with the vocabulary statistics epilogue (needs V, E, D % 32)
            nvid, k = int(rng.randint(10, 25)), int(rng.randint(4, 9))
        if case % 5 == 3:              # many videos x frames (>= 2048 items): the shared-slab attention kernel; with > 64 rows its launch
            nvid, T, K = int(rng.randint(8, 40)), int(rng.randint(40, 97)), int(rng.randint(1, 13))     # also carries the previous word's update
            k = int(rng.randint(2, 7))                                                                    # (17 .. 64 rows: shared-slab kernel, update as a launch of its own)
        precision = ["fp32", "split"][case % 2]
        opt = dict(O.default_options(**dims), stattn_precision=precision, lt_mode=int(rng.randint(2)))
        P = O.random_params(opt, seed=int(rng.randint(1 << 30)), dtype=np.float32)
        P['ff_logit_b'] = P['ff_logit_b'].copy(); P['ff_logit_b'][0] += float(rng.uniform(0.0, 4.0))     # word 0 = <eos>
        P64 = O.cast_params(P, np.float64)
        b = O.synthetic_batch(opt, B=nvid, T=T, K=K, t=3, seed=int(rng.randint(1 << 30)))
        if only is not None and case != only:
            continue
        model = stattn.Attention()
        tparams = model.init_tparams(P)
        f_init, f_next = model.build_sampler(tparams, opt, None, None)
        f_next.device_loop = False
        res = model.gen_sample_batch(tparams, opt, b['ctxg'], b['mask_ctxg'], b['ctxl'], b['ctxm'], k=k, maxlen=maxlen)
        rode = tparams.decoder.path_counts()['upd_rider'] > 0
        ok, why = True, ""
        for v in range(nvid):
            args = (b['ctxg'][v], b['mask_ctxg'][v], b['ctxl'][v], b['mask_ctxl'][v], b['ctxm'][v], b['mask_ctxm'][v])
            s_, sc, _, _ = model.gen_sample(tparams, f_init, f_next, *args, opt, None, k, maxlen=maxlen)
            a64 = tuple(a.astype(np.float64) for a in args)
            sr, scr, _, _ = O.gen_sample(lambda g_, m_: O.f_init(P64, opt, g_, m_), lambda *a: O.f_next(P64, opt, *a), *a64, k=k, maxlen=maxlen)
            bs, bsc = res[v]
            scr = np.asarray(scr, np.float64)
            if verbose and (bs != s_ or not np.allclose(bsc, np.asarray(sc, np.float32), rtol=1e-4, atol=1e-4)):
                print('video', v, '\n device', bs, list(bsc), '\n host  ', s_, list(sc), '\n oracle', sr, list(scr))

                def both(x, *a):          # every f_next call of the host loop against the oracle's on the same inputs
                    r = f_next(x, *a)
                    ro = O.f_next(P64, opt, x, *[None if q is None else np.asarray(q, np.float64) for q in a])
                    print('   f_next m=%d words %s: max |dp| %.2e  max |dh| %.2e' % (len(x), list(x), np.abs(r[0] - ro[0]).max(), np.abs(r[2] - ro[2]).max()))
                    return r
                both.decoder = f_next.decoder; both.device_loop = False
                model.gen_sample(tparams, f_init, both, *args, opt, None, k, maxlen=maxlen)
            if bs != s_ or not np.allclose(bsc, np.asarray(sc, np.float32), rtol=1e-4, atol=1e-4):
                # the two loops round their candidate costs differently (device: (score + lse) - logit, host: score - log p, both float32):
                # a near-tie at the pruning boundary may keep different survivors.  Accepted only when the device loop reproduces the
                # float64 oracle's hypotheses and scores exactly -- then it is the host loop that sits on the other side of a tie.
                # ... or when both loops hold the SAME hypotheses with the same scores and merely list two of them whose scores are within
                # 2e-4 of each other in the other order (seed 424242 case 119: 17.698643 / 17.698645 after four words, device order swapped
                # against host and oracle, with and without the riding update)
                # (the order-only rule checks the PARTNER -- a hypothesis that sits elsewhere in the other list must sit there at a
                # position whose score is within 2e-4 of its own; and the run reports how many cases needed either relaxation)
                dev_sc = dict(zip(map(tuple, bs), map(float, bsc)))
                host_pos = {tuple(h_): i_ for i_, h_ in enumerate(s_)}
                same_set = (sorted(map(tuple, bs)) == sorted(map(tuple, s_)) and len(host_pos) == len(s_) and
                            np.allclose([dev_sc[h_] for h_ in map(tuple, s_)], np.asarray(sc, np.float32), rtol=1e-4, atol=1e-4) and
                            all(tuple(x) == tuple(y) or abs(float(cx) - float(sc[host_pos[tuple(x)]])) < 2e-4 and abs(float(cx) - float(bsc[host_pos[tuple(x)]])) < 2e-4
                                for x, y, cx in zip(bs, s_, bsc)))
                oracle_eq = bs == sr and np.allclose(bsc, scr, rtol=1e-4, atol=1e-4)
                if same_set or oracle_eq:
                    relaxed['order_only' if same_set else 'device_equals_oracle'] += 1
                else:
                    ok, why = False, "device loop != host loop (video %d)" % v
            elif len(bs) != len(sr) or not np.allclose(sorted(bsc), sorted(scr), rtol=1e-4, atol=1e-4):
                # a near-tie at the beam boundary may swap which hypothesis survives; accept only if the oracle itself is that close
                gap = np.min(np.diff(np.sort(scr))) if len(scr) > 1 else 1.0
                if gap > 2e-4:
                    ok, why = False, "scores differ from the oracle (video %d)" % v
            elif bs[int(np.argmin(bsc))] != sr[int(np.argmin(scr))] and (np.sort(scr)[1] - np.sort(scr)[0] if len(scr) > 1 else 1.0) > 2e-4:
                ok, why = False, "best hypothesis differs (video %d)" % v
        bad += not ok
        print("%3d %-5s lt%d D=%3d E=%3d V=%3d sel=%d p2o=%d c2o=%d videos=%d T=%2d K=%2d beam=%d maxlen=%d  %s"
              % (case, precision, opt['lt_mode'], D, dims['dim_word'], dims['n_words'], dims['selector'], dims['prev2out'], dims['ctx2out'],
                 nvid, T, K, k, maxlen, ("ok" if ok else "FAIL: " + why) + (" (update rode)" if rode else "")), flush=True)
    print("beam cases %d  failures %d  accepted through a near-tie rule: %d order-only, %d device == float64 oracle != host loop"
          % (n, bad, relaxed['order_only'], relaxed['device_equals_oracle']))
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "beam1":      # one case of a beam sweep, verbosely: beam1 <n> <seed> <case>
        sys.exit(1 if run_beam(int(sys.argv[2]), int(sys.argv[3]), only=int(sys.argv[4]), verbose=True) else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "beam":
        sys.exit(1 if run_beam(int(sys.argv[2]) if len(sys.argv) > 2 else 30, int(sys.argv[3]) if len(sys.argv) > 3 else 2024) else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "trained":     # even cases small shapes, odd cases production-sized ones
        n_, seed_ = int(sys.argv[2]) if len(sys.argv) > 2 else 12, int(sys.argv[3]) if len(sys.argv) > 3 else 2024
        sys.exit(1 if run(n_, seed_, trained=True) + run(max(n_ // 3, 1), seed_ + 1, large=True, trained=True) else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "large":
        sys.exit(1 if run(int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 2024, large=True) else 0)
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 2024) else 0)
