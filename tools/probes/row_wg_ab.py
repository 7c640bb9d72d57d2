#!/usr/bin/env python
"""Beam search on the small path (at most 16 rows) with k update workgroups per video ("row workgroups", beam_inl.h) against the same
search with STATTN_NO_ROW_WG=1 (one workgroup per video): tokens, scores and final states must be BIT-equal (the arithmetic is the
same value for value).  Shapes: configs[0] dimensions with a small vocabulary and the full one, k = 2 .. 8, 1 .. 3 videos, <eos> likely."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
from oracle import stattn_oracle as O
bad = 0
for V, D, eos in ((400, 128, 2.5), (12000, 512, 6.0), (3001, 256, 0.0)):
    dims = dict(dim=D, dim_word=D, n_words=V, ctxg_dim=D, ctxl_dim=256, ctxm_dim=256, ctxglm_dim=D)
    opt = O.default_options(**dims)
    P = O.random_params(opt, seed=5 + V, dtype=np.float32)
    P['ff_logit_b'] = P['ff_logit_b'].copy(); P['ff_logit_b'][0] += eos
    model = stattn.Attention()
    f_init, f_next = model.build_sampler(model.init_tparams(P), opt, None, None)
    dec = f_next.decoder
    for k, nvid in ((2, 1), (3, 1), (5, 1), (8, 1), (5, 3), (4, 4), (2, 8), (7, 2), (8, 2)):
        b = O.synthetic_batch(opt, B=nvid, T=9, K=4, t=3, seed=100 + k + nvid)
        for maxlen in (1, 2, 5, 12):
            os.environ.pop('STATTN_NO_ROW_WG', None)
            r1 = dec.beam_search(b['ctxg'], b['mask_ctxg'], b['ctxl'], b['ctxm'], k=k, maxlen=maxlen)
            f1 = dec.beam_final_state()
            os.environ['STATTN_NO_ROW_WG'] = '1'
            r2 = dec.beam_search(k=k, maxlen=maxlen, resident=True)
            f2 = dec.beam_final_state()
            os.environ.pop('STATTN_NO_ROW_WG', None)
            ok = True
            for v in range(nvid):
                ok &= [list(x) for x in r1[v][0]] == [list(x) for x in r2[v][0]]
                ok &= np.array_equal(np.asarray(r1[v][1], np.float32), np.asarray(r2[v][1], np.float32))
                ok &= f1[v][0].shape == f2[v][0].shape and np.array_equal(f1[v][0], f2[v][0]) and np.array_equal(f1[v][1], f2[v][1])
            bad += not ok
            if not ok:
                print("V=%d D=%d k=%d nvid=%d maxlen=%d  DIFFERS" % (V, D, k, nvid, maxlen), r1[0][1], r2[0][1])
print("row-workgroup A/B: %s" % ("all bit-equal" if not bad else "%d cases differ" % bad))
sys.exit(1 if bad else 0)
