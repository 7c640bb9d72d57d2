#!/usr/bin/env python
"""Generates tests/golden/*.npz with the float64 CPU oracle (fixed seeds).

The reference (Python 2 + Theano) cannot be imported here or on the GPU box and ships no golden
vectors, so these fixtures freeze the oracle's outputs: they pin the oracle
against accidental edits and give the GPU tests size-stable expected values.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stattn_oracle as O          # noqa: E402
from oracle import stattn_oracle_grad as OG    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DIMS = dict(dim=64, dim_word=64, n_words=37, ctxg_dim=64, ctxl_dim=32, ctxm_dim=32, ctxglm_dim=64)


def main():
    opt = O.default_options(**DIMS)
    P32 = O.random_params(opt, seed=2024, dtype=np.float32)
    P = O.cast_params(P32, np.float64)
    # (1) f_init / f_next chain: 3 consecutive steps, m in {1, 3}
    b = O.synthetic_batch(opt, B=1, T=3, K=2, t=4, seed=7)
    g, l, m, gm = b['ctxg'][0], b['ctxl'][0], b['ctxm'][0], b['mask_ctxg'][0]
    out = dict(ctxg=g, ctxl=l, ctxm=m, ctxg_mask=gm)
    _, h0, c0 = O.f_init(P, opt, g.astype(np.float64), gm.astype(np.float64))
    out['h0'] = h0; out['c0'] = c0
    for mm in (1, 3):
        h = np.stack([h0 * (1 - 0.3 * i) for i in range(mm)]); c = np.stack([c0 * (1 + 0.2 * i) for i in range(mm)])
        x = np.array([-1, 5, 11][:mm], np.int64)
        out['m%d_h_in' % mm] = h; out['m%d_c_in' % mm] = c
        for s in range(3):
            (probs, _, h, c), r = O.f_next(P, opt, x, g.astype(np.float64), gm, l.astype(np.float64), None,
                                           m.astype(np.float64), None, h, c, extras=True)
            out['m%d_s%d_x' % (mm, s)] = x
            for k in ('alphal', 'alphag', 'alpham', 'alphalt', 'logit'):
                out['m%d_s%d_%s' % (mm, s, k)] = r[k]
            out['m%d_s%d_probs' % (mm, s)] = probs; out['m%d_s%d_h' % (mm, s)] = h; out['m%d_s%d_c' % (mm, s)] = c
            x = np.array([3 + s, 9, 2][:mm], np.int64)
    np.savez_compressed(os.path.join(HERE, 'sampler_chain.npz'), **out)
    # (2) build_model forward, t=4, m=3, ragged mask  (3) gradients of the full loss incl. decay + alpha reg
    batch = O.synthetic_batch(opt, B=3, T=3, K=2, t=4, seed=9)
    b64 = {k: (v if v.dtype == np.int64 else v.astype(np.float64)) for k, v in batch.items()}
    fwd = O.build_model_forward(P, opt, **b64)
    gr = OG.loss_and_grads(P, opt, batch, decay_c=1e-4, alpha_c=0.70602)
    out = {('in_' + k): v for k, v in batch.items()}
    for k in ('cost', 'probs', 'alphal', 'alphag', 'alpham', 'alphalt', 'logit', 'h', 'c', 'ctx'):
        out[k] = fwd[k]
    out['loss'] = np.float64(gr['loss'])
    for k, v in gr['grads'].items():
        out['grad_' + k] = v
    np.savez_compressed(os.path.join(HERE, 'train_graph.npz'), **out)
    np.savez_compressed(os.path.join(HERE, 'params.npz'), **P32)
    print('wrote', [f for f in sorted(os.listdir(HERE)) if f.endswith('.npz')])


if __name__ == '__main__':
    main()
