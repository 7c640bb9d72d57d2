#!/usr/bin/env python
"""configs[4]-shaped beam search with the update riding in the next word's attention launch against the same search with
STATTN_NO_UPDATE_RIDER=1 (update as a launch of its own, attention after the re-ordering): tokens must be equal; prints the rows whose
final states differ at all (a few ulps: children whose parent sat in another hypothesis slot of the shared-slab attention kernel)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stattn
from oracle import stattn_oracle as O
dims = dict(dim=1024, dim_word=512, n_words=2000, ctxg_dim=1024, ctxl_dim=512, ctxm_dim=512, ctxglm_dim=1024)
opt = O.default_options(**dims)
P = O.random_params(opt, seed=23, dtype=np.float32)
P['ff_logit_b'] = P['ff_logit_b'].copy(); P['ff_logit_b'][0] += 1.5
nvid, T, K, k = 32, 80, 32, 5
b = O.synthetic_batch(opt, B=nvid, T=T, K=K, t=3, seed=62)
model = stattn.Attention()
tparams = model.init_tparams(P)
f_init, f_next = model.build_sampler(tparams, opt, None, None)
dec = f_next.decoder
for maxlen in (1, 2, 3, 4, 6):
    os.environ.pop('STATTN_NO_UPDATE_RIDER', None)
    r1 = dec.beam_search(b['ctxg'], b['mask_ctxg'], b['ctxl'], b['ctxm'], k=k, maxlen=maxlen)
    f1 = dec.beam_final_state()
    os.environ['STATTN_NO_UPDATE_RIDER'] = '1'
    r2 = dec.beam_search(k=k, maxlen=maxlen, resident=True)
    f2 = dec.beam_final_state()
    bad = []
    for v in range(nvid):
        for j in range(f1[v][0].shape[0]):
            dh = np.abs(f1[v][0][j] - f2[v][0][j]).max(); dc = np.abs(f1[v][1][j] - f2[v][1][j]).max()
            if dh or dc: bad.append((v, j, float(dh), float(dc), int((f1[v][0][j] != f2[v][0][j]).sum())))
    print(maxlen, 'tokens equal', [x[0] for x in r1] == [x[0] for x in r2], 'bad rows', bad[:8], 'rows', [f1[v][0].shape[0] for v in range(4)])
