"""Child process of tests/test_gpu_z3_switches.py: one optimisation step and one batched beam search on seeded weights, in a fresh
process so that the library's product switches (csrc/switches.h, read once per process) can differ from the parent's.  Writes the
forward quantities, a few gradients, the captions and their scores, and the path counters to an npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DIMS = dict(dim=1024, dim_word=256, n_words=1900, ctxg_dim=1024, ctxl_dim=256, ctxm_dim=128, ctxglm_dim=1024)
GRADS = ('decoder_U', 'decoder_Wc', 'decoder_Wdl_att', 'decoder_Wclt_att', 'ff_local_W', 'ff_logit_W', 'ff_logit_lstm_W', 'Wemb', 'decoder_Ul_att')


def run():
    import stattn
    from oracle import stattn_oracle as O
    opt = O.default_options(**DIMS)
    P = O.random_params(opt, seed=31, dtype=np.float32)
    P['ff_logit_b'] = P['ff_logit_b'].copy(); P['ff_logit_b'][0] += 1.5          # some hypotheses end early
    batch = O.synthetic_batch(opt, B=24, T=7, K=6, t=5, seed=77)                  # 24 rows, D = 1024: riders + row-panel kernels
    model = stattn.Attention()
    tparams = model.init_tparams(P)
    f_init, f_next = model.build_sampler(tparams, opt, None, None)
    dec = f_next.decoder
    dec.set_batch(**batch)
    dec.forward_train()
    fw = dec.get_forward(logits=True)
    dec.backward(alpha_c=0.70602)
    res = {k: fw[k] for k in ('alphal', 'alphag', 'alpham', 'alphalt', 'logit', 'cost')}
    res.update({"g_" + k: dec.get_grad(k) for k in GRADS})
    res.update({"pc_" + k: np.int64(v) for k, v in dec.path_counts().items()})
    # 5 videos x beam 4 = 20 rows: the 17 .. 64-row word (vocabulary statistics on the wide kernel unless STATTN_WIDE_STATS_FROM=65)
    out = model.gen_sample_batch(tparams, opt, batch['ctxg'][:5], batch['mask_ctxg'][:5], batch['ctxl'][:5], batch['ctxm'][:5], k=4, maxlen=6)
    res["graph_replays"] = np.int64(dec.beam_graph_replays())
    res["stats_words"] = np.int64(dec.beam_vocab_stats_words())
    for v, (seqs, scores) in enumerate(out):
        order = np.argsort(np.asarray(scores), kind="stable")
        res["tok_%d" % v] = np.array([";".join(map(str, seqs[i])) for i in order])
        res["score_%d" % v] = np.asarray(scores, np.float64)[order]
    return res


if __name__ == "__main__":
    np.savez(sys.argv[1], **run())
