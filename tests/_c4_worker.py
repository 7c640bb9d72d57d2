"""Child process of tests/test_gpu_bf16.py::test_bf16_c4_bench_shape_*: the same forward + backward on the same seeded
weights and batch, in a fresh process so that the library's A/B switches (read once per process: STATTN_NO_RIDER, ...)
can differ from the parent's.  Writes alphas, logits, path counters and a few gradients to an npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

C4 = dict(dim=1024, dim_word=512, n_words=3000, ctxg_dim=1024, ctxl_dim=2048, ctxm_dim=2048, ctxglm_dim=1024)
GRADS = ('decoder_U', 'decoder_Wc', 'decoder_Wdl_att', 'decoder_Wclt_att', 'ff_local_W', 'ff_logit_W', 'Wemb', 'decoder_Ul_att')


def case(B, t, precision, seed=21):
    import stattn
    from oracle import stattn_oracle as O
    opt = O.default_options(**C4)
    P = O.random_params(opt, seed=seed, dtype=np.float32)
    batch = O.synthetic_batch(opt, B=B, T=40, K=16, t=t, seed=60 + B)
    dec = stattn.Decoder(opt, precision=precision)
    dec.set_params(P)
    return O, opt, P, batch, dec


def main():
    B, t, precision, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    O, opt, P, batch, dec = case(B, t, precision)
    dec.set_batch(**batch)
    dec.forward_train()
    fw = dec.get_forward(logits=True)
    dec.backward(alpha_c=0.70602)
    pc = dec.path_counts()
    res = {k: fw[k] for k in ('alphal', 'alphag', 'alpham', 'alphalt', 'logit', 'cost')}
    res.update({"g_" + k: dec.get_grad(k) for k in GRADS})
    res.update({"pc_" + k: np.int64(v) for k, v in pc.items()})
    np.savez(out, **res)


if __name__ == "__main__":
    main()
