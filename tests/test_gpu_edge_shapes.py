"""Degenerate and extreme shapes of the decoder path on the GPU against the oracle: one frame, one region, the
kernels' maxima (K = 64 regions, T = 256 frames), one row, a batch that is not a multiple of any tile, captions longer
than any bench config.  Forward at the 1e-4 bar, gradients relative to each tensor's largest entry."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS = dict(dim=64, dim_word=64, n_words=7, ctxg_dim=64, ctxl_dim=32, ctxm_dim=32, ctxglm_dim=64)


@pytest.mark.parametrize("B,T,K,t", [(1, 1, 1, 1), (1, 2, 1, 3), (3, 1, 5, 2), (2, 3, 64, 2), (5, 256, 1, 2),
                                     (130, 2, 2, 1), (2, 2, 2, 40)])
def test_edge_shapes_forward_and_gradients(B, T, K, t):
    import torch
    import stattn
    from oracle import stattn_oracle as O
    from oracle import stattn_oracle_grad as OG
    opt = O.default_options(**DIMS)
    P = O.random_params(opt, seed=B + T + K + t, dtype=np.float32)
    dec = stattn.Decoder(opt)
    dec.set_params(P)
    batch = O.synthetic_batch(opt, B=B, T=T, K=K, t=t, seed=5)
    dec.set_batch(**batch)
    dec.forward_train()
    out = dec.get_forward(logits=True)
    b64 = {k: (v if v.dtype == np.int64 else v.astype(np.float64)) for k, v in batch.items()}
    ref = O.build_model_forward(O.cast_params(P, np.float64), opt, **b64)
    for n in ('alphal', 'alphag', 'alpham', 'alphalt'):
        assert out[n].shape == ref[n].shape
        assert np.abs(out[n] - ref[n]).max() < 1e-4, n
    assert np.abs(out['logit'] - ref['logit'].reshape(out['logit'].shape)).max() < 1e-4
    np.testing.assert_allclose(out['cost'], ref['cost'], rtol=1e-4, atol=1e-4)
    dec.backward(nll_scale=1.0 / B, alpha_c=0.5)
    g = dec.get_grads()
    rg = OG.loss_and_grads(P, opt, batch, decay_c=0.0, alpha_c=0.5, dtype=torch.float64)['grads']
    gmax = max(np.abs(np.asarray(v)).max() for v in rg.values())
    for k in g:
        r = np.asarray(rg[k])
        # the scalar score biases c*_att have an exactly zero gradient (softmax shift invariance): what the GPU
        # returns there is the rounding of a sum of t*m*T terms, so it is measured against the overall gradient scale
        floor = 1e-4 * gmax if k in ('decoder_cg_att', 'decoder_cm_att', 'decoder_clt_att', 'decoder_cl_att') else 1e-6
        assert np.abs(g[k] - r).max() <= 2e-4 * np.abs(r).max() + floor, (k, np.abs(g[k] - r).max(), np.abs(r).max(), gmax)


def test_shapes_beyond_the_kernel_limits_are_refused():
    import stattn
    from oracle import stattn_oracle as O
    opt = O.default_options(**DIMS)
    dec = stattn.Decoder(opt)
    dec.set_params(O.random_params(opt, seed=1, dtype=np.float32))
    for T, K in ((257, 2), (2, 65)):
        batch = O.synthetic_batch(opt, B=1, T=T, K=K, t=2, seed=5)
        with pytest.raises((ValueError, stattn.NativeError)):
            dec.set_batch(**batch)
            dec.forward_train()


def test_large_odd_vocabulary_takes_the_multi_pass_softmax():
    """V = 20011 (> 256 x 48 register-resident values, not a multiple of anything): logits, probabilities, cost, the
    sampler's probabilities and the device beam search against the oracle."""
    import stattn
    from oracle import stattn_oracle as O
    dims = dict(DIMS, n_words=20011)
    opt = O.default_options(**dims)
    P = O.random_params(opt, seed=77, dtype=np.float32)
    P64 = O.cast_params(P, np.float64)
    dec = stattn.Decoder(opt)
    dec.set_params(P)
    batch = O.synthetic_batch(opt, B=300, T=3, K=2, t=2, seed=9)          # 600 rows: the 256-thread softmax variants
    dec.set_batch(**batch)
    dec.forward_train()
    out = dec.get_forward(logits=True)
    b64 = {k: (v if v.dtype == np.int64 else v.astype(np.float64)) for k, v in batch.items()}
    ref = O.build_model_forward(P64, opt, **b64)
    assert np.abs(out['logit'] - ref['logit'].reshape(out['logit'].shape)).max() < 1e-4
    assert np.abs(out['probs'] - ref['probs']).max() < 1e-5
    np.testing.assert_allclose(out['probs'].sum(-1), 1.0, atol=1e-4)
    np.testing.assert_allclose(out['cost'], ref['cost'], rtol=1e-4, atol=1e-4)
    model = stattn.Attention()
    tparams = model.init_tparams(P)
    f_init, f_next = model.build_sampler(tparams, opt, None, None)
    f_next.device_loop = False        # gen_sample below = the host-driven loop
    v = 0
    args = (batch['ctxg'][v], batch['mask_ctxg'][v], batch['ctxl'][v], batch['mask_ctxl'][v], batch['ctxm'][v], batch['mask_ctxm'][v])
    s, sc, _, _ = model.gen_sample(tparams, f_init, f_next, *args, opt, None, 3, maxlen=5)
    res = model.gen_sample_batch(tparams, opt, batch['ctxg'][:2], batch['mask_ctxg'][:2], batch['ctxl'][:2], batch['ctxm'][:2], k=3, maxlen=5)
    assert res[0][0] == s
    a64 = tuple(np.asarray(a, np.float64) for a in args)
    sr, scr, _, _ = O.gen_sample(lambda g_, m_: O.f_init(P64, opt, g_, m_), lambda *a: O.f_next(P64, opt, *a), *a64, k=3, maxlen=5)
    assert s[int(np.argmin(sc))] == sr[int(np.argmin(scr))]
