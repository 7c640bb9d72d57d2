"""Restatement-independent properties, asserted ON THE HIP PATH through the C ABI (the CPU
twins that pin the oracle are tests/test_oracle.py:66-131, 181).  None of these checks reads a number computed by the
oracle: each compares two runs of the HIP path whose results the reference's maths (model_attention.py:366-459, 583-717,
852-994) says must coincide.  oracle/ is imported only for its seeded weight / batch generators.

  * region permutation  => alphal permutes along K, everything else unchanged          (:371-383 sums over regions)
  * frame permutation   => alphag / alpham / alphalt (and alphal's frame axis) permute, logits unchanged (:389-430)
  * c*_att + constant   => nothing changes (softmax shift invariance, :380, 398, 411, 425)
  * t teacher-forced f_next calls == build_model's forward with use_noise = 0 (:719-850 is the one-step graph of :583-717)
  * beam k = 1          == the arg-max chain of f_next (:896-918)
  * rows of a batch are independent (only the regulariser couples them) and a masked step freezes the state (:454-457)
"""
from collections import OrderedDict

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMALL = dict(dim=128, dim_word=64, n_words=211, ctxg_dim=128, ctxl_dim=96, ctxm_dim=64, ctxglm_dim=128)
# D = 1024: the 128-thread attention kernel with the h.U rider; with 17 <= rows <= 64 the row-panel kernels
WIDE = dict(dim=1024, dim_word=128, n_words=500, ctxg_dim=1024, ctxl_dim=64, ctxm_dim=64, ctxglm_dim=1024)
ALPHAS = ('alphal', 'alphag', 'alpham', 'alphalt')
EQ = 2e-6        # two runs that differ in summation order only (absolute, on softmax outputs)
EQ_LOGIT = 2e-5


def _make(dims, B, T, K, t, seed=4, ragged=True, **optkw):
    import stattn
    from oracle import stattn_oracle as O
    opt = O.default_options(**{**dims, **optkw})
    P = O.random_params(opt, seed=seed, dtype=np.float32)
    batch = O.synthetic_batch(opt, B=B, T=T, K=K, t=t, seed=seed + 50, ragged=ragged)
    dec = stattn.Decoder(opt, lt_mode=1)
    dec.set_params(P)
    return opt, P, batch, dec


def _forward(dec, batch):
    dec.set_batch(**batch)
    dec.forward_train()
    return dec.get_forward(logits=True)


CASES = [(SMALL, 5, 5, 4, 4), (SMALL, 3, 4, 11, 3), (WIDE, 20, 6, 8, 3), (WIDE, 3, 5, 16, 3)]


@pytest.mark.parametrize("dims,B,T,K,t", CASES)
def test_region_permutation_permutes_alphal_only(dims, B, T, K, t):
    opt, P, batch, dec = _make(dims, B, T, K, t)
    o0 = _forward(dec, batch)
    pk = np.random.RandomState(1).permutation(K)
    b1 = dict(batch, ctxl=np.ascontiguousarray(batch['ctxl'][:, :, pk]))
    o1 = _forward(dec, b1)
    assert np.abs(o1['alphal'] - o0['alphal'][..., pk]).max() < EQ
    assert np.abs(o0['alphal'] - o0['alphal'][..., pk]).max() > 1e-3          # the permutation is visible at all
    for a in ALPHAS[1:]:
        assert np.abs(o1[a] - o0[a]).max() < EQ, a
    assert np.abs(o1['logit'] - o0['logit']).max() < EQ_LOGIT
    np.testing.assert_allclose(o1['cost'], o0['cost'], rtol=1e-5)


@pytest.mark.parametrize("dims,B,T,K,t", CASES)
def test_frame_permutation_permutes_the_temporal_attentions_only(dims, B, T, K, t):
    opt, P, batch, dec = _make(dims, B, T, K, t)
    o0 = _forward(dec, batch)
    pt = np.random.RandomState(2).permutation(T)
    b1 = dict(batch, ctxg=np.ascontiguousarray(batch['ctxg'][:, pt]), ctxl=np.ascontiguousarray(batch['ctxl'][:, pt]),
              ctxm=np.ascontiguousarray(batch['ctxm'][:, pt]))
    o1 = _forward(dec, b1)
    for a in ALPHAS[1:]:
        assert np.abs(o1[a] - o0[a][..., pt]).max() < EQ, a
        assert np.abs(o0[a] - o0[a][..., pt]).max() > 1e-4, a
    assert np.abs(o1['alphal'] - o0['alphal'][:, :, pt]).max() < EQ
    assert np.abs(o1['logit'] - o0['logit']).max() < EQ_LOGIT
    np.testing.assert_allclose(o1['cost'], o0['cost'], rtol=1e-5)


@pytest.mark.parametrize("dims,B,T,K,t", CASES[:3])
def test_attention_score_offsets_change_nothing(dims, B, T, K, t):
    opt, P, batch, dec = _make(dims, B, T, K, t)
    o0 = _forward(dec, batch)
    for k, shift in (('cg', 3.7), ('cm', -2.2), ('clt', 5.1), ('cl', -4.3)):
        dec.set_param('decoder_%s_att' % k, P['decoder_%s_att' % k] + np.float32(shift))
    o1 = _forward(dec, batch)
    for a in ALPHAS:
        assert np.abs(o1[a] - o0[a]).max() < EQ, a
    assert np.abs(o1['logit'] - o0['logit']).max() < EQ_LOGIT
    # and the gradient with respect to those offsets is zero
    dec.backward(alpha_c=0.70602)
    for k in ('cg', 'cm', 'clt', 'cl'):
        assert abs(float(dec.get_grad('decoder_%s_att' % k).reshape(-1)[0])) < 5e-6, k


@pytest.mark.parametrize("dims,B,T,K,t", [(SMALL, 3, 5, 4, 30), (WIDE, 18, 6, 8, 30)])
def test_teacher_forced_f_next_chain_equals_build_model(dims, B, T, K, t):
    """30 consecutive f_next calls per video (its own kernels: small-batch attention kernel, 64-column skinny GEMMs, readout
    per step) against ONE forward_train over the whole batch (batched readout, row-panel kernels at 18 rows), use_noise = 0,
    all-ones mask: probabilities, all four attention weights and the final state must coincide."""
    opt, P, batch, dec = _make(dims, B, T, K, t, ragged=False)
    batch['mask'][:] = 1.0
    o0 = _forward(dec, batch)
    st = dec.get_states()
    V = opt['n_words']
    probs0 = o0['probs'].reshape(t, B, V)
    worst = 0.0
    for b in range(B):
        g, l, m = batch['ctxg'][b], batch['ctxl'][b], batch['ctxm'][b]
        _, h, c = dec.f_init(g, batch['mask_ctxg'][b])
        h, c = h[None], c[None]
        w = np.array([-1], np.int64)
        with dec.video_scope(g, l, m):
            for s in range(t):
                (p, _, h, c), ex = dec.f_next(w, g, None, l, None, m, None, h, c, extras=True)
                worst = max(worst, float(np.abs(p[0] - probs0[s, b]).max()))
                for a in ALPHAS:
                    assert np.abs(ex[a][0] - o0[a][s, b]).max() < 1e-5, (a, b, s)
                assert np.abs(p[0] - probs0[s, b]).max() < 1e-5, (b, s)
                assert np.abs(h[0] - st['h'][s, b]).max() < 2e-5, (b, s)
                w = batch['x'][s:s + 1, b]
    assert worst < 1e-5


@pytest.mark.parametrize("dims,nvid,T,K", [(SMALL, 4, 5, 4), (WIDE, 3, 6, 8)])
def test_device_beam_search_with_one_beam_is_the_argmax_chain(dims, nvid, T, K):
    import stattn
    from oracle import stattn_oracle as O
    opt = O.default_options(**dims)
    P = O.random_params(opt, seed=9, dtype=np.float32)
    P['ff_logit_b'] = P['ff_logit_b'].copy(); P['ff_logit_b'][0] += 2.0          # <eos> ends some captions early
    b = O.synthetic_batch(opt, B=nvid, T=T, K=K, t=3, seed=19)
    model = stattn.Attention()
    tparams = model.init_tparams(P)
    maxlen = 12
    res = model.gen_sample_batch(tparams, opt, b['ctxg'], b['mask_ctxg'], b['ctxl'], b['ctxm'], k=1, maxlen=maxlen)
    dec = tparams.decoder
    for v in range(nvid):
        g, l, m = b['ctxg'][v], b['ctxl'][v], b['ctxm'][v]
        _, h, c = dec.f_init(g, b['mask_ctxg'][v])
        h, c = h[None], c[None]
        w = np.array([-1], np.int64)
        words, score = [], 0.0
        for _ in range(maxlen):
            p, _, h, c = dec.f_next(w, g, None, l, None, m, None, h, c)
            top2 = np.sort(p[0])[-2:]
            assert top2[1] - top2[0] > 1e-5                                     # no near-tie: the arg-max is well defined
            nw = int(p[0].argmax())
            words.append(nw); score -= float(np.log(p[0, nw]))
            if nw == 0:
                break
            w = np.array([nw], np.int64)
        (hyps, scores) = res[v]
        assert len(hyps) == 1 and list(hyps[0]) == words, (v, hyps, words)
        np.testing.assert_allclose(scores[0], score, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dims,B,T,K,t", [(SMALL, 6, 5, 4, 7), (WIDE, 20, 6, 8, 8)])
def test_rows_are_independent_and_masked_steps_freeze_the_state(dims, B, T, K, t):
    opt, P, batch, dec = _make(dims, B, T, K, t)
    o0 = _forward(dec, batch)
    st = dec.get_states()
    for b in range(B):
        ln = int(batch['mask'][:, b].sum())
        if ln < t:
            np.testing.assert_array_equal(st['h'][ln - 1, b], st['h'][-1, b])
            np.testing.assert_array_equal(st['c'][ln - 1, b], st['c'][-1, b])
    assert any(int(batch['mask'][:, b].sum()) < t for b in range(B))
    V = opt['n_words']
    for rows in ([1], [B - 1, 2]):
        sub = {k: np.ascontiguousarray(v[:, rows] if k in ('x', 'mask') else v[rows]) for k, v in batch.items()}
        o1 = _forward(dec, sub)
        for a in ALPHAS:
            assert np.abs(o1[a] - o0[a][:, rows]).max() < EQ * 2, a
        assert np.abs(o1['logit'].reshape(t, len(rows), V) - o0['logit'].reshape(t, B, V)[:, rows]).max() < EQ_LOGIT * 2
        np.testing.assert_allclose(o1['cost'], o0['cost'][rows], rtol=1e-5)


def test_zero_attention_weights_give_uniform_spatial_attention():
    """Wcl = Wdl = bl = Ul = 0  =>  alphal == 1/K exactly, whatever the features (analytic, :371-380)."""
    opt, P, batch, dec = _make(WIDE, 20, 6, 8, 3)
    for k in ('decoder_Wcl_att', 'decoder_Wdl_att', 'decoder_bl_att', 'decoder_Ul_att'):
        dec.set_param(k, np.zeros_like(P[k]))
    o = _forward(dec, batch)
    np.testing.assert_allclose(o['alphal'], 1.0 / 8, atol=1e-7)
    assert np.abs(o['alphag'] - 1.0 / 6).max() > 1e-3            # the other attentions still attend
