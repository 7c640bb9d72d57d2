"""Pins the CPU oracle (oracle/stattn_oracle.py) with checks that do not depend on
the restatement itself: analytic known answers, equivariances, an independent
torch-autograd restatement, and float64 central differences."""
from collections import OrderedDict

import numpy as np
import pytest

from oracle import stattn_oracle as O
from oracle import stattn_oracle_grad as OG

TINY = dict(dim=16, dim_word=8, n_words=11, ctxg_dim=16, ctxl_dim=12, ctxm_dim=10, ctxglm_dim=16)


def _setup(seed=0, dtype=np.float64, B=3, T=4, K=3, t=5, **kw):
    opt = O.default_options(**{**TINY, **kw})
    P = O.random_params(opt, seed=seed, dtype=dtype)
    batch = O.synthetic_batch(opt, B, T, K, t, seed=seed + 1, dtype=dtype)
    return opt, P, batch


def test_param_table_matches_survey_appendix_b():
    opt = O.default_options()
    shp = O.param_shapes(opt)
    assert len(shp) == 41                      # 41 arrays with config.py's options
    assert list(shp)[0] == 'Wemb' and list(shp)[-1] == 'ff_logit_b'
    n = sum(int(np.prod(s)) for s in shp.values())
    assert n == 42727141                       # SURVEY K-table X1: 42 727 141 fp32 @ D=1024,E=512,V=12k,F=4096
    assert shp['decoder_b_sel'] == ()


def test_zero_weights_known_answers():
    """All-zero weights => alphas uniform, probs 1/V, i=f=o=sigma(0)=.5, c'=.5c, h'=.5 tanh(.5c)."""
    opt, P, batch = _setup()
    Z = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
    T, K = 4, 3
    rng = np.random.RandomState(3)
    h = rng.standard_normal((2, 16)); c = rng.standard_normal((2, 16))
    x = np.array([3, -1])
    (probs, sample, h1, c1), r = O.f_next(Z, opt, x, batch['ctxg'][0], batch['mask_ctxg'][0], batch['ctxl'][0],
                                          None, batch['ctxm'][0], None, h, c, extras=True)
    np.testing.assert_allclose(r['alphal'], 1.0 / K)
    np.testing.assert_allclose(r['alphag'], 1.0 / T)
    np.testing.assert_allclose(r['alpham'], 1.0 / T)
    np.testing.assert_allclose(r['alphalt'], 1.0 / T)
    np.testing.assert_allclose(probs, 1.0 / 11)
    np.testing.assert_allclose(c1, 0.5 * c)              # g = tanh(0) = 0
    np.testing.assert_allclose(h1, 0.5 * np.tanh(0.5 * c))
    # CL = mean_K L where L = tanh(0) = 0
    np.testing.assert_allclose(r['CL'], 0.0)


def test_uniform_spatial_attention_gives_region_mean():
    """Zero attention weights only: alphal = 1/K and CL = mean_K L (analytic KAT)."""
    opt, P, batch = _setup()
    for k in ('decoder_Wcl_att', 'decoder_Wdl_att', 'decoder_bl_att', 'decoder_Ul_att'):
        P[k] = np.zeros_like(P[k])
    x = np.array([2]); h = np.zeros((1, 16)); c = np.zeros((1, 16))
    _, r = O.f_next(P, opt, x, batch['ctxg'][0], batch['mask_ctxg'][0], batch['ctxl'][0], None,
                    batch['ctxm'][0], None, h, c, extras=True)
    L = np.tanh(batch['ctxl'][0] @ P['ff_local_W'] + P['ff_local_b'])
    np.testing.assert_allclose(r['alphal'], 1.0 / 3)
    np.testing.assert_allclose(r['CL'][0], L.mean(1), rtol=1e-12)


def test_region_and_frame_permutation_equivariance():
    opt, P, batch = _setup(seed=5)
    x = np.array([4, 7]); rng = np.random.RandomState(9)
    h = rng.standard_normal((2, 16)); c = rng.standard_normal((2, 16))
    g, l, m = batch['ctxg'][1], batch['ctxl'][1], batch['ctxm'][1]
    gm = batch['mask_ctxg'][1]
    out0, r0 = O.f_next(P, opt, x, g, gm, l, None, m, None, h, c, extras=True)
    # permute regions: alphal permutes, everything else identical
    pk = np.array([2, 0, 1])
    out1, r1 = O.f_next(P, opt, x, g, gm, l[:, pk], None, m, None, h, c, extras=True)
    np.testing.assert_allclose(r1['alphal'], r0['alphal'][:, :, pk], rtol=1e-10)
    np.testing.assert_allclose(out1[0], out0[0], rtol=1e-10)
    np.testing.assert_allclose(out1[2], out0[2], rtol=1e-10)
    # permute frames: temporal alphas permute, h' identical
    pt = np.array([3, 1, 0, 2])
    out2, r2 = O.f_next(P, opt, x, g[pt], gm, l[pt], None, m[pt], None, h, c, extras=True)
    for a in ('alphag', 'alpham', 'alphalt'):
        np.testing.assert_allclose(r2[a], r0[a][:, pt], rtol=1e-10)
    np.testing.assert_allclose(out2[2], out0[2], rtol=1e-10)
    np.testing.assert_allclose(out2[0], out0[0], rtol=1e-10)


def test_softmax_shift_invariance_of_c_att():
    opt, P, batch = _setup(seed=6)
    x = np.array([4]); h = np.ones((1, 16)) * 0.1; c = np.zeros((1, 16))
    a = (batch['ctxg'][0], batch['mask_ctxg'][0], batch['ctxl'][0], None, batch['ctxm'][0], None)
    o0 = O.f_next(P, opt, x, *a, h, c)
    P2 = OrderedDict(P)
    for k in ('cg', 'cm', 'clt', 'cl'):
        P2['decoder_%s_att' % k] = P['decoder_%s_att' % k] + 3.7
    o1 = O.f_next(P2, opt, x, *a, h, c)
    np.testing.assert_allclose(o1[0], o0[0], rtol=1e-10)


def test_f_next_chain_equals_build_model():
    """t steps of f_next with teacher-forced words and all-ones mask == build_model, use_noise=0."""
    opt, P, batch = _setup(seed=7, B=2, t=4)
    batch['mask'][:] = 1.0
    fwd = O.build_model_forward(P, opt, **batch)
    t, B = batch['x'].shape
    for b in range(B):
        a = (batch['ctxg'][b], batch['mask_ctxg'][b], batch['ctxl'][b], None, batch['ctxm'][b], None)
        _, h, c = O.f_init(P, opt, a[0], a[1])
        h, c = h[None], c[None]
        w = np.array([-1])
        for s in range(t):
            (probs, _, h, c), r = O.f_next(P, opt, w, *a, h, c, extras=True)
            np.testing.assert_allclose(probs[0], fwd['probs'].reshape(t, B, -1)[s, b], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(r['alphal'][0], fwd['alphal'][s, b], rtol=1e-9)
            np.testing.assert_allclose(r['alphalt'][0], fwd['alphalt'][s, b], rtol=1e-9)
            w = batch['x'][s:s + 1, b]


def test_rows_independent_and_mask_freezes_state():
    opt, P, batch = _setup(seed=8, B=3, t=5)
    fwd = O.build_model_forward(P, opt, **batch)
    sub = {k: (v[:, 1:2] if k in ('x', 'mask') else v[1:2]) for k, v in batch.items()}
    f1 = O.build_model_forward(P, opt, **sub)
    np.testing.assert_allclose(f1['cost'][0], fwd['cost'][1], rtol=1e-10)
    # masked steps keep h: find a row with a masked tail
    for b in range(3):
        ln = int(batch['mask'][:, b].sum())
        if ln < 5:
            np.testing.assert_array_equal(fwd['h'][ln - 1, b], fwd['h'][-1, b])
            break


def test_torch_restatement_agrees_forward():
    opt, P, batch = _setup(seed=11)
    fwd = O.build_model_forward(P, opt, **batch)
    loss = O.total_loss(P, opt, fwd, decay_c=1e-3, alpha_c=0.7)
    tg = OG.loss_and_grads(P, opt, batch, decay_c=1e-3, alpha_c=0.7, want=('loss',))
    np.testing.assert_allclose(tg['cost'], fwd['cost'], rtol=1e-10)
    np.testing.assert_allclose(tg['alphal'], fwd['alphal'], rtol=1e-10)
    np.testing.assert_allclose(tg['alphalt'], fwd['alphalt'], rtol=1e-10)
    np.testing.assert_allclose(tg['loss'], loss, rtol=1e-12)


def test_autograd_gradient_matches_central_differences():
    opt, P, batch = _setup(seed=12, B=2, T=3, K=2, t=3)
    kw = dict(decay_c=1e-3, alpha_c=0.7)
    g = OG.loss_and_grads(P, opt, batch, **kw)['grads']
    rng = np.random.RandomState(0)

    def loss_at(Pm):
        return float(O.total_loss(Pm, opt, O.build_model_forward(Pm, opt, **batch), **kw))
    eps = 1e-6
    for name, v in P.items():
        flat_idx = rng.choice(v.size, size=min(3, v.size), replace=False)
        for fi in flat_idx:
            idx = np.unravel_index(fi, v.shape) if v.shape else ()
            Pp = OrderedDict(P); Pm = OrderedDict(P)
            vp = v.copy(); vm = v.copy()
            if v.shape:
                vp[idx] += eps; vm[idx] -= eps
            else:
                vp = vp + eps; vm = vm - eps
            Pp[name] = vp; Pm[name] = vm
            num = (loss_at(Pp) - loss_at(Pm)) / (2 * eps)
            ana = g[name][idx] if v.shape else g[name]
            assert abs(num - ana) <= 1e-6 * max(1.0, abs(num)), (name, idx, num, ana)


def test_dropout_masks_flow_through_both_restatements():
    opt, P, batch = _setup(seed=13, B=2, t=3)
    rng = np.random.RandomState(2)
    t, B = batch['x'].shape
    dp = rng.binomial(1, 0.5, (t, B, 48)).astype(np.float64)
    d1 = rng.binomial(1, 0.5, (t, B, 16)).astype(np.float64)
    d2 = rng.binomial(1, 0.5, (t, B, 8)).astype(np.float64)
    fwd = O.build_model_forward(P, opt, **batch, dp_mask=dp, d1=d1, d2=d2)
    tg = OG.loss_and_grads(P, opt, batch, dropout=dict(dp=dp, d1=d1, d2=d2), want=('loss',))
    np.testing.assert_allclose(tg['cost'], fwd['cost'], rtol=1e-10)


def test_gen_sample_greedy_is_argmax_chain_and_beam_sorted():
    opt, P, batch = _setup(seed=14)
    a = (batch['ctxg'][0], batch['mask_ctxg'][0], batch['ctxl'][0], None, batch['ctxm'][0], None)
    fi = lambda g, m: O.f_init(P, opt, g, m)
    fn = lambda *args: O.f_next(P, opt, *args)
    s, sc, _, _ = O.gen_sample(fi, fn, *a, k=1, maxlen=6)
    # greedy == repeated argmax
    _, h, c = fi(a[0], a[1]); h, c = h[None], c[None]; w = np.array([-1]); ref = []; score = 0.0
    for _ in range(6):
        p, _, h, c = fn(w, *a, h, c)
        wi = int(p[0].argmax()); ref.append(wi); score -= np.log(p[0, wi]); w = np.array([wi])
        if wi == 0:
            break
    assert s[0] == ref
    np.testing.assert_allclose(sc[0], score, rtol=1e-5)
    s5, sc5, _, _ = O.gen_sample(fi, fn, *a, k=5, maxlen=6)
    assert len(s5) == 5 and min(sc5) <= sc[0] + 1e-6


def test_clip_and_adadelta_follow_common_py():
    g = OrderedDict(a=np.array([3.0, 4.0]), b=np.array([12.0]))
    c = O.clip_grads(g, 1.0)                      # norm 13 -> scaled to 1
    np.testing.assert_allclose(np.sqrt(sum((v ** 2).sum() for v in c.values())), 1.0)
    assert O.clip_grads(g, 100.0) is g
    p = OrderedDict(a=np.array([1.0, 1.0])); gr = OrderedDict(a=np.array([0.5, -2.0]))
    rg2 = OrderedDict(a=np.zeros(2)); ru2 = OrderedDict(a=np.zeros(2))
    O.adadelta_update(p, gr, rg2, ru2)
    e_rg2 = 0.05 * gr['a'] ** 2
    ud = -np.sqrt(1e-6) / np.sqrt(e_rg2 + 1e-6) * gr['a']
    np.testing.assert_allclose(rg2['a'], e_rg2); np.testing.assert_allclose(p['a'], 1.0 + ud)
    np.testing.assert_allclose(ru2['a'], 0.05 * ud ** 2)


# ------------------------------------------------------------------ committed golden vectors
import os as _os

GOLD = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'golden')
GOLD_DIMS = dict(dim=64, dim_word=64, n_words=37, ctxg_dim=64, ctxl_dim=32, ctxm_dim=32, ctxglm_dim=64)


def _gold():
    P32 = OrderedDict(np.load(_os.path.join(GOLD, 'params.npz')))
    opt = O.default_options(**GOLD_DIMS)
    order = list(O.param_shapes(opt))
    return opt, OrderedDict((k, P32[k]) for k in order)


def test_oracle_reproduces_committed_golden_vectors():
    """tests/golden/*.npz were written by make_golden.py with the float64 oracle; the oracle must keep
    reproducing them (float64: to rounding; float32 arithmetic: within the 1e-4 parity bar)."""
    opt, P32 = _gold()
    P = O.cast_params(P32, np.float64)
    tg = np.load(_os.path.join(GOLD, 'train_graph.npz'))
    batch = {k[3:]: tg[k] for k in tg.files if k.startswith('in_')}
    b64 = {k: (v if v.dtype == np.int64 else v.astype(np.float64)) for k, v in batch.items()}
    fwd = O.build_model_forward(P, opt, **b64)
    for k in ('cost', 'probs', 'alphal', 'alphag', 'alpham', 'alphalt', 'logit', 'h', 'c', 'ctx'):
        np.testing.assert_allclose(fwd[k], tg[k], rtol=1e-10, atol=1e-12)
    f32 = O.build_model_forward(P32, opt, **batch)                      # the reference's floatX=float32 arithmetic
    for k in ('alphal', 'alphag', 'alpham', 'alphalt', 'logit'):
        assert np.abs(f32[k] - tg[k]).max() < 1e-4, k
    gr = OG.loss_and_grads(P, opt, batch, decay_c=1e-4, alpha_c=0.70602)
    np.testing.assert_allclose(gr['loss'], float(tg['loss']), rtol=1e-12)
    for k, v in gr['grads'].items():
        np.testing.assert_allclose(v, tg['grad_' + k], rtol=1e-9, atol=1e-13)
    sc = np.load(_os.path.join(GOLD, 'sampler_chain.npz'))
    g, l, m, gm = sc['ctxg'].astype(np.float64), sc['ctxl'].astype(np.float64), sc['ctxm'].astype(np.float64), sc['ctxg_mask']
    _, h0, c0 = O.f_init(P, opt, g, gm.astype(np.float64))
    np.testing.assert_allclose(h0, sc['h0'], rtol=1e-10)
    for mm in (1, 3):
        h, c = sc['m%d_h_in' % mm], sc['m%d_c_in' % mm]
        for s in range(3):
            (probs, _, h, c), r = O.f_next(P, opt, sc['m%d_s%d_x' % (mm, s)], g, gm, l, None, m, None, h, c, extras=True)
            np.testing.assert_allclose(probs, sc['m%d_s%d_probs' % (mm, s)], rtol=1e-10, atol=1e-14)
            np.testing.assert_allclose(r['alphal'], sc['m%d_s%d_alphal' % (mm, s)], rtol=1e-10)
