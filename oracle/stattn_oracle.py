"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

numpy restatement of the spatial-temporal-attention LSTM caption decoder of
tuyunbin/Video-Description-with-Spatial-Temporal-Attention (model_attention.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file, and only as the checker.  The product (the stattn package + the HIP
library) never imports it and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference is Python 2 + Theano; neither is importable in
the build container or on the GPU box, and the reference
has no numeric tests, golden vectors or published checkpoints.  This oracle is
therefore an argument from the reference *source*, line by line, not an
execution of it.  It is cross-checked three independent ways in tests/:
  (1) analytic known-answer tests that need no restatement (zero weights =>
      uniform attention, sigma(0) gates, 1/V probs; permutation equivariance;
      softmax shift invariance; f_next chain == build_model),
  (2) an independent torch-autograd restatement of the training loss
      (oracle/stattn_oracle_grad.py) -- forward equality and gradients,
  (3) central finite differences in float64 for the gradient.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Semantics of Theano ops that had to be inferred are listed in
SURVEY.md Appendix C and repeated where used.

dtype: every function computes in the dtype of `params` (float32 reproduces the
reference's floatX=float32 arithmetic order op by op; float64 is the "truth"
used for golden vectors).
"""
from collections import OrderedDict

import numpy as np


# ---------------------------------------------------------------------------
# small numeric helpers (Theano op semantics)
# ---------------------------------------------------------------------------
def _sigmoid(x):
    # theano.tensor.nnet.sigmoid; the fp32 clamp of Theano's ultra-fast variant is
    # irrelevant at 1e-4
    return 1.0 / (1.0 + np.exp(-x))


def _softmax_rows(e):
    # tensor.nnet.softmax: row-wise over the last axis, max-subtracted
    # (model_attention.py:380, 398, 411, 425, 708, 840; Appendix C.3)
    m = e.max(axis=-1, keepdims=True)
    p = np.exp(e - m)
    return p / p.sum(axis=-1, keepdims=True)


def _ff(params, prefix, x, activ):
    # Attention.fflayer, model_attention.py:89-92: activ(x . W + b)
    y = x @ params[prefix + '_W'] + params[prefix + '_b']
    if activ == 'tanh':
        return np.tanh(y)
    if activ == 'linear':
        return y
    if activ == 'rectifier':
        return np.maximum(0.0, y)
    raise ValueError(activ)


def default_options(**kw):
    """Options consumed by the hot path (SURVEY section 5, config.py:17-48)."""
    o = dict(dim=1024, dim_word=512, n_words=12000,
             ctxg_dim=1024, ctxl_dim=4096, ctxm_dim=4096, ctxglm_dim=1024,
             selector=True, use_dropout=True, prev2out=True, ctx2out=True,
             n_layers_out=1, n_layers_init=0, encoder='none')
    o.update(kw)
    # hard constraint of the reference graph: ff_global is commented out
    # (model_attention.py:553-554) so the global feature dim must equal dim.
    assert o['ctxg_dim'] == o['dim'] == o['ctxglm_dim']
    return o


def param_shapes(options):
    """Name -> shape in dict order (= npz order = gradient order).
    model_attention.py:518-581 (init_params) and 180-282 (param_init_lstm_cond)."""
    D, E, V = options['dim'], options['dim_word'], options['n_words']
    Dg, Fl, Fm = options['ctxg_dim'], options['ctxl_dim'], options['ctxm_dim']
    s = OrderedDict()
    s['Wemb'] = (V, E)                                   # :522
    s['ff_state_W'] = (Dg, D); s['ff_state_b'] = (D,)    # :549-550
    s['ff_memory_W'] = (Dg, D); s['ff_memory_b'] = (D,)  # :551-552
    s['ff_local_W'] = (Fl, D); s['ff_local_b'] = (D,)    # :556-557
    s['ff_motion_W'] = (Fm, D); s['ff_motion_b'] = (D,)  # :558-559
    p = 'decoder_'
    s[p + 'W'] = (E, 4 * D)                              # :189-193
    s[p + 'U'] = (D, 4 * D)                              # :196-200
    s[p + 'b'] = (4 * D,)                                # :203
    s[p + 'Wc'] = (D, 4 * D)                             # :206-207
    s[p + 'Wcg_att'] = (D, D)                            # :210
    s[p + 'Wcm_att'] = (D, D)                            # :214
    s[p + 'Wclt_att'] = (D, D)                           # :218
    s[p + 'Wdg_att'] = (D, D)                            # :222
    s[p + 'Wdm_att'] = (D, D)                            # :225
    s[p + 'Wdlt_att'] = (D, D)                           # :228
    s[p + 'bg_att'] = (D,)                               # :232
    s[p + 'bm_att'] = (D,)                               # :235
    s[p + 'blt_att'] = (D,)                              # :239
    s[p + 'Wcl_att'] = (D, D)                            # :243
    s[p + 'Wdl_att'] = (D, D)                            # :247
    s[p + 'bl_att'] = (D,)                               # :251
    s[p + 'Ug_att'] = (D, 1); s[p + 'cg_att'] = (1,)     # :255-258
    s[p + 'Um_att'] = (D, 1); s[p + 'cm_att'] = (1,)     # :260-263
    s[p + 'Ult_att'] = (D, 1); s[p + 'clt_att'] = (1,)   # :265-268
    s[p + 'Ul_att'] = (D, 1); s[p + 'cl_att'] = (1,)     # :271-274
    if options['selector']:
        s[p + 'W_sel'] = (D, 1); s[p + 'b_sel'] = ()     # :276-281
    s['ff_logit_lstm_W'] = (D, E); s['ff_logit_lstm_b'] = (E,)          # :566-568
    if options['ctx2out']:
        s['ff_logit_ctxglm_W'] = (options['ctxglm_dim'], E)             # :569-572
        s['ff_logit_ctxglm_b'] = (E,)
    s['ff_logit_W'] = (E, V); s['ff_logit_b'] = (V,)                    # :578-580
    return s


def random_params(options, seed=1234, dtype=np.float32, scale=None):
    """Random *test* weights with the reference's shapes.  Not the reference's init
    distribution (that lives in the product's init_params); the oracle only needs
    weights that exercise every term, so biases and c*_att are non-zero here.
    `scale` maps name -> std to make attention non-trivially peaked in tests."""
    rng = np.random.RandomState(seed)
    P = OrderedDict()
    for k, shp in param_shapes(options).items():
        fan_in = shp[0] if len(shp) == 2 else 1
        std = 1.0 / np.sqrt(max(fan_in, 1)) if len(shp) == 2 else 0.1
        if k == 'Wemb':
            std = 0.5
        if scale and k in scale:
            std = scale[k]
        P[k] = (std * rng.standard_normal(shp)).astype(dtype)
    return P


def cast_params(params, dtype):
    return OrderedDict((k, np.asarray(v, dtype=dtype)) for k, v in params.items())


# ---------------------------------------------------------------------------
# the decoder cell: lstm_cond_layer pre-amble + _step
# ---------------------------------------------------------------------------
def project_contexts(params, G, L, M, prefix='decoder_'):
    """model_attention.py:322-326: pctxg_, pctxl_, pctxm_.
    tensor.dot(ND, 2D) contracts the last axis of the left operand (Appendix C.2)."""
    PG = G @ params[prefix + 'Wcg_att'] + params[prefix + 'bg_att']
    PL = L @ params[prefix + 'Wcl_att'] + params[prefix + 'bl_att']
    PM = M @ params[prefix + 'Wcm_att'] + params[prefix + 'bm_att']
    return PG, PL, PM


def step(params, options, m_, x_, dp_, h_, c_, PG, PL, PM, G, L, M, prefix='decoder_'):
    """One decoder timestep: _step, model_attention.py:366-459.

    m_ (m,) mask; x_ (m,4D) = emb.W+b (:334-335); dp_ (m,3D) dropout multiplier on
    the i/f/o pre-activations (:444-447); h_, c_ (m,D); G (m|1,T,D), L (m|1,T,K,D),
    M (m|1,T,D) and their projections PG/PL/PM.  A leading dim of 1 broadcasts over
    the m hypotheses (sampler, :786-788, :330-332).
    Returns a dict with the reference's rval entries 0..10 plus the raw scores."""
    D = h_.shape[1]
    p = params
    # --- spatial attention over K regions, :371-383
    pstatel = h_ @ p[prefix + 'Wdl_att']                                   # :371
    tl = np.tanh(PL + pstatel[:, None, None, :])                           # :372-375
    el = (tl @ p[prefix + 'Ul_att'])[..., 0] + p[prefix + 'cl_att'][0]     # :377  (m,T,K)
    alphal = _softmax_rows(el)                                             # :380-381 over K
    CL = (L * alphal[:, :, :, None]).sum(2)                                # :383  (m,T,D)
    # --- temporal attention, global, :389-399 (no frame mask: Appendix C.5)
    pstateg = h_ @ p[prefix + 'Wdg_att']
    tg = np.tanh(PG + pstateg[:, None, :])
    eg = (tg @ p[prefix + 'Ug_att'])[..., 0] + p[prefix + 'cg_att'][0]     # (m,T)
    alphag = _softmax_rows(eg)
    cg = (G * alphag[:, :, None]).sum(1)                                   # :399
    # --- temporal attention, motion, :402-412
    pstatem = h_ @ p[prefix + 'Wdm_att']
    tm = np.tanh(PM + pstatem[:, None, :])
    em = (tm @ p[prefix + 'Um_att'])[..., 0] + p[prefix + 'cm_att'][0]
    alpham = _softmax_rows(em)
    cm = (M * alpham[:, :, None]).sum(1)                                   # :412
    # --- temporal attention over the spatially attended local ctx, :415-426
    pstatelt = h_ @ p[prefix + 'Wdlt_att']                                 # :415
    pctxlt = CL @ p[prefix + 'Wclt_att'] + p[prefix + 'blt_att']           # :416
    tlt = np.tanh(pctxlt + pstatelt[:, None, :])                           # :417-420
    elt = (tlt @ p[prefix + 'Ult_att'])[..., 0] + p[prefix + 'clt_att'][0]
    alphalt = _softmax_rows(elt)                                           # :425
    clt = (CL * alphalt[:, :, None]).sum(1)                                # :426
    # --- fusion by SUM, :430, and the selector gate, :432-435
    ctx = cg + cm + clt
    sel = None
    if options['selector']:
        sel = _sigmoid((h_ @ p[prefix + 'W_sel'])[:, 0] + p[prefix + 'b_sel'])  # :433-434
        ctx = sel[:, None] * ctx                                           # :435
    # --- LSTM, :437-457.  bias is inside x_; gate order i, f, o, c~ (:441-451)
    preact = h_ @ p[prefix + 'U'] + x_ + ctx @ p[prefix + 'Wc']            # :437-439
    i = preact[:, 0 * D:1 * D]
    f = preact[:, 1 * D:2 * D]
    o = preact[:, 2 * D:3 * D]
    if options['use_dropout']:                                             # :444-447
        i = i * dp_[:, 0 * D:1 * D]
        f = f * dp_[:, 1 * D:2 * D]
        o = o * dp_[:, 2 * D:3 * D]
    i = _sigmoid(i); f = _sigmoid(f); o = _sigmoid(o)                      # :448-450
    g = np.tanh(preact[:, 3 * D:4 * D])                                    # :451 (no dropout)
    c = f * c_ + i * g                                                     # :453
    c = m_[:, None] * c + (1.0 - m_)[:, None] * c_                         # :454
    h = o * np.tanh(c)                                                     # :456 (masked c)
    h = m_[:, None] * h + (1.0 - m_)[:, None] * h_                         # :457
    return dict(h=h, c=c, alphal=alphal, CL=CL, alphag=alphag, cg=cg, alpham=alpham,
                cm=cm, alphalt=alphalt, clt=clt, ctx=ctx, sel=sel,
                el=el, eg=eg, em=em, elt=elt, i=i, f=f, o=o, g=g, preact=preact)


def readout(params, options, h, emb, ctx, d1, d2):
    """model_attention.py:684-705 / 817-838.  d1, d2: dropout multipliers (0.5 at
    eval: common.py:94-99, non-inverted).  Returns (logit, pre-tanh z)."""
    assert options['n_layers_out'] == 1
    ph = h * d1 if options['use_dropout'] else h                           # :684-685
    z = _ff(params, 'ff_logit_lstm', ph, 'linear')                         # :687-688
    if options['prev2out']:
        z = z + emb                                                        # :689-690
    if options['ctx2out']:
        z = z + _ff(params, 'ff_logit_ctxglm', ctx, 'linear')              # :691-693
    a = np.tanh(z)                                                         # :694
    if options['use_dropout']:
        a = a * d2                                                         # :695-696
    logit = _ff(params, 'ff_logit', a, 'linear')                           # :704-705
    return logit, z


# ---------------------------------------------------------------------------
# sampler graph: f_init / f_next  (build_sampler, model_attention.py:719-850)
# ---------------------------------------------------------------------------
def f_init(params, options, ctxg, ctxg_mask):
    """model_attention.py:791-795.  ctxg (T,Dg), ctxg_mask (T,) -> [ctxg, h0(D,), c0(D,)]."""
    counts = ctxg_mask.sum(-1)                                             # :739
    mean = ctxg.sum(0) / counts                                            # :766
    h0 = _ff(params, 'ff_state', mean, 'tanh')                             # :776-777
    c0 = _ff(params, 'ff_memory', mean, 'tanh')                            # :778-779
    return [ctxg, h0, c0]


def project_video(params, options, ctxg, ctxl, ctxm):
    """The part of f_next the reference recomputes on EVERY call
    (model_attention.py:782-788 and 322-326): ff_local / ff_motion F->D and the
    three attention pre-projections.  Leading broadcast dim of 1 (:786-788)."""
    L = _ff(params, 'ff_local', ctxl, 'tanh')[None]                        # :782-783, :787
    M = _ff(params, 'ff_motion', ctxm, 'tanh')[None]                       # :784-785, :788
    G = ctxg[None]                                                         # :786 (ff_global commented out :780-781)
    PG, PL, PM = project_contexts(params, G, L, M)
    return G, L, M, PG, PL, PM


def f_next(params, options, x, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask, h, c,
           extras=False, cached=None):
    """model_attention.py:845-848 with use_noise=0.
    x (m,) int64 (-1 = first word), h,c (m,D) -> [probs(m,V), sample(m,), h', c'].
    ctxl_mask / ctxm_mask are unused by the graph (on_unused_input='ignore', :848).
    `cached` = project_video(...) result to skip the reference's per-call
    re-projection (the "fair" CPU variant of BASELINE.md section 3).
    next_sample: the reference draws from multinomial(next_probs) with Theano's MRG
    stream (:841); bit-parity with MRG is not a goal, the oracle returns argmax."""
    dt = params['Wemb'].dtype
    G, L, M, PG, PL, PM = cached if cached is not None else project_video(params, options, ctxg, ctxl, ctxm)
    m = x.shape[0]
    E = params['Wemb'].shape[1]
    D = h.shape[1]
    emb = np.where((x < 0)[:, None], np.zeros((1, E), dt), params['Wemb'][np.maximum(x, 0)])  # :803-804
    x_ = emb @ params['decoder_W'] + params['decoder_b']                   # :334-335
    dp = np.full((m, 3 * D), 0.5, dt)                                      # :469-472 use_noise=0
    ones = np.ones((m,), dt)                                               # :310-311, Appendix C.1
    r = step(params, options, ones, x_, dp, h, c, PG, PL, PM, G, L, M)
    half = dt.type(0.5)
    logit, _ = readout(params, options, r['h'], emb, r['ctx'], half, half)  # :817-838
    probs = _softmax_rows(logit)                                           # :840
    sample = probs.argmax(1).astype(np.int64)
    out = [probs, sample, r['h'], r['c']]
    if extras:
        r['logit'] = logit
        return out, r
    return out


# ---------------------------------------------------------------------------
# training graph: build_model forward (model_attention.py:583-717)
# ---------------------------------------------------------------------------
def build_model_forward(params, options, x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm,
                        dp_mask=None, d1=None, d2=None, keep=False):
    """x (t,m) int64, mask (t,m), ctxg (m,T,Dg), mask_ctxg (m,T), ctxl (m,T,K,F),
    ctxm (m,T,F).  dp_mask (t,m,3D), d1 (t,m,D), d2 (t,m,E): dropout multipliers;
    None => use_noise=0 => the constant 0.5 (:474-477, common.py:94-99).
    Returns dict(cost(m,), probs(t*m,V), alphal(t,m,T,K), alphag/m/lt(t,m,T), logit(t,m,V), ...)."""
    dt = params['Wemb'].dtype
    t, m = x.shape
    D, E = options['dim'], options['dim_word']
    emb = params['Wemb'][x.flatten()].reshape(t, m, E)                     # :613-614
    emb_shifted = np.zeros_like(emb)                                       # :615-617
    emb_shifted[1:] = emb[:-1]
    emb = emb_shifted
    counts = mask_ctxg.sum(-1)[:, None]                                    # :618
    mean = ctxg.sum(1) / counts                                            # :649
    h = _ff(params, 'ff_state', mean, 'tanh')                              # :657-658
    c = _ff(params, 'ff_memory', mean, 'tanh')                             # :659-660
    G = ctxg                                                               # :646 (ff_global commented out :661-662)
    L = _ff(params, 'ff_local', ctxl, 'tanh')                              # :664-665
    M = _ff(params, 'ff_motion', ctxm, 'tanh')                             # :666-667
    PG, PL, PM = project_contexts(params, G, L, M)                         # :322-326
    x_all = emb @ params['decoder_W'] + params['decoder_b']                # :334-335
    half = dt.type(0.5)
    if dp_mask is None:
        dp_mask = np.full((t, m, 3 * D), half, dt)
    hs, cs, ctxs = [], [], []
    al, ag, am, alt = [], [], [], []
    steps = []
    for tt in range(t):                                                    # theano.scan :495-512
        r = step(params, options, mask[tt], x_all[tt], dp_mask[tt], h, c, PG, PL, PM, G, L, M)
        h, c = r['h'], r['c']
        hs.append(h); cs.append(c); ctxs.append(r['ctx'])
        al.append(r['alphal']); ag.append(r['alphag']); am.append(r['alpham']); alt.append(r['alphalt'])
        if keep:
            steps.append(r)
    proj_h = np.stack(hs)                                                  # proj[0]  :678
    ctxs = np.stack(ctxs)                                                  # proj[10] :683
    logit, _ = readout(params, options, proj_h, emb, ctxs,
                       half if d1 is None else d1, half if d2 is None else d2)
    V = logit.shape[-1]
    probs = _softmax_rows(logit.reshape(t * m, V))                         # :708-709
    x_flat = x.flatten()
    cost = -np.log(probs[np.arange(t * m), x_flat] + dt.type(1e-8))        # :712 (C.13: not log-softmax)
    cost = (cost.reshape(t, m) * mask).sum(0)                              # :714-715
    out = dict(cost=cost, probs=probs, logit=logit,
               alphal=np.stack(al), alphag=np.stack(ag), alpham=np.stack(am), alphalt=np.stack(alt),
               h=proj_h, c=np.stack(cs), ctx=ctxs, L=L, M=M, PG=PG, PL=PL, PM=PM)
    if keep:
        out['steps'] = steps
    return out


def total_loss(params, options, fwd, decay_c=0.0, alpha_c=0.0):
    """model_attention.py:1129-1147: mean NLL + L2 decay + 'doubly stochastic'
    attention regulariser on the four alpha tensors.  alphas.sum(0) runs over ALL
    nsteps including masked ones."""
    dt = params['Wemb'].dtype
    loss = fwd['cost'].mean()                                              # :1129
    if decay_c > 0.0:
        wd = dt.type(0.0)
        for v in params.values():
            wd = wd + (v ** 2).sum()                                       # :1133-1134
        loss = loss + dt.type(decay_c) * wd                                # :1135-1136
    if alpha_c > 0.0:
        for name in ('alphag', 'alphal', 'alpham', 'alphalt'):             # :1140-1147
            a = fwd[name]
            loss = loss + dt.type(alpha_c) * ((1.0 - a.sum(0)) ** 2).sum(0).mean()
    return loss


def clip_grads(grads, clip_c):
    """model_attention.py:1194-1203: global-norm clip."""
    if clip_c <= 0.0:
        return grads
    g2 = sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in grads.values())
    if g2 > clip_c ** 2:
        s = clip_c / np.sqrt(g2)
        return OrderedDict((k, (g * g.dtype.type(s)).astype(g.dtype)) for k, g in grads.items())
    return grads


def adadelta_update(params, grads, rg2, ru2):
    """common.py:178-195.  f_grad_shared part: rg2 <- .95 rg2 + .05 g^2 (:184);
    f_update part: ud = -sqrt(ru2+1e-6)/sqrt(rg2+1e-6)*g (:189); ru2 <- .95 ru2 +
    .05 ud^2 (:190); p <- p + ud (:191).  `lr` is ignored by the reference (C.8).
    In place on the dicts; returns nothing."""
    for k in params:
        g = grads[k]
        dt = g.dtype.type
        rg2[k] = dt(0.95) * rg2[k] + dt(0.05) * g * g
        ud = -np.sqrt(ru2[k] + dt(1e-6)) / np.sqrt(rg2[k] + dt(1e-6)) * g
        ru2[k] = dt(0.95) * ru2[k] + dt(0.05) * ud * ud
        params[k] = params[k] + ud


# ---------------------------------------------------------------------------
# beam search driver: gen_sample (model_attention.py:852-994)
# ---------------------------------------------------------------------------
def gen_sample(f_init_fn, f_next_fn, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask,
               k=1, maxlen=30, stochastic=False, suppress_eos=False):
    """Restatement of gen_sample with `//` for the py2 `/` at :926 (Appendix C.9).
    f_init_fn(ctxg, mask) -> [ctxg, h0, c0]; f_next_fn(x, ctxg, gm, ctxl, lm, ctxm, mm, h, c)
    -> [probs, sample, h, c].  suppress_eos: bench/test aid that forbids word 0 so
    every hypothesis runs maxlen steps."""
    if k > 1:
        assert not stochastic                                              # :863-864
    sample, sample_score = [], []
    if stochastic:
        sample_score = 0
    live_k, dead_k = 1, 0
    hyp_samples = [[]] * live_k
    hyp_scores = np.zeros(live_k, np.float32)                              # :875
    rval = f_init_fn(ctxg_0, ctxg_mask)                                    # :880
    ctxg_0 = rval[0]
    next_state = rval[1].reshape(live_k, -1)                               # :887-892
    next_memory = rval[2].reshape(live_k, -1)
    next_w = -1 * np.ones((1,), np.int64)                                  # :893
    for _ in range(maxlen):                                                # :896
        rval = f_next_fn(next_w, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask,
                         next_state, next_memory)                          # :903
        next_p = rval[0]
        if suppress_eos:
            next_p = next_p.copy(); next_p[:, 0] = 0.0
        next_w, next_state, next_memory = rval[1], rval[2], rval[3]
        if stochastic:                                                     # :914-918
            sample.append(int(next_w[0]))
            sample_score += next_p[0, next_w[0]]
            if next_w[0] == 0:
                break
            continue
        with np.errstate(divide='ignore'):
            cand_scores = hyp_scores[:, None] - np.log(next_p)             # :921
        cand_flat = cand_scores.flatten()
        ranks_flat = cand_flat.argsort()[:(k - dead_k)]                    # :923
        voc_size = next_p.shape[1]
        trans_indices = ranks_flat // voc_size                             # :926
        word_indices = ranks_flat % voc_size                               # :927
        costs = cand_flat[ranks_flat]
        new_hyp_samples, new_hyp_scores = [], np.zeros(k - dead_k, np.float32)
        new_hyp_states, new_hyp_memories = [], []
        for idx, (ti, wi) in enumerate(zip(trans_indices, word_indices)):  # :939-945
            new_hyp_samples.append(hyp_samples[ti] + [int(wi)])
            new_hyp_scores[idx] = costs[idx]
            new_hyp_states.append(next_state[ti].copy())
            new_hyp_memories.append(next_memory[ti].copy())
        new_live_k = 0
        hyp_samples, hyp_scores, hyp_states, hyp_memories = [], [], [], []
        for idx in range(len(new_hyp_samples)):                            # :958-970
            if new_hyp_samples[idx][-1] == 0:
                sample.append(new_hyp_samples[idx])
                sample_score.append(new_hyp_scores[idx])
                dead_k += 1
            else:
                new_live_k += 1
                hyp_samples.append(new_hyp_samples[idx])
                hyp_scores.append(new_hyp_scores[idx])
                hyp_states.append(new_hyp_states[idx])
                hyp_memories.append(new_hyp_memories[idx])
        hyp_scores = np.array(hyp_scores)
        live_k = new_live_k
        if new_live_k < 1 or dead_k >= k:                                  # :974-977
            break
        next_w = np.array([w[-1] for w in hyp_samples], np.int64)          # :979
        next_state = np.array(hyp_states)
        next_memory = np.array(hyp_memories)
    if not stochastic and live_k > 0:                                      # :987-992
        for idx in range(live_k):
            sample.append(hyp_samples[idx])
            sample_score.append(hyp_scores[idx])
    return sample, sample_score, [next_state], [next_memory]               # one-element lists (n_layers_lstm = 1, :980-994)


def sampler_closures(params, options, out_dtype=np.float32, draw_seed=None):
    """(f_init, f_next) closures over this oracle with the calling convention of the compiled Theano functions
    (model_attention.py:791-795, :845-848): outputs in `out_dtype` (floatX = float32 in the reference), `sample`
    int64.  With `draw_seed` the returned sample is an inverse-CDF draw from the probabilities (stand-in for the
    multinomial of :841; seeded numpy stream, not MRG) instead of the arg-max.  Used to drive gen_sample drivers --
    the reference's own (tests/golden/make_ref_fixtures.py), this file's, and the product's host loop -- with
    identical numbers."""
    rng = np.random.RandomState(draw_seed) if draw_seed is not None else None
    f64 = lambda a: None if a is None else np.asarray(a, np.float64)

    def fi(ctxg, ctxg_mask):
        g, h0, c0 = f_init(params, options, f64(ctxg), f64(ctxg_mask))
        return [np.asarray(ctxg), h0.astype(out_dtype), c0.astype(out_dtype)]

    def fn(x, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask, h, c):
        probs, sample, h1, c1 = f_next(params, options, np.asarray(x, np.int64), f64(ctxg), f64(ctxg_mask), f64(ctxl),
                                       None, f64(ctxm), None, f64(h), f64(c))
        if rng is not None:
            u = rng.uniform(size=(probs.shape[0], 1))
            sample = np.minimum((np.cumsum(probs, axis=1) < u).sum(1), probs.shape[1] - 1).astype(np.int64)
        return [probs.astype(out_dtype), sample, h1.astype(out_dtype), c1.astype(out_dtype)]
    return fi, fn


# ---------------------------------------------------------------------------
# synthetic MSVD-shaped inputs
# ---------------------------------------------------------------------------
def synthetic_batch(options, B, T, K, t, seed=1234, dtype=np.float32, ragged=True):
    """prepare_data's output layout (data_engine.py:258-337): x (t,B) int64, mask
    (t,B) f32, ctxg (B,T,Dg), ctxl (B,T,K,F), ctxm (B,T,F), all masks = 1."""
    rng = np.random.RandomState(seed)
    V = options['n_words']
    ctxg = rng.standard_normal((B, T, options['ctxg_dim'])).astype(dtype)
    ctxl = rng.standard_normal((B, T, K, options['ctxl_dim'])).astype(dtype)
    ctxm = rng.standard_normal((B, T, options['ctxm_dim'])).astype(dtype)
    x = np.zeros((t, B), np.int64)
    mask = np.zeros((t, B), dtype)
    for b in range(B):
        ln = rng.randint(min(5, t - 1), t) if ragged else t - 1            # lengths in [5, t-1]
        if b == 0:
            ln = t - 1                                                     # maxlen = max(len)+1, data_engine.py:329
        x[:ln, b] = rng.randint(2, V, size=ln)
        mask[:ln + 1, b] = 1.0                                             # data_engine.py:331-335
    return dict(x=x, mask=mask, ctxg=ctxg, mask_ctxg=np.ones((B, T), dtype),
                ctxl=ctxl, mask_ctxl=np.ones((B, T, K), dtype),
                ctxm=ctxm, mask_ctxm=np.ones((B, T), dtype))
