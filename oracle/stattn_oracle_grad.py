"""
CPU ORACLE (gradient leg) -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

An independent torch restatement of the *training loss* of model_attention.py so
that reverse-mode autodiff (the analogue of the reference's `tensor.grad`,
model_attention.py:1193) yields the gradient oracle for the hand-written HIP
backward.  It is deliberately written separately from stattn_oracle.py (einsum /
torch idiom instead of numpy broadcasting) so the two restatements check each
other; tests/test_oracle.py asserts they agree on the forward and that the
autograd gradient matches float64 central differences.

PARITY UNPINNED (see stattn_oracle.py header): Theano is not importable here.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
from collections import OrderedDict

import numpy as np
import torch


def _p(params, name):
    return params['decoder_' + name]


def loss_and_grads(params_np, options, batch, decay_c=0.0, alpha_c=0.0, dropout=None,
                   dtype=torch.float64, nll_scale=None, want=('loss', 'grads')):
    """Total loss of model_attention.py:1129-1147 and d(loss)/d(every param).

    batch: dict with x(t,m) int64, mask(t,m), ctxg(m,T,D), mask_ctxg(m,T), ctxl(m,T,K,F), ctxm(m,T,F).
    dropout: None (use_noise=0 -> constant 0.5 multipliers) or dict(dp=(t,m,3D), d1=(t,m,D), d2=(t,m,E)).
    nll_scale: weight of sum_b cost[b]; default 1/m = cost.mean() (:1129).  The data-parallel
      tests pass 1/B_global here.
    Returns dict(loss, cost, grads (OrderedDict name->np.ndarray), alphas...)."""
    P = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)) for k, v in params_np.items())
    x = torch.as_tensor(batch['x'])
    mask = torch.tensor(batch['mask'], dtype=dtype)
    G = torch.tensor(batch['ctxg'], dtype=dtype)
    mg = torch.tensor(batch['mask_ctxg'], dtype=dtype)
    ctxl = torch.tensor(batch['ctxl'], dtype=dtype)
    ctxm = torch.tensor(batch['ctxm'], dtype=dtype)
    t, m = x.shape
    D, E = options['dim'], options['dim_word']

    # embedding lookup, shifted one step forward in time (:613-617)
    emb = P['Wemb'][x.reshape(-1)].reshape(t, m, E)
    emb = torch.cat([torch.zeros(1, m, E, dtype=dtype), emb[:-1]], 0)
    # init state from the masked-count mean of the global features (:618, :649, :657-660)
    mean = G.sum(1) / mg.sum(-1, keepdim=True)
    h = torch.tanh(mean @ P['ff_state_W'] + P['ff_state_b'])
    c = torch.tanh(mean @ P['ff_memory_W'] + P['ff_memory_b'])
    # F -> D projections of local / motion features (:664-667)
    L = torch.tanh(torch.einsum('btkf,fd->btkd', ctxl, P['ff_local_W']) + P['ff_local_b'])
    M = torch.tanh(torch.einsum('btf,fd->btd', ctxm, P['ff_motion_W']) + P['ff_motion_b'])
    # attention pre-projections (:322-326)
    PG = torch.einsum('btd,de->bte', G, _p(P, 'Wcg_att')) + _p(P, 'bg_att')
    PL = torch.einsum('btkd,de->btke', L, _p(P, 'Wcl_att')) + _p(P, 'bl_att')
    PM = torch.einsum('btd,de->bte', M, _p(P, 'Wcm_att')) + _p(P, 'bm_att')
    xproj = emb @ _p(P, 'W') + _p(P, 'b')                                   # :334-335

    if dropout is None:
        dp = torch.full((t, m, 3 * D), 0.5, dtype=dtype)
        d1 = torch.full((t, m, D), 0.5, dtype=dtype)
        d2 = torch.full((t, m, E), 0.5, dtype=dtype)
    else:
        dp = torch.tensor(dropout['dp'], dtype=dtype)
        d1 = torch.tensor(dropout['d1'], dtype=dtype)
        d2 = torch.tensor(dropout['d2'], dtype=dtype)

    hs, ctxs, als, ags, ams, alts = [], [], [], [], [], []
    for s in range(t):                                                      # scan :495-512, body :366-459
        sl = h @ _p(P, 'Wdl_att')
        el = torch.einsum('btkd,d->btk', torch.tanh(PL + sl[:, None, None, :]), _p(P, 'Ul_att')[:, 0]) + _p(P, 'cl_att')
        al = torch.softmax(el, dim=2)                                       # over K regions :380
        CL = torch.einsum('btk,btkd->btd', al, L)                           # :383
        sg = h @ _p(P, 'Wdg_att')
        eg = torch.einsum('btd,d->bt', torch.tanh(PG + sg[:, None, :]), _p(P, 'Ug_att')[:, 0]) + _p(P, 'cg_att')
        ag = torch.softmax(eg, dim=1)                                       # over T frames, unmasked :398
        cg = torch.einsum('bt,btd->bd', ag, G)
        sm = h @ _p(P, 'Wdm_att')
        em = torch.einsum('btd,d->bt', torch.tanh(PM + sm[:, None, :]), _p(P, 'Um_att')[:, 0]) + _p(P, 'cm_att')
        am = torch.softmax(em, dim=1)
        cm = torch.einsum('bt,btd->bd', am, M)
        slt = h @ _p(P, 'Wdlt_att')
        plt = torch.einsum('btd,de->bte', CL, _p(P, 'Wclt_att')) + _p(P, 'blt_att')   # :416
        elt = torch.einsum('btd,d->bt', torch.tanh(plt + slt[:, None, :]), _p(P, 'Ult_att')[:, 0]) + _p(P, 'clt_att')
        alt = torch.softmax(elt, dim=1)
        clt = torch.einsum('bt,btd->bd', alt, CL)                           # :426
        ctx = cg + cm + clt                                                 # :430
        if options['selector']:
            sel = torch.sigmoid(h @ _p(P, 'W_sel')[:, 0] + _p(P, 'b_sel'))  # :433
            ctx = sel[:, None] * ctx
        pre = h @ _p(P, 'U') + xproj[s] + ctx @ _p(P, 'Wc')                 # :437-439
        pi, pf, po, pg = pre[:, :D], pre[:, D:2 * D], pre[:, 2 * D:3 * D], pre[:, 3 * D:]
        if options['use_dropout']:                                          # :444-447
            pi = pi * dp[s, :, :D]; pf = pf * dp[s, :, D:2 * D]; po = po * dp[s, :, 2 * D:]
        i, f, o, g = torch.sigmoid(pi), torch.sigmoid(pf), torch.sigmoid(po), torch.tanh(pg)
        mk = mask[s][:, None]
        c_new = f * c + i * g
        c_new = mk * c_new + (1 - mk) * c                                   # :454
        h_new = o * torch.tanh(c_new)
        h_new = mk * h_new + (1 - mk) * h                                   # :457
        h, c = h_new, c_new
        hs.append(h); ctxs.append(ctx); als.append(al); ags.append(ag); ams.append(am); alts.append(alt)

    H = torch.stack(hs); C = torch.stack(ctxs)
    ph = H * d1 if options['use_dropout'] else H                            # :684-685
    z = ph @ P['ff_logit_lstm_W'] + P['ff_logit_lstm_b']
    if options['prev2out']:
        z = z + emb
    if options['ctx2out']:
        z = z + C @ P['ff_logit_ctxglm_W'] + P['ff_logit_ctxglm_b']
    a = torch.tanh(z)
    if options['use_dropout']:
        a = a * d2
    logit = a @ P['ff_logit_W'] + P['ff_logit_b']
    probs = torch.softmax(logit.reshape(t * m, -1), dim=1)                  # :708-709
    px = probs[torch.arange(t * m), x.reshape(-1)]
    cost = (-(torch.log(px + 1e-8)).reshape(t, m) * mask).sum(0)            # :712-715
    scale = (1.0 / m) if nll_scale is None else nll_scale
    loss = cost.sum() * scale                                               # cost.mean() :1129
    AL, AG, AM, ALT = torch.stack(als), torch.stack(ags), torch.stack(ams), torch.stack(alts)
    if decay_c > 0:                                                         # :1130-1136
        loss = loss + decay_c * sum((v ** 2).sum() for v in P.values())
    if alpha_c > 0:                                                         # :1138-1147
        for A in (AG, AL, AM, ALT):
            loss = loss + alpha_c * ((1.0 - A.sum(0)) ** 2).sum(0).mean()
    out = dict(loss=float(loss.detach()), cost=cost.detach().numpy(), probs=probs.detach().numpy(),
               alphal=AL.detach().numpy(), alphag=AG.detach().numpy(), alpham=AM.detach().numpy(),
               alphalt=ALT.detach().numpy(), logit=logit.detach().numpy())
    if 'grads' in want:
        loss.backward()
        out['grads'] = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().copy())
                                   for k, v in P.items())
    return out
