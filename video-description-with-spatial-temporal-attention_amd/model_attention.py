"""Host-side mirror of the reference's `model_attention.Attention` for the decoder path.

Same method names, positional orders and return orders as the reference
(model_attention.py:42-994), so its callers -- Attention.train (:1034), metrics.py:28-40,126 --
keep working against this class:

    model   = Attention()
    params  = model.init_params(options)                 # :518-581
    tparams = model.init_tparams(params)                 # :70-78
    trng, use_noise, x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm, \
        alphals, alphags, alphams, alphalts, cost, extra = model.build_model(tparams, options)   # :583-717
    f_init, f_next = model.build_sampler(tparams, options, use_noise, trng)                      # :719-850
    sample, score, h, c = model.gen_sample(tparams, f_init, f_next, ctxg, ctxg_mask, ctxl, ctxl_mask,
                                           ctxm, ctxm_mask, options, None, k, maxlen)            # :852-994

The Theano graph is replaced by libstattn.so (hand-written gfx950 kernels behind a C ABI);
the symbolic variables build_model returns are inert name tags that `function()` below
recognises when it stands in for `theano.function` (f_log_probs, :1126; f_alpha*, :1167-1188)."""
import warnings
from collections import OrderedDict

import numpy

try:                                    # package import (stattn.model_attention)
    from . import common
    from ._native import Decoder
except ImportError:                     # top-level import with the package dir on sys.path
    import common
    from _native import Decoder

from_common = ('zipp', 'unzip', 'itemlist', 'norm_weight', 'ortho_weight', 'load_params')
zipp, unzip, itemlist = common.zipp, common.unzip, common.itemlist
norm_weight, ortho_weight, load_params = common.norm_weight, common.ortho_weight, common.load_params


def _p(pp, name):
    return '%s_%s' % (pp, name)


def validate_options(options):
    """model_attention.py:35-40"""
    if options['ctx2out']:
        warnings.warn('Feeding context to output directly seems to hurt.')
    if options['dim_word'] > options['dim']:
        warnings.warn('dim_word should only be as large as dim.')
    return options


class Sym(object):
    """Inert stand-in for a Theano symbolic variable returned by build_model."""

    def __init__(self, name, neg=False):
        self.name, self.neg = name, neg

    def __neg__(self):
        return Sym(self.name, not self.neg)

    def __repr__(self):
        return ('-' if self.neg else '') + self.name


class TParams(OrderedDict):
    """OrderedDict name -> SharedVar plus the native decoder they live in once bound."""
    decoder = None


class Attention(object):
    def __init__(self, channel=None):
        self.channel = channel
        self.layers = {'ff': ('param_init_fflayer', 'fflayer'),
                       'lstm_cond': ('param_init_lstm_cond', 'lstm_cond_layer')}
        self._options = None
        self.engine = None                # the reference's train() hangs its data engine here (:1067); pred_probs reads it
        self.device = 0
        self.stream = None

    # ---------------------------------------------------------------- parameters
    def load_params(self, path, params):
        return common.load_params(path, params)

    def param_init_fflayer(self, options, params, prefix='ff', nin=None, nout=None):
        """model_attention.py:80-87"""
        params[_p(prefix, 'W')] = norm_weight(nin, nout, scale=0.01)
        params[_p(prefix, 'b')] = numpy.zeros((nout,)).astype('float32')
        return params

    def param_init_lstm_cond(self, options, params, prefix='lstm_cond', nin=None, dim=None, dimctxglm=None,
                             dimctxg=None, dimctxl=None, dimctxm=None):
        """model_attention.py:180-282: same arrays, same order, same draws from common.rng_numpy."""
        nin = options['dim'] if nin is None else nin
        dim = options['dim'] if dim is None else dim
        dimctxglm = options['dim'] if dimctxglm is None else dimctxglm
        params[_p(prefix, 'W')] = numpy.concatenate([norm_weight(nin, dim) for _ in range(4)], axis=1)   # :189-193
        params[_p(prefix, 'U')] = numpy.concatenate([ortho_weight(dim) for _ in range(4)], axis=1)       # :196-200
        params[_p(prefix, 'b')] = numpy.zeros((4 * dim,)).astype('float32')                              # :203
        params[_p(prefix, 'Wc')] = norm_weight(dimctxglm, dim * 4)                                       # :206
        params[_p(prefix, 'Wcg_att')] = norm_weight(dimctxg, ortho=False)                                # :210
        params[_p(prefix, 'Wcm_att')] = norm_weight(dimctxm, ortho=False)                                # :214
        params[_p(prefix, 'Wclt_att')] = norm_weight(dimctxl, ortho=False)                               # :218
        params[_p(prefix, 'Wdg_att')] = norm_weight(dim, dimctxg)                                        # :222
        params[_p(prefix, 'Wdm_att')] = norm_weight(dim, dimctxm)                                        # :225
        params[_p(prefix, 'Wdlt_att')] = norm_weight(dim, dimctxl)                                       # :228
        params[_p(prefix, 'bg_att')] = numpy.zeros((dimctxg,)).astype('float32')                         # :232
        params[_p(prefix, 'bm_att')] = numpy.zeros((dimctxm,)).astype('float32')                         # :235
        params[_p(prefix, 'blt_att')] = numpy.zeros((dimctxl,)).astype('float32')                        # :239
        params[_p(prefix, 'Wcl_att')] = norm_weight(dimctxl, ortho=False)                                # :243
        params[_p(prefix, 'Wdl_att')] = norm_weight(dim, dimctxl)                                        # :247
        params[_p(prefix, 'bl_att')] = numpy.zeros((dimctxl,)).astype('float32')                         # :251
        for nm, dd in (('g', dimctxg), ('m', dimctxm), ('lt', dimctxl), ('l', dimctxl)):                 # :255-274
            params[_p(prefix, 'U%s_att' % nm)] = norm_weight(dd, 1)
            params[_p(prefix, 'c%s_att' % nm)] = numpy.zeros((1,)).astype('float32')
        if options['selector']:                                                                          # :276-281
            params[_p(prefix, 'W_sel')] = norm_weight(dim, 1)
            params[_p(prefix, 'b_sel')] = numpy.float32(0.)
        return params

    def init_params(self, options):
        """model_attention.py:518-581 with encoder == 'none' and n_layers_init == 0."""
        if options.get('encoder', 'none') not in ('none', None):
            raise ValueError("encoder must be 'none': the lstm encoder branches of the reference are broken")
        if options.get('n_layers_init', 0) != 0:
            raise ValueError('n_layers_init must be 0 (model_attention.py:546-548 uses undefined names)')
        self._options = dict(options)
        params = OrderedDict()
        params['Wemb'] = norm_weight(options['n_words'], options['dim_word'])                            # :522
        ctxg_dim, ctxl_dim, ctxm_dim = options['ctxg_dim'], options['ctxl_dim'], options['ctxm_dim']
        params = self.param_init_fflayer(options, params, prefix='ff_state', nin=ctxg_dim, nout=options['dim'])
        params = self.param_init_fflayer(options, params, prefix='ff_memory', nin=ctxg_dim, nout=options['dim'])
        params = self.param_init_fflayer(options, params, prefix='ff_local', nin=ctxl_dim, nout=options['dim'])
        params = self.param_init_fflayer(options, params, prefix='ff_motion', nin=ctxm_dim, nout=options['dim'])
        params = self.param_init_lstm_cond(options, params, prefix='decoder', nin=options['dim_word'],
                                           dim=options['dim'], dimctxg=options['dim'], dimctxl=options['dim'],
                                           dimctxm=options['dim'], dimctxglm=options['dim'])             # :561-563
        params = self.param_init_fflayer(options, params, prefix='ff_logit_lstm', nin=options['dim'],
                                         nout=options['dim_word'])
        if options['ctx2out']:
            params = self.param_init_fflayer(options, params, prefix='ff_logit_ctxglm',
                                             nin=options['ctxglm_dim'], nout=options['dim_word'])
        if options.get('n_layers_out', 1) > 1:
            raise ValueError('only n_layers_out == 1 is supported')
        params = self.param_init_fflayer(options, params, prefix='ff_logit', nin=options['dim_word'],
                                         nout=options['n_words'])
        return params

    def init_tparams(self, params, force_cpu=False):
        """model_attention.py:70-78.  `force_cpu` is accepted for signature compatibility; there is
        no CPU execution path -- the parameters always live in HBM."""
        tparams = TParams()
        for kk, pp in params.items():
            tparams[kk] = common.SharedVar(pp, kk)
        return tparams

    def _bind(self, tparams, options):
        """Create the native decoder for these parameters on first use (options arrive only with
        build_model / build_sampler, exactly like the reference's graph construction)."""
        if getattr(tparams, 'decoder', None) is not None:
            return tparams.decoder
        dec = Decoder(options, device=self.device, stream=self.stream)
        dec.set_params(OrderedDict((k, v.get_value()) for k, v in tparams.items()))
        for v in tparams.values():
            v.bind(dec)
        tparams.decoder = dec
        return dec

    # ---------------------------------------------------------------- graphs
    def build_model(self, tparams, options):
        """model_attention.py:583-717.  Returns the reference's 16-tuple; the tensors are name tags
        (see `function`).  `use_noise.set_value(1.)` switches the dropout draws on (:1248)."""
        dec = self._bind(tparams, options)
        trng = common.rng_theano
        use_noise = common.SharedScalar(0., on_change=dec.set_use_noise)
        names = ('x', 'mask', 'ctxg', 'mask_ctxg', 'ctxl', 'mask_ctxl', 'ctxm', 'mask_ctxm')
        x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm = [Sym(n) for n in names]
        alphals, alphags, alphams, alphalts = Sym('alphal'), Sym('alphag'), Sym('alpham'), Sym('alphalt')
        cost = Sym('cost')
        extra = [Sym('probs'), alphals, alphags, alphams, alphalts]
        self._train_decoder = dec
        return (trng, use_noise, x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm,
                alphals, alphags, alphams, alphalts, cost, extra)

    def function(self, inputs, outputs, tparams=None, **kwargs):
        """Stand-in for theano.function over build_model's variables, e.g.
        f_log_probs = function([x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm], -cost) (:1126)
        f_alphal    = function([...8 inputs...], [alphals, 0.0])                                     (:1167)"""
        dec = tparams.decoder if tparams is not None else self._train_decoder
        single = not isinstance(outputs, (list, tuple))
        outs = [outputs] if single else list(outputs)
        want_probs = any(isinstance(o, Sym) and o.name == 'probs' for o in outs)
        want_alpha = any(isinstance(o, Sym) and o.name.startswith('alpha') for o in outs)

        def fn(x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm):
            dec.set_batch(x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm)
            dec.forward_train()
            r = dec.get_forward(probs=want_probs, alphas=want_alpha)
            vals = []
            for o in outs:
                if isinstance(o, Sym):
                    v = r[o.name]
                    vals.append(-v if o.neg else v)
                else:
                    vals.append(o)
            return vals[0] if single else vals
        return fn

    def build_sampler(self, tparams, options, use_noise, trng, mode=None):
        """model_attention.py:719-850 -> (f_init, f_next), same signatures and return orders."""
        dec = self._bind(tparams, options)

        def f_init(ctxg_0, ctxg_mask):                                      # :791-795
            return dec.f_init(ctxg_0, ctxg_mask)

        def f_next(x, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask, init_state, init_memory):
            return dec.f_next(x, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask,
                              init_state, init_memory)                      # :845-848
        f_next.decoder = dec
        # gen_sample runs its beam / greedy loop on the device when handed this very f_next (k <= 8, not stochastic);
        # set f_next.device_loop = False to drive f_next from the host word by word like the reference does
        f_next.device_loop = True
        return f_init, f_next

    # ---------------------------------------------------------------- gradient / update functions
    def build_train_functions(self, tparams, options, decay_c=0., alpha_c=0., clip_c=0., global_batch=None,
                              group=None, return_grads=False):
        """f_grad_shared, f_update -- what `eval(optimizer)(lr, tparams, grads, inps, cost, extra + grads)` returns
        for optimizer='adadelta' (model_attention.py:1207-1209, common.py:178-195), with the loss assembled as in
        :1129-1147 (mean NLL + L2 decay + doubly-stochastic attention regulariser) and the global-norm clip of
        :1194-1203.

        f_grad_shared(x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm) -> [cost, probs, alphals, alphags,
        alphams, alphalts] (+ the gradient list in parameter order when return_grads=True: 171 MB of device->host
        copies at the MSVD config, which the reference pays on every update, :1260-1266).  The returned gradients
        are those of mean-NLL + regulariser; the L2 term and the clip are applied inside f_update (same update).
        `global_batch` / `group`: data-parallel use -- every rank passes its row shard, gradients are summed with one
        RCCL all-reduce (stattn.dp)."""
        try:
            from . import dp
        except ImportError:
            import dp
        dec = self._bind(tparams, options)
        reducer = dp.GradReducer(dec, group) if global_batch else None    # rank-distinct dropout seed inside

        def f_grad_shared(x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm):
            dec.set_batch(x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm)
            dec.forward_train()
            gb = global_batch if global_batch else x.shape[1]
            dec.backward(nll_scale=1.0 / gb, alpha_c=alpha_c)
            r = dec.get_forward(probs=True, alphas=True)
            loss = dec.get_loss(decay_c)
            if reducer is not None:
                reducer.allreduce()                       # ordered with the library's stream on both sides
                loss = reducer.global_loss(loss, dec, decay_c)
            out = [numpy.float32(loss), r['probs'], r['alphal'], r['alphag'], r['alpham'], r['alphalt']]
            if return_grads:
                out += list(dec.get_grads().values())
            return out

        def f_update(lr):
            # common.py:193: the reference's adadelta ignores lr too
            dec.update(decay_c=decay_c, clip_c=clip_c)
        return f_grad_shared, f_update

    def train(self, *args, **kwargs):
        """Two call forms.
        (1) The reference's: `model.train(**state.attention)` (model_attention.py:1034-1078, called from
            train_from_scratch, :1558-1562) -- the keyword names and defaults of the reference, plus `engine=` to inject
            the data engine (the h5 / pkl loader `data_engine.Movie2Caption` is out of scope: DESIGN.md section 13).
            Returns (train_err, valid_err, test_err) like :1556.  See `train_reference`.
        (2) `model.train(batches, options, ...)`: the loop over ready-made prepare_data() 8-tuples (`fit`)."""
        if args or 'batches' in kwargs or 'options' in kwargs:
            return self.fit(*args, **kwargs)
        return self.train_reference(**kwargs)

    def train_reference(self,
                        random_seed=1234, dim_word=256, ctxglm_dim=-1, ctxg_dim=-1, ctxl_dim=-1, ctxm_dim=-1, dim=1000,
                        n_layers_out=1, n_layers_init=1, encoder='none', encoder_dim=100, prev2out=False, ctx2out=False,
                        patience=10, max_epochs=5000, dispFreq=100, decay_c=0., alpha_c=0., alpha_entropy_r=0., lrate=0.01,
                        selector=False, n_words=100000, maxlen=100, optimizer='adadelta', clip_c=2., batch_size=64,
                        valid_batch_size=64, save_model_dir='./', validFreq=10, saveFreq=10, sampleFreq=10, metric='blue',
                        dataset='youtube2text', video_feature='googlenet', use_dropout=False, reload_=False, from_dir=None,
                        K=10, OutOf=240, verbose=True, debug=True, engine=None):
        """Attention.train with the reference's keyword surface (model_attention.py:1034-1557; names, defaults and
        meaning as there, so config.py's `attention` block passes through unchanged).  What it keeps of the loop:
        `model_options` = the keyword arguments (pickled to `<save_model_dir>model_options.pkl`, :1083-1084) with the
        four feature dims taken from the engine (:1093-1096); reload from `<from_dir>/model_best_so_far.npz` (:1109-1113);
        epochs over `engine.kf_train` with prepare_data per minibatch (:1239-1254), use_noise = 1, f_grad_shared +
        f_update (:1259-1278), NaN check; every `validFreq` updates: model_current.npz, pred_probs on train / valid / test
        with use_noise = 0 (:1392-1426), a `history_errs` row in the reference's 22-column layout (:1462-1469; the
        caption-metric columns are 0 -- coco-caption scoring is out of scope), train_valid_test.txt, the reference's
        best-model / early-stopping rule on column 6 (:1477-1499: nothing is saved at the first validation, exactly like
        there); at the end the best parameters are put back (:1519-1520), valid / test errors recomputed and
        model_best.npz written (:1543-1546).  `debug=True` (the reference's default!) stops after ONE update and
        skips the error passes (:1504-1509, 1525).  `optimizer` must be 'adadelta' (the one the path implements;
        `lrate` is ignored by it in the reference too, common.py:193).  With `verbose`, every `dispFreq` updates the four
        attention min / max ratios and regulariser values are printed (:1291-1306) and every `sampleFreq` updates beam-5 captions of
        up to ten training and validation videos next to their ground truth (:1312-1368); every validation appends the ratios to
        `alpha{l,g,m,lt}_ratio.txt` (:1372-1390).  The caption metrics (coco-caption) are not reproduced.  `self.channel.save()` is called where the reference calls it."""
        import os
        import pickle as pkl
        try:
            from . import data_engine
        except ImportError:
            import data_engine
        if optimizer != 'adadelta':
            raise NotImplementedError("optimizer %r: the decoder path implements common.adadelta (config.py:34)" % (optimizer,))
        if engine is None:
            raise NotImplementedError("pass engine= (an object with the attributes of data_engine.Movie2Caption, e.g. "
                                      "stattn.data_engine.MemoryEngine): the h5 / pkl feature loader is out of scope")
        common.reset_rngs(random_seed)
        model_options = dict(locals())
        for k_ in ('self', 'engine', 'os', 'pkl', 'data_engine'):
            model_options.pop(k_, None)
        model_options = validate_options(model_options)
        if save_model_dir and not os.path.isdir(save_model_dir):
            os.makedirs(save_model_dir)
        with open('%smodel_options.pkl' % save_model_dir, 'wb') as f:
            pkl.dump(model_options, f)
        self.engine = engine
        for k_ in ('ctxglm_dim', 'ctxg_dim', 'ctxl_dim', 'ctxm_dim'):
            model_options[k_] = getattr(engine, k_)
        params = self.init_params(model_options)
        if reload_:
            saved = os.path.join(from_dir, 'model_best_so_far.npz')
            assert os.path.isfile(saved)
            params = common.load_params(saved, params)
        tparams = self.init_tparams(params)
        rv = self.build_model(tparams, model_options)
        use_noise, inps, cost = rv[1], list(rv[2:10]), rv[14]
        self.f_init, self.f_next = self.build_sampler(tparams, model_options, use_noise, rv[0])
        f_log_probs = self.function(inps, -cost, tparams=tparams)
        # f_alphal / f_alphag / f_alpham / f_alphalt (:1167-1188): [alphas, alpha_c * mean_{T(,K)} sum_b (1 - sum_t alpha)^2]
        f_alphas = self.function(inps, [rv[10], rv[11], rv[12], rv[13]], tparams=tparams)

        def alpha_ratios(batch):
            """min / max attention-weight ratio and regulariser value per attention, as logged at :1291-1306 and :1372-1390"""
            out = []
            for al in f_alphas(*batch):
                reg = alpha_c * float(((1. - al.sum(0)) ** 2).sum(0).mean()) if alpha_c > 0. else 0.
                out.append((float(al.min(-1).mean() / al.max(-1).mean()), reg))
            return out

        def words(seq):                                              # ids up to the first <eos> (0), unknown ids as UNK
            out = []
            for w in seq:
                if int(w) == 0:
                    break
                out.append(engine.word_idict.get(int(w), 'UNK'))
            return ' '.join(out)

        def sample_execute(from_which, batch):                      # :1312-1368
            print('------------- sampling from %s ----------' % from_which)
            if from_which == 'valid' and len(engine.kf_valid) > 2:
                idx = engine.kf_valid[numpy.random.randint(1, len(engine.kf_valid) - 1)]
                batch = data_engine.prepare_data(engine, [engine.valid[i] for i in idx])
            x_s, _, g_s, gm_s, l_s, lm_s, m_s, mm_s = batch
            for jj in range(min(10, x_s.shape[1])):
                sample, score, _, _ = self.gen_sample(tparams, self.f_init, self.f_next, g_s[jj], gm_s[jj], l_s[jj], lm_s[jj], m_s[jj],
                                                      mm_s[jj], model_options, trng=rv[0], k=5, maxlen=30, stochastic=False)
                best = int(numpy.argmin(score))
                print('cost', score[best])
                print('Truth ', jj, ': ', words(x_s[:, jj]))
                print('Sample ( 0 ) ', jj, ': ', words(sample[best]))

        f_grad_shared, f_update = self.build_train_functions(tparams, model_options, decay_c, alpha_c, clip_c)
        history_errs = []
        if reload_:
            history_errs = numpy.load(saved)['history_errs'].tolist()
        best_p, bad_counter, uidx, estop = None, 0, 0, False
        train_err = valid_err = test_err = -1
        ratio_log = [[], [], [], []]
        for eidx in range(max_epochs):
            train_costs = []
            for idx in engine.kf_train:
                tags = [engine.train[i] for i in idx]
                uidx += 1
                use_noise.set_value(1.)
                batch = data_engine.prepare_data(engine, tags)
                if batch[0] is None:
                    continue
                c = f_grad_shared(*batch)[0]
                if numpy.isnan(c) or numpy.isinf(c):
                    raise FloatingPointError('NaN detected in cost')       # the reference drops into pdb (:1274-1276)
                f_update(lrate)
                train_costs.append(c)
                if dispFreq and numpy.mod(uidx, dispFreq) == 0 and verbose:
                    print('Epoch ', eidx, 'Update ', uidx, 'Train cost', c)
                    for name, (ratio, reg) in zip(('alphal', 'alphag', 'alpham', 'alphalt'), alpha_ratios(batch)):
                        print('%s ratio %.3f, reg %.3f' % (name, ratio, reg))
                if sampleFreq and sampleFreq > 0 and numpy.mod(uidx, sampleFreq) == 0 and verbose:
                    use_noise.set_value(0.)
                    sample_execute('train', batch)
                    sample_execute('valid', batch)
                if validFreq != -1 and numpy.mod(uidx, validFreq) == 0:
                    use_noise.set_value(0.)
                    for lst, name, (ratio, _) in zip(ratio_log, ('alphal', 'alphag', 'alpham', 'alphalt'), alpha_ratios(batch)):
                        lst.append(ratio)                     # (:1372-1390; the reference appends alphag's ratio to alpham's log)
                        numpy.savetxt(save_model_dir + '%s_ratio.txt' % name, lst)
                    numpy.savez(save_model_dir + 'model_current.npz', history_errs=history_errs, **common.unzip(tparams))
                    use_noise.set_value(0.)
                    train_err = train_perp = valid_err = valid_perp = test_err = test_perp = -1
                    if not debug:
                        train_err, train_perp = self.pred_probs('train', f_log_probs, verbose=verbose)
                        valid_err, valid_perp = self.pred_probs('valid', f_log_probs, verbose=verbose)
                        test_err, test_perp = self.pred_probs('test', f_log_probs, verbose=verbose)
                    history_errs.append([eidx, uidx, train_err, train_perp, valid_perp, test_perp, valid_err, test_err] + [0.] * 14)
                    numpy.savetxt(save_model_dir + 'train_valid_test.txt', history_errs, fmt='%.3f')
                    if len(history_errs) > 1 and valid_err < numpy.array(history_errs)[:-1, 6].min():
                        best_p = common.unzip(tparams)
                        bad_counter = 0
                        numpy.savez(save_model_dir + 'model_best_so_far.npz', history_errs=history_errs, **best_p)
                        with open('%smodel_options.pkl' % save_model_dir, 'wb') as f:
                            pkl.dump(model_options, f)
                    elif len(history_errs) > 1 and valid_err >= numpy.array(history_errs)[:-1, 6].min():
                        bad_counter += 1
                        if bad_counter > patience:
                            estop = True
                            break
                    if self.channel:
                        self.channel.save()
                if debug:
                    break
            if estop or debug:
                break
        if best_p is not None:
            common.zipp(best_p, tparams)
        use_noise.set_value(0.)
        valid_err = test_err = 0
        if not debug:
            valid_err, _ = self.pred_probs('valid', f_log_probs, verbose=verbose)
            test_err, _ = self.pred_probs('test', f_log_probs, verbose=verbose)     # (commented out in the reference, :1530-1532)
        final = best_p if best_p is not None else common.unzip(tparams)
        numpy.savez(save_model_dir + 'model_best.npz', train_err=train_err, valid_err=valid_err, test_err=test_err,
                    history_errs=history_errs, **final)
        if history_errs != []:
            numpy.savetxt(save_model_dir + 'train_valid_test.txt', numpy.asarray(history_errs), fmt='%.4f')
        self.tparams = tparams
        return train_err, valid_err, test_err

    def fit(self, batches, options, valid_batches=None, max_epochs=1, decay_c=0., alpha_c=0., clip_c=0.,
            patience=10, validFreq=-1, dispFreq=0, save_model_dir=None, reload_=False, from_dir=None, params=None):
        """Minimal counterpart of the optimisation loop of Attention.train (model_attention.py:1239-1517): epochs over
        `batches` (a re-iterable of prepare_data() 8-tuples, or a callable returning a fresh iterator per epoch), f_grad_shared + f_update per minibatch with use_noise = 1,
        validation NLL with use_noise = 0 (pred_probs), early stopping on it with `patience` (:1494-1503), and the
        reference's checkpoint files: model_best_so_far.npz = numpy.savez(path, history_errs=..., **params)
        (:1488-1490), reloaded with load_params when reload_ (:1109-1113).  Returns (tparams, history_errs)."""
        import os
        common.reset_rngs(1234)
        if params is None:
            params = self.init_params(options)
        history_errs = []
        if reload_:
            saved = os.path.join(from_dir, 'model_best_so_far.npz')
            params = common.load_params(saved, params)
            history_errs = numpy.load(saved)['history_errs'].tolist()
        tparams = self.init_tparams(params)
        rv = self.build_model(tparams, options)
        use_noise, inps, cost = rv[1], list(rv[2:10]), rv[14]
        f_log_probs = self.function(inps, -cost, tparams=tparams)
        f_grad_shared, f_update = self.build_train_functions(tparams, options, decay_c, alpha_c, clip_c)
        best_p, bad_counter, uidx, estop = None, 0, 0, False
        # `batches` / `valid_batches`: a re-iterable (list, dataset object) or a callable returning a fresh iterator -- the
        # way to stream minibatches like the reference does (one prepare_data() 8-tuple alive at a time, :1250-1251).  A
        # one-shot generator cannot be replayed and holding all of it would keep every batch's features in host memory.
        def replayable(src, what):
            if callable(src):
                return src
            if iter(src) is not src:
                return lambda: src
            used = [False]

            def once():                 # a one-shot generator serves ONE pass; only a second pass is an error
                if used[0]:
                    raise ValueError("%s is a one-shot generator but is needed %s: pass a list / re-iterable, or a callable "
                                     "that returns a fresh iterator" % (what, "once per epoch" if what == 'batches' else "at every validation"))
                used[0] = True
                return src
            return once
        epoch_batches = replayable(batches, 'batches')
        valid_iter = replayable(valid_batches, 'valid_batches') if valid_batches is not None else None
        for eidx in range(max_epochs):
            for batch in epoch_batches():
                if batch[0] is None:              # "Minibatch with zero sample under length" (:1252-1254)
                    continue
                uidx += 1
                use_noise.set_value(1.)
                c = f_grad_shared(*batch)[0]
                if numpy.isnan(c) or numpy.isinf(c):
                    raise FloatingPointError('NaN detected in cost')       # the reference drops into pdb (:1274-1276)
                f_update(0.)
                if dispFreq and uidx % dispFreq == 0:
                    print('Epoch', eidx, 'Update', uidx, 'Train cost', c)
                if valid_batches is not None and validFreq > 0 and uidx % validFreq == 0:
                    use_noise.set_value(0.)
                    valid_err = self.pred_probs(valid_iter(), f_log_probs)[0]
                    history_errs.append([eidx, uidx, float(c), float(valid_err)])
                    if best_p is None or valid_err <= numpy.array(history_errs)[:, 3].min():
                        best_p = common.unzip(tparams)
                        bad_counter = 0
                        if save_model_dir:
                            numpy.savez(os.path.join(save_model_dir, 'model_best_so_far.npz'),
                                        history_errs=history_errs, **best_p)
                    elif len(history_errs) > patience and valid_err >= numpy.array(history_errs)[:-patience, 3].min():
                        bad_counter += 1
                        if bad_counter > patience:
                            estop = True
                            break
            if estop:
                break
        use_noise.set_value(0.)
        if save_model_dir:
            numpy.savez(os.path.join(save_model_dir, 'model_current.npz'), history_errs=history_errs, **common.unzip(tparams))
        return tparams, history_errs

    # ---------------------------------------------------------------- beam search driver
    def gen_sample(self, tparams, f_init, f_next, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask, options,
                   trng=None, k=1, maxlen=30, stochastic=False, restrict_voc=False):
        """Host-side driver of the sampler with the contract of model_attention.py:852-994: beam search of width k
        (k = 1: greedy), or ancestral sampling when `stochastic`.  Returns (sample, sample_score, next_state,
        next_memory): hypotheses that ended with <eos> in the order they ended, then the ones still alive;
        sample_score = summed -log p, no length normalisation (metrics.py:130 takes the argmin); next_state /
        next_memory are one-element lists (n_layers_lstm = 1), like the reference's.

        The candidate table of one step is the (live, V) matrix `cost[i] - log p[i, w]`; its (k - finished) smallest
        entries are taken with one argsort over the flattened table and split back into (parent row, word) with
        integer division (the reference's Python-2 `/` at :926).  Survivors are gathered with fancy indexing.
        When `f_next` belongs to this package, the video is staged once for the whole loop (Decoder.video_scope)."""
        if k > 1 and stochastic:
            raise AssertionError('Beam search does not support stochastic sampling')
        if restrict_voc:
            raise NotImplementedError()
        dec = getattr(f_next, 'decoder', None)
        if dec is not None and stochastic and getattr(f_next, 'device_loop', False) and dec.precision != 'bf16':
            # ancestral sampling on the device as well (stattn_sample_search: Gumbel-max in the logits launch, no m x V
            # copy per word); `sample` is one flat word list and the score the summed probabilities, like :913-918
            try:
                (sample, score), = dec.sample_search(ctxg_0[None], ctxg_mask[None], ctxl_0[None], ctxm_0[None], maxlen=maxlen)
            except ValueError as e:                  # no row-panel path for this shape / STATTN_NO_PANELS: the host loop below
                if 'row-panel' not in str(e):
                    raise
            else:
                (hh, cc), = dec.beam_final_state()
                return sample, score, [hh], [cc]
        if dec is not None and not stochastic and k <= 8 and getattr(f_next, 'device_loop', False):
            # the whole loop on the device (stattn_beam_search: hipGraph-captured word sequence, no per-word host
            # round trip); k = 1 is the greedy decode of :896-918
            (sample, score), = dec.beam_search(ctxg_0[None], ctxg_mask[None], ctxl_0[None], ctxm_0[None], k=k, maxlen=maxlen)
            (hh, cc), = dec.beam_final_state()
            return sample, list(score), [hh], [cc]
        if dec is None:
            return self._decode_loop(f_init, f_next, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask, k, maxlen, stochastic)
        with dec.video_scope(ctxg_0, ctxl_0, ctxm_0):
            return self._decode_loop(f_init, f_next, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask, k, maxlen, stochastic)

    @staticmethod
    def _decode_loop(f_init, f_next, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask, k, maxlen, stochastic):
        ended, ended_cost = [], []                       # finished hypotheses, in order of death
        prefixes, cost = [[]], numpy.zeros(1, dtype='float32')
        drawn, drawn_score = [], 0
        _, h0, c0 = f_init(ctxg_0, ctxg_mask)
        state, memory = [h0[None, :]], [c0[None, :]]
        words = numpy.full((1,), -1, dtype='int64')      # -1: no previous word (:803-804)
        for _step in range(maxlen):
            probs, words, h, c = f_next(words, ctxg_0, ctxg_mask, ctxl_0, ctxl_mask, ctxm_0, ctxm_mask, state[0], memory[0])
            state, memory = [h], [c]
            if stochastic:                               # :913-918 (the score sums p, not log p, as the reference does)
                drawn.append(words[0])
                drawn_score += probs[0, words[0]]
                if words[0] == 0:
                    break
                continue
            with numpy.errstate(divide='ignore'):
                table = (cost[:, None] - numpy.log(probs)).ravel()
            best = table.argsort()[:k - len(ended)]
            parent, word = best // probs.shape[1], best % probs.shape[1]
            grown_cost = table[best].astype('float32')
            alive = word != 0
            for i in numpy.flatnonzero(~alive):          # <eos>: the hypothesis leaves the beam (:956-960)
                ended.append(prefixes[parent[i]] + [int(word[i])])
                ended_cost.append(grown_cost[i])
            prefixes = [prefixes[p] + [int(w)] for p, w in zip(parent[alive], word[alive])]
            cost = grown_cost[alive]
            if not prefixes or len(ended) >= k:
                break
            words = word[alive].astype('int64')
            state, memory = [h[parent[alive]]], [c[parent[alive]]]
        if stochastic:
            return drawn, drawn_score, state, memory
        return ended + prefixes, ended_cost + list(cost), state, memory

    def gen_sample_batch(self, tparams, options, ctxgs, ctxg_masks, ctxls, ctxms, k=5, maxlen=30, suppress_eos=False):
        """Batched counterpart of the evaluation loop of metrics.py:121-135 (one gen_sample per video): all videos
        and their beams advance together on the device.  Returns [(sample, sample_score), ...] per video with
        gen_sample's ordering, so `sample[numpy.argmin(score)]` picks the caption exactly as metrics.py:130 does."""
        dec = self._bind(tparams, options)
        return dec.beam_search(ctxgs, ctxg_masks, ctxls, ctxms, k=k, maxlen=maxlen, suppress_eos=suppress_eos)

    # ---------------------------------------------------------------- teacher-forced scoring
    def pred_probs(self, batches, f_log_probs, verbose=False):
        """Teacher-forced scoring of a split with the contract of model_attention.py:996-1032: f_log_probs returns -cost
        per caption; returns (mean NLL per caption, perplexity = 2 ** (sum NLL / number of words / ln 2)).
        `batches` is either the reference's `whichset` ('train' / 'valid' / 'test': the caption tags and minibatch index
        lists of `self.engine`, each minibatch assembled by prepare_data, :1003-1017) or any iterable of prepare_data()
        8-tuples."""
        if isinstance(batches, str):
            try:
                from . import data_engine
            except ImportError:
                import data_engine
            if batches not in ('train', 'valid', 'test'):
                raise NotImplementedError()
            tags, index_lists = getattr(self.engine, batches), getattr(self.engine, 'kf_' + batches)
            batches = (data_engine.prepare_data(self.engine, [tags[i] for i in index]) for index in index_lists)
        nll, nwords = [], 0.0
        for batch in batches:
            if batch[0] is None:                 # prepare_data found no usable caption (data_engine.py:318)
                continue
            nll.append(-numpy.asarray(f_log_probs(*batch), dtype='float64'))
            nwords += float(numpy.asarray(batch[1]).sum())
        nll = numpy.concatenate(nll) if nll else numpy.zeros(0)
        return nll.mean(), 2 ** (nll.sum() / nwords / numpy.log(2))


def train_from_scratch(state, channel):
    """model_attention.py:1558-1562, the function train_model.py:82 calls: `state.attention` (config.py's block, any
    mapping or attribute bag) goes to Attention.train as keyword arguments.  One key beyond the reference's:
    `state.attention['engine']` (or `state.engine`) injects the data engine, because the h5 / pkl loader is out of scope."""
    import time
    t0 = time.time()
    print('training an attention model')
    model = Attention(channel)
    kw = dict(state['attention'] if isinstance(state, dict) else state.attention)
    if 'engine' not in kw:
        eng = state.get('engine') if isinstance(state, dict) else getattr(state, 'engine', None)
        if eng is not None:
            kw['engine'] = eng
    rv = model.train(**kw)
    print('training time in total %.4f sec' % (time.time() - t0))
    return rv
