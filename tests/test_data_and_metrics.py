"""Host-side callers either side of the decoder path: prepare_data (data_engine.py:258-337), the feature-mask rule
(:169-218), _seqs2words and the sample-file writer (metrics.py:103-146).  The CPU tests pin the layout rules; the
GPU test runs the writer end to end and checks its captions against the oracle's beam search."""
import os

import numpy as np
import pytest

import stattn
from stattn import data_engine, metrics


def _engine(maxlen=None, n_words=8, signature='youtube2text'):
    rng = np.random.RandomState(0)
    T, K = 4, 3
    feats = {}
    for v, vid in enumerate(['vid1', 'vid2', 'vid3'] if signature == 'youtube2text' else ['a_b_1', 'a_b_2', 'c_3']):
        g = rng.standard_normal((T, 6)).astype(np.float32)
        l = rng.standard_normal((T, K, 5)).astype(np.float32)
        m = rng.standard_normal((T, 7)).astype(np.float32)
        if v == 1:                      # a short video: the last frame is zero padding (pad_frames, :83-91)
            g[-1] = 0; l[-1] = 0; m[-1] = 0
        feats[vid] = (g, l, m)
    worddict = {'a': 2, 'man': 3, 'is': 4, 'cooking': 5, 'dog': 6, 'running': 7, 'quickly': 8, 'outside': 9}
    vids = list(feats)
    caps = {vids[0]: [{'cap_id': '0', 'tokenized': 'a man is cooking'}, {'cap_id': '1', 'tokenized': 'a dog is running quickly outside'}],
            vids[1]: [{'cap_id': '0', 'tokenized': 'a dog'}],
            vids[2]: [{'cap_id': '5', 'tokenized': 'man is running'}]}
    return data_engine.MemoryEngine(feats, caps, worddict, n_words=n_words, maxlen=maxlen, signature=signature,
                                    valid_ids=vids[:2], test_ids=vids[2:])


def test_ctx_mask_rule():
    x = np.zeros((2, 3, 4), np.float32)
    x[0, 0, 1] = 2.0
    x[0, 1, 3] = 1.0           # outside the first `dim` = 3 features -> still padding
    x[1, 2, 0] = 1.0; x[1, 2, 1] = -1.0    # features cancel: the reference's sum != 0 test calls this padding too
    m = data_engine.ctx_mask(x, 3)
    assert m.dtype == np.float32
    np.testing.assert_array_equal(m, [[1, 0, 0], [0, 0, 0]])
    assert data_engine.ctx_mask(x[0], 4).tolist() == [1.0, 1.0, 0.0]
    assert data_engine.ctx_mask(np.ones((2, 3, 5, 4)), 4).shape == (2, 3, 5)
    with pytest.raises(NotImplementedError):
        data_engine.ctx_mask(np.ones(3), 1)


def test_prepare_data_layout_unk_and_length_filter():
    eng = _engine()
    out = data_engine.prepare_data(eng, ['vid1_0', 'vid2_0', 'vid1_1'])
    x, x_mask, yg, yg_mask, yl, yl_mask, ym, ym_mask = out
    assert x.dtype == np.int64 and x_mask.dtype == np.float32
    assert x.shape == (7, 3)                                   # longest caption (6 words) + the <eos> row
    # ids >= n_words (8) become UNK = 1
    assert x[:, 0].tolist() == [2, 3, 4, 5, 0, 0, 0]
    assert x[:, 1].tolist() == [2, 6, 0, 0, 0, 0, 0]
    assert x[:, 2].tolist() == [2, 6, 4, 7, 1, 1, 0]
    assert x_mask.sum(0).tolist() == [5.0, 3.0, 7.0]           # len + 1 ones per column
    assert x_mask[:5, 0].all() and not x_mask[5:, 0].any()
    assert yg.shape == (3, 4, 6) and yl.shape == (3, 4, 3, 5) and ym.shape == (3, 4, 7)
    np.testing.assert_array_equal(yg[0], eng.get_video_global_features('vid1'))
    np.testing.assert_array_equal(yg[2], yg[0])                # both captions of vid1 carry vid1's features
    assert yg_mask.tolist() == [[1, 1, 1, 1], [1, 1, 1, 0], [1, 1, 1, 1]]
    assert yl_mask.shape == (3, 4, 3) and yl_mask[1, -1].sum() == 0 and yl_mask[1, :-1].all()
    assert ym_mask[1].tolist() == [1, 1, 1, 0]
    # captions with len >= maxlen are thrown away; nothing left -> five Nones
    eng5 = _engine(maxlen=5)
    x5 = data_engine.prepare_data(eng5, ['vid1_0', 'vid2_0', 'vid1_1'])[0]
    assert x5.shape == (5, 2)
    assert data_engine.prepare_data(_engine(maxlen=2), ['vid1_0', 'vid1_1']) == (None,) * 5
    with pytest.raises(AssertionError):
        data_engine.prepare_data(eng, ['vid2_9'])


def test_lsmdc_ids_keep_their_underscores():
    eng = _engine(signature='lsmdc')
    assert data_engine.split_id('lsmdc', 'a_b_1_0') == ('a_b_1', '0')
    x = data_engine.prepare_data(eng, ['a_b_1_0', 'c_3_5'])[0]
    assert x.shape == (5, 2) and x[:3, 1].tolist() == [3, 4, 7]
    with pytest.raises(NotImplementedError):
        data_engine.split_id('other', 'x_1')


def test_seqs2words():
    idict = {0: '<eos>', 1: 'UNK', 2: 'a', 3: 'man', 4: 'runs'}
    assert metrics.seqs2words([[2, 3, 4, 0, 3], [2, 99, 0], [0, 2], [3, 4]], idict) == ['a man runs', 'a UNK', '', 'man runs']


def test_prepare_data_for_blue_lists():
    eng = _engine()
    g, gm, l, lm, m, mm = eng.prepare_data_for_blue('valid')
    assert len(g) == 2 and g[1].shape == (4, 6) and gm[1].tolist() == [1, 1, 1, 0] and lm[0].shape == (4, 3) and mm[1][-1] == 0
    assert len(eng.prepare_data_for_blue('test')[0]) == 1


@pytest.mark.gpu
def test_sample_files_match_oracle_beam_search(tmp_path):
    from oracle import stattn_oracle as O
    opt = O.default_options(dim=128, dim_word=64, n_words=211, ctxg_dim=128, ctxl_dim=96, ctxm_dim=64, ctxglm_dim=128)
    P = O.random_params(opt, seed=15, dtype=np.float32)
    P['ff_logit_b'] = P['ff_logit_b'].copy(); P['ff_logit_b'][0] += 2.0
    P64 = O.cast_params(P, np.float64)
    nvid, T, K = 5, 5, 4
    b = O.synthetic_batch(opt, B=nvid, T=T, K=K, t=3, seed=71)
    vids = ['v%d' % i for i in range(nvid)]
    feats = dict((vid, (b['ctxg'][i], b['ctxl'][i], b['ctxm'][i])) for i, vid in enumerate(vids))
    worddict = dict(('w%d' % i, i) for i in range(2, opt['n_words']))
    eng = data_engine.MemoryEngine(feats, {}, worddict, n_words=opt['n_words'], valid_ids=vids[:3], test_ids=vids[3:])
    model = stattn.Attention()
    tparams = model.init_tparams(P)
    f_init, f_next = model.build_sampler(tparams, opt, None, None)
    sv, st = metrics.generate_sample_gpu_single_process('attention', None, opt, eng, model, f_init, f_next,
                                                        save_dir=str(tmp_path / 'host'), beam=3, whichset='both')
    sv2, st2 = metrics.generate_sample_gpu_single_process('attention', None, opt, eng, model, f_init, f_next,
                                                          save_dir=str(tmp_path / 'dev'), beam=3, whichset='both',
                                                          batched=True, tparams=tparams)
    assert (sv, st) == (sv2, st2) and list(sv) == vids[:3] and list(st) == vids[3:]       # build_sample_pairs, metrics.py:79-83
    assert all(v == [{'image_id': vid, 'caption': v[0]['caption']}] for d in (sv, st) for vid, v in d.items())
    sv, st = [[v[0]['caption'] for v in d.values()] for d in (sv, st)]
    for d in ('host', 'dev'):
        assert open(os.path.join(str(tmp_path), d, 'valid_samples.txt')).read() == '\n'.join(sv) + '\n'
        assert open(os.path.join(str(tmp_path), d, 'test_samples.txt')).read().splitlines() == st
    # the same captions from the oracle's gen_sample (k = 3, maxlen = metrics.MAXLEN) + the same id -> word rule
    for i, vid in enumerate(vids):
        a64 = tuple(np.asarray(a, np.float64) for a in (b['ctxg'][i], b['mask_ctxg'][i], b['ctxl'][i], b['mask_ctxl'][i],
                                                        b['ctxm'][i], b['mask_ctxm'][i]))
        sr, scr, _, _ = O.gen_sample(lambda g_, m_: O.f_init(P64, opt, g_, m_), lambda *a: O.f_next(P64, opt, *a), *a64,
                                     k=3, maxlen=metrics.MAXLEN)
        want = metrics.seqs2words([sr[int(np.argmin(scr))]], eng.word_idict)[0]
        assert (sv + st)[i] == want, (vid, (sv + st)[i], want)
    only_test = metrics.generate_sample_gpu_single_process('attention', None, opt, eng, model, f_init, f_next,
                                                           save_dir=str(tmp_path / 't'), beam=1, whichset='test')
    assert only_test[0] is None and len(only_test[1]) == 2 and not os.path.exists(str(tmp_path / 't' / 'valid_samples.txt'))


# ------------------------------------------------------------------ the reference's training entry point (row b / f1)
REF_TRAIN_KEYWORDS = ['random_seed', 'dim_word', 'ctxglm_dim', 'ctxg_dim', 'ctxl_dim', 'ctxm_dim', 'dim', 'n_layers_out', 'n_layers_init',
                      'encoder', 'encoder_dim', 'prev2out', 'ctx2out', 'patience', 'max_epochs', 'dispFreq', 'decay_c', 'alpha_c',
                      'alpha_entropy_r', 'lrate', 'selector', 'n_words', 'maxlen', 'optimizer', 'clip_c', 'batch_size',
                      'valid_batch_size', 'save_model_dir', 'validFreq', 'saveFreq', 'sampleFreq', 'metric', 'dataset',
                      'video_feature', 'use_dropout', 'reload_', 'from_dir', 'K', 'OutOf', 'verbose', 'debug']     # model_attention.py:1034-1076


def _train_engine(n_vid=6, T=4, K=3, D=128, Fl=96, Fm=64):
    rng = np.random.RandomState(3)
    words = ['w%d' % i for i in range(2, 26)]
    worddict = dict((w, i + 2) for i, w in enumerate(words))
    feats, caps, tags = {}, {}, []
    for v in range(n_vid):
        vid = 'vid%d' % (v + 1)
        feats[vid] = (rng.standard_normal((T, D)).astype(np.float32), rng.standard_normal((T, K, Fl)).astype(np.float32),
                      rng.standard_normal((T, Fm)).astype(np.float32))
        caps[vid] = [{'cap_id': str(c), 'tokenized': ' '.join(rng.choice(words, size=rng.randint(2, 6)))} for c in range(2)]
        tags += ['%s_%d' % (vid, c) for c in range(2)]
    return data_engine.MemoryEngine(feats, caps, worddict, n_words=30, maxlen=30, train=tags[:8], valid=tags[8:10], test=tags[10:],
                                    mb_size_train=4, mb_size_test=2, train_ids=list(feats)[:4], valid_ids=list(feats)[4:5],
                                    test_ids=list(feats)[5:])


def test_train_keyword_surface_is_the_reference_s():
    import inspect
    from stattn import model_attention
    sig = inspect.signature(model_attention.Attention.train_reference)
    names = [n for n in sig.parameters if n != 'self']
    assert names == REF_TRAIN_KEYWORDS + ['engine']
    ref_defaults = dict(random_seed=1234, dim_word=256, dim=1000, n_layers_init=1, patience=10, max_epochs=5000, clip_c=2., batch_size=64,
                        validFreq=10, optimizer='adadelta', use_dropout=False, debug=True, K=10, OutOf=240)
    for k, v in ref_defaults.items():
        assert sig.parameters[k].default == v, k
    eng = _train_engine()
    assert eng.kf_train == [[0, 1, 2, 3], [4, 5, 6, 7]] and eng.kf_valid == [[0, 1]] and eng.ctxglm_dim == 128
    with pytest.raises(NotImplementedError):              # no engine: the h5 / pkl loader is out of scope, said loudly
        model_attention.train_from_scratch({'attention': dict(dim=128)}, None)
    with pytest.raises(TypeError):                        # unknown keywords are rejected like any Python call
        model_attention.Attention().train(no_such_option=1)


@pytest.mark.gpu
def test_train_from_scratch_with_the_reference_config_block(tmp_path, capsys):
    """train_model.py:82 -> model_attention.train_from_scratch(state, channel) -> Attention.train(**state.attention)
    (model_attention.py:1558-1562) with config.py's keys (dims shrunk) and an injected MemoryEngine."""
    from stattn import model_attention

    class Channel(object):
        saves = 0

        def save(self):
            Channel.saves += 1
    save_dir = str(tmp_path) + os.sep
    attention = dict(reload_=False, save_model_dir=save_dir, from_dir=None, dataset='youtube2text', video_feature='googlenet',
                     dim_word=64, ctxglm_dim=-1, ctxg_dim=-1, ctxl_dim=-1, ctxm_dim=-1, dim=128, n_layers_out=1, n_layers_init=0,
                     encoder_dim=300, prev2out=True, ctx2out=True, patience=20, max_epochs=4, decay_c=1e-4, alpha_entropy_r=0.,
                     alpha_c=0.70602, lrate=0.0002, selector=True, n_words=30, maxlen=30, optimizer='adadelta', clip_c=10.,
                     batch_size=4, valid_batch_size=2, dispFreq=4, validFreq=2, saveFreq=-1, sampleFreq=4, metric='everything',
                     use_dropout=True, K=4, OutOf=None, verbose=True, debug=False)               # config.py:13-52
    state = {'attention': attention, 'engine': _train_engine()}
    train_err, valid_err, test_err = model_attention.train_from_scratch(state, Channel())
    assert all(np.isfinite([train_err, valid_err, test_err])) and valid_err > 0 and test_err > 0
    logged = capsys.readouterr().out
    assert 'alphalt ratio' in logged and 'sampling from valid' in logged and logged.count('Truth ') >= 8 and 'Sample ( 0 )' in logged
    for f in ('model_options.pkl', 'model_current.npz', 'model_best.npz', 'train_valid_test.txt', 'alphal_ratio.txt', 'alphalt_ratio.txt'):
        assert os.path.isfile(save_dir + f), f
    hist = np.loadtxt(save_dir + 'train_valid_test.txt')
    assert hist.shape == (4, 22) and list(hist[:, 1]) == [2, 4, 6, 8]           # 8 updates, validated every 2; 22 columns (:1462-1469)
    assert hist[-1, 2] < hist[0, 2]                                              # the train NLL went down
    assert Channel.saves == 4
    assert np.loadtxt(save_dir + 'alphag_ratio.txt').shape == (4,)
    best = np.load(save_dir + 'model_best.npz')
    assert {'train_err', 'valid_err', 'test_err', 'history_errs', 'Wemb', 'decoder_U', 'ff_logit_W'} <= set(best.files)
    import pickle
    mo = pickle.load(open(save_dir + 'model_options.pkl', 'rb'))
    assert mo['dim'] == 128 and mo['alpha_c'] == 0.70602 and 'engine' not in mo
    # reload_: continues from model_best_so_far.npz when the run produced one
    if os.path.isfile(save_dir + 'model_best_so_far.npz'):
        state['attention'].update(reload_=True, from_dir=str(tmp_path), max_epochs=1)
        r2 = model_attention.train_from_scratch(state, None)
        assert np.isfinite(r2[1])
    # the reference's default debug=True: one update, no error passes
    state['attention'].update(reload_=False, debug=True, max_epochs=3)
    assert model_attention.train_from_scratch(state, None) == (-1, 0, 0)
