"""Reference-EXECUTED fixtures (tests/golden/ref_*.npz, written by tests/golden/make_ref_fixtures.py from the reference's
own pure-numpy functions run in the build container) against (a) the oracle's restatement and (b) the product's host
code.  Everything here is CPU-only; the GPU counterparts (device beam search, host loop around the HIP f_next) are in
tests/test_gpu_parity.py::test_reference_executed_gen_sample_*.

Pinned here: gen_sample (model_attention.py:852-994), init_params / param_init_lstm_cond / norm_weight / ortho_weight
(:518-581, :180-282, common.py:110-134), prepare_data + get_sub_frames + the mask rule (data_engine.py:83-135, :169-218,
:258-337), generate_minibatch_idx (common.py:287), pred_probs (:996-1032), generate_sample_gpu_single_process with
_seqs2words and build_sample_pairs (metrics.py:79-83, :103-152).  NOT pinned by anything: the Theano graph."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

import stattn
from stattn import common, data_engine, metrics
from oracle import stattn_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLDEN_DIMS = dict(dim=64, dim_word=64, n_words=37, ctxg_dim=64, ctxl_dim=32, ctxm_dim=32, ctxglm_dim=64)


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def tweak_readout(P32, logit_scale, eos_bias):
    P = dict(P32)
    P['ff_logit_W'] = P['ff_logit_W'] * np.float32(logit_scale)
    P['ff_logit_b'] = P['ff_logit_b'].copy()
    P['ff_logit_b'][0] += np.float32(eos_bias)
    return P


def unpad(fx, tag):
    return [row[:n].tolist() for row, n in zip(fx[tag + '_sample'], fx[tag + '_len'])]


# ------------------------------------------------------------------------------------------------ a1 / a2
@pytest.mark.parametrize("tag", ["a", "b"])
def test_init_params_equals_the_reference_draw_sequence(tag):
    fx = load('ref_init_params.npz')
    opt = json.loads(str(fx[tag + '__options']))
    common.reset_rngs(1234)
    params = stattn.Attention().init_params(opt)
    assert list(params) == [str(k) for k in fx[tag + '__order']]
    for k, v in params.items():
        ref = fx[tag + '/' + k]
        assert np.asarray(v).dtype == np.float32 and np.shape(v) == ref.shape, k
        np.testing.assert_array_equal(np.asarray(v), ref, err_msg=k)      # bit-equal: same draws, same SVDs, same order
    # the oracle's parameter table (names, order, shapes) is the reference's too
    shapes = O.param_shapes(O.default_options(**{k: opt[k] for k in ('dim', 'dim_word', 'n_words', 'ctxg_dim', 'ctxl_dim', 'ctxm_dim',
                                                                    'ctxglm_dim', 'selector', 'ctx2out')}))
    assert list(shapes) == list(params) and all(tuple(shapes[k]) == np.shape(params[k]) for k in params)


def test_weight_initialisers_equal_the_reference():
    fx = load('ref_init_params.npz')
    common.reset_rngs(1234)
    np.testing.assert_array_equal(common.ortho_weight(7), fx['seq/ortho_7'])
    np.testing.assert_array_equal(common.norm_weight(5, 9), fx['seq/norm_5x9'])
    np.testing.assert_array_equal(common.norm_weight(6), fx['seq/norm_6_square'])
    np.testing.assert_array_equal(common.norm_weight(6, ortho=False), fx['seq/norm_6_square_noortho'])
    np.testing.assert_array_equal(common.norm_weight(4, 1, scale=0.5), fx['seq/norm_4x1_scale'])


# ------------------------------------------------------------------------------------------------ a10
def _gen_sample_cases():
    fx = load('ref_gen_sample.npz')
    return fx, json.loads(str(fx['cases']))


def _check_case(fx, c, got, exact=True, tol=0.0):
    sample, score, hs, cs = got
    tag = c['tag']
    assert isinstance(hs, list) and isinstance(cs, list) and len(hs) == 1 and len(cs) == 1
    if c['stochastic']:
        assert [int(w) for w in sample] == fx[tag + '_sample'].tolist()
        np.testing.assert_allclose(score, fx[tag + '_score'], rtol=1e-6 if exact else 1e-4)
    else:
        assert [[int(w) for w in s] for s in sample] == unpad(fx, tag), (c, sample)
        sc = np.asarray(score)
        if exact:
            assert sc.dtype == np.float32
            np.testing.assert_array_equal(sc, fx[tag + '_score'])
        else:
            np.testing.assert_allclose(sc, fx[tag + '_score'], rtol=0, atol=tol)
    assert np.shape(hs[0]) == fx[tag + '_state'].shape, (c, np.shape(hs[0]))
    if exact:
        np.testing.assert_array_equal(hs[0], fx[tag + '_state'])
        np.testing.assert_array_equal(cs[0], fx[tag + '_memory'])
    else:
        np.testing.assert_allclose(hs[0], fx[tag + '_state'], rtol=0, atol=tol)
        np.testing.assert_allclose(cs[0], fx[tag + '_memory'], rtol=0, atol=tol)


@pytest.mark.parametrize("driver", ["oracle", "product_host_loop"])
def test_gen_sample_reproduces_the_reference_execution(driver):
    """80 runs of the reference's gen_sample (k = 1, 3, 5, stochastic; 70 hypotheses that end with <eos> at different
    steps, early exits on dead_k >= k, maxlen cut-offs) -- same closures, so samples, float32 scores and the returned
    next_state / next_memory must be identical, not close."""
    fx, cases = _gen_sample_cases()
    opt = O.default_options(**GOLDEN_DIMS)
    P32 = dict(load('params.npz'))
    model = stattn.Attention()
    for c in cases:
        v = c['video']
        P64 = O.cast_params(tweak_readout(P32, c['logit_scale'], c['eos_bias']), np.float64)
        fi, fn = O.sampler_closures(P64, opt, np.float32, draw_seed=c['draw_seed'])
        args = (fx['ctxg'][v], fx['mask_ctxg'][v], fx['ctxl'][v], fx['mask_ctxl'][v], fx['ctxm'][v], fx['mask_ctxm'][v])
        if driver == "oracle":
            got = O.gen_sample(fi, fn, *args, k=c['k'], maxlen=c['maxlen'], stochastic=c['stochastic'])
        else:       # f_next is a foreign callable: Attention.gen_sample runs its numpy loop around it (no device involved)
            got = model.gen_sample(None, fi, fn, *args, opt, None, c['k'], c['maxlen'], c['stochastic'])
        _check_case(fx, c, got)


# ------------------------------------------------------------------------------------------------ f2
def _engine_from_meta(fx, key, n_words, maxlen, **kw):
    meta = json.loads(str(fx[key + '__meta']))
    feats = OrderedDict((v, tuple(fx['%s/raw/%s/%s' % (key, v, nm)] for nm in ('g', 'l', 'm'))) for v in meta['videos'])
    eng = data_engine.MemoryEngine(feats, meta['captions'], meta['worddict'], n_words=n_words, maxlen=maxlen,
                                   signature=meta['signature'], n_frames=meta['n_frames'], **kw)
    return eng, meta


@pytest.mark.parametrize("key", ["youtube2text", "lsmdc"])
def test_prepare_data_equals_the_reference(key):
    fx = load('ref_data.npz')
    names = ('x', 'x_mask', 'yg', 'yg_mask', 'yl', 'yl_mask', 'ym', 'ym_mask')
    for n_words, maxlen in ((9, None), (100, 5), (100, 2)):
        eng, meta = _engine_from_meta(fx, key, n_words, maxlen)
        res = data_engine.prepare_data(eng, meta['ids'])
        tag = '%s/nw%d_ml%s' % (key, n_words, maxlen)
        assert len(res) == int(fx[tag + '/n_out'])                        # 8, or the five Nones of :318
        for nm, arr in zip(names, res):
            if tag + '/' + nm in fx.files:
                ref = fx[tag + '/' + nm]
                assert arr.dtype == ref.dtype and arr.shape == ref.shape, (tag, nm, arr.dtype, ref.dtype)
                np.testing.assert_array_equal(arr, ref, err_msg=tag + '/' + nm)
            else:
                assert arr is None
    # per-split feature lists for sampling (prepare_data_for_blue, :137-167), incl. frame sub-sampling and zero padding
    vl = meta['videos']
    eng, _ = _engine_from_meta(fx, key, 100, None, valid_ids=vl[:2], test_ids=vl[2:], train_ids=vl)
    for split in ('valid', 'test', 'train'):
        six = eng.prepare_data_for_blue(split)
        for nm, lst in zip(('g', 'gm', 'l', 'lm', 'm', 'mm'), six):
            ref = fx['%s/blue/%s/%s' % (key, split, nm)]
            assert len(lst) == len(ref)
            if len(lst):
                got = np.asarray(lst)
                assert got.dtype == ref.dtype
                np.testing.assert_array_equal(got, ref)


def test_generate_minibatch_idx_equals_the_reference():
    want = json.loads(str(load('ref_data.npz')['minibatch_idx']))
    for key, ref in want.items():
        n, b = (int(v) for v in key.split('_'))
        assert common.generate_minibatch_idx(n, b) == ref
    with pytest.raises(AssertionError):
        common.generate_minibatch_idx(3, 4)


# ------------------------------------------------------------------------------------------------ a14
def test_pred_probs_equals_the_reference():
    fx = load('ref_pred_probs.npz')
    meta = json.loads(str(fx['meta']))
    feats = OrderedDict((v, tuple(fx['raw/%s/%s' % (v, nm)] for nm in ('g', 'l', 'm'))) for v in ('vid1', 'vid2', 'vid3'))
    eng = data_engine.MemoryEngine(feats, meta['captions'], meta['worddict'], n_words=100, n_frames=meta['n_frames'])
    eng.valid, eng.kf_valid = meta['tags'], common.generate_minibatch_idx(len(meta['tags']), meta['mb'])

    def f_log_probs(x, mask, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask):
        return -(0.37 * mask.sum(0) + 0.011 * x.sum(0) + 0.05 * np.abs(ctxg).mean((1, 2))).astype(np.float32)
    model = stattn.Attention()
    model.engine = eng
    nll, perp = model.pred_probs('valid', f_log_probs, verbose=False)     # the reference's own calling convention
    np.testing.assert_allclose(nll, float(fx['mean_nll']), rtol=1e-12)
    np.testing.assert_allclose(perp, float(fx['perplexity']), rtol=1e-12)
    batches = [data_engine.prepare_data(eng, [meta['tags'][i] for i in idx]) for idx in eng.kf_valid]
    nll2, perp2 = model.pred_probs(batches, f_log_probs)
    assert (nll2, perp2) == (nll, perp)
    with pytest.raises(NotImplementedError):
        model.pred_probs('dev', f_log_probs)


# ------------------------------------------------------------------------------------------------ f4
def test_sample_files_equal_the_reference(tmp_path):
    """The reference's generate_sample_gpu_single_process ran end to end (its gen_sample, argmin pick, _seqs2words,
    file writer, build_sample_pairs) around the oracle's f_init / f_next; the product's function around the same
    closures must write the same bytes and return the same pairs."""
    fx = load('ref_metrics.npz')
    meta = json.loads(str(fx['meta']))
    opt = O.default_options(**GOLDEN_DIMS)
    P64 = O.cast_params(tweak_readout(dict(load('params.npz')), float(fx['logit_scale']), float(fx['eos_bias'])), np.float64)
    fi, fn = O.sampler_closures(P64, opt, np.float32)
    vids = meta['valid_ids'] + meta['test_ids']
    feats = OrderedDict((v, tuple(fx['raw/%s/%s' % (v, nm)] for nm in ('g', 'l', 'm'))) for v in vids)
    eng = data_engine.MemoryEngine(feats, {}, meta['worddict'], n_words=opt['n_words'], n_frames=int(fx['n_frames']),
                                   valid_ids=meta['valid_ids'], test_ids=meta['test_ids'])
    assert metrics.MAXLEN == meta['maxlen']
    pairs = metrics.generate_sample_gpu_single_process('attention', None, opt, eng, stattn.Attention(), fi, fn,
                                                       save_dir=str(tmp_path), beam=int(fx['beam']), whichset='both')
    files = json.loads(str(fx['files']))
    for split in ('valid', 'test'):
        with open(os.path.join(str(tmp_path), '%s_samples.txt' % split)) as f:
            assert f.read() == files[split]
    want = json.loads(str(fx['returned']))
    assert [[[k, v] for k, v in p.items()] for p in pairs] == want
    assert any(len(l.split()) == metrics.MAXLEN for l in files['test'].split('\n'))     # a caption that hit MAXLEN
    assert '' in files['valid'].split('\n')[:-1]                                       # and one that is just <eos>


def test_seqs2words_equals_the_reference_incl_its_published_samples():
    fx = load('ref_metrics.npz')
    meta = json.loads(str(fx['meta']))
    widict = dict((i, w) for w, i in meta['worddict'].items()); widict[0] = '<eos>'; widict[1] = 'UNK'
    caps = json.loads(str(fx['seqs_caps']))
    assert metrics.seqs2words(caps, widict) == json.loads(str(fx['seqs_words']))
    # reference quirk kept: `w > len(word_idict)` (metrics.py:116) lets w == len through to a KeyError
    with pytest.raises(KeyError):
        metrics.seqs2words([[len(widict)]], widict)
    # the captions the reference published under test/*.txt, as ids of its real msvd_data/worddict.pkl
    pub = json.loads(str(fx['published']))
    idict = dict((int(i), w) for i, w in pub['idict'].items())
    for split, rec in pub['splits'].items():
        class Sized(dict):                  # only the words in use are stored; len() must still be the real dictionary's
            def __len__(self):
                return rec['n_dict']
        assert metrics.seqs2words(rec['ids'], Sized(idict)) == rec['text']
