#!/usr/bin/env python
"""Reference-EXECUTED fixtures: runs the reference's own pure-numpy functions and stores what they return.

Build container only.  The reference (Python 2 + Theano) cannot be imported, but the functions below touch
nothing of Theano.  This script reads the reference sources where they lie (/root/reference, never copied),
runs lib2to3's mechanical py2 -> py3 rewrite (print, xrange, iteritems, ...) over each file in memory, parses
the result with `ast`, lifts out ONLY the named function / method definitions, and executes them:

  model_attention.py  _p :31; Attention.__init__ :43, get_layer :51, load_params :60, param_init_fflayer :80,
                      param_init_lstm_cond :180-282, init_params :518-581, gen_sample :852-994, pred_probs :996-1032
  common.py           ortho_weight :110, norm_weight :124, generate_minibatch_idx :287, flatten_list_of_list :316
  data_engine.py      Movie2Caption.pad_frames :83, extract_frames_equally_spaced :93, get_sub_frames :117,
                      prepare_data_for_blue :137, get_ctx{g,l,m}_mask :169-218; prepare_data :258-337
  metrics.py          MAXLEN :14, build_sample_pairs :79, generate_sample_gpu_single_process :103-146 (with its nested _seqs2words :109)

The only edits beyond lib2to3 are the two Python-2 integer divisions it cannot see (`ranks_flat / voc_size`,
model_attention.py:926, and `dataset_size / minibatch_size`, common.py:291), turned into `//` on the AST.

What is written under tests/golden/ is DATA ONLY (inputs and the values the reference code returned); no
reference source text is stored in any form.  The Theano graph itself (f_init / f_next / build_model) cannot
run: where the reference drivers need those callables they are handed the oracle's (oracle/stattn_oracle.py),
so these fixtures pin gen_sample, the parameter factory, the batch assembly, the sample-file writer and
pred_probs to reference-executed values, and leave the graph rows "parity unpinned" (DESIGN.md section 3).

Run from the repo root:  python tests/golden/make_ref_fixtures.py"""
import ast
import copy
import io
import json
import os
import pickle
import sys
import tempfile
import types
import warnings
from collections import OrderedDict

import numpy
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('STATTN_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
from oracle import stattn_oracle as O          # noqa: E402

GOLDEN_DIMS = dict(dim=64, dim_word=64, n_words=37, ctxg_dim=64, ctxl_dim=32, ctxm_dim=32, ctxglm_dim=64)   # = params.npz


# ----------------------------------------------------------------------------------------------------------------
# source -> py3 AST -> selected definitions
# ----------------------------------------------------------------------------------------------------------------
def py3_tree(fname):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    with open(os.path.join(REF, fname)) as f:
        src = f.read().expandtabs(8)                 # the files mix tabs and spaces (python 2: tab = 8 columns)
    return ast.parse(str(tool.refactor_string(src + '\n', fname)))


class Py2IntDiv(ast.NodeTransformer):
    """`a / b` on two ints is floor division in python 2: rewrite the named sites."""
    SITES = {('ranks_flat', 'voc_size'), ('dataset_size', 'minibatch_size')}

    def __init__(self):
        self.hits = []

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if (isinstance(node.op, ast.Div) and isinstance(node.left, ast.Name) and isinstance(node.right, ast.Name)
                and (node.left.id, node.right.id) in self.SITES):
            self.hits.append((node.left.id, node.right.id))
            node.op = ast.FloorDiv()
        return node


def top(tree, name):
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name == name:
            return node
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in node.targets):
            return node
    raise KeyError(name)


def class_subset(tree, cls, methods):
    node = copy.deepcopy(top(tree, cls))
    have = {n.name: n for n in node.body if isinstance(n, ast.FunctionDef)}
    node.body = [have[m] for m in methods]
    return node


def nested(fn_node, name):
    for n in ast.walk(fn_node):
        if isinstance(n, ast.FunctionDef) and n.name == name and n is not fn_node:
            return copy.deepcopy(n)
    raise KeyError(name)


def run(nodes, ns, label):
    fix = Py2IntDiv()
    mod = ast.Module(body=[fix.visit(copy.deepcopy(n)) for n in nodes], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, '<reference %s>' % label, 'exec'), ns)
    return fix.hits


def load_reference():
    """Returns namespaces holding the executed reference definitions."""
    t_common, t_model = py3_tree('common.py'), py3_tree('model_attention.py')
    t_de, t_metrics = py3_tree('data_engine.py'), py3_tree('metrics.py')
    common = {'numpy': numpy, 'rng_numpy': numpy.random.RandomState(1234)}       # common.py:25
    hits = run([top(t_common, n) for n in ('ortho_weight', 'norm_weight', 'generate_minibatch_idx',
                                           'flatten_list_of_list')], common, 'common.py')
    assert hits == [('dataset_size', 'minibatch_size')], hits
    de = {'numpy': numpy}
    run([class_subset(t_de, 'Movie2Caption', ['pad_frames', 'extract_frames_equally_spaced', 'get_sub_frames',
                                              'prepare_data_for_blue', 'get_ctxg_mask', 'get_ctxl_mask', 'get_ctxm_mask']),
         top(t_de, 'prepare_data')], de, 'data_engine.py')
    model = {'numpy': numpy, 'copy': copy, 'sys': sys, 'OrderedDict': OrderedDict, 'warnings': warnings,
             'norm_weight': common['norm_weight'], 'ortho_weight': common['ortho_weight'],          # from common import *
             'common': types.SimpleNamespace(flatten_list_of_list=common['flatten_list_of_list']),
             'data_engine': types.SimpleNamespace(prepare_data=de['prepare_data'])}
    hits = run([top(t_model, '_p'),
                class_subset(t_model, 'Attention', ['__init__', 'get_layer', 'load_params', 'param_init_fflayer',
                                                    'param_init_lstm_cond', 'init_params', 'gen_sample', 'pred_probs',
                                                    # get_layer evals 'self.fflayer' / 'self.lstm_cond_layer' for their
                                                    # attribute only; these two Theano graph builders are defined, never called
                                                    'fflayer', 'lstm_cond_layer'])],
               model, 'model_attention.py')
    assert hits == [('ranks_flat', 'voc_size')], hits
    metrics = {'numpy': numpy, 'OrderedDict': OrderedDict}
    gsp = top(t_metrics, 'generate_sample_gpu_single_process')
    run([top(t_metrics, 'MAXLEN'), top(t_metrics, 'build_sample_pairs'), gsp], metrics, 'metrics.py')
    seqs_ns = {'numpy': numpy}
    run([nested(gsp, '_seqs2words')], seqs_ns, 'metrics.py:_seqs2words')

    def seqs2words(caps, word_idict):
        seqs_ns['engine'] = types.SimpleNamespace(word_idict=word_idict)        # the closure variable of :109-119
        return seqs_ns['_seqs2words'](caps)
    return types.SimpleNamespace(common=common, model=model, de=de, metrics=metrics, seqs2words=seqs2words)


class quiet(object):
    """the reference functions print progress lines"""

    def __enter__(self):
        self._o = sys.stdout
        sys.stdout = io.StringIO()

    def __exit__(self, *a):
        sys.stdout = self._o


# ----------------------------------------------------------------------------------------------------------------
# inputs shared with the tests (the tests rebuild NOTHING from here: everything they need is stored in the fixtures)
# ----------------------------------------------------------------------------------------------------------------
def ref_engine(R, raw, captions, worddict, n_words, maxlen, n_frames, signature, dims, valid_ids=(), test_ids=(), train_ids=()):
    """An instance of the reference's Movie2Caption without its h5 / pkl loading (__init__, load_data and the
    _filter_* readers are not executed): get_video_*_features = get_sub_frames(raw array), which is what the
    reference's _filter_googlenet / _filter_rcnn / _filter_c3d do after reading the file (data_engine.py:39-60)."""
    cls = R.de['Movie2Caption']

    class Engine(cls):
        def get_video_global_features(self, vid):
            return self.get_sub_frames(raw[vid][0])

        def get_video_local_features(self, vid):
            return self.get_sub_frames(raw[vid][1])

        def get_video_motion_features(self, vid):
            return self.get_sub_frames(raw[vid][2])
    e = Engine.__new__(Engine)
    e.signature, e.CAP, e.worddict, e.n_words, e.maxlen = signature, captions, worddict, n_words, maxlen
    e.K, e.OutOf = n_frames, None
    e.ctxg_dim, e.ctxl_dim, e.ctxm_dim = dims
    e.word_idict = dict((v, k) for k, v in worddict.items())
    e.word_idict[0] = '<eos>'
    e.word_idict[1] = 'UNK'                                                      # data_engine.py:240-245
    e.valid_ids, e.test_ids, e.train_ids = list(valid_ids), list(test_ids), list(train_ids)
    return e


def tweak_readout(P32, logit_scale, eos_bias):
    """golden parameters with a sharper vocabulary distribution (ff_logit_W * logit_scale) and a likelier <eos>
    (ff_logit_b[0] + eos_bias), both in float32: hypotheses then end at different steps."""
    P = dict(P32)
    P['ff_logit_W'] = P['ff_logit_W'] * np.float32(logit_scale)
    P['ff_logit_b'] = P['ff_logit_b'].copy()
    P['ff_logit_b'][0] += np.float32(eos_bias)
    return P


def pad_ragged(seqs, fill=-1):
    n = max([len(s) for s in seqs] + [1])
    out = np.full((len(seqs), n), fill, np.int64)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
    return out, np.array([len(s) for s in seqs], np.int64)


# ----------------------------------------------------------------------------------------------------------------
def fixture_init_params(R):
    """init_params (:518-581) under RandomState(1234), two option sets (selector / ctx2out on and off; dim_word == dim
    takes the orthogonal branch of norm_weight for ff_logit_lstm_W)."""
    out = {}
    sets = {'a': dict(dim=32, dim_word=16, n_words=23, ctxg_dim=32, ctxl_dim=24, ctxm_dim=20, ctxglm_dim=32,
                      selector=True, ctx2out=True),
            'b': dict(dim=16, dim_word=16, n_words=11, ctxg_dim=16, ctxl_dim=8, ctxm_dim=12, ctxglm_dim=16,
                      selector=False, ctx2out=False)}
    for tag, o in sets.items():
        opt = dict(o, encoder='none', n_layers_init=0, n_layers_out=1, prev2out=True, use_dropout=True)
        R.common['rng_numpy'] = numpy.random.RandomState(1234)
        with quiet():
            params = R.model['Attention']().init_params(opt)
        out[tag + '__options'] = json.dumps(opt, sort_keys=True)
        out[tag + '__order'] = np.array(list(params))
        for k, v in params.items():
            out[tag + '/' + k] = np.asarray(v)
            assert np.asarray(v).dtype == np.float32, (k, np.asarray(v).dtype)
    # the raw initialisers, called in sequence from a fresh stream
    R.common['rng_numpy'] = numpy.random.RandomState(1234)
    out['seq/ortho_7'] = R.common['ortho_weight'](7)
    out['seq/norm_5x9'] = R.common['norm_weight'](5, 9)
    out['seq/norm_6_square'] = R.common['norm_weight'](6)
    out['seq/norm_6_square_noortho'] = R.common['norm_weight'](6, ortho=False)
    out['seq/norm_4x1_scale'] = R.common['norm_weight'](4, 1, scale=0.5)
    np.savez_compressed(os.path.join(HERE, 'ref_init_params.npz'), **out)


def fixture_gen_sample(R):
    """gen_sample (:852-994) executed as written, driven by the oracle's f_init / f_next (float32 outputs like the
    compiled Theano functions) on the committed golden parameters; <eos> made likely so hypotheses die at different
    steps.  Stored per case: the returned samples, float32 scores and final next_state / next_memory."""
    opt = O.default_options(**GOLDEN_DIMS)
    P32 = dict(np.load(os.path.join(HERE, 'params.npz')))
    vids = O.synthetic_batch(opt, B=4, T=5, K=3, t=3, seed=70)
    out = dict(ctxg=vids['ctxg'], ctxl=vids['ctxl'], ctxm=vids['ctxm'], mask_ctxg=vids['mask_ctxg'],
               mask_ctxl=vids['mask_ctxl'], mask_ctxm=vids['mask_ctxm'])
    cases = []
    model = R.model['Attention']()
    for logit_scale, eos_bias in ((1.0, 0.0), (1.0, 0.5), (1.0, 0.7), (8.0, 1.0)):
        P = tweak_readout(P32, logit_scale, eos_bias)
        P64 = O.cast_params(P, np.float64)
        for v in range(4):
            for k, maxlen, draw in ((1, 7, None), (3, 9, None), (5, 12, None), (5, 3, None), (1, 10, 77 + v)):
                fi, fn = O.sampler_closures(P64, opt, np.float32, draw_seed=draw)
                args = (vids['ctxg'][v], vids['mask_ctxg'][v], vids['ctxl'][v], vids['mask_ctxl'][v],
                        vids['ctxm'][v], vids['mask_ctxm'][v])
                sample, score, hs, cs = model.gen_sample(None, fi, fn, *args, opt, None, k, maxlen, draw is not None)
                tag = 'case%02d' % len(cases)
                cases.append(dict(tag=tag, video=v, k=k, maxlen=maxlen, logit_scale=logit_scale, eos_bias=eos_bias, draw_seed=draw,
                                  stochastic=draw is not None))
                if draw is not None:                       # stochastic: sample is one flat word list, score one number
                    out[tag + '_sample'] = np.asarray(sample, np.int64)
                    out[tag + '_score'] = np.asarray(score)
                else:
                    out[tag + '_sample'], out[tag + '_len'] = pad_ragged([[int(w) for w in s] for s in sample])
                    sc = np.asarray(score)
                    assert sc.dtype == np.float32, sc.dtype
                    out[tag + '_score'] = sc
                assert len(hs) == 1 and len(cs) == 1
                out[tag + '_state'] = np.asarray(hs[0])
                out[tag + '_memory'] = np.asarray(cs[0])
    out['cases'] = json.dumps(cases)
    n_dead = sum(int((out[c['tag'] + '_sample'][:, :].max(1) >= 0).sum() and
                     sum(1 for i, n in enumerate(out[c['tag'] + '_len']) if out[c['tag'] + '_sample'][i, n - 1] == 0))
                 for c in cases if not c['stochastic'])
    assert n_dead > 40, n_dead                             # the death bookkeeping is exercised
    np.savez_compressed(os.path.join(HERE, 'ref_gen_sample.npz'), **out)
    return n_dead, len(cases)


def make_raw_videos(rng, dims, frames, regions):
    raw = OrderedDict()
    for vid, n in frames.items():
        raw[vid] = (rng.standard_normal((n, dims[0])).astype(np.float32),
                    rng.standard_normal((n, regions, dims[1])).astype(np.float32),
                    rng.standard_normal((n, dims[2])).astype(np.float32))
    return raw


def fixture_data(R):
    """prepare_data (:258-337) with the frame sub-sampling / zero padding of get_sub_frames (:83-135) and the mask
    rule (:169-218), plus generate_minibatch_idx (common.py:287-301)."""
    rng = np.random.RandomState(11)
    dims, K = (6, 5, 7), 4
    out = {}
    for signature, frames in (('youtube2text', OrderedDict([('vid1', 9), ('vid2', 2), ('vid3', 4), ('vid4', 13)])),
                              ('lsmdc', OrderedDict([('a_b_1', 6), ('c_2', 3)]))):
        raw = make_raw_videos(rng, dims, frames, regions=3)
        first = list(raw)[0]
        raw[first][0][1, :] = 0.0                                   # a real frame whose features are all zero -> mask 0
        raw[first][1][0, 1, 0] = 1.0; raw[first][1][0, 1, 1:] = 0.0; raw[first][1][0, 1, 1] = -1.0   # features cancel -> mask 0
        worddict = OrderedDict((w, i + 2) for i, w in enumerate('a man is cooking dog running quickly outside the woman'.split()))
        vl = list(raw)
        caps = {vl[0]: [{'cap_id': '0', 'tokenized': 'a man is cooking'},
                        {'cap_id': '3', 'tokenized': 'a dog is running quickly outside the woman'}],
                vl[1]: [{'cap_id': '0', 'tokenized': 'a dog'}, {'cap_id': '1', 'tokenized': 'the woman is running'}]}
        for extra in vl[2:]:
            caps[extra] = [{'cap_id': '7', 'tokenized': 'man is running quickly'}]
        ids = ['%s_%s' % (v, c['cap_id']) for v in vl for c in caps[v]]
        key = signature
        out[key + '__meta'] = json.dumps(dict(signature=signature, worddict=worddict, captions=caps, ids=ids, n_frames=K,
                                              dims=dims, videos=vl))
        for v in vl:
            for nm, arr in zip(('g', 'l', 'm'), raw[v]):
                out['%s/raw/%s/%s' % (key, v, nm)] = arr
        for n_words, maxlen in ((9, None), (100, 5), (100, 2)):
            eng = ref_engine(R, raw, caps, worddict, n_words, maxlen, K, signature, dims)
            res = R.de['prepare_data'](eng, ids)
            tag = '%s/nw%d_ml%s' % (key, n_words, maxlen)
            out[tag + '/n_out'] = np.int64(len(res))
            for nm, arr in zip(('x', 'x_mask', 'yg', 'yg_mask', 'yl', 'yl_mask', 'ym', 'ym_mask'), res):
                if arr is not None:
                    out[tag + '/' + nm] = arr
        eng = ref_engine(R, raw, caps, worddict, 100, None, K, signature, dims, valid_ids=vl[:2], test_ids=vl[2:], train_ids=vl)
        for split in ('valid', 'test', 'train'):
            six = eng.prepare_data_for_blue(split)
            for nm, lst in zip(('g', 'gm', 'l', 'lm', 'm', 'mm'), six):
                out['%s/blue/%s/%s' % (key, split, nm)] = np.asarray(lst) if len(lst) else np.zeros((0,), np.float32)
    mb = {}
    with quiet():
        for n, b in ((10, 5), (11, 4), (7, 7), (64, 10), (5, 1)):
            mb['%d_%d' % (n, b)] = R.common['generate_minibatch_idx'](n, b)
    out['minibatch_idx'] = json.dumps(mb)
    np.savez_compressed(os.path.join(HERE, 'ref_data.npz'), **out)


def fixture_metrics(R):
    """generate_sample_gpu_single_process (metrics.py:103-146) end to end: the reference's loop over a split, its
    gen_sample, argmin pick, _seqs2words and file writer; f_init / f_next are the oracle's on the golden parameters.
    Stored: the video features, the dictionary and the text of the two files it wrote.  Also _seqs2words on its own,
    on hand-made id lists (ids beyond the dictionary print as UNK, :116-117) and on the reference's published sample files
    test/*.txt through the reference's real msvd_data/worddict.pkl (only the words those files use are stored)."""
    opt = O.default_options(**GOLDEN_DIMS)
    P64 = O.cast_params(tweak_readout(dict(np.load(os.path.join(HERE, 'params.npz'))), 8.0, 1.0), np.float64)
    fi, fn = O.sampler_closures(P64, opt, np.float32)
    rng = np.random.RandomState(23)
    dims, K = (GOLDEN_DIMS['ctxg_dim'], GOLDEN_DIMS['ctxl_dim'], GOLDEN_DIMS['ctxm_dim']), 5
    raw = make_raw_videos(rng, dims, OrderedDict([('vid1', 5), ('vid2', 3), ('vid3', 8), ('vid4', 5), ('vid5', 6)]), regions=3)
    words = ('a man is cooking dog running quickly outside the woman playing guitar cat riding horse slicing onion '
             'potato water boy girl jumping on in with and two are people car road ball street bike food table').split()
    assert len(set(words)) >= GOLDEN_DIMS['n_words'] - 2
    worddict = OrderedDict((w, i + 2) for i, w in enumerate(words[:GOLDEN_DIMS['n_words'] - 2]))
    eng = ref_engine(R, raw, {}, worddict, GOLDEN_DIMS['n_words'], None, K, 'youtube2text', dims,
                     valid_ids=['vid1', 'vid2'], test_ids=['vid3', 'vid4', 'vid5'])
    model = R.model['Attention']()
    out = dict(logit_scale=np.float32(8.0), eos_bias=np.float32(1.0), n_frames=np.int64(K), beam=np.int64(3),
               meta=json.dumps(dict(worddict=worddict, valid_ids=eng.valid_ids, test_ids=eng.test_ids, maxlen=R.metrics['MAXLEN'])))
    for v in raw:
        for nm, arr in zip(('g', 'l', 'm'), raw[v]):
            out['raw/%s/%s' % (v, nm)] = arr
    with tempfile.TemporaryDirectory() as d:
        with quiet():
            pairs = R.metrics['generate_sample_gpu_single_process']('attention', None, opt, eng, model, fi, fn,
                                                                    save_dir=d, beam=3, whichset='both')
        files = {}
        for split in ('valid', 'test'):
            with open(os.path.join(d, '%s_samples.txt' % split)) as f:
                files[split] = f.read()
    out['files'] = json.dumps(files)
    out['returned'] = json.dumps([list(p.items()) for p in pairs])       # (samples_valid, samples_test): OrderedDicts (:79-83)
    # _seqs2words alone
    widict = dict(eng.word_idict)
    caps = [[2, 3, 4, 0, 5], [0], [], [36, 38, 39, 1000, 2], [1, 1, 0], [5, 6, 7, 8, 9, 10]]
    out['seqs_caps'] = json.dumps(caps)
    out['seqs_words'] = json.dumps(R.seqs2words(caps, widict))
    # the reference's own sample files through the reference's own dictionary
    with open(os.path.join(REF, 'msvd_data', 'worddict.pkl'), 'rb') as f:
        real = pickle.load(f, encoding='latin1')
    real_idict = dict((v, k) for k, v in real.items()); real_idict[0] = '<eos>'; real_idict[1] = 'UNK'
    pub = {}
    for split in ('valid', 'test'):
        with open(os.path.join(REF, 'test', '%s_samples.txt' % split)) as f:
            lines = f.read().split('\n')
        lines = [l for l in lines[:60]]
        ids = [[real[w] for w in l.split(' ') if w] + [0] for l in lines]
        pub[split] = dict(ids=ids, text=R.seqs2words(ids, real_idict), n_dict=len(real_idict))
        assert pub[split]['text'] == [' '.join(w for w in l.split(' ') if w) for l in lines]
    used = sorted(set(w for s in pub.values() for cap in s['ids'] for w in cap))
    out['published'] = json.dumps(dict(splits=pub, idict=dict((str(i), real_idict[i]) for i in used)))
    np.savez_compressed(os.path.join(HERE, 'ref_metrics.npz'), **out)
    return files


def fixture_pred_probs(R):
    """pred_probs (:996-1032): the loop over a split's minibatches, prepare_data per minibatch, mean NLL and perplexity,
    with a closed-form stand-in for f_log_probs (the Theano function cannot run)."""
    rng = np.random.RandomState(5)
    dims, K = (6, 5, 7), 4
    raw = make_raw_videos(rng, dims, OrderedDict([('vid1', 9), ('vid2', 2), ('vid3', 4)]), regions=2)
    worddict = OrderedDict((w, i + 2) for i, w in enumerate('a man is cooking dog running quickly outside'.split()))
    caps = {'vid1': [{'cap_id': str(i), 'tokenized': t} for i, t in enumerate(['a man is cooking', 'a dog', 'man is running quickly outside'])],
            'vid2': [{'cap_id': '0', 'tokenized': 'a dog is running'}, {'cap_id': '1', 'tokenized': 'dog'}],
            'vid3': [{'cap_id': '4', 'tokenized': 'a man is running outside quickly'}, {'cap_id': '5', 'tokenized': 'is'}]}
    tags = ['vid1_0', 'vid2_0', 'vid1_1', 'vid3_4', 'vid1_2', 'vid2_1', 'vid3_5']
    eng = ref_engine(R, raw, caps, worddict, 100, None, K, 'youtube2text', dims)
    with quiet():
        eng.valid, eng.kf_valid = tags, R.common['generate_minibatch_idx'](len(tags), 3)

    def f_log_probs(x, mask, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask):
        return -(0.37 * mask.sum(0) + 0.011 * x.sum(0) + 0.05 * np.abs(ctxg).mean((1, 2))).astype(np.float32)
    model = R.model['Attention']()
    model.engine = eng
    with quiet():
        nll, perp = model.pred_probs('valid', f_log_probs, verbose=False)
    out = dict(meta=json.dumps(dict(worddict=worddict, captions=caps, tags=tags, n_frames=K, dims=dims, mb=3)),
               mean_nll=np.float64(nll), perplexity=np.float64(perp))
    for v in raw:
        for nm, arr in zip(('g', 'l', 'm'), raw[v]):
            out['raw/%s/%s' % (v, nm)] = arr
    np.savez_compressed(os.path.join(HERE, 'ref_pred_probs.npz'), **out)
    return nll, perp


def main():
    R = load_reference()
    fixture_init_params(R)
    print('gen_sample: %d hypotheses ended with <eos> over %d cases' % fixture_gen_sample(R))
    fixture_data(R)
    print('sample files:', fixture_metrics(R))
    print('pred_probs:', fixture_pred_probs(R))
    print('wrote', sorted(f for f in os.listdir(HERE) if f.startswith('ref_')))


if __name__ == '__main__':
    main()
