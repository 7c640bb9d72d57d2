"""Batch assembly for the decoder path: the counterpart of `data_engine.prepare_data` (data_engine.py:258-337) and
of the feature-mask rule (data_engine.py:169-218), plus an in-memory engine for tests, demos and benches.

`prepare_data(engine, IDs)` takes any object with the reference engine's attributes -- `signature`, `CAP`,
`worddict`, `n_words`, `maxlen`, `get_video_{global,local,motion}_features`, `get_ctx{g,l,m}_mask` -- and returns
the 8-tuple `x, x_mask, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask` that f_grad_shared / f_log_probs /
`Decoder.set_batch` consume.  The h5/pkl-backed `Movie2Caption` loader itself is out of scope (its feature files
are not part of the reference repository); `MemoryEngine` below stands in for it with features held in arrays."""
import numpy

try:
    from .common import generate_minibatch_idx
except ImportError:
    from common import generate_minibatch_idx


def ctx_mask(ctx, dim):
    """1.0 where a frame (or region) has any non-zero feature among its first `dim`, else 0.0 -- how the reference
    tells real frames from zero padding (data_engine.py:169-218; all three feature kinds use the same rule)."""
    ctx = numpy.asarray(ctx)
    if ctx.ndim not in (2, 3, 4, 5):
        raise NotImplementedError('ctx_mask: unsupported rank %d' % ctx.ndim)
    if ctx.ndim in (4, 5):
        body = ctx[:, :, :, :dim]
    else:
        body = ctx[..., :dim]
    return (body.sum(axis=-1) != 0).astype('int32').astype('float32')


def split_id(signature, ID):
    """'vid123_7' -> ('vid123', '7'); lsmdc video ids contain underscores themselves (data_engine.py:276-282)."""
    if signature == 'youtube2text':
        vid, cap = ID.split('_')
        return vid, cap
    if signature == 'lsmdc':
        parts = ID.split('_')
        return '_'.join(parts[:-1]), parts[-1]
    raise NotImplementedError(signature)


def _tokens(engine, vid, cap_id):
    for cap in engine.CAP[vid]:
        if cap['cap_id'] == cap_id:
            return cap['tokenized'].split(' ')
    raise AssertionError('caption %s of video %s not found' % (cap_id, vid))


def prepare_data(engine, IDs):
    """data_engine.py:258-337.  Word ids outside the vocabulary become 1 (UNK); captions with `len >= engine.maxlen`
    are dropped; `x` is (max_len + 1, n) int64 zero padded (the extra row is the <eos> = 0 target), `x_mask` float32
    with `len + 1` ones per column.  When nothing survives the length filter the reference returns FIVE Nones
    (:316) -- kept, callers test `x is None`."""
    rows = []
    for ID in IDs:
        vid, cap_id = split_id(engine.signature, ID)
        words = _tokens(engine, vid, cap_id)
        seq = [engine.worddict[w] if engine.worddict[w] < engine.n_words else 1 for w in words]
        rows.append((seq, engine.get_video_global_features(vid), engine.get_video_local_features(vid),
                     engine.get_video_motion_features(vid)))
    if engine.maxlen is not None:
        rows = [r for r in rows if len(r[0]) < engine.maxlen]
        if not rows:
            return None, None, None, None, None
    yg = numpy.asarray([r[1] for r in rows])
    yl = numpy.asarray([r[2] for r in rows])
    ym = numpy.asarray([r[3] for r in rows])
    lengths = [len(r[0]) for r in rows]
    x = numpy.zeros((max(lengths) + 1, len(rows)), dtype='int64')
    x_mask = numpy.zeros(x.shape, dtype='float32')
    for j, (seq, _, _, _) in enumerate(rows):
        x[:len(seq), j] = seq
        x_mask[:len(seq) + 1, j] = 1.
    return x, x_mask, yg, engine.get_ctxg_mask(yg), yl, engine.get_ctxl_mask(yl), ym, engine.get_ctxm_mask(ym)


def sub_frames(frames, n_frames):
    """get_sub_frames (data_engine.py:117-135) for array features: a video with fewer than `n_frames` frames is padded
    with all-zero frames (pad_frames, :83-91; the mask rule then reads them as padding), any other one is cut into
    `n_frames` nearly equal runs (numpy.array_split) and the first frame of each run is kept (:93-100)."""
    frames = numpy.asarray(frames)
    n = len(frames)
    if n < n_frames:
        return numpy.concatenate([frames, numpy.zeros((n_frames - n,) + frames.shape[1:], frames.dtype)], axis=0)
    return frames[[run[0] for run in numpy.array_split(numpy.arange(n), n_frames)]]


class MemoryEngine(object):
    """The slice of `Movie2Caption` (data_engine.py:9-256) the decoder path touches, over in-memory data:
    features[vid] = (global (T, ctxg_dim), local (T, K, ctxl_dim), motion (T, ctxm_dim)) float32 arrays and
    captions[vid] = [{'cap_id': str, 'tokenized': 'a man is ...'}, ...].  With `n_frames` (the reference's K, config.py:47)
    the arrays are whole videos of any length and every access goes through `sub_frames`, like the reference's
    _filter_googlenet / _filter_rcnn / _filter_c3d (:39-60)."""

    def __init__(self, features, captions, worddict, n_words, maxlen=None, signature='youtube2text',
                 train_ids=(), valid_ids=(), test_ids=(), n_frames=None,
                 train=(), valid=(), test=(), mb_size_train=None, mb_size_test=None):
        self.K = n_frames
        self.signature = signature
        self.CAP = captions
        self.worddict = worddict
        self.word_idict = dict((i, w) for w, i in worddict.items())
        self.word_idict[0] = '<eos>'
        self.word_idict[1] = 'UNK'
        self.n_words = n_words
        self.maxlen = maxlen
        self._features = features
        g, l, m = next(iter(features.values()))
        self.ctxg_dim, self.ctxl_dim, self.ctxm_dim = g.shape[-1], l.shape[-1], m.shape[-1]
        self.train_ids, self.valid_ids, self.test_ids = list(train_ids), list(valid_ids), list(test_ids)
        # what Attention.train / pred_probs walk (data_engine.py:225-227, 251-256): caption tags 'vid_cap' per split and
        # their minibatch index lists; ctxglm_dim = ctxg_dim (the fused dimension, :247)
        self.ctxglm_dim = self.ctxg_dim
        self.train, self.valid, self.test = list(train), list(valid), list(test)
        self.mb_size_train, self.mb_size_test = mb_size_train, mb_size_test
        for name, tags, mb in (('train', self.train, mb_size_train), ('valid', self.valid, mb_size_test), ('test', self.test, mb_size_test)):
            setattr(self, 'kf_' + name, generate_minibatch_idx(len(tags), min(mb, len(tags))) if tags and mb else [])

    def get_sub_frames(self, frames, jpegs=False):
        return frames if self.K is None else sub_frames(frames, self.K)

    def get_video_global_features(self, vid):
        return self.get_sub_frames(self._features[vid][0])

    def get_video_local_features(self, vid):
        return self.get_sub_frames(self._features[vid][1])

    def get_video_motion_features(self, vid):
        return self.get_sub_frames(self._features[vid][2])

    def get_ctxg_mask(self, ctxg):
        return ctx_mask(ctxg, self.ctxg_dim)

    def get_ctxl_mask(self, ctxl):
        return ctx_mask(ctxl, self.ctxl_dim)

    def get_ctxm_mask(self, ctxm):
        return ctx_mask(ctxm, self.ctxm_dim)

    def prepare_data_for_blue(self, whichset):
        """data_engine.py:137-167: per video of a split, the three feature arrays and their masks (six lists)."""
        ids = {'valid': self.valid_ids, 'test': self.test_ids, 'train': self.train_ids}[whichset]
        out = ([], [], [], [], [], [])
        for vid in ids:
            g, l, m = (self.get_video_global_features(vid), self.get_video_local_features(vid),
                       self.get_video_motion_features(vid))
            for lst, v in zip(out, (g, self.get_ctxg_mask(g), l, self.get_ctxl_mask(l), m, self.get_ctxm_mask(m))):
                lst.append(v)
        return out
