"""Host-side counterparts of the helpers of the reference's common.py that the decoder path
uses: the numpy RNG habit (common.py:16-25), zipp / unzip / itemlist (common.py:78-91), the
weight initialisers (common.py:110-134) and load_params (common.py:149-155).

Theano shared variables become `SharedVar` objects: get_value() / set_value() copy between
host numpy arrays and the device-resident parameter store of libstattn.so."""
from collections import OrderedDict

import numpy


def get_two_rngs(seed=None):
    """common.py:16-23.  The second generator (Theano's MRG stream) is the library's own
    counter-based generator; its seed is what is returned."""
    seed = 1234 if seed is None else seed
    return numpy.random.RandomState(seed), seed


rng_numpy, rng_theano = get_two_rngs()


def reset_rngs(seed=1234):
    global rng_numpy, rng_theano
    rng_numpy, rng_theano = get_two_rngs(seed)


class SharedVar(object):
    """theano.shared stand-in: a named parameter living on the device once bound."""

    def __init__(self, value, name):
        self.name = name
        self._host = numpy.asarray(value, dtype=numpy.float32)
        self._dec = None

    def bind(self, decoder):
        self._dec = decoder

    def get_value(self, borrow=False):
        if self._dec is not None:
            return self._dec.get_param(self.name)
        return self._host.copy()

    def set_value(self, value, borrow=False):
        value = numpy.asarray(value, dtype=numpy.float32)
        if value.shape != self._host.shape:
            raise ValueError("%s: shape %s != %s" % (self.name, value.shape, self._host.shape))
        self._host = value
        if self._dec is not None:
            self._dec.set_param(self.name, value)

    @property
    def shape(self):
        return self._host.shape


class SharedScalar(object):
    """`use_noise = theano.shared(numpy.float32(0.))` (model_attention.py:585)."""

    def __init__(self, value=0.0, on_change=None):
        self._v = numpy.float32(value)
        self._cb = on_change

    def get_value(self):
        return self._v

    def set_value(self, v):
        self._v = numpy.float32(v)
        if self._cb is not None:
            self._cb(float(self._v))


def zipp(params, tparams):
    """push parameters to the device (common.py:78-80)"""
    for kk, vv in params.items():
        tparams[kk].set_value(vv)


def unzip(zipped):
    """pull parameters from the device (common.py:83-87)"""
    new_params = OrderedDict()
    for kk, vv in zipped.items():
        new_params[kk] = vv.get_value()
    return new_params


def itemlist(tparams):
    return [vv for kk, vv in tparams.items()]


def ortho_weight(ndim):
    """common.py:110-122: left singular vectors of a Gaussian matrix."""
    W = rng_numpy.randn(ndim, ndim)
    u, _, _ = numpy.linalg.svd(W)
    return u.astype('float32')


def norm_weight(nin, nout=None, scale=0.01, ortho=True):
    """common.py:124-134: orthogonal when square (and ortho), else scale * randn."""
    if nout is None:
        nout = nin
    if nout == nin and ortho:
        W = ortho_weight(nin)
    else:
        W = scale * rng_numpy.randn(nin, nout)
    return W.astype('float32')


def load_params(path, params):
    """common.py:149-155: npz key = parameter name."""
    pp = numpy.load(path)
    for kk in params:
        if kk not in pp:
            raise Warning('%s is not in the archive' % kk)
        params[kk] = pp[kk]
    return params


def generate_minibatch_idx(dataset_size, minibatch_size):
    """common.py:287-301: consecutive index lists of `minibatch_size`, the remainder as a last, shorter one."""
    assert dataset_size >= minibatch_size
    full = dataset_size - dataset_size % minibatch_size
    idx = [list(range(s, s + minibatch_size)) for s in range(0, full, minibatch_size)]
    if full < dataset_size:
        idx.append(list(range(full, dataset_size)))
    return idx


def flatten_list_of_list(l):
    return [item for sublist in l for item in sublist]
