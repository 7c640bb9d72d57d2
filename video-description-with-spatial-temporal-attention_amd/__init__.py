"""stattn -- MI355X-native spatial-temporal-attention LSTM caption decoder.

Host-side mirror of the reference's operator surface for ONE path (model_attention.py's
decoder): `model_attention.Attention` keeps init_params / init_tparams / build_model /
build_sampler -> (f_init, f_next) / gen_sample / pred_probs; `common` keeps zipp / unzip /
itemlist / the weight initialisers; `data_engine.prepare_data` and `metrics` (sample files) are the
host-side callers either side of the path.  All arithmetic runs in libstattn.so (hand-written
gfx950 HIP kernels behind the C ABI of include/stattn.h); there is no CPU fallback."""
from . import _native, common, data_engine, dp, metrics, model_attention  # noqa: F401
from ._native import Decoder, NativeError, library_path  # noqa: F401
from .model_attention import Attention  # noqa: F401

__all__ = ["Attention", "Decoder", "NativeError", "common", "data_engine", "metrics", "model_attention", "library_path"]
