"""ctypes binding of libstattn.so (C ABI: include/stattn.h).

This is the stub a maintainer of the reference would add where model_attention.py calls
`theano.function(...)`: every method below forwards to exactly one C entry point.  The
library is loaded lazily and LOUDLY: a missing .so or a box without a HIP device raises
NativeError -- there is no CPU fallback anywhere in the product path."""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class NativeError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libstattn.so")


class _Options(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("dim", "dim_word", "n_words", "ctxg_dim", "ctxl_dim", "ctxm_dim",
                 "selector", "use_dropout", "prev2out", "ctx2out", "lt_mode", "precision")] + [("reserved", C.c_int32 * 4)]


_F = C.POINTER(C.c_float)
_I64 = C.POINTER(C.c_int64)
_H = C.c_void_p

_SIGNATURES = {
    # name: (restype, argtypes)
    "stattn_create": (C.c_int, [C.POINTER(_Options), C.c_int, C.c_void_p, C.POINTER(_H)]),
    "stattn_destroy": (None, [_H]),
    "stattn_last_error": (C.c_char_p, [_H]),
    "stattn_version": (C.c_char_p, []),
    "stattn_sync": (C.c_int, [_H]),
    "stattn_param_count": (C.c_int, [_H]),
    "stattn_param_name": (C.c_char_p, [_H, C.c_int]),
    "stattn_param_shape": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int64 * 2), C.POINTER(C.c_int)]),
    "stattn_set_param": (C.c_int, [_H, C.c_char_p, _F, C.c_size_t]),
    "stattn_get_param": (C.c_int, [_H, C.c_char_p, _F, C.c_size_t]),
    "stattn_get_grad": (C.c_int, [_H, C.c_char_p, _F, C.c_size_t]),
    "stattn_param_buffer_dev": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "stattn_grad_buffer_dev": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "stattn_set_use_noise": (C.c_int, [_H, C.c_float]),
    "stattn_set_seed": (C.c_int, [_H, C.c_uint64]),
    "stattn_set_dropout_masks": (C.c_int, [_H, _F, _F, _F, C.c_int, C.c_int]),
    "stattn_f_init": (C.c_int, [_H, _F, _F, C.c_int, _F, _F]),
    "stattn_f_next": (C.c_int, [_H, _I64, C.c_int, _F, _F, _F, _F, _F, _F, C.c_int, C.c_int, _F, _F,
                                _F, _I64, _F, _F, _F, _F, _F, _F, _F]),
    "stattn_set_video": (C.c_int, [_H, _F, _F, _F, C.c_int, C.c_int]),
    "stattn_beam_stage": (C.c_int, [_H, C.c_int, _F, _F, _F, _F, C.c_int, C.c_int]),
    "stattn_beam_search": (C.c_int, [_H, C.c_int, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     _I64, _F, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stattn_beam_final_state": (C.c_int, [_H, _F, _F, C.POINTER(C.c_int32)]),
    "stattn_sample_search": (C.c_int, [_H, C.c_int, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, _I64, _F, C.POINTER(C.c_int32)]),
    "stattn_set_batch": (C.c_int, [_H, _I64, _F, C.c_int, C.c_int, _F, _F, _F, _F, _F, _F, C.c_int, C.c_int]),
    "stattn_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "stattn_host_free": (C.c_int, [C.c_void_p]),
    "stattn_prefetch_batch": (C.c_int, [_H, _I64, _F, C.c_int, C.c_int, _F, _F, _F, _F, _F, _F, C.c_int, C.c_int]),
    "stattn_swap_batch": (C.c_int, [_H]),
    "stattn_forward_train": (C.c_int, [_H]),
    "stattn_get_forward": (C.c_int, [_H, _F, _F, _F, _F, _F, _F, _F]),
    "stattn_get_states": (C.c_int, [_H, _F, _F, _F]),
    "stattn_backward": (C.c_int, [_H, C.c_float, C.c_float]),
    "stattn_get_loss": (C.c_int, [_H, C.c_float, C.c_float, _F]),
    "stattn_update": (C.c_int, [_H, C.c_float, C.c_float]),
    "stattn_reset_optimizer": (C.c_int, [_H]),
    "stattn_comm_unique_id": (C.c_int, [C.c_void_p]),
    "stattn_comm_init": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "stattn_comm_destroy": (C.c_int, [_H]),
    "stattn_comm_info": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "stattn_comm_set_overlap": (C.c_int, [_H, C.c_int]),
    "stattn_allreduce_grads": (C.c_int, [_H]),
    "stattn_broadcast_params": (C.c_int, [_H, C.c_int]),
    "stattn_allreduce_scalars": (C.c_int, [_H, _F, C.c_int]),
    "stattn_comm_stats": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _F]),
    "stattn_comm_library_path": (C.c_char_p, []),
    "stattn_set_profiling": (C.c_int, [_H, C.c_int]),
    "stattn_get_kernel_ms": (C.c_int, [_H, C.c_int, _F, C.POINTER(C.c_int)]),
}

# csrc/stattn_dbg.h: development entry points (kernels in isolation for tests/ and tools/), not the drop-in surface
_DBG_SIGNATURES = {
    "stattn_dbg_gemm": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                  _F, _F, _F, _F, C.c_int, _F]),
    "stattn_dbg_time_gemm": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]),
    "stattn_dbg_counter": (C.c_long, [_H, C.c_int]),
    "stattn_dbg_time_gemm_bf16": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]),
    "stattn_dbg_time_skinny": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]),
    "stattn_dbg_grad_regions": (C.c_int, [C.POINTER(_Options), C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "stattn_dbg_redzone_enabled": (C.c_int, []),
    "stattn_dbg_redzone_buffers": (C.c_long, [_H]),
    "stattn_dbg_redzone_check": (C.c_int, [_H]),
    "stattn_dbg_redzone_poke": (C.c_int, [_H, C.c_char_p, C.c_long]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
REDZONE_CHECKS = [0]        # red-zone scans run by this process (STATTN_DBG_REDZONE=1): tests/test_gpu_z1_redzone.py reads it at session end
DEBUG_SYMBOLS = tuple(_DBG_SIGNATURES)


def _elf_dynamic_strings(path, tag):
    """Values of the string-valued dynamic entries `tag` (1 = DT_NEEDED, 14 = DT_SONAME) of a 64-bit little-endian ELF
    shared object; [] when the file is not one or cannot be parsed."""
    import struct
    try:
        with open(path, "rb") as f:
            data = f.read()
        if data[:6] != b"\x7fELF\x02\x01":
            return []
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
        out = []
        for sec in secs:
            if sec[1] != 6:                                   # SHT_DYNAMIC
                continue
            stroff = secs[sec[6]][4]                          # sh_link -> .dynstr
            for o in range(sec[4], sec[4] + sec[5], 16):
                t, v = struct.unpack_from("<qQ", data, o)
                if t == 0:
                    break
                if t == tag:
                    end = data.index(b"\0", stroff + v)
                    out.append(data[stroff + v:end].decode())
        return out
    except Exception:
        return []


def _hip_runtimes_mapped():
    """Distinct libamdhip64 files mapped into this process."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({os.path.realpath(l.split()[-1]) for l in f if "libamdhip64" in l and "/" in l})
    except OSError:
        return []


def _one_hip_runtime():
    """One HIP runtime per process.  libstattn.so needs `libamdhip64.so.7`; torch ships its own copy under torch/lib
    and asks for it as `libamdhip64.so`, so whichever of the two libraries is loaded SECOND brings a second HIP + HSA
    runtime when libstattn came first (torch first is fine: its copy carries the soname libstattn asks for).  The
    second runtime cannot open the GPU again, and whatever binds to it -- torch.cuda, torch's RCCL -- finds no device.
    So when torch is installed (not necessarily imported) and no HIP runtime is loaded yet, torch's copy is loaded
    first and libstattn binds to it -- but ONLY when that copy's SONAME is the very name libstattn's DT_NEEDED asks for
    (a torch wheel built against another ROCm major would not satisfy the loader, and preloading it would CREATE the
    two-runtime process this function exists to prevent).  STATTN_NO_HIP_PRELOAD=1 switches the preload off."""
    if os.environ.get("STATTN_NO_HIP_PRELOAD"):
        return
    if _hip_runtimes_mapped():
        return
    needed = [n for n in _elf_dynamic_strings(library_path(), 1) if n.startswith("libamdhip64.so")]
    if not needed:
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand) and needed[0] in _elf_dynamic_strings(cand, 14):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen libstattn.so and declare every prototype.  Loading needs no GPU."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise NativeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C %s/csrc`).  stattn has no CPU fallback." % (path, _HERE))
    _one_hip_runtime()
    lib = C.CDLL(path)
    mapped = _hip_runtimes_mapped()
    if len(mapped) > 1:                    # two HIP + HSA runtimes: whatever binds to the second one finds no device
        import warnings
        warnings.warn("two HIP runtimes are mapped into this process (%s): import torch before stattn, or align the ROCm "
                      "versions; RCCL / torch.cuda may report 'no ROCm-capable device'" % ", ".join(mapped), RuntimeWarning)
    for name, (res, args) in list(_SIGNATURES.items()) + list(_DBG_SIGNATURES.items()):
        fn = getattr(lib, name)            # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(_F)


def _f32(a, name, shape=None):
    """Theano-strict input check: float32, C-contiguous (TypeError otherwise, like theano.function)."""
    if not isinstance(a, np.ndarray) or a.dtype != np.float32:
        raise TypeError("%s must be a float32 numpy array (got %s)" % (name, getattr(a, "dtype", type(a))))
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return np.ascontiguousarray(a)


def _i64(a, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.int64:
        raise TypeError("%s must be an int64 numpy array (got %s)" % (name, getattr(a, "dtype", type(a))))
    return np.ascontiguousarray(a)


OPTION_KEYS = ("dim", "dim_word", "n_words", "ctxg_dim", "ctxl_dim", "ctxm_dim",
               "selector", "use_dropout", "prev2out", "ctx2out")

KERNEL_CLASSES = ("spatial", "hproj", "lt_gemm", "temporal", "lstm", "prologue", "readout", "gemm_nn", "select")


def grad_regions(options):
    """(regions, nflat): the [offset, offset + length) ranges, in floats, of the flat gradient buffer that stattn_backward hands to the
    data-parallel all-reduce, in the order it completes them (csrc/handle.h GRAD_REGIONS).  Computed on the host: needs no GPU."""
    lib = load_library()
    o = _Options()
    for k in OPTION_KEYS:
        setattr(o, k, int(options[k]))
    off = (C.c_size_t * 8)(); ln = (C.c_size_t * 8)(); n = C.c_int(); nflat = C.c_size_t()
    rc = lib.stattn_dbg_grad_regions(C.byref(o), 8, off, ln, C.byref(n), C.byref(nflat))
    if rc != 0:
        raise NativeError("stattn_dbg_grad_regions: %s" % lib.stattn_last_error(None).decode())
    return [(int(off[i]), int(ln[i])) for i in range(n.value)], int(nflat.value)


class Decoder(object):
    """One native decoder instance = one GPU + one stream (stattn_handle)."""

    def __init__(self, options, device=0, stream=None, lt_mode=None, precision=None):
        lib = load_library()
        self._lib = lib
        self._h = _H()
        o = _Options()
        for k in OPTION_KEYS:
            if k not in options:
                raise ValueError("options lacks '%s'" % k)
            setattr(o, k, int(options[k]))
        # reference graph limits: reject what the reference itself cannot run
        if options.get("n_layers_init", 0) != 0:
            raise ValueError("n_layers_init must be 0 (model_attention.py:546-548 references undefined names)")
        if options.get("n_layers_out", 1) != 1:
            raise ValueError("only n_layers_out == 1 is supported (config.py:23)")
        if options.get("encoder", "none") not in ("none", None):
            raise ValueError("encoder must be 'none' (the lstm encoder branches are broken: model_attention.py:626-634)")
        if lt_mode is None:
            lt_mode = int(os.environ.get("STATTN_LT_MODE", options.get("lt_mode", 1)))
        o.lt_mode = int(lt_mode)
        self.lt_mode = int(lt_mode)
        # 'fp32' (default, the parity configuration) or 'bf16' (bf16-MFMA forward/decode path, BASELINE configs[3])
        if precision is None:
            precision = os.environ.get("STATTN_PRECISION", options.get("stattn_precision", "fp32"))
        if precision not in ("fp32", "bf16", "split"):
            raise ValueError("precision must be 'fp32', 'bf16' or 'split' (fp32 results, GEMMs on the bf16 matrix cores)")
        o.precision = {"fp32": 0, "bf16": 1, "split": 2}[precision]
        self.precision = precision
        self.options = dict(options)
        rc = lib.stattn_create(C.byref(o), int(device), C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc != 0:
            msg = lib.stattn_last_error(None).decode()
            self._h = _H()
            if rc == -1:
                raise ValueError(msg)
            raise NativeError("stattn_create failed (%d): %s" % (rc, msg))
        self.D, self.E, self.V = o.dim, o.dim_word, o.n_words
        self.Fl, self.Fm = o.ctxl_dim, o.ctxm_dim
        self._shapes = OrderedDict()
        self._redzone = bool(lib.stattn_dbg_redzone_enabled())
        self.redzone_checks = 0
        for i in range(lib.stattn_param_count(self._h)):
            dims = (C.c_int64 * 2)()
            nd = C.c_int()
            self._chk(lib.stattn_param_shape(self._h, i, C.byref(dims), C.byref(nd)))
            self._shapes[lib.stattn_param_name(self._h, i).decode()] = tuple(dims[j] for j in range(nd.value))
        self._batch = None

    # -- plumbing
    def _chk(self, rc):
        if rc == 0:
            # STATTN_DBG_REDZONE=1 (csrc/handle.h): after EVERY library call, every canary byte around every device buffer is verified
            if getattr(self, "_redzone", False) and self._h.value:
                self.redzone_checks += 1
                REDZONE_CHECKS[0] += 1
                rc = self._lib.stattn_dbg_redzone_check(self._h)
                if rc != 0:
                    raise NativeError("libstattn red zone: %s" % self._lib.stattn_last_error(self._h).decode())
            return
        msg = self._lib.stattn_last_error(self._h).decode()
        if rc == -1:
            raise ValueError(msg)
        raise NativeError("libstattn error %d: %s" % (rc, msg))

    def redzone_buffers(self):
        """Device buffers currently guarded by red zones (0 unless the process started with STATTN_DBG_REDZONE=1)."""
        return int(self._lib.stattn_dbg_redzone_buffers(self._h))

    def redzone_poke(self, name, offset):
        """Test hook: damage one canary byte of the named buffer; the next library call must then raise."""
        rc = self._lib.stattn_dbg_redzone_poke(self._h, name.encode(), int(offset))
        if rc != 0:
            raise NativeError(self._lib.stattn_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.stattn_destroy(self._h)
            self._h = _H()
            for p in getattr(self, "_pinned", []):
                self._lib.stattn_host_free(C.c_void_p(p))
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self._lib.stattn_sync(self._h))

    # -- parameters
    def param_shapes(self):
        return OrderedDict(self._shapes)

    def set_param(self, name, value):
        v = np.asarray(value, dtype=np.float32)
        if name not in self._shapes:
            raise KeyError(name)
        if tuple(v.shape) != self._shapes[name]:
            raise ValueError("%s: shape %s, expected %s" % (name, v.shape, self._shapes[name]))
        v = np.ascontiguousarray(v).reshape(-1)      # (0-d arrays become 1 element)
        self._chk(self._lib.stattn_set_param(self._h, name.encode(), _fp(v), v.size))

    def get_param(self, name):
        out = np.empty(self._shapes[name], np.float32)
        self._chk(self._lib.stattn_get_param(self._h, name.encode(), _fp(out), out.size))
        return out

    def get_grad(self, name):
        out = np.empty(self._shapes[name], np.float32)
        self._chk(self._lib.stattn_get_grad(self._h, name.encode(), _fp(out), out.size))
        return out

    def set_params(self, params):
        for k in self._shapes:
            if k not in params:
                raise KeyError("parameter '%s' missing" % k)
            self.set_param(k, params[k])

    def get_params(self):
        return OrderedDict((k, self.get_param(k)) for k in self._shapes)

    def _dev_buffer(self, fn):
        p = C.c_void_p()
        n = C.c_size_t()
        self._chk(fn(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def param_buffer_dev(self):
        return self._dev_buffer(self._lib.stattn_param_buffer_dev)

    def grad_buffer_dev(self):
        return self._dev_buffer(self._lib.stattn_grad_buffer_dev)

    def set_use_noise(self, v):
        self._chk(self._lib.stattn_set_use_noise(self._h, float(v)))

    def set_seed(self, seed):
        self._chk(self._lib.stattn_set_seed(self._h, int(seed)))

    def set_dropout_masks(self, dp, d1, d2):
        if dp is None:
            self._chk(self._lib.stattn_set_dropout_masks(self._h, None, None, None, 0, 0))
            return
        t, m = dp.shape[0], dp.shape[1]
        dp = _f32(dp, "dp", (t, m, 3 * self.D)); d1 = _f32(d1, "d1", (t, m, self.D)); d2 = _f32(d2, "d2", (t, m, self.E))
        self._chk(self._lib.stattn_set_dropout_masks(self._h, _fp(dp), _fp(d1), _fp(d2), t, m))

    # -- sampler (model_attention.py:791-795, 845-848)
    def f_init(self, ctxg, ctxg_mask):
        ctxg = _f32(ctxg, "ctxg")
        if ctxg.ndim != 2 or ctxg.shape[1] != self.D:
            raise ValueError("ctxg must be (T, %d)" % self.D)
        T = ctxg.shape[0]
        ctxg_mask = _f32(ctxg_mask, "ctxg_mask", (T,))
        h0 = np.empty((self.D,), np.float32); c0 = np.empty((self.D,), np.float32)
        self._chk(self._lib.stattn_f_init(self._h, _fp(ctxg), _fp(ctxg_mask), T, _fp(h0), _fp(c0)))
        return [ctxg, h0, c0]

    def set_video(self, ctxg, ctxl, ctxm):
        """Stage one video for the sampler (stattn_set_video): features -> HBM, projected once.  f_next calls made with
        `resident=True` (or inside video_scope with the same arrays) then skip the per-call upload + F->D projection
        that the reference graph repeats on every call (model_attention.py:782-788)."""
        ctxg = _f32(ctxg, "ctxg"); ctxl = _f32(ctxl, "ctxl"); ctxm = _f32(ctxm, "ctxm")
        if ctxl.ndim != 3 or ctxl.shape[2] != self.Fl:
            raise ValueError("ctxl must be (T, K, %d)" % self.Fl)
        T, K = ctxl.shape[0], ctxl.shape[1]
        if ctxg.shape != (T, self.D) or ctxm.shape != (T, self.Fm):
            raise ValueError("ctxg/ctxm shapes do not match ctxl's T")
        self._chk(self._lib.stattn_set_video(self._h, _fp(ctxg), _fp(ctxl), _fp(ctxm), T, K))

    def video_scope(self, ctxg, ctxl, ctxm):
        """Context manager for a decode loop over ONE video: stages it once, and f_next calls inside the scope that are
        handed these very array objects run on the resident copy.  The scope is explicit and ends with the loop, so
        nothing is ever inferred from pointers or contents: outside a scope every f_next call re-projects like the
        reference."""
        dec = self

        class _Scope(object):
            def __enter__(self_s):
                dec.set_video(ctxg, ctxl, ctxm)
                dec._scope = (ctxg, ctxl, ctxm)
                return dec

            def __exit__(self_s, *exc):
                dec._scope = None
                return False
        return _Scope()

    def f_next(self, x, ctxg, ctxg_mask, ctxl, ctxl_mask, ctxm, ctxm_mask, h, c, extras=False, resident=False):
        x = _i64(x, "x")
        if x.ndim != 1:
            raise ValueError("x must be a vector")
        m = x.shape[0]
        sc = getattr(self, "_scope", None)
        if sc is not None and ctxg is sc[0] and ctxl is sc[1] and ctxm is sc[2]:
            resident = True
        ctxg = _f32(ctxg, "ctxg"); ctxl = _f32(ctxl, "ctxl"); ctxm = _f32(ctxm, "ctxm")
        if ctxl.ndim != 3 or ctxl.shape[2] != self.Fl:
            raise ValueError("ctxl must be (T, K, %d)" % self.Fl)
        T, K = ctxl.shape[0], ctxl.shape[1]
        if ctxg.shape != (T, self.D) or ctxm.shape != (T, self.Fm):
            raise ValueError("ctxg/ctxm shapes do not match ctxl's T")
        h = _f32(h, "init_state", (m, self.D)); c = _f32(c, "init_memory", (m, self.D))
        probs = np.empty((m, self.V), np.float32); sample = np.empty((m,), np.int64)
        ho = np.empty((m, self.D), np.float32); co = np.empty((m, self.D), np.float32)
        al = ag = am = alt = lg = None
        if extras:
            al = np.empty((m, T, K), np.float32); ag = np.empty((m, T), np.float32)
            am = np.empty((m, T), np.float32); alt = np.empty((m, T), np.float32)
            lg = np.empty((m, self.V), np.float32)
        pg, pl, pm = (None, None, None) if resident else (_fp(ctxg), _fp(ctxl), _fp(ctxm))
        self._chk(self._lib.stattn_f_next(
            self._h, x.ctypes.data_as(_I64), m, pg, None, pl, None, pm, None, T, K,
            _fp(h), _fp(c), _fp(probs), sample.ctypes.data_as(_I64), _fp(ho), _fp(co),
            _fp(al), _fp(ag), _fp(am), _fp(alt), _fp(lg)))
        out = [probs, sample, ho, co]
        if extras:
            return out, dict(alphal=al, alphag=ag, alpham=am, alphalt=alt, logit=lg)
        return out

    def _beam_shapes(self, ctxg, ctxg_mask, ctxl, ctxm):
        ctxl = _f32(ctxl, "ctxl")
        if ctxl.ndim != 4 or ctxl.shape[3] != self.Fl:
            raise ValueError("ctxl must be (nvid, T, K, %d)" % self.Fl)
        nvid, T, K = ctxl.shape[0], ctxl.shape[1], ctxl.shape[2]
        ctxg = _f32(ctxg, "ctxg", (nvid, T, self.D)); ctxg_mask = _f32(ctxg_mask, "ctxg_mask", (nvid, T))
        ctxm = _f32(ctxm, "ctxm", (nvid, T, self.Fm))
        return ctxg, ctxg_mask, ctxl, ctxm, nvid, T, K

    def beam_stage(self, ctxg, ctxg_mask, ctxl, ctxm):
        """Stage a batch of videos for beam_search(resident=True): inputs resident in HBM (stattn_beam_stage)."""
        ctxg, ctxg_mask, ctxl, ctxm, nvid, T, K = self._beam_shapes(ctxg, ctxg_mask, ctxl, ctxm)
        self._chk(self._lib.stattn_beam_stage(self._h, nvid, _fp(ctxg), _fp(ctxg_mask), _fp(ctxl), _fp(ctxm), T, K))
        self._staged = (nvid, T, K)

    def beam_search(self, ctxg=None, ctxg_mask=None, ctxl=None, ctxm=None, k=5, maxlen=30, suppress_eos=False, resident=False):
        """gen_sample for a batch of videos, device-side (stattn_beam_search).  Returns a list with one
        (samples, scores) pair per video, ordered like gen_sample's return value.  k = 1 is the greedy mode.
        resident=True decodes the videos staged by beam_stage() (no host arrays, nothing re-uploaded)."""
        if resident:
            if getattr(self, "_staged", None) is None:
                raise ValueError("beam_search(resident=True) needs beam_stage() first")
            nvid, T, K = self._staged
            pg = pk = pl = pm = None
        else:
            ctxg, ctxg_mask, ctxl, ctxm, nvid, T, K = self._beam_shapes(ctxg, ctxg_mask, ctxl, ctxm)
            pg, pk, pl, pm = _fp(ctxg), _fp(ctxg_mask), _fp(ctxl), _fp(ctxm)
            self._staged = (nvid, T, K)
        tok = np.empty((nvid, k, maxlen), np.int64); sc = np.empty((nvid, k), np.float32)
        ln = np.empty((nvid, k), np.int32); cnt = np.empty((nvid,), np.int32)
        self._chk(self._lib.stattn_beam_search(self._h, nvid, pg, pk, pl, pm, T, K, int(k),
                                               int(maxlen), int(bool(suppress_eos)), tok.ctypes.data_as(_I64), _fp(sc),
                                               ln.ctypes.data_as(C.POINTER(C.c_int32)), cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        self._beam_shape = (nvid, int(k))
        out = []
        for v in range(nvid):
            samples = [tok[v, j, :ln[v, j]].tolist() for j in range(cnt[v])]
            out.append((samples, sc[v, :cnt[v]].copy()))
        return out

    def sample_search(self, ctxg=None, ctxg_mask=None, ctxl=None, ctxm=None, maxlen=30, resident=False):
        """gen_sample(stochastic=True) for a batch of at most 16 videos, device-side (stattn_sample_search): one
        (word list, score) pair per video -- the words include the closing <eos> when one was drawn, the score is the
        sum of the drawn words' probabilities (model_attention.py:913-918)."""
        if resident:
            if getattr(self, "_staged", None) is None:
                raise ValueError("sample_search(resident=True) needs beam_stage() first")
            nvid, T, K = self._staged
            pg = pk = pl = pm = None
        else:
            ctxg, ctxg_mask, ctxl, ctxm, nvid, T, K = self._beam_shapes(ctxg, ctxg_mask, ctxl, ctxm)
            pg, pk, pl, pm = _fp(ctxg), _fp(ctxg_mask), _fp(ctxl), _fp(ctxm)
            self._staged = (nvid, T, K)
        tok = np.empty((nvid, maxlen), np.int64); sc = np.empty((nvid,), np.float32); ln = np.empty((nvid,), np.int32)
        self._chk(self._lib.stattn_sample_search(self._h, nvid, pg, pk, pl, pm, T, K, int(maxlen), tok.ctypes.data_as(_I64),
                                                 _fp(sc), ln.ctypes.data_as(C.POINTER(C.c_int32))))
        self._beam_shape = (nvid, 1)
        return [(tok[v, :ln[v]].tolist(), float(sc[v])) for v in range(nvid)]

    def beam_final_state(self):
        """(next_state, next_memory) of gen_sample for every video of the last beam_search: a list of
        ((rows, D) h, (rows, D) c) pairs (stattn_beam_final_state)."""
        nvid, k = self._beam_shape
        hh = np.empty((nvid, k, self.D), np.float32); cc = np.empty((nvid, k, self.D), np.float32)
        rows = np.empty((nvid,), np.int32)
        self._chk(self._lib.stattn_beam_final_state(self._h, _fp(hh), _fp(cc), rows.ctypes.data_as(C.POINTER(C.c_int32))))
        return [(hh[v, :rows[v]].copy(), cc[v, :rows[v]].copy()) for v in range(nvid)]

    # -- training graph
    def set_batch(self, x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm):
        x = _i64(x, "x")
        if x.ndim != 2:
            raise ValueError("x must be (t, m)")
        t, m = x.shape
        mask = _f32(mask, "mask", (t, m))
        ctxl = _f32(ctxl, "ctxl")
        if ctxl.ndim != 4 or ctxl.shape[0] != m or ctxl.shape[3] != self.Fl:
            raise ValueError("ctxl must be (m, T, K, %d)" % self.Fl)
        T, K = ctxl.shape[1], ctxl.shape[2]
        ctxg = _f32(ctxg, "ctxg", (m, T, self.D)); mask_ctxg = _f32(mask_ctxg, "mask_ctxg", (m, T))
        ctxm = _f32(ctxm, "ctxm", (m, T, self.Fm))
        self._chk(self._lib.stattn_set_batch(self._h, x.ctypes.data_as(_I64), _fp(mask), t, m, _fp(ctxg), _fp(mask_ctxg),
                                             _fp(ctxl), None, _fp(ctxm), None, T, K))
        self._batch = (t, m, T, K)

    def _check_batch(self, x, mask, ctxg, mask_ctxg, ctxl, ctxm):
        x = _i64(x, "x")
        if x.ndim != 2:
            raise ValueError("x must be (t, m)")
        t, m = x.shape
        mask = _f32(mask, "mask", (t, m))
        ctxl = _f32(ctxl, "ctxl")
        if ctxl.ndim != 4 or ctxl.shape[0] != m or ctxl.shape[3] != self.Fl:
            raise ValueError("ctxl must be (m, T, K, %d)" % self.Fl)
        T, K = ctxl.shape[1], ctxl.shape[2]
        ctxg = _f32(ctxg, "ctxg", (m, T, self.D)); mask_ctxg = _f32(mask_ctxg, "mask_ctxg", (m, T))
        ctxm = _f32(ctxm, "ctxm", (m, T, self.Fm))
        return x, mask, ctxg, mask_ctxg, ctxl, ctxm, t, m, T, K

    def prefetch_batch(self, x, mask, ctxg, mask_ctxg, ctxl, mask_ctxl, ctxm, mask_ctxm):
        """Start copying the NEXT minibatch to the shadow buffer set (asynchronous for pinned arrays, see
        pinned_empty); the arrays must stay alive and unmodified until swap_batch()."""
        x, mask, ctxg, mask_ctxg, ctxl, ctxm, t, m, T, K = self._check_batch(x, mask, ctxg, mask_ctxg, ctxl, ctxm)
        self._pending_refs = (x, mask, ctxg, mask_ctxg, ctxl, ctxm)
        self._chk(self._lib.stattn_prefetch_batch(self._h, x.ctypes.data_as(_I64), _fp(mask), t, m, _fp(ctxg), _fp(mask_ctxg),
                                                  _fp(ctxl), None, _fp(ctxm), None, T, K))
        self._pending = (t, m, T, K)

    def swap_batch(self):
        self._chk(self._lib.stattn_swap_batch(self._h))
        self._batch = self._pending
        self._live_refs = getattr(self, "_pending_refs", None)

    def pinned_empty(self, shape, dtype=np.float32):
        """numpy array over page-locked host memory (hipHostMalloc): H2D copies from it are truly asynchronous."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        rc = self._lib.stattn_host_alloc(nbytes, C.byref(p))
        if rc != 0:
            raise NativeError("stattn_host_alloc failed: %s" % self._lib.stattn_last_error(None).decode())
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)           # freed in close()
        return arr

    def forward_train(self):
        self._chk(self._lib.stattn_forward_train(self._h))

    def get_forward(self, probs=True, alphas=True, logits=False):
        t, m, T, K = self._batch
        cost = np.empty((m,), np.float32)
        pr = np.empty((t * m, self.V), np.float32) if probs else None
        lg = np.empty((t * m, self.V), np.float32) if logits else None
        al = np.empty((t, m, T, K), np.float32) if alphas else None
        ag = np.empty((t, m, T), np.float32) if alphas else None
        am = np.empty((t, m, T), np.float32) if alphas else None
        alt = np.empty((t, m, T), np.float32) if alphas else None
        self._chk(self._lib.stattn_get_forward(self._h, _fp(cost), _fp(pr), _fp(al), _fp(ag), _fp(am), _fp(alt), _fp(lg)))
        return dict(cost=cost, probs=pr, alphal=al, alphag=ag, alpham=am, alphalt=alt, logit=lg)

    def get_states(self):
        t, m, T, K = self._batch
        hs = np.empty((t, m, self.D), np.float32); cs = np.empty((t, m, self.D), np.float32)
        ctx = np.empty((t, m, self.D), np.float32)
        self._chk(self._lib.stattn_get_states(self._h, _fp(hs), _fp(cs), _fp(ctx)))
        return dict(h=hs, c=cs, ctx=ctx)

    # -- gradients and optimizer (f_grad_shared / f_update)
    def backward(self, nll_scale=None, alpha_c=0.0):
        if nll_scale is None:
            nll_scale = 1.0 / self._batch[1]          # cost.mean()  (model_attention.py:1129)
        self._chk(self._lib.stattn_backward(self._h, float(nll_scale), float(alpha_c)))
        self._nll_scale = float(nll_scale)

    def get_loss(self, decay_c=0.0):
        v = C.c_float()
        self._chk(self._lib.stattn_get_loss(self._h, self._nll_scale, float(decay_c), C.byref(v)))
        return v.value

    def get_grads(self):
        return OrderedDict((k, self.get_grad(k)) for k in self._shapes)

    def update(self, decay_c=0.0, clip_c=0.0):
        self._chk(self._lib.stattn_update(self._h, float(decay_c), float(clip_c)))

    # -- data parallel (in-library RCCL; stattn.dp drives the rendezvous)
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        """Rendezvous token (bytes) made by rank 0 and handed to every rank's comm_init."""
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        rc = self._lib.stattn_comm_unique_id(buf)
        if rc != 0:
            raise NativeError("stattn_comm_unique_id failed: %s" % self._lib.stattn_last_error(None).decode())
        return buf.raw

    def comm_init(self, rank, nranks, token):
        if len(token) != self.COMM_ID_BYTES:
            raise ValueError("token must be %d bytes" % self.COMM_ID_BYTES)
        buf = C.create_string_buffer(bytes(token), self.COMM_ID_BYTES)
        self._chk(self._lib.stattn_comm_init(self._h, int(rank), int(nranks), buf))

    def comm_destroy(self):
        self._chk(self._lib.stattn_comm_destroy(self._h))

    def comm_info(self):
        r = C.c_int(); n = C.c_int()
        self._chk(self._lib.stattn_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def comm_stats(self):
        """{'ranks', 'overlap', 'regions', 'exposed_ms'} of the last backward / all-reduce pair: ranks of the RCCL
        communicator (0: none), overlap mode, regions of the gradient buffer that were summed on the side stream while
        backward still ran, and the time the compute stream spent inside stattn_allreduce_grads (HIP events)."""
        n = C.c_int(); ov = C.c_int(); rg = C.c_int(); ms = C.c_float()
        self._chk(self._lib.stattn_comm_stats(self._h, C.byref(n), C.byref(ov), C.byref(rg), C.byref(ms)))
        return dict(ranks=n.value, overlap=ov.value, regions=rg.value, exposed_ms=float(ms.value))

    def comm_library_path(self):
        return self._lib.stattn_comm_library_path().decode()

    def comm_set_overlap(self, mode):
        """0 = one all-reduce after backward, 1 = regions reduced while backward runs (default), 2 = same, forced
        even in a one-rank communicator (test hook)."""
        self._chk(self._lib.stattn_comm_set_overlap(self._h, int(mode)))

    def allreduce_grads(self):
        self._chk(self._lib.stattn_allreduce_grads(self._h))

    def broadcast_params(self, root=0):
        self._chk(self._lib.stattn_broadcast_params(self._h, int(root)))

    def allreduce_scalars(self, vals):
        v = np.ascontiguousarray(np.asarray(vals, dtype=np.float32).reshape(-1))
        self._chk(self._lib.stattn_allreduce_scalars(self._h, _fp(v), v.size))
        return v

    def reset_optimizer(self):
        self._chk(self._lib.stattn_reset_optimizer(self._h))

    # -- kernel-level entry points
    def gemm(self, A, B, bias=None, add=None, act=0, alpha=1.0, kind=0, transA=False, transB=False):
        A = _f32(A, "A"); B = _f32(B, "B")
        K, M = (A.shape if transA else A.shape[::-1])
        N = B.shape[0] if transB else B.shape[1]
        if (B.shape[1] if transB else B.shape[0]) != K:
            raise ValueError("inner dimensions differ")
        out = np.empty((M, N), np.float32)
        bias = None if bias is None else _f32(bias, "bias", (N,))
        add = None if add is None else _f32(add, "add", (M, N))
        self._chk(self._lib.stattn_dbg_gemm(self._h, int(kind), int(transA), int(transB), M, N, K, float(alpha),
                                            _fp(A), _fp(B), _fp(bias), _fp(add), int(act), _fp(out)))
        return out

    def time_gemm(self, M, N, K, iters=20, transA=False, transB=False):
        ms = C.c_float()
        self._chk(self._lib.stattn_dbg_time_gemm(self._h, int(transA), int(transB), M, N, K, iters, C.byref(ms)))
        return ms.value

    def beam_graph_replays(self):
        """hipGraph replays (two words each) in the last beam_search; 0 = the kernels were launched eagerly."""
        return int(self._lib.stattn_dbg_counter(self._h, 0))

    def path_counts(self):
        """Which kernels the decoder steps ran on: steps of the last forward_train (or f_next calls since) whose
        attention launch carried the h.U rider / that used the row-panel kernels, and the same for the reverse steps of
        the last backward.  Tests assert with it that a shape exercises the path it is meant to."""
        names = ("fwd_rider", "fwd_panel", "bwd_rider", "bwd_panel", "upd_rider", "upd_rowwg")
        return {n: int(self._lib.stattn_dbg_counter(self._h, i + 1)) for i, n in enumerate(names)}

    def beam_vocab_stats_words(self):
        """Words of the last beam search whose vocabulary launch ended in the statistics epilogue (no logits, softmax or top-k launch)."""
        return int(self._lib.stattn_dbg_counter(self._h, 7))

    def time_gemm_bf16(self, M, N, K, tile=0, iters=20):
        ms = C.c_float()
        self._chk(self._lib.stattn_dbg_time_gemm_bf16(self._h, M, N, K, int(tile), iters, C.byref(ms)))
        return ms.value

    def time_skinny(self, M, N, K, nseg=1, variant=0, iters=50):
        ms = C.c_float()
        self._chk(self._lib.stattn_dbg_time_skinny(self._h, M, N, K, nseg, variant, iters, C.byref(ms)))
        return ms.value

    def set_profiling(self, on):
        self._chk(self._lib.stattn_set_profiling(self._h, int(bool(on))))

    def kernel_ms(self):
        out = OrderedDict()
        for i, name in enumerate(KERNEL_CLASSES):
            ms = C.c_float(); n = C.c_int()
            self._chk(self._lib.stattn_get_kernel_ms(self._h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def gemm_launch_ms(self):
        """Average duration of each of the first 16 plain GEMM launches of a forward pass, in launch order."""
        out = []
        for i in range(16):
            ms = C.c_float(); n = C.c_int()
            self._chk(self._lib.stattn_get_kernel_ms(self._h, len(KERNEL_CLASSES) + i, C.byref(ms), C.byref(n)))
            if n.value:
                out.append(ms.value)
        return out

    BWD_KERNELS = ("lstm_bwd", "panel_dctx_dhU", "temporal_bwd", "spatial_bwd", "reduce_T", "panel_dhW", "ctxgrad")

    def bwd_gemm_launch_ms(self):
        """Average duration of every LDS-tiled GEMM launch of a backward pass (profiling on), in launch order."""
        out = []
        for i in range(24):
            ms = C.c_float(); n = C.c_int()
            self._chk(self._lib.stattn_get_kernel_ms(self._h, len(KERNEL_CLASSES) + 16 + i, C.byref(ms), C.byref(n)))
            if n.value:
                out.append(ms.value)
        return out

    def bwd_kernel_ms(self):
        """(avg ms, launches) of the kernels of a reverse-scan step and of the deferred context-gradient kernel."""
        out = OrderedDict()
        for i, name in enumerate(self.BWD_KERNELS):
            ms = C.c_float(); n = C.c_int()
            self._chk(self._lib.stattn_get_kernel_ms(self._h, len(KERNEL_CLASSES) + 16 + 24 + i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out
