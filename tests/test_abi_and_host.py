"""CPU-side tests: the C-ABI library loads and exports every symbol include/stattn.h declares,
fails loudly without a GPU, and the host-side mirror of the reference surface behaves."""
import os
import re

import numpy as np
import pytest

import stattn
from stattn import _native, common


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(path=("include", "stattn.h")):
    src = open(os.path.join(ROOT, *path)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(stattn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()
    declared = _header_symbols()
    assert len(declared) >= 20
    for s in declared:
        assert hasattr(lib, s), "libstattn.so lacks %s" % s
    assert sorted(_native.EXPORTED_SYMBOLS) == declared      # the ctypes stub binds exactly the header
    assert not [s for s in declared if "_dbg_" in s]         # development entry points live in csrc/stattn_dbg.h
    dbg = _header_symbols(("video-description-with-spatial-temporal-attention_amd", "csrc", "stattn_dbg.h"))
    assert sorted(_native.DEBUG_SYMBOLS) == dbg and all(hasattr(lib, s) for s in dbg)
    assert b"gfx950" in lib.stattn_version()


def test_create_fails_loudly_without_gpu_or_with_bad_options():
    import torch
    opt = dict(dim=128, dim_word=64, n_words=50, ctxg_dim=128, ctxl_dim=64, ctxm_dim=64, selector=True,
               use_dropout=True, prev2out=True, ctx2out=True)
    if not torch.cuda.is_available():
        with pytest.raises(_native.NativeError, match="no CPU fallback|HIP"):
            stattn.Decoder(opt)
    with pytest.raises(ValueError):
        stattn.Decoder(dict(opt, dim=100))                   # not a multiple of 64
    with pytest.raises(ValueError):
        stattn.Decoder(dict(opt, ctxg_dim=256))              # reference graph needs ctxg_dim == dim
    with pytest.raises(ValueError):
        stattn.Decoder(dict(opt, use_dropout=False))         # reference's False branch is broken
    with pytest.raises(ValueError):
        stattn.Decoder(dict(opt, n_layers_init=1))
    with pytest.raises(ValueError):
        stattn.Decoder(dict(opt, encoder='lstm_bi'))


def test_init_params_follows_reference_order_shapes_and_init_rules():
    from oracle import stattn_oracle as O
    opt = O.default_options(dim=64, dim_word=32, n_words=40, ctxg_dim=64, ctxl_dim=48, ctxm_dim=16, ctxglm_dim=64)
    common.reset_rngs(1234)
    p = stattn.Attention().init_params(opt)
    shp = O.param_shapes(opt)
    assert list(p) == list(shp)
    for k in shp:
        assert np.shape(p[k]) == shp[k], k
        assert np.asarray(p[k]).dtype == np.float32
    # square norm_weight(ortho=True) -> orthogonal (common.py:130-131); *_att context weights gaussian
    for k in ('ff_state_W', 'decoder_Wdg_att', 'decoder_Wdl_att'):
        np.testing.assert_allclose(p[k] @ p[k].T, np.eye(64), atol=1e-5)
    for blk in range(4):
        u = p['decoder_U'][:, blk * 64:(blk + 1) * 64]
        np.testing.assert_allclose(u @ u.T, np.eye(64), atol=1e-5)
    assert abs(p['decoder_Wcg_att'].std() - 0.01) < 0.002
    assert not p['decoder_b'].any() and p['decoder_b_sel'].shape == ()
    # deterministic in the module-global RandomState(1234) (common.py:25)
    common.reset_rngs(1234)
    q = stattn.Attention().init_params(opt)
    for k in p:
        np.testing.assert_array_equal(p[k], q[k])
    with pytest.raises(ValueError):
        stattn.Attention().init_params(dict(opt, encoder='lstm_uni'))


def test_gen_sample_host_logic_against_oracle_driver_with_a_fake_step():
    """The beam-search bookkeeping is host code: drive both restatements with the same toy f_next."""
    from oracle import stattn_oracle as O
    V, D = 9, 4
    rng = np.random.RandomState(0)
    table = rng.dirichlet(np.ones(V) * 0.5, size=V + 1).astype(np.float32)

    def f_init(g, m):
        return [g, np.zeros(D, np.float32), np.zeros(D, np.float32)]

    def f_next(x, g, gm, l, lm, mo, mm, h, c):
        p = table[x + 1]
        return [p, p.argmax(1).astype(np.int64), h + 1, c - 1]
    a = (np.zeros((3, D), np.float32), np.ones(3, np.float32), None, None, None, None)
    model = stattn.Attention()
    for k in (1, 3, 5):
        s, sc, hs, cs = model.gen_sample(None, f_init, f_next, *a, {}, None, k, maxlen=7)
        sr, scr, hr, cr = O.gen_sample(f_init, f_next, *a, k=k, maxlen=7)
        assert s == sr
        np.testing.assert_allclose(np.asarray(sc), np.asarray(scr))
        # one-element lists like the reference's (n_layers_lstm = 1, model_attention.py:980-994)
        assert isinstance(hs, list) and isinstance(cs, list) and len(hs) == 1 and len(cs) == 1
        np.testing.assert_array_equal(hs[0], hr[0])
    s, sc, _, _ = model.gen_sample(None, f_init, f_next, *a, {}, None, 1, maxlen=5, stochastic=True)
    assert len(s) <= 5
    with pytest.raises(AssertionError):
        model.gen_sample(None, f_init, f_next, *a, {}, None, 3, maxlen=5, stochastic=True)


def test_shared_var_and_zipp_unzip_unbound():
    sv = common.SharedVar(np.ones((2, 3)), 'w')
    assert sv.get_value().dtype == np.float32
    tp = {'w': sv}
    common.zipp({'w': np.zeros((2, 3))}, tp)
    assert not common.unzip(tp)['w'].any()
    with pytest.raises(ValueError):
        sv.set_value(np.zeros((3, 2)))


def test_one_hip_runtime_whichever_of_libstattn_and_torch_is_loaded_first():
    """libstattn.so needs libamdhip64.so.7, torch's libraries ask for their bundled copy as libamdhip64.so: loaded in the
    order (libstattn, torch) a process used to end up with TWO HIP + HSA runtimes, and torch's RCCL -- bound to the
    uninitialised one -- failed ncclCommInitRank with "no ROCm-capable device" (found by the two-rank run of
    tests/test_gpu_dp2.py).  _native.load_library() loads torch's copy first when torch is installed.  Needs no GPU."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import stattn\n"
        "from stattn import _native\n"
        "_native.load_library()\n"
        "try:\n"
        "    import torch\n"
        "except ImportError:\n"
        "    torch = None\n"
        "hip = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "print(len(hip), hip)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for order in ("stattn_first", "torch_first"):
        src = code if order == "stattn_first" else "import torch\n" + code
        out = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.split()[0] == "1", (order, out.stdout)


# ---- the product build's surface: runtime switches and kernels ---------------------------------
CSRC = os.path.join(ROOT, "video-description-with-spatial-temporal-attention_amd", "csrc")


def _product_switches():
    src = open(os.path.join(CSRC, "switches.h")).read()
    body = src[src.index("#define STATTN_PRODUCT_SWITCHES(X)"):src.index("inline bool sw_is_product")]
    return re.findall(r"X\((STATTN_[A-Z0-9_]+)\)", body)


def test_product_library_reads_at_most_ten_environment_switches():
    listed = _product_switches()
    assert 0 < len(listed) <= 10 and len(set(listed)) == len(listed), listed
    used_product, used_tool = set(), set()
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".cpp", ".h")) or f == "switches.h":
            continue
        src = open(os.path.join(CSRC, f)).read()
        code = re.sub(r"//[^\n]*", "", src)
        # no raw getenv in product sources: every variable goes through sw_product / sw_tool (csrc/experimental/ is not compiled in)
        assert not re.search(r"(?<![\w:])getenv\s*\(", code), "%s reads the environment directly" % f
        used_product |= set(re.findall(r'sw_product\("(STATTN_[A-Z0-9_]+)"\)', code))
        used_tool |= set(re.findall(r'sw_tool\("(STATTN_[A-Z0-9_]+)"\)', code))
    assert used_product <= set(listed), used_product - set(listed)       # a name off the list would be a dead switch
    assert set(listed) - used_product == set(), set(listed) - used_product   # and the list names nothing the library does not read
    assert not (used_tool & set(listed))
    # the product Makefile does not define STATTN_PROBES / STATTN_EXPERIMENTAL unless asked on the command line
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-DSTATTN_EXPERIMENTAL" not in mk and "$(if $(PROBES),-DSTATTN_PROBES)" in mk


def test_product_library_contains_no_experimental_kernel():
    """csrc/experimental/*.inl (kernels that have never been through the GPU parity suite) are compiled only under
    -DSTATTN_EXPERIMENTAL=1; the shipped libstattn.so must not carry their host stubs or device symbols."""
    names = set()
    for f in os.listdir(os.path.join(CSRC, "experimental")):
        names |= set(re.findall(r"__global__[^\n]*?void\s+(\w+)\s*\(", open(os.path.join(CSRC, "experimental", f)).read()))
    assert names, "no experimental kernels found: delete csrc/experimental and this test together"
    blob = open(_native.library_path(), "rb").read()
    for n in names:
        assert n.encode() not in blob, "libstattn.so contains the experimental kernel %s" % n
    assert b"spatial2_kernel" in blob and b"lstm_panel_kernel" in blob          # (the scan does see kernel names)
