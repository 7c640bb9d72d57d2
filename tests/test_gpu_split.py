"""fp32 GEMM on the bf16 matrix cores (csrc/gemm_split.hip, precision='split').  Every fp32 operand is the exact sum of
three bf16 terms; six of the nine term products are accumulated in fp32.  The claim to check: against float64 the result
is as close as the fp32-MFMA kernel's -- this is not a reduced-precision path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMALL = dict(dim=128, dim_word=64, n_words=211, ctxg_dim=128, ctxl_dim=96, ctxm_dim=64, ctxglm_dim=128)


@pytest.fixture(scope="module")
def dec():
    import stattn
    from oracle import stattn_oracle as O
    return stattn.Decoder(O.default_options(**SMALL))


def _operands(rng, M, N, K, tA, tB, wide):
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    if wide:            # magnitudes over 12 decades inside one dot product: the split must be exact at every exponent
        A *= np.exp(rng.uniform(-14, 14, A.shape)).astype(np.float32)
        B *= np.exp(rng.uniform(-14, 14, B.shape)).astype(np.float32)
    return A, B


SHAPES = [(128, 128, 16, 0, 0), (64, 128, 64, 0, 0), (200, 256, 96, 0, 0), (333, 128, 1000, 0, 1), (130, 256, 72, 0, 1),
          (1000, 1024, 520, 0, 0), (512, 256, 192, 1, 0), (260, 128, 200, 1, 0), (1024, 1024, 512, 0, 1),
          (256, 128, 4096, 0, 0), (76, 384, 36, 1, 0), (1024, 1024, 4096, 1, 0), (5, 128, 4, 0, 0),
          # >= 384 tiles of 128 x 128: the large-tile kernel (the shapes above run the 64 x 64 one)
          (2048, 3072, 96, 0, 0), (2048, 3072, 80, 1, 0), (2100, 3072, 72, 0, 1), (2176, 3072, 64, 1, 0), (3072, 2048, 40, 0, 0)]


@pytest.mark.parametrize("M,N,K,tA,tB", SHAPES)
@pytest.mark.parametrize("wide", [False, True])
def test_split_gemm_is_as_accurate_as_the_fp32_mfma_kernel(dec, M, N, K, tA, tB, wide):
    rng = np.random.RandomState(M + 3 * N + 7 * K + tA + 2 * tB)
    A, B = _operands(rng, M, N, K, tA, tB, wide)
    ref = (A.T if tA else A).astype(np.float64) @ (B.T if tB else B).astype(np.float64)
    mag = np.abs(A.T if tA else A).astype(np.float64) @ np.abs(B.T if tB else B).astype(np.float64)   # sum |a||b|
    got = dec.gemm(A, B, kind=4, transA=bool(tA), transB=bool(tB))
    f32 = dec.gemm(A, B, kind=0, transA=bool(tA), transB=bool(tB))
    e_split = np.abs(got - ref) / mag
    e_f32 = np.abs(f32 - ref) / mag
    # forward error bound of an fp32 dot product: a small multiple of 2^-24 relative to sum |a||b|
    assert e_split.max() < 4 * 2.0 ** -24 * max(1.0, np.log2(K)), e_split.max()
    # and not worse than the fp32 pipe on the same operands (1.5x slack on the maximum, 1.25x on the mean)
    assert e_split.max() <= 1.5 * e_f32.max() + 2.0 ** -26, (e_split.max(), e_f32.max())
    assert e_split.mean() <= 1.25 * e_f32.mean() + 2.0 ** -28, (e_split.mean(), e_f32.mean())


def test_split_gemm_epilogue_and_exact_integers(dec):
    rng = np.random.RandomState(5)
    M, N, K = 192, 256, 160
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32)
    ref = np.tanh(0.1 * (A.astype(np.float64) @ B.astype(np.float64)) + bias + add)
    got = dec.gemm(A, B, bias=bias, add=add, act=1, alpha=0.1, kind=4)
    np.testing.assert_allclose(got, ref, atol=2e-6, rtol=0)
    # 24-bit integers: all three terms of the split are needed, and then every product and sum is exact
    Ai = rng.randint(-2 ** 11, 2 ** 11, (96, 32)).astype(np.float32) * 4097.0           # 12 + 12 significant bits
    Bi = np.zeros((32, 128), np.float32); Bi[np.arange(32), np.arange(32) * 3] = 1.0; Bi[:, 127] = 1.0
    out = dec.gemm(Ai, Bi, kind=4)
    np.testing.assert_array_equal(out[:, np.arange(32) * 3][:, 1:], Ai[:, 1:])           # selector columns: exact copies
    with pytest.raises(ValueError):
        dec.gemm(np.ones((8, 16), np.float32), np.ones((16, 64), np.float32), kind=4)    # N % 128 != 0


def _pair(dims, seed, **kw):
    import stattn
    from oracle import stattn_oracle as O
    opt = O.default_options(**{**dims, **kw})
    P = O.random_params(opt, seed=seed, dtype=np.float32)
    dec = stattn.Decoder(opt, precision="split", lt_mode=kw.get("lt_mode", 1))
    dec.set_params(P)
    return O, opt, P, O.cast_params(P, np.float64), dec


MEDIUM = dict(dim=256, dim_word=128, n_words=1024, ctxg_dim=256, ctxl_dim=512, ctxm_dim=256, ctxglm_dim=256)


@pytest.mark.parametrize("dims,B,T,K,t,kw", [(SMALL, 5, 5, 4, 6, {}), (MEDIUM, 9, 26, 8, 7, {}), (MEDIUM, 7, 6, 3, 5, dict(lt_mode=0))])
def test_split_handle_meets_the_fp32_parity_bar(dims, B, T, K, t, kw):
    """precision='split' against the float64 oracle at the fp32 tolerance of the parity configuration (1e-4)."""
    O, opt, P, P64, dec = _pair(dims, 6, **kw)
    batch = O.synthetic_batch(opt, B=B, T=T, K=K, t=t, seed=31)
    dec.set_batch(**batch)
    dec.forward_train()
    out = dec.get_forward(logits=True)
    ref = O.build_model_forward(P64, opt, **{k: (v if v.dtype == np.int64 else v.astype(np.float64)) for k, v in batch.items()})
    for name in ('alphal', 'alphag', 'alpham', 'alphalt'):
        assert np.abs(out[name] - ref[name]).max() < 1e-4, name
    assert np.abs(out['logit'] - ref['logit'].reshape(out['logit'].shape)).max() < 1e-4
    np.testing.assert_allclose(out['cost'], ref['cost'], rtol=1e-4, atol=1e-4)


def test_split_handle_gradients_match_autograd_oracle():
    from oracle import stattn_oracle_grad as OG
    O, opt, P, P64, dec = _pair(SMALL, 3)
    batch = O.synthetic_batch(opt, B=6, T=5, K=4, t=6, seed=11)
    dec.set_batch(**batch)
    dec.forward_train()
    dec.backward(alpha_c=0.70602)
    got = dec.get_grads()
    ref = OG.loss_and_grads(P, opt, batch, alpha_c=0.70602)
    for k in got:
        scale = np.abs(np.asarray(ref['grads'][k])).max()
        assert np.abs(got[k] - ref['grads'][k]).max() <= 1e-4 * scale + 5e-6, k


def test_split_and_fp32_handles_agree_at_configs1_size():
    """BASELINE.json configs[1] at full size (batch 64, T=26, K=8, feat 4096, hidden 1024, vocab 12k, 30 steps), same weights
    and minibatch on a precision='fp32' and a precision='split' handle: forward quantities agree far inside the 1e-4 bar,
    every gradient to 5e-5 of its scale (half the bar against the float64 oracle; bias gradients are sums over 13 312 rows), and five optimisation steps follow the same loss trajectory."""
    import stattn
    import bench
    c = bench.CONFIGS["c2"]
    options = bench.make_options(c)
    decs = [stattn.Decoder(options, precision=p) for p in ("fp32", "split")]
    params = bench.fast_params(decs[0].param_shapes(), 1234)
    batch = bench.synthetic_batch(c, 77)
    outs, grads, losses = [], [], []
    for d in decs:
        d.set_params(params)
        d.set_batch(**batch)
        d.set_use_noise(0.0)
        d.forward_train()
        outs.append(d.get_forward(logits=True))
        d.backward(alpha_c=0.70602)
        grads.append(d.get_grads())
        tr = []
        for _ in range(5):
            d.forward_train(); d.backward(alpha_c=0.70602)
            tr.append(d.get_loss(1e-4))
            d.update(decay_c=1e-4, clip_c=10.0)
        losses.append(tr)
    a, b = outs
    for name in ('alphal', 'alphag', 'alpham', 'alphalt'):
        assert np.abs(a[name] - b[name]).max() < 2e-6, name
    assert np.abs(a['logit'] - b['logit']).max() < 2e-5
    np.testing.assert_allclose(a['cost'], b['cost'], rtol=1e-6, atol=1e-5)
    for k in grads[0]:
        scale = np.abs(grads[0][k]).max()
        assert np.abs(grads[0][k] - grads[1][k]).max() <= 5e-5 * scale + 5e-6, k        # floor: the softmax offsets c_* have an exactly-zero gradient
    np.testing.assert_allclose(losses[0], losses[1], rtol=2e-6)
    assert losses[0][-1] < losses[0][0]
