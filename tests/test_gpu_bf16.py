"""bf16-MFMA path (BASELINE.json configs[3]).  The kernel rounds its operands to bf16 (round to nearest even) and
accumulates in fp32, so against float64 on the SAME bf16-rounded operands it must agree to fp32-accumulation accuracy;
against unrounded float64 the error is the bf16 input quantisation (2^-9 relative per operand)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMALL = dict(dim=128, dim_word=64, n_words=211, ctxg_dim=128, ctxl_dim=96, ctxm_dim=64, ctxglm_dim=128)


def bf16_round(x):
    """float32 -> nearest-even bfloat16, returned as float32 (numpy has no bf16)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


@pytest.fixture(scope="module")
def dec():
    import stattn
    from oracle import stattn_oracle as O
    return stattn.Decoder(O.default_options(**SMALL))


@pytest.mark.parametrize("M,N,K,transB", [(64, 64, 64, False), (200, 128, 96, False), (333, 192, 1000, True),
                                          (128, 256, 72, True), (1000, 1024, 520, False)])
def test_bf16_gemm_matches_float64_on_rounded_operands(dec, M, N, K, transB):
    rng = np.random.RandomState(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if transB else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32)
    Bm = B.T if transB else B
    ref = bf16_round(A).astype(np.float64) @ bf16_round(Bm).astype(np.float64)
    got = dec.gemm(A, B, kind=2, transB=transB)
    scale = np.sqrt(K)
    assert np.abs(got - ref).max() < 2e-6 * K + 1e-5, np.abs(got - ref).max()       # fp32 accumulation only
    # asymmetric operands catch a transposed C write or a wrong k mapping (identity A)
    exact = A.astype(np.float64) @ Bm.astype(np.float64)
    assert np.abs(got - exact).max() < 0.02 * scale                                  # bf16 quantisation of the inputs
    got2 = dec.gemm(A, B, bias=bias, add=add, act=1, kind=2, transB=transB)
    np.testing.assert_allclose(got2, np.tanh(ref + bias + add), atol=2e-5 * max(1, K / 64), rtol=0)


def test_bf16_gemm_identity_and_rejects_bad_shapes(dec):
    K = 128
    A = np.eye(K, dtype=np.float32)[:96]                       # 96 x 128 selector
    B = (np.arange(K * 64, dtype=np.float32).reshape(K, 64) % 251) - 125.0   # small integers: exact in bf16
    np.testing.assert_array_equal(dec.gemm(A, B, kind=2), B[:96])
    with pytest.raises(ValueError):
        dec.gemm(np.ones((8, 12), np.float32), np.ones((12, 64), np.float32), kind=2)    # K % 8 != 0
    with pytest.raises(ValueError):
        dec.gemm(np.ones((8, 16), np.float32), np.ones((16, 32), np.float32), kind=2)    # N % 64 != 0
