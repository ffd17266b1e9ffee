"""CPU tests of the oracle: against the committed golden vectors (made by oracle/make_golden.py from
the reference's own translation unit), against that translation unit directly when oracle/_ref is
built, and against the analytical identities the reference states in prose
(TODO/TODO_seed_delta_debug_toy_model.md:46-54: W=I exact, diagonal exact, circulant+base exact)."""
import numpy as np
import pytest

import oracle_py as O
from conftest import bits, golden_names, load_golden
from llama_cpp_amd.payload import Payload, make_payload, make_x


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden_bit_exact(name):
    p, x, y_ref = load_golden(name)
    for nth in (1, 3):
        y = O.forward(p, x, nth=nth)
        assert np.array_equal(bits(y), bits(y_ref)), f"{name}: oracle differs from the reference output"


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("scheme", ["coo", "block"])
@pytest.mark.parametrize("base", [False, True])
@pytest.mark.parametrize("dims", [(96, 256), (256, 96), (200, 520), (520, 200), (1152, 6912 // 8)])
def test_oracle_matches_reference_tu(scheme, base, dims):
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme=scheme, K=20, block=16, base=base, d13_random=True, seed=n_in)
    x = make_x(n_in, 3, seed=n_out)
    y = O.forward(p, x, nth=3)
    y_ref = O.forward(p, x, nth=2, use_ref=True)
    assert np.array_equal(bits(y), bits(y_ref))


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")
def test_fp16_conversions_match_reference():
    lib, ref = O.oracle_lib(), O.ref_lib()
    for h in range(0, 65536, 3):
        a, b = lib.sdo_fp16_to_fp32(h), ref.sdref_fp16_to_fp32(h)
        assert (a != a and b != b) or np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(3000).astype(np.float32) * s
                           for s in (1e-8, 1e-5, 1e-3, 1, 100, 70000)])
    for v in vals:
        assert lib.sdo_fp32_to_fp16(float(v)) == ref.sdref_fp32_to_fp16(float(v))


def test_fp16_conversions_match_numpy():
    lib = O.oracle_lib()
    hs = np.arange(65536, dtype=np.uint16)
    f = hs.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 7):
        a = lib.sdo_fp16_to_fp32(h)
        assert (a != a and f[h] != f[h]) or np.float32(a).view(np.uint32) == f[h].view(np.uint32)
    for v in (0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e30, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, 0.1):
        with np.errstate(over="ignore"):
            assert lib.sdo_fp32_to_fp16(v) == int(np.float32(v).astype(np.float16).view(np.uint16))


def test_hadamard_matches_matrix():
    lib = O.oracle_lib()
    n = 64
    H = np.array([[(-1) ** bin(i & j).count("1") for j in range(n)] for i in range(n)], dtype=np.float64)
    v = np.arange(n, dtype=np.float32) - 17.0          # small integers: exact in fp32
    a = v.copy()
    lib.sdo_hadamard(a.ctypes.data, n)
    assert np.array_equal(a.astype(np.float64), H @ v.astype(np.float64))


def test_identity_weight_is_exact():
    """A1: W = I as a K=1 COO residual reproduces x exactly."""
    n = 64
    p = Payload(n, n, d_idx=np.arange(n, dtype=np.int16)[:, None].copy(),
                d_val=np.ones((n, 1), dtype=np.float16))
    x = make_x(n, 2)
    assert np.array_equal(bits(O.forward(p, x)), bits(x))


def test_xor_circulant_base_is_exact():
    """A4: a weight that IS XOR-circulant, W[p,q] = h[p^q] with integer h, is reproduced exactly by
    the base alone (d2 = FWHT(h)/L, zero residual)."""
    L = 32
    rng = np.random.default_rng(3)
    h = rng.integers(-3, 4, size=L).astype(np.float32)
    d2 = h.copy()
    O.oracle_lib().sdo_hadamard(d2.ctypes.data, L)
    d2 = (d2 / L).astype(np.float32)
    assert np.array_equal(d2.astype(np.float16).astype(np.float32), d2)      # representable
    p = Payload(L, L, d_idx=np.full((L, 1), -1, dtype=np.int16), d_val=np.zeros((L, 1), dtype=np.float16),
                base_d1=np.ones((1, L), np.float16), base_d2=d2.astype(np.float16)[None],
                base_d3=np.ones((1, L), np.float16), base_perm1=np.arange(L, dtype=np.int16)[None].copy())
    x = rng.integers(-4, 5, size=(2, L)).astype(np.float32)
    W = np.array([[h[pp ^ q] for q in range(L)] for pp in range(L)], dtype=np.float64)
    y = O.forward(p, x)
    assert np.array_equal(y.astype(np.float64), x.astype(np.float64) @ W.T)
    # and the closed-form tile generator agrees with the transform
    assert np.array_equal(O.w0_tile(p, 0, L, 0, L).astype(np.float64), W)


@pytest.mark.parametrize("dims", [(64, 200), (200, 64)])
def test_w0_closed_form_matches_transform(dims):
    """W0 tile definition vs the transform applied to unit vectors."""
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme="coo", K=4, base=True, d13_random=True, row_scale=False, seed=5)
    p.d_val[:] = 0
    W0 = O.w0_tile(p, 0, n_out, 0, n_in)
    y = O.forward(p, np.eye(n_in, dtype=np.float32))          # column i of W0 = op(e_i)
    np.testing.assert_allclose(y.T, W0, rtol=0, atol=2e-6 * np.abs(W0).max())


def test_threads_do_not_change_results():
    p = make_payload(300, 100, scheme="block", K=48, block=16, base=True, seed=9)
    x = make_x(300, 5)
    y1 = O.forward(p, x, nth=1)
    for nth in (2, 5, 8):
        assert np.array_equal(bits(O.forward(p, x, nth=nth)), bits(y1))


def test_nan_repair_with_q8_0_dense_weight():
    """NaN guard with a quantised dense weight: the repaired output is the double-accumulated dense
    dot product of the dequantised row (src/llama-seeddelta.cpp:294-361)."""
    n_in, n_out = 64, 8
    rng = np.random.default_rng(1)
    qs = rng.integers(-127, 128, size=(n_out, n_in // 32, 32), dtype=np.int8)
    d = (0.01 * (1 + rng.random((n_out, n_in // 32)))).astype(np.float16)
    raw = np.zeros((n_out, n_in // 32, 34), dtype=np.uint8)
    raw[:, :, :2] = d.view(np.uint8).reshape(n_out, n_in // 32, 2)
    raw[:, :, 2:] = qs.view(np.uint8)
    p = make_payload(n_in, n_out, scheme="coo", K=4, seed=2)
    p.d_val[3, 0] = np.float16(np.inf)
    x = make_x(n_in, 2)
    y = O.forward(p, x, w_ref_q8=raw)
    w_row = (qs[3].astype(np.float32) * d[3].astype(np.float32)[:, None]).reshape(-1)
    for t in range(2):
        want = np.float32((w_row.astype(np.float64) * x[t].astype(np.float64)).sum())
        assert abs(y[t, 3] - want) <= 1e-6 * abs(want) + 1e-7
    if O.ref_available():
        y_ref = O.forward(p, x, w_ref_q8=raw, use_ref=True)
        assert np.array_equal(bits(y), bits(y_ref))
    # reference quirk: a single-token x is 1-D for ggml_n_dims, the dense fallback declines
    # (src/llama-seeddelta.cpp:323-325) and the guard writes 0.0
    y1 = O.forward(p, x[:1], w_ref_q8=raw)
    assert y1[0, 3] == 0.0
    if O.ref_available():
        assert np.array_equal(bits(y1), bits(O.forward(p, x[:1], w_ref_q8=raw, use_ref=True)))
