"""CPU tests of the gated FFN block oracle (oracle/sd_oracle.c: sdo_glu, oracle_py.ffn_forward) against
the committed fixtures of the reference's ggml graph (tests/golden/ffn_*.npz) and, when oracle/_ref is
built (build container), against the reference's own CPU backend run live.

Bars: GELU form (f16 table, thresholds) bit-exact; SiLU form within 5e-7 relative of the reference
(its x86 SIMD body evaluates expf with a polynomial; the oracle restates the scalar definition
ggml_silu_f32, vec.h:1061), and the tolerance is carried through ffn_down.
"""
import numpy as np
import pytest

import oracle_py as O
from conftest import bits, ffn_golden_names, load_ffn_golden

SILU_REL = 5e-7


def dense_abs(p):
    """|W| of a payload (n_out, n_in) in float64, by pushing the identity through the oracle."""
    eye = np.eye(p.n_in, dtype=np.float32)
    return np.abs(O.forward(p, eye).astype(np.float64)).T


def test_glu_table_and_thresholds_bit_exact_vs_reference():
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(3)
    g = (6.0 * rng.standard_normal(50000)).astype(np.float32)
    u = rng.standard_normal(50000).astype(np.float32)
    g[:10] = [10.0, -10.0, 0.0, -0.0, np.nan, np.inf, -np.inf, np.nextafter(np.float32(10), np.float32(0)),
              np.nextafter(np.float32(-10), np.float32(0)), 65504.0]
    a, b = O.glu(O.ACT_GELU, g, u), O.glu(O.ACT_GELU, g, u, use_ref=True)
    assert np.array_equal(bits(a)[~np.isnan(b)], bits(b)[~np.isnan(b)])
    assert np.array_equal(np.isnan(a), np.isnan(b))


def test_glu_silu_within_tolerance_of_reference():
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(4)
    g = (6.0 * rng.standard_normal(50000)).astype(np.float32)
    u = rng.standard_normal(50000).astype(np.float32)
    g[:6] = [0.0, -0.0, 88.0, -88.0, np.inf, 1e-30]
    a, b = O.glu(O.ACT_SILU, g, u), O.glu(O.ACT_SILU, g, u, use_ref=True)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    assert np.all(np.abs(a[fin] - b[fin]) <= SILU_REL * np.abs(b[fin]) + 1e-37)


def test_glu_against_the_stock_native_simd_build():
    """A stock reference build compiles ggml-cpu with -march=native (GGML_NATIVE), which selects the SIMD bodies of
    ggml_vec_swiglu_f32 (ggml_v_silu / ggml_v_expf, a polynomial expf) instead of the scalar ggml_silu_f32 the plain
    -O3 pin runs. oracle/_ref/libsdref_native.so is that build of the same files. Stated tolerance of the oracle
    (and hence of the HIP path, which evaluates x / (1 + expf(-x)) like the scalar body) against it: SILU_REL
    relative per element; the GELU table form has no SIMD variant and stays bit-exact."""
    if not O.ref_native_available():
        pytest.skip("oracle/_ref/libsdref_native.so not built (needs /root/reference; build container only)")
    rng = np.random.default_rng(9)
    g = (6.0 * rng.standard_normal(100003)).astype(np.float32)          # odd length: SIMD body + scalar tail
    u = rng.standard_normal(100003).astype(np.float32)
    g[:8] = [0.0, -0.0, 88.0, -88.0, 20.0, -20.0, 1e-30, -1e-30]
    a, b = O.glu(O.ACT_SILU, g, u), O.glu_native(O.ACT_SILU, g, u)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    rel = np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1e-37)
    assert rel.max() <= SILU_REL, rel.max()
    a, b = O.glu(O.ACT_GELU, g, u), O.glu_native(O.ACT_GELU, g, u)
    assert np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("name", ffn_golden_names())
def test_ffn_oracle_vs_reference_graph_fixture(name):
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden(name)
    h, y = O.ffn_forward(gate, up, down, x, act)
    if act == O.ACT_GELU:
        assert np.array_equal(bits(h), bits(h_ref))
        assert np.array_equal(bits(y), bits(y_ref))
    else:
        assert np.all(np.abs(h - h_ref) <= SILU_REL * np.abs(h_ref) + 1e-37)
        # through ffn_down: |dy| <= |W_down| . |dh|  (+ one rounding of the differing partial sums)
        bound = dense_abs(down) @ (SILU_REL * np.abs(h_ref.astype(np.float64))).T
        assert np.all(np.abs(y.astype(np.float64) - y_ref) <= 4.0 * bound.T + 1e-30)


def test_ffn_oracle_vs_live_reference_graph_threads():
    """the reference graph on 4 threads equals the fixture path on 1 (the ops are thread-invariant)"""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden("ffn_gelu_block_nobase_t3")
    h, y = O.ffn_forward_graph(gate, up, down, x, act, nth=4)
    assert np.array_equal(bits(h), bits(h_ref)) and np.array_equal(bits(y), bits(y_ref))
