"""The drop-in test: the reference's own graph constructors and ggml custom-op machinery, with
src/llama-seeddelta.cpp swapped for llama.cpp_amd/shim/llama-seeddelta-hip.cpp
(oracle/_ref/libsdshim.so = reference ggml objects + oracle/ref_harness.cpp + the shim TU, linked to
libseeddelta_hip.so). The harness builds the node through llama_seeddelta_mul_mat*(), pulls the
callback out of op_params and runs it the way ggml-cpu does — so what is checked here is exactly
what a llama.cpp build with the swapped TU would compute.

Bars: same as test_gpu_forward (fast path, TOL = 2e-5 relative to the term magnitude); repaired
non-finite outputs (dense row in double, or 0) are bit-exact.
"""
import numpy as np
import pytest

import oracle_py as O
from conftest import bits, golden_names, load_golden
from llama_cpp_amd.payload import make_payload, make_x
from test_gpu_forward import check_fast

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shim(hip):
    if not O.shim_available():
        pytest.fail("oracle/_ref/libsdshim.so is missing: run __graft_entry__.build() where /root/reference exists")
    return O.shim_lib()


@pytest.mark.parametrize("name", golden_names())
def test_shim_matches_reference_goldens(hip, shim, name):
    p, x, y_ref = load_golden(name)
    y = O.forward_shim(p, x, nth=3)          # extra ggml worker threads must be harmless
    if "nonfinite" in name:
        # positions the reference repaired: dense row·x in double when w_ref is there (T >= 2), else 0
        w = hip.SeedDeltaWeight(p)
        bad = ~np.isfinite(w.forward_host(x))            # unrepaired device result
        assert bad.any() and np.all(np.isfinite(y))
        assert np.array_equal(bits(y)[bad], bits(y_ref)[bad])
        assert np.allclose(y[~bad], y_ref[~bad], rtol=1e-5, atol=1e-6)
        return
    check_fast(p, x, y, y_ref)


def test_shim_nonfinite_repair_uses_dense_row(shim):
    """Inf in the residual values: with the dense weight on the host and a 2-D x the output becomes
    the dense product (src/llama-seeddelta.cpp:363-392); the count is reported."""
    p = make_payload(64, 96, scheme="coo", K=8, seed=11)
    p.d_val = p.d_val.copy()
    p.d_val[5, 0] = np.inf
    p.d_val[17, 3] = -np.inf
    rng = np.random.default_rng(0)
    p.w_ref = rng.standard_normal((96, 64)).astype(np.float32)
    x = np.abs(make_x(64, 2)) + 0.1
    before = shim.llama_seeddelta_hip_nonfinite_total()
    y = O.forward_shim(p, x)
    y_ref = O.forward(p, x)                  # oracle pinned against the reference for this very case
    assert shim.llama_seeddelta_hip_nonfinite_total() - before == 4
    assert np.array_equal(bits(y[:, [5, 17]]), bits(y_ref[:, [5, 17]]))
    dense = (p.w_ref.astype(np.float64) @ x.astype(np.float64).T).T.astype(np.float32)
    assert np.allclose(y[:, [5, 17]], dense[:, [5, 17]], rtol=1e-6)


def test_shim_single_token_is_1d_and_declines_dense_repair(shim):
    """At T=1 ggml sees a 1-D x, the reference's dense fallback declines (:323) and writes 0."""
    p = make_payload(64, 96, scheme="coo", K=8, seed=12)
    p.d_val = p.d_val.copy()
    p.d_val[9, 1] = np.inf
    p.w_ref = np.ones((96, 64), dtype=np.float32)
    x = np.abs(make_x(64, 1)) + 0.1
    y = O.forward_shim(p, x)
    assert y[0, 9] == 0.0
    assert np.array_equal(bits(y[:, 9]), bits(O.forward(p, x)[:, 9]))


def test_shim_f16_activations(shim):
    """x may be F16 (read_x_f16_or_f32, src/llama-seeddelta.cpp:42-50): widened exactly on the host."""
    p = make_payload(512, 256, scheme="block", K=64, block=16, base=True, seed=13)
    x16 = make_x(512, 3).astype(np.float16)
    y_ref = O.forward(p, x16)
    y = O.forward_shim(p, x16)
    check_fast(p, x16.astype(np.float32), y, y_ref)


@pytest.mark.parametrize("scheme,base,dims", [("coo", False, (1024, 768)), ("block", False, (768, 1024)),
                                              ("coo", True, (1024, 2816)), ("block", True, (2816, 1024))])
@pytest.mark.parametrize("T", [1, 7])
def test_shim_random_payloads(shim, scheme, base, dims, T):
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme=scheme, K=64, block=32, base=base, d13_random=base, seed=T + n_in)
    x = make_x(n_in, T, seed=5)
    check_fast(p, x, O.forward_shim(p, x, nth=2), O.forward(p, x, nth=4))


@pytest.mark.parametrize("name", ["ffn_silu_base_coo_t5", "ffn_gelu_base_block_t1", "ffn_gelu_block_nobase_t3"])
def test_shim_nodes_inside_the_reference_ffn_graph(shim, name):
    """The whole gated FFN as ONE ggml graph computed by the reference's CPU backend
    (ggml_graph_compute, 2 threads), its three SeedDelta nodes served by the drop-in TU and its own
    swiglu / geglu between them (src/llama-graph.cpp:811-953) — against the same graph with the
    reference's op TU (the fixture)."""
    from conftest import load_ffn_golden
    from test_gpu_ffn import check_h, check_y, h_bounds
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden(name)
    h, y = O.ffn_forward_graph(gate, up, down, x, act, nth=2, shim=True)
    smooth, full, _ = h_bounds(gate, up, x, act)
    check_h(h, smooth, full, h_ref.astype(np.float64))
    check_y(y, y_ref.astype(np.float64), down, h_ref, full)


@pytest.mark.parametrize("name", ["ffn_silu_base_coo_t1", "ffn_gelu_base_block_t1", "ffn_silu_base_coo_t5", "ffn_gelu_block_nobase_t3"])
def test_shim_whole_ffn_node(shim, name):
    """The optional fifth constructor (one ggml node for the whole gated FFN) inside a
    graph computed by the reference's CPU backend, against the reference's three-node graph."""
    from conftest import load_ffn_golden
    from test_gpu_ffn import check_y, h_bounds
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden(name)
    y = O.ffn_node_shim(gate, up, down, x, act, nth=2)
    assert y is not None
    _, full, _ = h_bounds(gate, up, x, act)
    check_y(y, y_ref.astype(np.float64), down, h_ref, full)


def test_shim_whole_ffn_node_declines_when_dense_repair_is_possible(shim):
    """more than one token and a dense weight still loaded: the reference would repair from the dense
    row, so the constructor keeps the three-node form; with one token it does not (the reference's
    guard declines 1-D activations, src/llama-seeddelta.cpp:323)."""
    from conftest import load_ffn_golden
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden("ffn_silu_base_coo_t5")
    dense = np.zeros((gate.n_out, gate.n_in), dtype=np.float32)
    assert O.ffn_node_shim(gate, up, down, x, act, dense_ref=dense) is None
    assert O.ffn_node_shim(gate, up, down, x[:1], act, dense_ref=dense) is not None
