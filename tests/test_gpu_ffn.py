"""GPU parity tests of the gated FFN block (sdh_ffn_gate_up / sdh_ffn_forward / sdh_glu), called through
the C ABI, against the reference's ggml-graph fixtures (tests/golden/ffn_*.npz) and the CPU oracle.

Bars:
  * verification mode (SDH_FWD_EXACT_ORDER), GELU form: h and y bit-identical to the reference graph
    (nodes bit-exact, the f16 table is built with the host's tanhf exactly as ggml builds it);
  * verification mode, SiLU form: h within 1e-6 relative (device expf / division vs the reference's
    polynomial expf), y within that error pushed through |W_down|;
  * fast path (incl. the single gate|up launch at T = 1): node outputs differ from the CPU's by the
    summation-order tolerance of test_gpu_forward (2e-5 * sum|terms|); that error is pushed through
    the activation's slope and |W_down|. The GELU form is a step function of f16(g): an element whose
    g sits within the node tolerance of an f16 rounding boundary may land on the neighbouring table
    entry (one f16 ulp of input and output); the bound allows that for every element and the test
    also requires that at most 1 % of the elements need it.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_py as O
from conftest import bits, ffn_golden_names, load_ffn_golden
from llama_cpp_amd.payload import make_payload, make_x
from test_gpu_forward import fast_tolerance

pytestmark = pytest.mark.gpu

F16_ULP = 2.0 ** -10


def dense_abs(p):
    eye = np.eye(p.n_in, dtype=np.float32)
    return np.abs(O.forward(p, eye).astype(np.float64)).T            # (n_out, n_in)


def h_bounds(gate, up, x, act):
    """-> (smooth bound, full bound, h_ref) for h = act(g) * u given node tolerances on g and u."""
    g = O.forward(gate, x).astype(np.float64)
    u = O.forward(up, x).astype(np.float64)
    tg, tu = fast_tolerance(gate, x, None), fast_tolerance(up, x, None)
    h = O.glu(act, g.astype(np.float32), u.astype(np.float32)).astype(np.float64)
    a = np.divide(h, u, out=np.zeros_like(h), where=u != 0)          # act(g)
    slope = 1.13                                                      # max |silu'|, |gelu'| < 1.13
    smooth = slope * tg * np.abs(u) + np.abs(a) * tu + 1e-6 * np.abs(h) + 1e-30
    full = smooth
    if act == O.ACT_GELU:
        full = smooth + (slope * np.abs(g) * F16_ULP + np.abs(a) * F16_ULP) * np.abs(u)
    return smooth, full, h


def check_h(h_gpu, smooth, full, h_ref):
    err = np.abs(h_gpu.astype(np.float64) - h_ref)
    assert np.all(err <= full), f"max err/bound {np.max(err / full):.3g}"
    assert np.mean(err > smooth) <= 0.01


def check_y(y_gpu, y_ref, down, h_ref, h_bound):
    t_down = fast_tolerance(down, h_ref.astype(np.float32), None)
    bound = t_down + (dense_abs(down) @ h_bound.T).T
    err = np.abs(y_gpu.astype(np.float64) - y_ref)
    assert np.all(err <= bound), f"max err/bound {np.max(err / bound):.3g}"


@pytest.mark.parametrize("name", ffn_golden_names())
def test_ffn_exact_order_vs_reference_graph(hip, name):
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden(name)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    h = hip.ffn_gate_up_host(wg, wu, x, act, hip.FWD_EXACT_ORDER)
    y = hip.ffn_forward_host(wg, wu, wd, x, act, hip.FWD_EXACT_ORDER)
    assert not hip.ffn_last_fused()
    if act == O.ACT_GELU:
        assert np.array_equal(bits(h), bits(h_ref))
        assert np.array_equal(bits(y), bits(y_ref))
    else:
        hb = 1e-6 * np.abs(h_ref.astype(np.float64)) + 1e-37
        assert np.all(np.abs(h.astype(np.float64) - h_ref) <= hb)
        bound = (dense_abs(down) @ hb.T).T * 4.0 + 1e-30
        assert np.all(np.abs(y.astype(np.float64) - y_ref) <= bound)


@pytest.mark.parametrize("name", ffn_golden_names())
def test_ffn_fast_path_vs_reference_graph(hip, name):
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden(name)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    smooth, full, h_or = h_bounds(gate, up, x, act)
    h = hip.ffn_gate_up_host(wg, wu, x, act)
    # 1..4 tokens with tall base gate/up of one shape and L >= 1024: the single gate|up launch must be taken
    expect_fused = x.shape[0] <= 4 and gate.base_d1 is not None and gate.base_d1.shape[1] >= 1024
    assert hip.ffn_last_fused() == expect_fused
    check_h(h, smooth, full, h_ref.astype(np.float64))
    y = hip.ffn_forward_host(wg, wu, wd, x, act)
    assert hip.ffn_last_fused() == expect_fused
    check_y(y, y_ref.astype(np.float64), down, h_ref, full)
    assert wg.nonfinite_count() == 0 and wu.nonfinite_count() == 0 and wd.nonfinite_count() == 0


def fold_node_tol(p, x):
    """per (token, output) bound of a node evaluated as a folded f16 contraction (SDH_FWD_MFMA_FOLD): the f16 roundings of x,
    of W0 and of W0 + D against the dense |W|.|x| (2e-3: the 1.5e-3 of tests/test_gpu_prefill_full.py's fold_bound with
    room for |W0 + D| < |W0| + |D| where they overlap), plus the fast path's ordering tolerance"""
    return 2.0e-3 * (np.abs(x.astype(np.float64)) @ dense_abs(p).T) + 75.0 * fast_tolerance(p, x, None)


@pytest.mark.parametrize("act", [O.ACT_SILU, O.ACT_GELU])
@pytest.mark.parametrize("scheme,K,block", [("coo", 32, 16), ("block", 64, 16)])
def test_ffn_folded_prefill(hip, act, scheme, K, block):
    """sdh_ffn_gate_up / sdh_ffn_forward with SDH_FWD_MFMA_FOLD and many tokens: gate, up and down each as ONE folded
    contraction in the permuted order, the activation in the up node's epilogue (act(gate) * up written by the GEMM)."""
    n_embd, n_ff, T = 1024, 2816, 200
    kw = dict(scheme=scheme, K=K, block=block, base=True)
    gate, up, down = (make_payload(n_embd, n_ff, seed=31, **kw), make_payload(n_embd, n_ff, seed=32, **kw),
                      make_payload(n_ff, n_embd, seed=33, **kw))
    x = make_x(n_embd, T, seed=5)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    g = O.forward(gate, x).astype(np.float64)
    u = O.forward(up, x).astype(np.float64)
    tg, tu = fold_node_tol(gate, x), fold_node_tol(up, x)
    h_ref32 = O.glu(act, g.astype(np.float32), u.astype(np.float32))
    h_ref = h_ref32.astype(np.float64)
    a = np.divide(h_ref, u, out=np.zeros_like(h_ref), where=u != 0)
    hb = 1.13 * tg * np.abs(u) + np.abs(a) * tu + 1e-6 * np.abs(h_ref) + 1e-30
    if act == O.ACT_GELU:
        hb = hb + (1.13 * np.abs(g) * F16_ULP + np.abs(a) * F16_ULP) * np.abs(u)
    h = hip.ffn_gate_up_host(wg, wu, x, act, hip.FWD_MFMA_FOLD)
    err = np.abs(h.astype(np.float64) - h_ref)
    assert np.all(err <= hb), f"h: max err/bound {np.max(err / hb):.3g}"
    assert wg.fold_layout() & 2 and wu.fold_layout() & 2       # both nodes ran as permuted-order contractions
    y = hip.ffn_forward_host(wg, wu, wd, x, act, hip.FWD_MFMA_FOLD)
    y_ref = O.forward(down, h_ref32).astype(np.float64)
    bound = fold_node_tol(down, h_ref32) + (dense_abs(down) @ hb.T).T
    erry = np.abs(y.astype(np.float64) - y_ref)
    assert np.all(erry <= bound), f"y: max err/bound {np.max(erry / bound):.3g}"
    assert wd.fold_layout() & 2
    assert np.max(erry / bound) > 1e-4                          # not the exact forms
    y0 = hip.ffn_forward_host(wg, wu, wd, x, act)               # the default path is another one, and close
    assert not np.array_equal(bits(y), bits(y0))
    assert np.abs(y - y0).max() <= 2e-2 * np.abs(y0).max()
    assert wg.nonfinite_count() == 0 and wu.nonfinite_count() == 0 and wd.nonfinite_count() == 0
    # from 224 tokens on the block takes the folded forms by itself; SDH_FWD_NO_MFMA keeps the fp32 ones
    x3 = make_x(n_embd, 300, seed=6)
    y3 = hip.ffn_forward_host(wg, wu, wd, x3, act)
    assert np.array_equal(bits(y3), bits(hip.ffn_forward_host(wg, wu, wd, x3, act, hip.FWD_MFMA_FOLD)))
    y3f = hip.ffn_forward_host(wg, wu, wd, x3, act, hip.FWD_NO_MFMA)
    assert not np.array_equal(bits(y3), bits(y3f)) and np.abs(y3 - y3f).max() <= 2e-2 * np.abs(y3f).max()
    h3 = hip.ffn_gate_up_host(wg, wu, x3, act)
    assert np.array_equal(bits(h3), bits(hip.ffn_gate_up_host(wg, wu, x3, act, hip.FWD_MFMA_FOLD)))
    # above 256 tokens gate and up are ONE two-pass launch; up to 256 they are two launches: the same bits token by token
    h2 = hip.ffn_gate_up_host(wg, wu, x3[:256], act, hip.FWD_MFMA_FOLD)
    assert np.array_equal(bits(h3[:256]), bits(h2))


@pytest.mark.parametrize("act", [O.ACT_SILU, O.ACT_GELU])
@pytest.mark.parametrize("scheme,K,block", [("coo", 32, 16), ("coo", 128, 16), ("block", 64, 16), ("block", 128, 32)])
@pytest.mark.parametrize("dims", [(1024, 2816), (2048, 5632), (1152, 6912)])
def test_gate_up_single_launch_matches_node_by_node(hip, act, scheme, K, block, dims):
    """T = 1: the gate|up launch against the node-by-node device path (SDH_FWD_NO_FUSE) and the oracle."""
    n_embd, n_ff = dims
    kw = dict(scheme=scheme, K=K, block=block, base=True, d13_random=True)
    gate, up = make_payload(n_embd, n_ff, seed=21, **kw), make_payload(n_embd, n_ff, seed=22, **kw)
    wg, wu = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up)
    x = (3.0 * make_x(n_embd, 1, seed=9)).astype(np.float32)
    smooth, full, h_ref = h_bounds(gate, up, x, act)
    for rep in range(3):                                  # epochs advance, granules are reused
        h1 = hip.ffn_gate_up_host(wg, wu, x, act)
        assert hip.ffn_last_fused()
        check_h(h1, smooth, full, h_ref)
    h2 = hip.ffn_gate_up_host(wg, wu, x, act, hip.FWD_NO_FUSE)
    assert not hip.ffn_last_fused()
    check_h(h2, smooth, full, h_ref)
    # the plain ops on the same handles still work after the pair launches (shared granule buffers)
    yg = wg.forward_host(x)
    assert np.all(np.abs(yg - O.forward(gate, x)) <= fast_tolerance(gate, x, None))


@pytest.mark.parametrize("act", [O.ACT_SILU, O.ACT_GELU])
@pytest.mark.parametrize("scheme,K,block", [("coo", 32, 16), ("coo", 128, 16), ("block", 64, 16), ("block", 128, 32), ("block", 1024, 32)])
@pytest.mark.parametrize("dims", [(1024, 2816), (2048, 5632), (4096, 9000), (4096, 14336)])
def test_gate_up_small_batch_single_launch_is_bit_identical_to_node_by_node(hip, act, scheme, K, block, dims):
    """2..4 tokens: the gate|up launch (base blocks of both tensors for every token, rows of both tensors per wave,
    COO with x staged in LDS when it fits) gives every token the bits of its own one-token gate|up launch and of the
    node-by-node device path; padded token columns included."""
    torch = pytest.importorskip("torch")
    n_embd, n_ff = dims
    if K * 1 > 512 and n_embd < 2048:
        pytest.skip("rows longer than the input")
    kw = dict(scheme=scheme, K=K, block=block, base=True, d13_random=True)
    gate, up = make_payload(n_embd, n_ff, seed=31, **kw), make_payload(n_embd, n_ff, seed=32, **kw)
    wg, wu = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up)
    xs = (2.0 * make_x(n_embd, 4, seed=19)).astype(np.float32)
    single = []
    for t in range(4):
        single.append(hip.ffn_gate_up_host(wg, wu, xs[t:t + 1], act))
    fused1 = hip.ffn_last_fused()                            # long rows x many rows exceed the co-residency limit: node by node
    assert fused1 or K > 512
    single = np.concatenate(single, axis=0)
    two_steps = K > 512                                      # rows of two steps: four token slots would spill, 3-4 tokens go node by node
    for T in (2, 3, 4):
        for rep in range(2):
            h = hip.ffn_gate_up_host(wg, wu, xs[:T], act)
            assert hip.ffn_last_fused() == (fused1 and not (two_steps and T > 2)), (T, rep)
            assert np.array_equal(bits(h), bits(single[:T])), (T, rep)
        hn = hip.ffn_gate_up_host(wg, wu, xs[:T], act, hip.FWD_NO_FUSE)
        assert not hip.ffn_last_fused()
        assert np.array_equal(bits(hn), bits(single[:T]))
    smooth, full, h_ref = h_bounds(gate, up, xs, act)
    check_h(hip.ffn_gate_up_host(wg, wu, xs, act), smooth, full, h_ref)
    # device buffers with padded columns
    ldx = n_embd + 8
    xd = torch.zeros((3, ldx), dtype=torch.float32, device="cuda")
    xd[:, :n_embd] = torch.from_numpy(xs[1:4]).cuda()
    hd = torch.zeros((3, n_ff), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    hip.ffn_gate_up_device(wg, wu, act, xd.data_ptr(), ldx, 3, hd.data_ptr(), n_ff, 0, st.cuda_stream)
    st.synchronize()
    assert hip.ffn_last_fused() == (fused1 and not two_steps)
    assert np.array_equal(bits(hd.cpu().numpy()), bits(single[1:4]))
    assert wg.nonfinite_count() == 0 and wu.nonfinite_count() == 0


@pytest.mark.parametrize("scheme,K,block", [("coo", 128, 16), ("coo", 32, 16), ("block", 128, 16)])
@pytest.mark.parametrize("dims", [(2048, 5632), (4096, 14336)])
def test_ffn_two_to_four_tokens_keep_one_token_results(hip, scheme, K, block, dims):
    """sdh_ffn_forward with 2..4 tokens: the gate|up launch also leaves h token-interleaved and a COO down launch gathers
    from that copy (one 16-byte gather per entry for all tokens). Every token's output is its one-token output bit for
    bit, call after call (the interleaved copy is rewritten per call), and equals the node-by-node path's."""
    n_embd, n_ff = dims
    kw = dict(scheme=scheme, K=K, block=block, base=True, d13_random=True)
    gate, up = make_payload(n_embd, n_ff, seed=61, **kw), make_payload(n_embd, n_ff, seed=62, **kw)
    down = make_payload(n_ff, n_embd, seed=63, **kw)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    xs = make_x(n_embd, 8, seed=29)
    single = np.concatenate([hip.ffn_forward_host(wg, wu, wd, xs[t:t + 1], O.ACT_GELU) for t in range(8)], axis=0)
    for T in (4, 2, 3, 4):
        for t0 in (0, 4):
            y = hip.ffn_forward_host(wg, wu, wd, xs[t0:t0 + T], O.ACT_GELU)
            assert hip.ffn_last_fused()
            assert np.array_equal(bits(y), bits(single[t0:t0 + T])), (T, t0)
    yn = hip.ffn_forward_host(wg, wu, wd, xs[:3], O.ACT_GELU, hip.FWD_NO_FUSE)
    assert np.array_equal(bits(yn), bits(single[:3]))
    assert wg.nonfinite_count() == 0 and wu.nonfinite_count() == 0 and wd.nonfinite_count() == 0


@pytest.mark.parametrize("scheme,K,block", [("coo", 128, 16), ("block", 128, 16)])
def test_ffn_five_to_eight_tokens_keep_one_token_results(hip, scheme, K, block):
    """sdh_ffn_forward with 5..8 tokens goes node by node through the LDS rows kernels (base terms of gate and up from
    one launch): every token's output is its one-token output bit for bit, and the non-finite counters cover the
    whole call."""
    n_embd, n_ff = 2048, 5632
    kw = dict(scheme=scheme, K=K, block=block, base=True, d13_random=True)
    gate, up = make_payload(n_embd, n_ff, seed=41, **kw), make_payload(n_embd, n_ff, seed=42, **kw)
    down = make_payload(n_ff, n_embd, seed=43, **kw)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    xs = make_x(n_embd, 8, seed=23)
    single = np.concatenate([hip.ffn_forward_host(wg, wu, wd, xs[t:t + 1], O.ACT_SILU) for t in range(8)], axis=0)
    for T in (5, 6, 7, 8):
        y = hip.ffn_forward_host(wg, wu, wd, xs[:T], O.ACT_SILU)
        assert not hip.ffn_last_fused()                          # node by node through the LDS rows kernels
        assert np.array_equal(bits(y), bits(single[:T])), T
        assert wg.nonfinite_count() == 0 and wu.nonfinite_count() == 0 and wd.nonfinite_count() == 0
    y9 = hip.ffn_forward_host(wg, wu, wd, np.concatenate([xs, xs[:1]]), O.ACT_SILU)       # 9 tokens: two token tiles
    assert not hip.ffn_last_fused()
    smooth, full, h_ref = h_bounds(gate, up, xs, O.ACT_SILU)
    y_ref = O.forward(down, O.glu(O.ACT_SILU, O.forward(gate, xs), O.forward(up, xs)))
    check_y(y9[:8], y_ref.astype(np.float64), down, h_ref.astype(np.float32), full)
    # a non-finite gate output in every token: the gate node's counter reports the whole call
    bad = make_payload(n_embd, n_ff, seed=41, **kw)
    if scheme == "coo":
        bad.d_val[7, 0] = np.float16(np.inf); bad.d_idx[7, 0] = 3
    else:
        bad.b_val[7, 0, 3] = np.float16(np.inf); bad.b_idx[7, 0] = 0
    wb = hip.SeedDeltaWeight(bad)
    xb = xs[:7].copy()
    xb[:, 3] = 1.0                                          # inf * x is non-finite for every token
    y = hip.ffn_forward_host(wb, wu, wd, xb, O.ACT_SILU, hip.FWD_ZERO_NONFINITE)
    assert wb.nonfinite_count() == 7 and wu.nonfinite_count() == 0
    assert np.all(np.isfinite(y))


def test_gate_up_declines_gracefully(hip):
    """shapes the single launch does not cover fall back to node by node and stay correct"""
    x = make_x(512, 1, seed=2)
    for kw_g, kw_u in [(dict(scheme="coo", K=32, base=True), dict(scheme="block", K=32, block=16, base=True)),   # kinds differ
                       (dict(scheme="coo", K=32, base=False), dict(scheme="coo", K=32, base=False)),              # no base
                       (dict(scheme="coo", K=32, base=True), dict(scheme="coo", K=64, base=True))]:               # row shapes differ
        gate, up = make_payload(512, 1536, seed=1, **kw_g), make_payload(512, 1536, seed=2, **kw_u)
        wg, wu = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up)
        smooth, full, h_ref = h_bounds(gate, up, x, O.ACT_SILU)
        h = hip.ffn_gate_up_host(wg, wu, x, O.ACT_SILU)
        assert not hip.ffn_last_fused()
        check_h(h, smooth, full, h_ref)


def test_ffn_nonfinite_nodes_are_counted_and_zeroed(hip):
    gate, up = make_payload(1024, 2048, seed=5, scheme="coo", K=16, base=True), make_payload(1024, 2048, seed=6, scheme="coo", K=16, base=True)
    down = make_payload(2048, 1024, seed=7, scheme="coo", K=16, base=True)
    gate.d_val[7, 0] = np.float16(np.inf)
    gate.d_idx[7, 0] = 3
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    x = make_x(1024, 1, seed=3)
    x[0, 3] = 1.0
    h = hip.ffn_gate_up_host(wg, wu, x, O.ACT_SILU, hip.FWD_ZERO_NONFINITE)
    assert hip.ffn_last_fused()
    assert wg.nonfinite_count() == 1 and wu.nonfinite_count() == 0
    assert h[0, 7] == 0.0 and np.all(np.isfinite(h))     # silu(0) * u = 0, as the reference's guard + swiglu give
    y = hip.ffn_forward_host(wg, wu, wd, x, O.ACT_SILU, hip.FWD_ZERO_NONFINITE)
    assert np.all(np.isfinite(y)) and wd.nonfinite_count() == 0


def test_ffn_device_io_and_strides(hip):
    gate, up, down, x, act, h_ref, y_ref = load_ffn_golden("ffn_silu_base_coo_t5")
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    T, n_embd = x.shape
    ldx, ldy = n_embd + 8, n_embd + 4
    xd = torch.zeros((T, ldx), dtype=torch.float32, device="cuda")
    xd[:, :n_embd] = torch.from_numpy(x).cuda()
    yd = torch.full((T, ldy), 7.0, dtype=torch.float32, device="cuda")
    hip.ffn_forward_device(wg, wu, wd, act, xd.data_ptr(), ldx, T, yd.data_ptr(), ldy, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    assert np.all(y[:, n_embd:] == 7.0)
    y_host = hip.ffn_forward_host(wg, wu, wd, x, act)
    assert np.array_equal(bits(y[:, :n_embd]), bits(y_host))


@pytest.mark.parametrize("act", [O.ACT_SILU, O.ACT_GELU])
def test_ffn_block_under_graph_capture(hip, act):
    """One decode FFN block (gate|up launch + fused down launch) recorded into a HIP graph and replayed
    with changing inputs; eager blocks on the same handles run in between. Captured launches use the
    reserved hand-off tag (sdh_decode.hip), so every replay must equal the eager block bit for bit."""
    kw = dict(scheme="block", K=64, block=16, base=True, d13_random=True)
    gate, up = make_payload(2048, 5632, seed=41, **kw), make_payload(2048, 5632, seed=42, **kw)
    down = make_payload(5632, 2048, seed=43, **kw)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    xs = make_x(2048, 4, seed=12)
    want = [hip.ffn_forward_host(wg, wu, wd, xs[i:i + 1], act) for i in range(4)]
    assert hip.ffn_last_fused()
    xd = torch.zeros((1, 2048), dtype=torch.float32, device="cuda")
    yd = torch.zeros((1, 2048), dtype=torch.float32, device="cuda")
    ye = torch.zeros((1, 2048), dtype=torch.float32, device="cuda")
    xe = torch.from_numpy(xs[3:4]).cuda()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hip.ffn_forward_device(wg, wu, wd, act, xd.data_ptr(), 2048, 1, yd.data_ptr(), 2048, 0, side.cuda_stream)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            hip.ffn_forward_device(wg, wu, wd, act, xd.data_ptr(), 2048, 1, yd.data_ptr(), 2048, 0,
                                   torch.cuda.current_stream().cuda_stream)
            assert hip.ffn_last_fused()                  # the recorded block is the two-launch one
        for i in range(3):
            xd.copy_(torch.from_numpy(xs[i:i + 1]))
            g.replay()
            g.replay()
            if i == 1:
                hip.ffn_forward_device(wg, wu, wd, act, xe.data_ptr(), 2048, 1, ye.data_ptr(), 2048, 0, side.cuda_stream)
            side.synchronize()
            assert np.array_equal(bits(yd.cpu().numpy()), bits(want[i])), f"replay {i}"
        assert np.array_equal(bits(ye.cpu().numpy()), bits(want[3]))
    assert wg.nonfinite_count() == 0 and wu.nonfinite_count() == 0 and wd.nonfinite_count() == 0


@pytest.mark.parametrize("scheme", ["block", "coo"])
def test_ffn_small_batch_block_under_graph_capture(hip, scheme):
    """Three tokens per call: the several-token gate|up launch (COO: x staged in LDS) and the fused several-token down
    launch recorded into a HIP graph and replayed with changing inputs, an eager call on the same handles in between;
    the captured launches clear the granules they read, so every replay equals the eager result bit for bit."""
    kw = dict(scheme=scheme, K=64, block=16, base=True, d13_random=True)
    gate, up = make_payload(2048, 5632, seed=51, **kw), make_payload(2048, 5632, seed=52, **kw)
    down = make_payload(5632, 2048, seed=53, **kw)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    xs = make_x(2048, 12, seed=14)
    want = [hip.ffn_forward_host(wg, wu, wd, xs[3 * i:3 * i + 3], O.ACT_SILU) for i in range(4)]
    assert hip.ffn_last_fused()
    xd = torch.zeros((3, 2048), dtype=torch.float32, device="cuda")
    yd = torch.zeros((3, 2048), dtype=torch.float32, device="cuda")
    ye = torch.zeros((3, 2048), dtype=torch.float32, device="cuda")
    xe = torch.from_numpy(xs[9:12]).cuda()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hip.ffn_forward_device(wg, wu, wd, O.ACT_SILU, xd.data_ptr(), 2048, 3, yd.data_ptr(), 2048, 0, side.cuda_stream)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            hip.ffn_forward_device(wg, wu, wd, O.ACT_SILU, xd.data_ptr(), 2048, 3, yd.data_ptr(), 2048, 0,
                                   torch.cuda.current_stream().cuda_stream)
            assert hip.ffn_last_fused()
        for i in range(3):
            xd.copy_(torch.from_numpy(xs[3 * i:3 * i + 3]))
            g.replay()
            g.replay()
            if i == 1:
                hip.ffn_forward_device(wg, wu, wd, O.ACT_SILU, xe.data_ptr(), 2048, 3, ye.data_ptr(), 2048, 0, side.cuda_stream)
            side.synchronize()
            assert np.array_equal(bits(yd.cpu().numpy()), bits(want[i])), f"replay {i}"
        assert np.array_equal(bits(ye.cpu().numpy()), bits(want[3]))
    assert wg.nonfinite_count() == 0 and wu.nonfinite_count() == 0 and wd.nonfinite_count() == 0


def test_glu_kernel_vs_oracle(hip):
    rng = np.random.default_rng(8)
    n = 100003
    g = (6.0 * rng.standard_normal(n)).astype(np.float32)
    u = rng.standard_normal(n).astype(np.float32)
    g[:6] = [10.0, -10.0, 0.0, -0.0, np.inf, -np.inf]
    gd, ud = torch.from_numpy(g).cuda(), torch.from_numpy(u).cuda()
    yd = torch.empty_like(gd)
    for act in (O.ACT_GELU, O.ACT_SILU):
        hip.glu_device(act, gd.data_ptr(), ud.data_ptr(), yd.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        y, ref = yd.cpu().numpy(), O.glu(act, g, u)
        if act == O.ACT_GELU:
            assert np.array_equal(bits(y), bits(ref))
        else:
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(y), fin)
            assert np.all(np.abs(y[fin] - ref[fin]) <= 1e-6 * np.abs(ref[fin]) + 1e-37)


def test_ffn_full_size_mistral_layer(hip):
    """Mistral-7B layer shape, T = 1: gate|up launch + down against the oracle."""
    kw = dict(scheme="coo", K=128, base=True)
    gate, up = make_payload(4096, 14336, seed=31, **kw), make_payload(4096, 14336, seed=32, **kw)
    down = make_payload(14336, 4096, seed=33, **kw)
    wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
    x = make_x(4096, 1, seed=4)
    smooth, full, h_ref = h_bounds(gate, up, x, O.ACT_SILU)
    h = hip.ffn_gate_up_host(wg, wu, x, O.ACT_SILU)
    assert hip.ffn_last_fused()
    check_h(h, smooth, full, h_ref)
    y = hip.ffn_forward_host(wg, wu, wd, x, O.ACT_SILU)
    y2 = hip.ffn_forward_host(wg, wu, wd, x, O.ACT_SILU, hip.FWD_NO_FUSE)
    y_or = O.forward(down, h_ref.astype(np.float32))
    # |W_down| . |dh| without materialising W_down: bound dh by its norm and the row norms of the two parts
    scale = np.abs(y_or).max()
    assert np.max(np.abs(y - y_or)) <= 2e-4 * scale and np.max(np.abs(y2 - y_or)) <= 2e-4 * scale
