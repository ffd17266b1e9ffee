"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the committed
golden vectors of the reference.

Bars (BASELINE.json north_star):
  * index logic and W0 tiles: bit-exact;
  * FFN outputs in verification mode (SDH_FWD_EXACT_ORDER): bit-exact with the CPU op;
  * FFN outputs on the fast path: fp32 accumulation in a different order, so
        |y_gpu - y_cpu| <= TOL * |rs| * (|base| + sum_r |val_r * x_r|),   TOL = 2e-5
    (n-term fp32 sums differ by at most ~n*eps*sum|terms| between orders; n <= 1536, eps = 6e-8;
    observed differences are ~100x smaller). The base term itself is bit-exact on both paths.
"""
import numpy as np
import pytest

import oracle_py as O
from conftest import abs_term_sum, bits, golden_names, load_golden
from llama_cpp_amd.payload import Payload, config_payload, make_payload, make_x

pytestmark = pytest.mark.gpu

TOL = 2e-5


def fast_tolerance(p, x, y_ref):
    s = abs_term_sum(p, x)
    if p.base_d1 is not None:
        q = Payload(p.n_in, p.n_out, base_d1=p.base_d1, base_d2=p.base_d2, base_d3=p.base_d3,
                    base_perm1=p.base_perm1, base_perm2=p.base_perm2)
        if p.kind == 3:
            q.b_idx, q.b_val = np.full_like(p.b_idx, -1), p.b_val
        else:
            q.d_idx, q.d_val = np.full_like(p.d_idx, -1), p.d_val
        s = s + np.abs(O.forward(q, x).astype(np.float64))
    rs = np.abs(p.row_scale.astype(np.float64))[None, :] if p.row_scale is not None else 1.0
    return TOL * rs * s + 1e-30


def check_fast(p, x, y_gpu, y_ref):
    tol = fast_tolerance(p, x, y_ref)
    err = np.abs(y_gpu.astype(np.float64) - y_ref.astype(np.float64))
    assert np.all(np.isfinite(y_gpu) == np.isfinite(y_ref))
    fin = np.isfinite(y_ref)
    assert np.all(err[fin] <= tol[fin]), f"max err/tol = {np.max(err[fin] / tol[fin]):.3g}"


def nonfinite_case(name):
    return "nonfinite" in name


@pytest.mark.parametrize("name", golden_names())
def test_golden_exact_order_is_bit_exact(hip, name):
    p, x, y_ref = load_golden(name)
    w = hip.SeedDeltaWeight(p)
    flags = hip.FWD_EXACT_ORDER
    if nonfinite_case(name):
        if p.w_ref is not None:
            pytest.skip("dense-weight repair is done by the shim on the host (see test_shim)")
        flags |= hip.FWD_ZERO_NONFINITE
    y = w.forward_host(x, flags)
    assert np.array_equal(bits(y), bits(y_ref)), f"{name}: max abs diff {np.abs(y - y_ref).max()}"


@pytest.mark.parametrize("name", golden_names())
def test_golden_fast_path_within_tolerance(hip, name):
    p, x, y_ref = load_golden(name)
    if nonfinite_case(name):
        pytest.skip("covered by test_nonfinite_counter")
    w = hip.SeedDeltaWeight(p)
    check_fast(p, x, w.forward_host(x), y_ref)


@pytest.mark.parametrize("scheme,K,block", [("coo", 32, 16), ("coo", 128, 16), ("block", 64, 16),
                                            ("block", 128, 32), ("block", 24, 8)])
@pytest.mark.parametrize("base", [False, True])
@pytest.mark.parametrize("dims", [(256, 1024), (1024, 256), (1152, 864)])
@pytest.mark.parametrize("T", [1, 5])
def test_random_payloads_vs_oracle(hip, scheme, K, block, base, dims, T):
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme=scheme, K=K, block=block, base=base, d13_random=base, seed=K + n_in)
    x = make_x(n_in, T, seed=T)
    y_ref = O.forward(p, x, nth=4)
    w = hip.SeedDeltaWeight(p)
    assert w.kind == p.kind
    assert w.payload_bytes == p.payload_bytes()
    y_exact = w.forward_host(x, hip.FWD_EXACT_ORDER)
    assert np.array_equal(bits(y_exact), bits(y_ref))
    check_fast(p, x, w.forward_host(x), y_ref)
    assert w.nonfinite_count() == 0


@pytest.mark.parametrize("idx_dtype", [np.int16, np.int32])
@pytest.mark.parametrize("val_dtype", [np.float16, np.float32])
@pytest.mark.parametrize("scheme", ["coo", "block"])
def test_dtype_combinations(hip, idx_dtype, val_dtype, scheme):
    p = make_payload(512, 384, scheme=scheme, K=64, block=16, base=True, idx_dtype=idx_dtype,
                     val_dtype=val_dtype, rs_dtype=np.float32, base_dtype=np.float32, perm_dtype=np.int32, seed=3)
    x = make_x(512, 3)
    y_ref = O.forward(p, x)
    w = hip.SeedDeltaWeight(p)
    assert np.array_equal(bits(w.forward_host(x, hip.FWD_EXACT_ORDER)), bits(y_ref))
    check_fast(p, x, w.forward_host(x), y_ref)


def test_strided_host_tensors_are_repacked(hip):
    """ggml tensors may be views: nb[] is honoured at upload."""
    p = make_payload(256, 300, scheme="coo", K=16, seed=4)
    big_idx = np.zeros((300, 40), dtype=np.int16)
    big_val = np.zeros((300, 40), dtype=np.float16)
    big_idx[:, 3:19], big_val[:, 3:19] = p.d_idx, p.d_val
    q = Payload(256, 300, d_idx=big_idx[:, 3:19], d_val=big_val[:, 3:19], row_scale=p.row_scale)
    x = make_x(256, 2)
    xs = np.zeros((2, 300), dtype=np.float32)
    xs[:, :256] = x
    y_ref = O.forward(p, x)
    w = hip.SeedDeltaWeight(q)
    assert np.array_equal(bits(w.forward_host(xs[:, :256], hip.FWD_EXACT_ORDER)), bits(y_ref))


@pytest.mark.parametrize("dims", [(64, 200), (200, 64), (512, 1536), (1536, 512)])
def test_w0_tile_bit_exact(hip, dims):
    """The regenerated dense base tile (integer XOR index into G_b) against the oracle's closed form,
    and the closed form against the transform itself."""
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme="coo", K=4, base=True, d13_random=True, seed=11)
    w = hip.SeedDeltaWeight(p)
    o0, o1, i0, i1 = n_out // 5, n_out, 3, n_in - 1
    tile = w.w0_tile(o0, o1, i0, i1)
    ref = O.w0_tile(p, o0, o1, i0, i1)
    assert np.array_equal(bits(tile), bits(ref))


def test_w0_tile_rejects_non_bijective_perm(hip):
    p = make_payload(64, 128, scheme="coo", K=4, base=True, seed=1)
    p.base_perm1[0, 5] = p.base_perm1[0, 6]
    w = hip.SeedDeltaWeight(p)
    with pytest.raises(hip.SeedDeltaError):
        w.w0_tile(0, 8, 0, 8)
    # ... while the transform form handles it like the reference does (gather semantics)
    x = make_x(64, 2)
    assert np.array_equal(bits(w.forward_host(x, hip.FWD_EXACT_ORDER)), bits(O.forward(p, x)))


def test_nonfinite_counter_and_zeroing(hip):
    p, x, y_ref = load_golden("coo_nonfinite_no_wref")
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x)                       # left as computed
    n_bad = int((~np.isfinite(y)).sum())
    assert n_bad > 0 and w.nonfinite_count() == n_bad
    y0 = w.forward_host(x, hip.FWD_ZERO_NONFINITE)
    assert w.nonfinite_count() == n_bad
    assert np.all(y0[~np.isfinite(y)] == 0.0) and np.all(np.isfinite(y0))
    # a clean payload afterwards reports zero (per-call counter)
    p2 = make_payload(64, 96, scheme="coo", K=8)
    w2 = hip.SeedDeltaWeight(p2)
    w2.forward_host(make_x(64, 1))
    assert w2.nonfinite_count() == 0
    for _ in range(300):                        # ring of per-call counters wraps cleanly
        w.forward_host(x[:1])
    assert w.nonfinite_count() == int((~np.isfinite(y[:1])).sum())


def test_validation_mirrors_try_add_seeddelta(hip):
    """src/llama-context.cpp:145-268: bad sub-payloads are dropped, not fatal, unless nothing is left."""
    p = make_payload(128, 256, scheme="coo", K=8, base=True, seed=2)
    x = make_x(128, 2)
    # L not a power of two / too small -> base dropped, op becomes plain COO
    q = Payload(128, 256, d_idx=p.d_idx, d_val=p.d_val, row_scale=p.row_scale,
                base_d1=p.base_d1[:, :100].copy(), base_d2=p.base_d2[:, :100].copy())
    w = hip.SeedDeltaWeight(q)
    assert w.kind == 0
    ref = Payload(128, 256, d_idx=p.d_idx, d_val=p.d_val, row_scale=p.row_scale)
    assert np.array_equal(bits(w.forward_host(x, hip.FWD_EXACT_ORDER)), bits(O.forward(ref, x)))
    # row_scale of the wrong length is ignored
    q = Payload(128, 256, d_idx=p.d_idx, d_val=p.d_val, row_scale=p.row_scale[:100].copy())
    ref = Payload(128, 256, d_idx=p.d_idx, d_val=p.d_val)
    assert np.array_equal(bits(hip.SeedDeltaWeight(q).forward_host(x, hip.FWD_EXACT_ORDER)), bits(O.forward(ref, x)))
    # mismatched idx/val -> nothing usable -> error
    with pytest.raises(hip.SeedDeltaError):
        hip.SeedDeltaWeight(Payload(128, 256, d_idx=p.d_idx, d_val=p.d_val[:, :4].copy()))
    # B too small -> base dropped
    q = Payload(128, 256, d_idx=p.d_idx, d_val=p.d_val, base_d1=p.base_d1[:1].copy())
    assert hip.SeedDeltaWeight(q).kind == 0


def test_reference_constructor_mirrors(hip):
    p = make_payload(96, 160, scheme="block", K=32, block=16, base=True, seed=6)
    x = make_x(96, 2)
    y = hip.llama_seeddelta_mul_mat_base_block(x, None, p.base_d1, p.base_d2, p.base_d3, p.base_perm1, None,
                                               p.b_idx, p.b_val, p.row_scale, flags=hip.FWD_EXACT_ORDER)
    assert np.array_equal(bits(y), bits(O.forward(p, x)))
    ctx = hip.SeedDeltaContext()
    assert ctx.try_add("blk.0.ffn_gate.weight", p)
    assert ctx.mul_mat("blk.0.ffn_up.weight", x) is None          # not registered -> dense path
    check_fast(p, x, ctx.mul_mat("blk.0.ffn_gate.weight", x), O.forward(p, x))


# ---- full-size configs: size-independent properties + sampled oracle rows ------------------------
@pytest.mark.parametrize("cfg", ["cfg3_mistral_down_block32_k1536", "cfg2_gemma4b_gate_coo_k128",
                                 "cfg4_mistral_gate_block16_k128", "cfg1_gemma1b_down_block16_k128"])
def test_full_size_config_properties(hip, cfg):
    p = config_payload(cfg)
    w = hip.SeedDeltaWeight(p)
    x = make_x(p.n_in, 2, seed=5)
    y = w.forward_host(x)
    # (1) oracle on the whole tensor (the CPU finishes these in well under a second per token)
    y_ref = O.forward(p, x, nth=8)
    check_fast(p, x, y, y_ref)
    # (2) linearity: f(a*x1 + x2) == a*f(x1) + f(x2) within rounding
    a = np.float32(0.5)
    xc = (a * x[0] + x[1])[None, :].astype(np.float32)
    yc = w.forward_host(xc)
    lin = a * y[0] + y[1]
    scale = np.abs(y).max()
    assert np.abs(yc[0] - lin).max() <= 1e-4 * scale
    # (3) zero input -> zero output; deterministic across calls
    assert not w.forward_host(np.zeros((1, p.n_in), np.float32)).any()
    assert np.array_equal(bits(w.forward_host(x)), bits(y))
    # (4) base term bit-exact at full size: drop the residual by marking every index invalid
    q = Payload(p.n_in, p.n_out, row_scale=p.row_scale, base_d1=p.base_d1, base_d2=p.base_d2,
                base_d3=p.base_d3, base_perm1=p.base_perm1)
    if p.kind == 3:
        q.b_idx, q.b_val = np.full_like(p.b_idx[:, :1], -1), p.b_val[:, :1].copy()
    else:
        q.d_idx, q.d_val = np.full_like(p.d_idx[:, :8], -1), p.d_val[:, :8].copy()
    yb = hip.SeedDeltaWeight(q).forward_host(x[:1], hip.FWD_NO_DIRECT)      # the FWHT form (the direct launch sums in another order)
    assert np.array_equal(bits(yb), bits(O.forward(q, x[:1])))


@pytest.mark.parametrize("cfg", ["cfg3_mistral_down_block32_k1536", "cfg2_gemma4b_gate_coo_k128",
                                 "cfg4_mistral_gate_block16_k128", "cfg4_mistral_down_coo_k128",
                                 "cfg1_gemma1b_gate_block16_k64", "cfg1_gemma1b_down_block16_k128"])
def test_fused_decode_matches_two_launch_path(hip, cfg):
    """Batch-1 decode with a base runs as one fused launch (base workgroups hand their output to the
    residual waves through epoch-tagged granules). Same arithmetic, so it must be bit-identical to the
    two-launch path, call after call with changing inputs (a stale granule would show up here)."""
    p = config_payload(cfg)
    w = hip.SeedDeltaWeight(p)
    xs = make_x(p.n_in, 6, seed=21)
    for i in range(6):
        x = xs[i:i + 1]
        y_fused = w.forward_host(x, hip.FWD_NO_DIRECT)      # the fused launch (by default one token may take the direct one)
        y_two = w.forward_host(x, hip.FWD_NO_FUSE)
        assert np.array_equal(bits(y_fused), bits(y_two)), f"{cfg} call {i}"
        assert w.nonfinite_count() == 0
    check_fast(p, xs[5:6], y_fused, O.forward(p, xs[5:6], nth=8))
    # many back-to-back launches without host synchronisation in between (device I/O)
    torch = pytest.importorskip("torch")
    xd = torch.from_numpy(xs).cuda()
    yd = torch.zeros((6, p.n_out), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    sp = torch.cuda.current_stream().cuda_stream
    for rep in range(50):
        for i in range(6):
            w.forward_device(xd[i].data_ptr(), p.n_in, 1, yd[i].data_ptr(), p.n_out, hip.FWD_NO_DIRECT, stream=sp)
    torch.cuda.synchronize()
    y_all = yd.cpu().numpy()
    for i in range(6):
        assert np.array_equal(bits(y_all[i:i + 1]), bits(w.forward_host(xs[i:i + 1], hip.FWD_NO_FUSE)))


@pytest.mark.parametrize("cfg", ["cfg3_mistral_down_block32_k1536", "cfg2_gemma4b_gate_coo_k128",
                                 "cfg1_gemma1b_gate_block16_k64"])
def test_fused_decode_under_graph_capture(hip, cfg):
    """The fused launch recorded into a HIP graph: a replay repeats its arguments, so captured launches
    use the reserved hand-off tag and their consumers clear the granules they read. Replays with
    changing inputs, replays back to back without synchronisation, and eager launches of the same
    weight in between must all be bit-identical to the two-launch path."""
    torch = pytest.importorskip("torch")
    p = config_payload(cfg)
    w = hip.SeedDeltaWeight(p)
    xs = make_x(p.n_in, 5, seed=33)
    want = [w.forward_host(xs[i:i + 1], hip.FWD_NO_FUSE) for i in range(5)]
    xd = torch.zeros((1, p.n_in), dtype=torch.float32, device="cuda")
    yd = torch.zeros((1, p.n_out), dtype=torch.float32, device="cuda")
    xe = torch.from_numpy(xs[4:5]).cuda()
    ye = torch.zeros((1, p.n_out), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        w.forward_device(xd.data_ptr(), p.n_in, 1, yd.data_ptr(), p.n_out, hip.FWD_NO_DIRECT, stream=side.cuda_stream)   # census, eager
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            w.forward_device(xd.data_ptr(), p.n_in, 1, yd.data_ptr(), p.n_out, hip.FWD_NO_DIRECT,
                             stream=torch.cuda.current_stream().cuda_stream)
        for i in range(4):
            xd.copy_(torch.from_numpy(xs[i:i + 1]), non_blocking=False)
            g.replay()
            if i == 2:                       # an eager launch of the same weight between two replays
                w.forward_device(xe.data_ptr(), p.n_in, 1, ye.data_ptr(), p.n_out, hip.FWD_NO_DIRECT, stream=side.cuda_stream)
            side.synchronize()
            assert np.array_equal(bits(yd.cpu().numpy()), bits(want[i])), f"{cfg} replay {i}"
        assert np.array_equal(bits(ye.cpu().numpy()), bits(want[4]))
        for _ in range(200):                 # back to back, no host synchronisation in between
            g.replay()
        side.synchronize()
        assert np.array_equal(bits(yd.cpu().numpy()), bits(want[3]))
    assert w.nonfinite_count() == 0


@pytest.mark.parametrize("cfg", ["cfg3_mistral_down_block32_k1536", "cfg2_gemma4b_gate_coo_k128", "cfg4_mistral_gate_block16_k128",
                                 "cfg4_mistral_down_coo_k128", "cfg1_gemma1b_gate_block16_k64", "cfg1_gemma1b_down_block16_k128"])
def test_small_batch_decode_is_one_launch_and_bit_identical(hip, cfg):
    """2..4 tokens (speculative / batched decode): base workgroups per token and one pass over the payload
    for all tokens, in one launch. Per token the arithmetic is the one-token kernel's, so every token's
    output equals its own batch-1 result through the fused launch bit for bit (SDH_FWD_NO_DIRECT: by default one token
    may take the direct launch, whose base term is summed in another order), call after call, with strided token columns as well; against the oracle the fast-path tolerance holds."""
    torch = pytest.importorskip("torch")
    p = config_payload(cfg)
    w = hip.SeedDeltaWeight(p)
    xs = make_x(p.n_in, 12, seed=77)
    single = np.concatenate([w.forward_host(xs[t:t + 1], hip.FWD_NO_DIRECT) for t in range(12)], axis=0)   # the fused one-token launch
    for T in (2, 3, 4):
        for rep in range(3):
            x = xs[rep * T:(rep + 1) * T]
            y = w.forward_host(x)
            assert np.array_equal(bits(y), bits(single[rep * T:(rep + 1) * T])), f"{cfg} T={T} call {rep}"
            assert w.nonfinite_count() == 0
        check_fast(p, x, y, O.forward(p, x, nth=8))
        check_fast(p, x, w.forward_host(x, hip.FWD_NO_FUSE), O.forward(p, x, nth=8))
    # device I/O with padded columns, back to back without synchronisation, then under graph capture
    ldx, ldy = p.n_in + 8, p.n_out + 4
    xd = torch.zeros((4, ldx), dtype=torch.float32, device="cuda")
    xd[:, :p.n_in] = torch.from_numpy(xs[4:8]).cuda()
    yd = torch.zeros((4, ldy), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(20):
            w.forward_device(xd.data_ptr(), ldx, 4, yd.data_ptr(), ldy, stream=side.cuda_stream)
        side.synchronize()
        assert np.array_equal(bits(yd.cpu().numpy()[:, :p.n_out]), bits(single[4:8]))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            w.forward_device(xd.data_ptr(), ldx, 4, yd.data_ptr(), ldy, stream=torch.cuda.current_stream().cuda_stream)
        xd[:, :p.n_in] = torch.from_numpy(xs[8:12]).cuda()
        for _ in range(3):
            g.replay()
        side.synchronize()
        assert np.array_equal(bits(yd.cpu().numpy()[:, :p.n_out]), bits(single[8:12]))
    assert w.nonfinite_count() == 0


def test_device_io_path_matches_host_io(hip):
    torch = pytest.importorskip("torch")
    p = make_payload(1024, 512, scheme="block", K=64, block=32, base=True, seed=8)
    w = hip.SeedDeltaWeight(p)
    x = make_x(1024, 4)
    y_host = w.forward_host(x)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty((4, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    w.forward_device(xd.data_ptr(), 1024, 4, yd.data_ptr(), 512, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(bits(yd.cpu().numpy()), bits(y_host))


@pytest.mark.parametrize("scheme,K,block", [("coo", 64, 16), ("coo", 20, 16), ("block", 64, 16), ("block", 96, 32), ("block", 24, 12)])
@pytest.mark.parametrize("base", [False, True])
@pytest.mark.parametrize("dims", [(256, 1024), (1024, 256), (1152, 864), (2500, 300)])
@pytest.mark.parametrize("T", [6, 8, 9, 37, 130])
def test_prefill_spmm_is_bit_exact(hip, scheme, K, block, base, dims, T):
    """T >= 6 takes the token-tiled SpMM kernel: a lane per (row, token) walks the entries in the
    reference's order with separately rounded multiply and add, so the fast path itself is
    bit-identical to the CPU op; the per-token row kernels (SDH_FWD_NO_SPMM) stay within tolerance."""
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme=scheme, K=K, block=block, base=base, d13_random=base, seed=K + n_in + T)
    x = make_x(n_in, T, seed=T)
    y_ref = O.forward(p, x, nth=8)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x)
    assert np.array_equal(bits(y), bits(y_ref)), f"max abs diff {np.abs(y - y_ref).max()}"
    check_fast(p, x, w.forward_host(x, hip.FWD_NO_SPMM), y_ref)


ROW_KERNEL_CASES = [
    # scheme, K, block, n_in, n_out, idx dtype, val dtype
    ("coo", 64, 16, 1152, 864, np.int16, np.float16),
    ("coo", 256, 16, 1152, 864, np.int16, np.float16),
    ("block", 64, 16, 1152, 864, np.int16, np.float16),
    ("block", 512, 32, 2048, 320, np.int16, np.float16),        # one step, four tokens in the LDS
    ("block", 1024, 32, 2048, 320, np.int16, np.float16),       # two steps
    ("block", 1536, 32, 2048, 320, np.int16, np.float16),       # three steps
    ("block", 2048, 32, 2048, 320, np.int16, np.float16),       # four steps
    ("block", 1536, 32, 10240, 200, np.int16, np.float16),      # long input: two tokens in the LDS
    ("block", 2048, 64, 12288, 100, np.int16, np.float16),
    ("block", 768, 16, 2048, 320, np.int32, np.float32),        # fp32 values, int32 block indices
    ("block", 1536, 32, 10240, 200, np.int32, np.float32),
    ("block", 2048, 32, 2048, 320, np.int16, np.float32),       # fp32 values at four steps: per-token kernel
]


@pytest.mark.parametrize("case", ROW_KERNEL_CASES, ids=lambda c: f"{c[0]}-K{c[1]}-b{c[2]}-{c[3]}x{c[4]}-{np.dtype(c[5]).name}-{np.dtype(c[6]).name}")
@pytest.mark.parametrize("base", [False, True])
@pytest.mark.parametrize("T", [5, 9, 10, 11, 13, 130])
def test_row_kernels_many_tokens_equal_one_token_calls(hip, case, base, T):
    """Calls that stay on the row kernels (SpMM off or not applicable). From 10 tokens on, block payloads with
    rows longer than 256 entries keep the activations of two or four tokens in the LDS and walk the rows with
    a different reduction data path; each token's arithmetic and the pairing of the lane sums are the
    one-token kernel's, so every token's output equals its own one-token call bit for bit — ragged token
    counts (last tile with 1..3 tokens), bad block indices and a ragged last block included."""
    scheme, K, block, n_in, n_out, idx_dtype, val_dtype = case
    kw = dict(idx_dtype=idx_dtype, val_dtype=val_dtype)
    if val_dtype is np.float32:
        kw.update(rs_dtype=np.float32, base_dtype=np.float32, perm_dtype=np.int32)
    p = make_payload(n_in, n_out, scheme=scheme, K=K, block=block, base=base, d13_random=base, seed=K + T, **kw)
    if scheme == "block":
        p.b_idx = p.b_idx.copy(); p.b_idx[::7, 1] = -1; p.b_idx[::5, 2] = n_in // block + 3
    x = make_x(n_in, T, seed=T + 1)
    w = hip.SeedDeltaWeight(p)
    flags = hip.FWD_NO_SPMM | hip.FWD_NO_FUSE
    pick = sorted(set(range(min(T, 6))) | {T - 1, T - 2})
    single = np.concatenate([w.forward_host(x[t:t + 1], hip.FWD_NO_FUSE) for t in pick], axis=0)
    y = w.forward_host(x, flags)
    assert np.array_equal(bits(y[pick]), bits(single))
    check_fast(p, x, y, O.forward(p, x, nth=8))
    assert w.nonfinite_count() == 0


def test_prefill_spmm_dtypes_and_bad_indices(hip):
    for idx_dtype, val_dtype in ((np.int32, np.float32), (np.int16, np.float32), (np.int32, np.float16)):
        for scheme in ("coo", "block"):
            p = make_payload(512, 384, scheme=scheme, K=64, block=16, base=True, idx_dtype=idx_dtype, val_dtype=val_dtype,
                             rs_dtype=np.float32, base_dtype=np.float32, perm_dtype=np.int32, seed=3)
            if scheme == "coo":
                p.d_idx = p.d_idx.copy(); p.d_idx[::7, 3] = -1; p.d_idx[::5, 9] = 600
            else:
                p.b_idx = p.b_idx.copy(); p.b_idx[::7, 1] = -1; p.b_idx[::5, 2] = 40
            x = make_x(512, 19)
            w = hip.SeedDeltaWeight(p)
            assert np.array_equal(bits(w.forward_host(x)), bits(O.forward(p, x, nth=4))), (idx_dtype, val_dtype, scheme)


@pytest.mark.parametrize("dims", [(256, 1024), (1024, 256), (1152, 864), (864, 1152), (320, 1000), (1000, 320)])
@pytest.mark.parametrize("scheme", ["coo", "block"])
@pytest.mark.parametrize("T", [5, 40, 130, 300])
def test_prefill_w0_mfma_path(hip, dims, scheme, T):
    """SDH_FWD_W0_MFMA: the base as a tiled GEMM (W0 and f16 x tiles in LDS; L >= 128, more than 4 tokens) or, for
    short base blocks, as W0 fragments regenerated in registers — closed form in fp32, contracted in f16 on MFMA with
    fp32 accumulation. Shapes with ragged row tiles (n_out % 128), ragged K slabs (n_in % 64), partial last base
    blocks and ragged token tiles (T % 256), d1/d3 not all ones. f16 operands: each product carries up to
    2 * 2^-11 relative error, so |y - y_ref| <= 1.5e-3 * |rs| * sum_i |W0[o,i] x[i]| + the fast-path
    tolerance of the residual."""
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme=scheme, K=32, block=16, base=True, d13_random=True, seed=n_in + T)
    x = make_x(n_in, T, seed=T + 1)
    y_ref = O.forward(p, x, nth=8)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x, hip.FWD_W0_MFMA)
    W0 = np.abs(O.w0_tile(p, 0, n_out, 0, n_in).astype(np.float64))
    rs = np.abs(p.row_scale.astype(np.float64))[None, :] if p.row_scale is not None else 1.0
    bound = 1.5e-3 * rs * (np.abs(x.astype(np.float64)) @ W0.T) + fast_tolerance(p, x, y_ref)
    err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    assert np.all(err <= bound), f"max err/bound = {np.max(err / bound):.3g}"
    assert np.max(err / bound) > 1e-4      # the MFMA path really ran (it is not bit-exact)


def test_null_stream_is_the_default_stream(hip):
    """Device I/O with stream NULL runs on HIP's legacy default stream (include/seeddelta_hip.h, Streams): x is
    produced and y consumed on torch's default stream with NO host synchronisation around the call. A library
    that re-routed NULL to a private stream would read x before the producer finished."""
    torch = pytest.importorskip("torch")
    p = config_payload("cfg4_mistral_down_coo_k128")
    w = hip.SeedDeltaWeight(p)
    xs = make_x(p.n_in, 4, seed=9)
    want = [w.forward_host(xs[i:i + 1]) for i in range(4)]
    big = torch.randn((8192, 8192), device="cuda")
    xd = torch.zeros((1, p.n_in), dtype=torch.float32, device="cuda")
    yd = torch.zeros((1, p.n_out), dtype=torch.float32, device="cuda")
    src = [torch.from_numpy(xs[i:i + 1]).cuda() for i in range(4)]
    outs = []
    torch.cuda.synchronize()
    for i in range(4):
        _ = big @ big                                  # keeps the default stream busy ahead of the producer
        xd.copy_(src[i] * 1.0)                         # producer of x, default stream
        w.forward_device(xd.data_ptr(), p.n_in, 1, yd.data_ptr(), p.n_out, stream=0)
        outs.append(yd.clone())                        # consumer of y, default stream
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(bits(outs[i].cpu().numpy()), bits(want[i])), i
    assert hip.library_stream() != 0


DIRECT_CFGS = ["cfg3_mistral_down_block32_k1536", "cfg2_gemma4b_gate_coo_k128", "cfg4_mistral_gate_coo_k128",
               "cfg4_mistral_gate_block16_k128", "cfg1_gemma1b_gate_block16_k64"]


@pytest.mark.parametrize("cfg", DIRECT_CFGS)
def test_direct_decode_one_token(hip, cfg):
    """One token, closed-form base without d1: by default every 16-row workgroup forms the base term of its own rows
    (csrc/sdh_direct.hip) instead of waiting for a base workgroup's hand-off. fp32 in another summation order than the
    FWHT: the fast path's tolerance against the oracle, not the fused launch's bits (a difference in the low bits is
    what shows the launch ran); call after call with changing inputs, back to back on a stream, and replayed from a
    HIP graph the results are its own, bit for bit. SDH_FWD_NO_DIRECT keeps the fused launch."""
    torch = pytest.importorskip("torch")
    p = config_payload(cfg)
    w = hip.SeedDeltaWeight(p)
    xs = (1.5 * make_x(p.n_in, 6, seed=41)).astype(np.float32)
    ys = []
    for i in range(6):
        x = xs[i:i + 1]
        y = w.forward_host(x)
        assert hip.last_decode_kind() == 1, "the direct launch did not run"
        y_ref = O.forward(p, x, nth=8)
        check_fast(p, x, y, y_ref)
        y_fused = w.forward_host(x, hip.FWD_NO_DIRECT)
        assert hip.last_decode_kind() == 2
        assert np.array_equal(bits(y_fused), bits(w.forward_host(x, hip.FWD_NO_FUSE)))
        assert not np.array_equal(bits(y), bits(y_fused))
        assert np.all(np.abs(y.astype(np.float64) - y_fused) <= fast_tolerance(p, x, y_ref))
        assert w.nonfinite_count() == 0
        ys.append(y)
    xd = torch.from_numpy(xs).cuda()
    yd = torch.zeros((6, p.n_out), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for rep in range(30):
            for i in range(6):
                w.forward_device(xd[i].data_ptr(), p.n_in, 1, yd[i].data_ptr(), p.n_out, stream=side.cuda_stream)
        side.synchronize()
        y_all = yd.cpu().numpy()
        for i in range(6):
            assert np.array_equal(bits(y_all[i:i + 1]), bits(ys[i]))
        xg = torch.zeros((1, p.n_in), dtype=torch.float32, device="cuda")
        yg = torch.zeros((1, p.n_out), dtype=torch.float32, device="cuda")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            w.forward_device(xg.data_ptr(), p.n_in, 1, yg.data_ptr(), p.n_out, stream=torch.cuda.current_stream().cuda_stream)
        for i in range(3):
            xg.copy_(xd[i:i + 1])
            g.replay()
            side.synchronize()
            assert np.array_equal(bits(yg.cpu().numpy()), bits(ys[i])), f"{cfg} replay {i}"


@pytest.mark.parametrize("kw", [dict(scheme="coo", K=64, d13_random=True), dict(scheme="block", K=64, block=16, d13_random=True),
                                dict(scheme="coo", K=64, perm=False), dict(scheme="coo", K=40)])
@pytest.mark.parametrize("dims", [(2816, 1024), (1024, 2816), (4096, 4096)])
def test_direct_decode_declines_or_holds(hip, kw, dims):
    """Payloads the direct launch does not take (d1 present; wide with d3; rows that need the scalar kernel) keep the fused
    launch's bits; the ones it takes (no permutation = identity, square shapes) hold the fast path's tolerance."""
    p = make_payload(dims[0], dims[1], base=True, seed=8, **kw)
    w = hip.SeedDeltaWeight(p)
    x = (2.0 * make_x(p.n_in, 1, seed=12)).astype(np.float32)
    y = w.forward_host(x)
    check_fast(p, x, y, O.forward(p, x, nth=4))
    if kw.get("d13_random"):
        assert np.array_equal(bits(y), bits(w.forward_host(x, hip.FWD_NO_DIRECT)))
    assert w.nonfinite_count() == 0


def test_direct_decode_captured_without_a_warm_up_call(hip):
    """The direct launch needs nothing from an earlier eager call (its tables are built by sdh_weight_create, it has no
    residency census): the very first call on a fresh handle can be recorded into a HIP graph, and replays give the
    eager result of the same launch."""
    torch = pytest.importorskip("torch")
    p = config_payload("cfg4_mistral_gate_coo_k128")
    w = hip.SeedDeltaWeight(p)
    xs = make_x(p.n_in, 3, seed=55)
    xg = torch.zeros((1, p.n_in), dtype=torch.float32, device="cuda")
    yg = torch.full((1, p.n_out), float("nan"), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            w.forward_device(xg.data_ptr(), p.n_in, 1, yg.data_ptr(), p.n_out, stream=torch.cuda.current_stream().cuda_stream)
        assert hip.last_decode_kind() == 1
        outs = []
        for i in range(3):
            xg.copy_(torch.from_numpy(xs[i:i + 1]))
            g.replay()
            side.synchronize()
            outs.append(yg.cpu().numpy().copy())
    for i in range(3):
        assert np.array_equal(bits(outs[i]), bits(w.forward_host(xs[i:i + 1])))
        check_fast(p, xs[i:i + 1], outs[i], O.forward(p, xs[i:i + 1], nth=8))
