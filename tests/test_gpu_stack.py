"""The persistent FFN stack (sdh_ffn_stack_*, csrc/sdh_stack.hip): n gated FFN blocks at T = 1 in one launch, every
hand-off on the device. Bar: bit-identical to the same layers evaluated one by one with sdh_ffn_forward on the fast
path (which tests/test_gpu_ffn.py pins to the reference's ggml graph), and within the propagated fast-path tolerance
of the oracle's chained FFN (node sequence of build_ffn, src/llama-graph.cpp:826-932)."""
import numpy as np
import pytest

import oracle_py as O
from conftest import bits
from llama_cpp_amd.payload import make_payload, make_x

pytestmark = pytest.mark.gpu


def make_layers(hip, n_layers, n_embd, n_ff, scheme, K, block=16, seed=100, d13_random=False):
    ps, ws = [], []
    for l in range(n_layers):
        kw = dict(scheme=scheme, K=K, block=block, base=True, d13_random=d13_random)
        trip = (make_payload(n_embd, n_ff, seed=seed + 3 * l, **kw), make_payload(n_embd, n_ff, seed=seed + 3 * l + 1, **kw),
                make_payload(n_ff, n_embd, seed=seed + 3 * l + 2, **kw))
        # keep the chain's activations O(1): scale the down values so that |x_{l+1}| ~ |x_l|
        ps.append(trip)
        ws.append(tuple(hip.SeedDeltaWeight(p) for p in trip))
    return ps, ws


def chain_node_by_node(hip, torch, ws, x, act):
    """the same layers through sdh_ffn_forward, one call per layer, device buffers. SDH_FWD_NO_DIRECT: the stack kernel's base
    terms come from base workgroups running the FWHT, as in the fused launches; the direct one-token launches that
    sdh_ffn_forward takes by default sum the base term in another order (same bound, other low bits)."""
    n_embd = x.shape[1]
    a = torch.from_numpy(x).cuda()
    b = torch.empty_like(a)
    outs = []
    for wg, wu, wd in ws:
        hip.ffn_forward_device(wg, wu, wd, act, a.data_ptr(), n_embd, 1, b.data_ptr(), n_embd, hip.FWD_NO_DIRECT, 0)
        torch.cuda.synchronize()
        outs.append(b.cpu().numpy().copy())
        a, b = b, a
    return outs


@pytest.mark.parametrize("dims,scheme,K,block", [((1024, 2304), "coo", 32, 16), ((1024, 2304), "block", 64, 16),
                                                 ((2048, 5632), "coo", 128, 16), ((4096, 14336), "coo", 128, 16),
                                                 ((4096, 14336), "block", 128, 16), ((2560, 10240), "coo", 128, 16),
                                                 ((1152, 6912), "block", 64, 16), ((4096, 14336), "coo", 64, 16)])
@pytest.mark.parametrize("act", [0, 1])
def test_stack_is_bit_identical_to_layer_by_layer(hip, dims, scheme, K, block, act):
    torch = pytest.importorskip("torch")
    n_embd, n_ff = dims
    n_layers = 3
    ps, ws = make_layers(hip, n_layers, n_embd, n_ff, scheme, K, block, seed=7 * K + n_embd, d13_random=(n_embd == 1024))
    st = hip.FfnStack(ws)
    xd = torch.empty((1, n_embd), dtype=torch.float32, device="cuda")
    yd = torch.empty((1, n_embd), dtype=torch.float32, device="cuda")
    for call in range(3):                                   # tags advance call after call; inputs change
        x = (2.0 * make_x(n_embd, 1, seed=40 + call)).astype(np.float32)
        want = chain_node_by_node(hip, torch, ws, x, act)
        xd.copy_(torch.from_numpy(x))
        yd.fill_(float("nan"))
        torch.cuda.synchronize()
        st.forward_device(act, xd.data_ptr(), yd.data_ptr())
        nf, timeouts = st.status()
        assert timeouts == 0 and nf == 0
        y = yd.cpu().numpy()
        assert np.array_equal(bits(y), bits(want[-1])), (dims, scheme, call, float(np.abs(y - want[-1]).max()))
    # prefixes of the stack: 1 and 2 layers
    for n in (1, 2):
        sub = hip.FfnStack(ws[:n])
        sub.forward_device(act, xd.data_ptr(), yd.data_ptr())
        assert sub.status() == (0, 0)
        assert np.array_equal(bits(yd.cpu().numpy()), bits(want[n - 1])), (dims, scheme, n)
        sub.close()
    # against the oracle's chain (fast-path tolerance, propagated through the layers: coarse relative bound)
    xo = x
    for (pg, pu, pd) in ps:
        _, xo = O.ffn_forward(pg, pu, pd, xo, act)
    assert np.abs(y - xo).max() <= 2e-3 * max(np.abs(xo).max(), 1e-30)
    st.close()
    for t in ws:
        for w in t:
            w.close()


def test_stack_back_to_back_without_host_sync(hip):
    torch = pytest.importorskip("torch")
    n_embd, n_ff = 4096, 14336
    ps, ws = make_layers(hip, 4, n_embd, n_ff, "coo", 128, seed=900)
    st = hip.FfnStack(ws)
    xs = make_x(n_embd, 6, seed=5)
    xd = torch.from_numpy(xs).cuda()
    yd = torch.zeros((6, n_embd), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(20):
        for i in range(6):
            st.forward_device(0, xd[i].data_ptr(), yd[i].data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    assert st.status() == (0, 0)
    y = yd.cpu().numpy()
    for i in range(6):
        want = chain_node_by_node(hip, torch, ws, xs[i:i + 1], 0)[-1]
        assert np.array_equal(bits(y[i:i + 1]), bits(want)), i
    st.close()


def test_stack_declines_shapes_it_is_not_built_for(hip):
    ps, ws = make_layers(hip, 1, 1024, 2304, "coo", 32)
    nb = hip.SeedDeltaWeight(make_payload(1024, 2304, scheme="coo", K=32, base=False, seed=1))
    with pytest.raises(hip.SeedDeltaError, match=r"\(-4\)"):
        hip.FfnStack([(nb, ws[0][1], ws[0][2])])
    with pytest.raises(hip.SeedDeltaError, match=r"\(-4\)"):
        hip.FfnStack([(ws[0][0], ws[0][0], ws[0][2])])          # gate and up one handle
    big = [hip.SeedDeltaWeight(make_payload(a, b, scheme="coo", K=512, base=True, seed=3)) for a, b in ((4096, 14336), (4096, 14336), (14336, 4096))]
    with pytest.raises(hip.SeedDeltaError, match="register-resident"):      # rows this long do not fit the register-resident passes
        hip.FfnStack([tuple(big)])
    odd = hip.SeedDeltaWeight(make_payload(1024, 2304, scheme="coo", K=20, base=True, seed=2))   # K not a multiple of 8
    with pytest.raises(hip.SeedDeltaError, match=r"\(-4\)"):
        hip.FfnStack([(odd, odd, ws[0][2])])


def test_stack_call_under_contention_reports_its_own_timeouts(hip):
    """The stack kernel is only correct while all its workgroups are resident at once (header: one tenant per GPU while
    a stack call runs). A call made while another stream saturates the GPU either still finds its workgroups co-resident
    (timeouts == 0: the output must be right) or has hand-off waits time out (bounded at 500 us each): then
    sdh_ffn_stack_status says so FOR THAT CALL — and the next call, on an idle GPU, reports 0 again and is right."""
    torch = pytest.importorskip("torch")
    n_embd, n_ff = 4096, 14336
    ps, ws = make_layers(hip, 3, n_embd, n_ff, "coo", 128, seed=1300)
    st = hip.FfnStack(ws)
    x = make_x(n_embd, 1, seed=9)
    want = chain_node_by_node(hip, torch, ws, x, 0)[-1]
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((1, n_embd), float("nan"), dtype=torch.float32, device="cuda")
    s, side = torch.cuda.Stream(), torch.cuda.Stream()
    st.forward_device(0, xd.data_ptr(), yd.data_ptr(), stream=s.cuda_stream)        # census + a clean call
    assert st.status() == (0, 0)
    a = torch.randn((8192, 8192), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(12):
            b = a @ a                                                               # ~10 ms each: the GPU is full for the call below
    with torch.cuda.stream(s):                                                      # on the call's stream: ordered before it, whatever the other streams hold
        yd.fill_(float("nan"))
    st.forward_device(0, xd.data_ptr(), yd.data_ptr(), stream=s.cuda_stream)
    nf, timeouts = st.status()
    y = yd.cpu().numpy()
    if timeouts == 0:
        assert np.array_equal(bits(y), bits(want))
    else:
        print(f"\ncontended stack call: {timeouts} hand-off waits timed out, {nf} non-finite node outputs")
    side.synchronize()
    del b
    with torch.cuda.stream(s):
        yd.fill_(float("nan"))
    st.forward_device(0, xd.data_ptr(), yd.data_ptr(), stream=s.cuda_stream)
    assert st.status() == (0, 0)                                                    # per call, not cumulative
    assert np.array_equal(bits(yd.cpu().numpy()), bits(want))
    st.close()
