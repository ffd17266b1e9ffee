"""sdh_forward_noalloc (include/seeddelta_hip.h): the entry point for callers that cannot allocate. Its base-term
scratch is a static device array; results must equal sdh_forward's bit for bit. Collected last (test_zz_*) so
that a failure here cannot hide the parity tests."""
import numpy as np
import pytest

from conftest import bits
from llama_cpp_amd.payload import Payload, config_payload, make_payload, make_x

pytestmark = pytest.mark.gpu

SDH_ERR_INVALID, SDH_ERR_TOO_LARGE = -1, -5


def run_noalloc(torch, w, x, flags=0, stream=0):
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((x.shape[0], w.n_out), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rc = w.forward_device_noalloc(xd.data_ptr(), w.n_in, x.shape[0], yd.data_ptr(), w.n_out, flags, stream)
    torch.cuda.synchronize()
    return rc, yd.cpu().numpy()


@pytest.mark.parametrize("cfg", ["cfg4_mistral_down_coo_k128", "cfg4_mistral_gate_block16_k128", "cfg1_gemma1b_down_block16_k128"])
def test_noalloc_matches_forward_two_calls_in_a_row(hip, cfg):
    torch = pytest.importorskip("torch")
    p = config_payload(cfg)
    w = hip.SeedDeltaWeight(p)
    ref = hip.SeedDeltaWeight(p)                     # the existing entry point runs on a handle of its own
    assert hip.load_library().sdh_noalloc_scratch_bytes() == 8 * 1024 * 1024
    # first call: 700 tokens = more than one scratch-sized chunk (B*L*4 = 64 KiB per token -> 128 tokens per chunk)
    # second call: smaller (3 tokens), on the scratch the first call left behind
    free0 = None
    for T, seed in ((700, 1), (3, 2), (1, 3), (130, 4)):
        x = make_x(p.n_in, T, seed=seed)
        want = ref.forward_host(x, hip.FWD_NO_FUSE)
        if free0 is None:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
        rc, y = run_noalloc(torch, w, x)
        assert rc == 0, hip.load_library().sdh_last_error()
        assert np.array_equal(bits(y), bits(want)), (cfg, T, float(np.abs(y - want).max()))
        assert np.array_equal(bits(y), bits(ref.forward_host(x)))          # and the default path (fused launch at T <= 4)
        assert w.nonfinite_count() == 0
    # the handle never grew a workspace: free device memory is what it was before the first call
    # (torch's own allocations for xd / yd come from its cache or are returned to it; allow its block granularity)
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) <= 64 << 20
    # explicit stream and the verification order
    st = torch.cuda.Stream()
    x = make_x(p.n_in, 9, seed=8)
    rc, y = run_noalloc(torch, w, x, hip.FWD_EXACT_ORDER, st.cuda_stream)
    st.synchronize()
    assert rc == 0 and np.array_equal(bits(y), bits(ref.forward_host(x, hip.FWD_EXACT_ORDER)))
    w.close(); ref.close()


def test_noalloc_without_base_and_rejections(hip):
    torch = pytest.importorskip("torch")
    p = make_payload(512, 768, scheme="coo", K=32, base=False, seed=4)
    w = hip.SeedDeltaWeight(p)
    x = make_x(512, 20)
    rc, y = run_noalloc(torch, w, x)
    assert rc == 0 and np.array_equal(bits(y), bits(w.forward_host(x)))
    # host pointers / the MFMA base are not served by this entry point
    rc, _ = run_noalloc(torch, w, x, hip.FWD_HOST_IO)
    assert rc == SDH_ERR_INVALID
    rc, _ = run_noalloc(torch, w, x, hip.FWD_W0_MFMA)
    assert rc == SDH_ERR_INVALID
    w.close()
    # a base whose single-token term exceeds the scratch: B*L*4 = 2 * 2^20 * 4 = 8 MiB + ... > 8 MiB needs B*L > 2^21
    L, B = 1 << 20, 3
    q = Payload(64, 128, d_idx=np.zeros((128, 8), np.int16), d_val=np.ones((128, 8), np.float16),
                base_d1=np.ones((B, L), np.float16), base_d2=np.full((B, L), 1.0 / L, np.float16),
                base_perm1=np.tile(np.arange(L, dtype=np.int32), (B, 1)))
    wq = hip.SeedDeltaWeight(q)
    assert wq.kind == 2
    rc, _ = run_noalloc(torch, wq, make_x(64, 1))
    assert rc == SDH_ERR_TOO_LARGE
    assert b"static scratch" in hip.load_library().sdh_last_error()
    wq.close()
