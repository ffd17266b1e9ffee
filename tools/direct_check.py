"""Development check of the direct decode launch (csrc/sdh_direct.hip): one token, default dispatch against
SDH_FWD_NO_DIRECT (the fused launch) and the oracle, with the fast path's tolerance; then the launch period of both on
the cfg3 shape (32 rotating payloads would need 420 MB: here 8, enough to leave L2)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_py as O
import llama_cpp_amd.seeddelta as hip
from llama_cpp_amd.payload import config_payload, make_payload, make_x
from test_gpu_forward import fast_tolerance

hip.init(0)
NO_DIRECT = 0x100
cases = [("cfg3", config_payload("cfg3_mistral_down_block32_k1536")),
         ("gate_coo", config_payload("cfg4_mistral_gate_coo_k128")),
         ("down_coo", config_payload("cfg4_mistral_down_coo_k128")),
         ("gate_b16", config_payload("cfg4_mistral_gate_block16_k128")),
         ("down_b16", config_payload("cfg4_mistral_down_block16_k128")),
         ("small_tall", make_payload(1024, 2816, scheme="coo", K=32, base=True, seed=3)),
         ("small_wide", make_payload(2816, 1024, scheme="block", K=64, block=16, base=True, seed=4)),
         ("d13", make_payload(2816, 1024, scheme="coo", K=64, base=True, d13_random=True, seed=5))]
bad = 0
for name, p in cases:
    x = (2.0 * make_x(p.n_in, 1, seed=11)).astype(np.float32)
    y_ref = O.forward(p, x, nth=8)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x)
    y0 = w.forward_host(x, NO_DIRECT)
    tol = fast_tolerance(p, x, y_ref)
    e = np.abs(y.astype(np.float64) - y_ref) / tol
    e0 = np.abs(y0.astype(np.float64) - y_ref) / tol
    same = np.array_equal(y.view(np.uint32), y0.view(np.uint32))
    print(f"{name:12s} direct err/tol {e.max():.3g}  fused err/tol {e0.max():.3g}  same bits {same}  nf {w.nonfinite_count()}", flush=True)
    if e.max() > 1 or not np.all(np.isfinite(y)):
        bad += 1
print("BAD" if bad else "OK", bad)
