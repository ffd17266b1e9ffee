"""Timeline of the direct decode launch (csrc/sdh_direct.hip built with -DSDH_DD_STAMPS): per workgroup, the first base
wave and the first row wave stamp the 100 MHz counter; printed relative to the earliest stamp of the launch, in us."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import llama_cpp_amd.seeddelta as hip
from llama_cpp_amd.payload import config_payload, make_x

hip.init(0)
lib = hip.load_library()
ws = [hip.SeedDeltaWeight(config_payload("cfg3_mistral_down_block32_k1536", seed=100 + i)) for i in range(4)]
p = config_payload("cfg3_mistral_down_block32_k1536")
x = make_x(p.n_in, 1, seed=3)
for rep in range(6):
    for w in ws:
        y = w.forward_host(x)
n_wg = p.n_out // 16
buf = np.zeros(n_wg * 16, dtype=np.uint64)
rc = lib.sdh_debug_direct_stamps(buf.ctypes.data_as(ctypes.c_void_p), n_wg)
assert rc == 0, rc
t = buf.reshape(n_wg, 2, 8).astype(np.float64)
t0 = t[:, :, 0].min()
t = (t - t0) / 100.0
names = {0: ["start", "x stored", "barrier 1", "segments done", "partials written", "barrier 2"],
         1: ["start", "x stored", "barrier 1", "row sums done", "barrier 2", "stored"]}
for role in (0, 1):
    print("base wave" if role == 0 else "row wave")
    for k, nm in enumerate(names[role]):
        v = t[:, role, k]
        print(f"  {nm:18s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
