"""dev: in-kernel timeline of the persistent FFN stack (needs the -DSDH_STK_STAMPS build:
SEEDDELTA_HIP_LIB=tools/bin/variants/stkstamps/libseeddelta_hip.so python tools/stack_stamps.py)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import llama_cpp_amd  # noqa: E402,F401
from llama_cpp_amd import seeddelta  # noqa: E402
from llama_cpp_amd.payload import make_payload  # noqa: E402

NL = int(os.environ.get("NL", "8"))
n_embd, n_ff, K = 4096, 14336, 128
seeddelta.init(0)
layers = []
for i in range(NL):
    ps = [make_payload(n_embd, n_ff, scheme="coo", K=K, base=True, seed=9000 + 3 * i),
          make_payload(n_embd, n_ff, scheme="coo", K=K, base=True, seed=9001 + 3 * i),
          make_payload(n_ff, n_embd, scheme="coo", K=K, base=True, seed=9002 + 3 * i)]
    ps[2].row_scale = (ps[2].row_scale.astype(np.float32) * np.float32(0.05)).astype(np.float16)
    layers.append(tuple(seeddelta.SeedDeltaWeight(p) for p in ps))
st = seeddelta.FfnStack(layers)
x = torch.randn((1, n_embd), device="cuda")
y = torch.empty_like(x)
s = torch.cuda.Stream()
for _ in range(5):
    st.forward_device(0, x.data_ptr(), y.data_ptr(), 0, s.cuda_stream)
s.synchronize()
print("status", st.status())
buf = np.zeros((7, NL, 8), dtype=np.uint64)
lib = seeddelta.load_library()
lib.sdh_ffn_stack_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.sdh_ffn_stack_debug_stamps(st._h, buf.ctypes.data) == 0
t0 = buf[buf > 0].min()
rel = (buf.astype(np.int64) - int(t0)) / 100.0          # us
names = ["base gate b0", "base up b0", "base down b0", "down wg first", "down wg last", "gu wg first", "gu wg last"]
ev = {0: ["top", "input", "xform", "publ", "P1", "F1", "back"], 1: ["top", "h in", "sums", "VD in", "publ", "pref"], 2: ["top", "x stg", "sums", "V in", "publ", "pref"]}
for smp in range(7):
    kind = 0 if smp < 3 else (1 if smp < 5 else 2)
    print(names[smp], ev[kind])
    for l in range(NL):
        print("  L%-2d " % l + " ".join("%8.2f" % rel[smp, l, k] for k in range(len(ev[kind]))))
print("per-layer period (gate base publish to publish):", np.diff(rel[0, :, 3]))
# shader clock the base workgroup saw: slot 7 holds s_memtime beside the real-time stamp of slot 0
for smp in range(3):
    dt_clk = np.diff(buf[smp, :, 7].astype(np.int64))
    dt_rt = np.diff(buf[smp, :, 0].astype(np.int64))
    print(names[smp], "shader clock between layer tops, GHz:", np.round(dt_clk / np.maximum(dt_rt, 1) * 0.1, 3))
