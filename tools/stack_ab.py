"""dev: us per layer of the persistent FFN stack (Mistral-7B layers, COO K=128 + base) under the environment's
SDH_STACK_* knobs; one line of output. NL layers (default 32), REPS calls."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import llama_cpp_amd  # noqa: E402,F401
from llama_cpp_amd import seeddelta  # noqa: E402
from llama_cpp_amd.payload import make_payload  # noqa: E402

NL, REPS = int(os.environ.get("NL", "32")), int(os.environ.get("REPS", "40"))
SCHEME = os.environ.get("SCHEME", "coo")
n_embd, n_ff, K = 4096, 14336, 128
seeddelta.init(0)
layers = []
for i in range(NL):
    ps = [make_payload(n_embd, n_ff, scheme=SCHEME, K=K, block=16, base=True, seed=9000 + 3 * i),
          make_payload(n_embd, n_ff, scheme=SCHEME, K=K, block=16, base=True, seed=9001 + 3 * i),
          make_payload(n_ff, n_embd, scheme=SCHEME, K=K, block=16, base=True, seed=9002 + 3 * i)]
    ps[2].row_scale = (ps[2].row_scale.astype(np.float32) * np.float32(0.05)).astype(np.float16)
    layers.append(tuple(seeddelta.SeedDeltaWeight(p) for p in ps))
st = seeddelta.FfnStack(layers)
x = torch.randn((1, n_embd), device="cuda")
y = torch.empty_like(x)
s = torch.cuda.Stream()
for _ in range(3):
    st.forward_device(0, x.data_ptr(), y.data_ptr(), 0, s.cuda_stream)
s.synchronize()
seeddelta.profile_span_begin(s.cuda_stream)
t0 = time.perf_counter()
for _ in range(REPS):
    st.forward_device(0, x.data_ptr(), y.data_ptr(), 0, s.cuda_stream)
ms, _ = seeddelta.profile_span_end(s.cuda_stream)
s.synchronize()
host = (time.perf_counter() - t0) * 1e3
knobs = {k: v for k, v in os.environ.items() if k.startswith("SDH_STACK")}
print("us/layer %.2f (host %.2f) status %s %s %s" % (ms * 1e3 / (REPS * NL), host * 1e3 / (REPS * NL), st.status(), SCHEME, knobs), flush=True)
