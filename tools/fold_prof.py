"""Driver for profiling the folded prefill contraction: python tools/fold_prof.py <workload> [reps] [flags]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llama_cpp_amd  # noqa: F401
from llama_cpp_amd import seeddelta
from llama_cpp_amd.payload import config_payload

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_mistral_gate_coo_k128"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else seeddelta.FWD_MFMA_FOLD
T = int(os.environ.get("T", "512"))
seeddelta.init(0)
p = config_payload(name)
w = seeddelta.SeedDeltaWeight(p)
x = torch.randn((T, p.n_in), dtype=torch.float32, device="cuda")
y = torch.empty((T, p.n_out), dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
for _ in range(reps):
    w.forward_device(x.data_ptr(), p.n_in, T, y.data_ptr(), p.n_out, flags=flags, stream=st.cuda_stream)
st.synchronize()
print("done", name, reps)
