"""Time the folded prefill call per workload with events on the launch stream (library span hooks).
python tools/fold_time.py [flags] ; SEEDDELTA_HIP_LIB selects an A/B build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llama_cpp_amd  # noqa: F401
from llama_cpp_amd import seeddelta
from llama_cpp_amd.payload import config_payload

flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else seeddelta.FWD_MFMA_FOLD
T = int(os.environ.get("T", "512"))
seeddelta.init(0)
st = torch.cuda.Stream()
out = []
for name in os.environ.get("WL", "cfg4_mistral_gate_coo_k128,cfg4_mistral_down_coo_k128,cfg3_mistral_down_block32_k1536").split(","):
    p = config_payload(name)
    w = seeddelta.SeedDeltaWeight(p)
    x = torch.randn((T, p.n_in), dtype=torch.float32, device="cuda")
    y = torch.empty((T, p.n_out), dtype=torch.float32, device="cuda")
    fn = lambda: w.forward_device(x.data_ptr(), p.n_in, T, y.data_ptr(), p.n_out, flags=flags, stream=st.cuda_stream)
    for _ in range(int(os.environ.get("WARM", "300"))):
        fn()
    st.synchronize()
    seeddelta.profile_span_begin(st.cuda_stream)
    for _ in range(30):
        fn()
    ms, _ = seeddelta.profile_span_end(st.cuda_stream)
    out.append("%s %.1f us" % (name.split("_", 1)[1], ms / 30 * 1e3))
    w.close()
print(os.environ.get("SEEDDELTA_HIP_LIB", "default").split("/")[-2] if os.environ.get("SEEDDELTA_HIP_LIB") else "default", " | ".join(out))
