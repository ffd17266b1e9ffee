"""us per one-token call, direct launch against the fused one (SDH_FWD_NO_DIRECT), for ad-hoc shapes:
    python3 tools/direct_time.py n_in n_out scheme K [block]   (8 rotating payloads, 2000 calls on one stream)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import llama_cpp_amd.seeddelta as hip
from llama_cpp_amd.payload import make_payload, make_x

hip.init(0)
shapes = [tuple(sys.argv[1:])] if len(sys.argv) > 4 else [("1152", "6912", "block", "128", "16"), ("1152", "6912", "coo", "128"), ("2048", "8192", "coo", "256"),
                                                          ("2560", "10240", "block", "128", "16"), ("3072", "8192", "coo", "64"), ("6912", "1152", "block", "512", "32")]
for sh in shapes:
    n_in, n_out, scheme, K = int(sh[0]), int(sh[1]), sh[2], int(sh[3])
    block = int(sh[4]) if len(sh) > 4 else 16
    ws = [hip.SeedDeltaWeight(make_payload(n_in, n_out, scheme=scheme, K=K, block=block, base=True, seed=70 + i)) for i in range(8)]
    x = make_x(n_in, 1, seed=3)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros((1, n_out), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    res = {}
    for fl, nm in ((0, "direct"), (hip.FWD_NO_DIRECT, "fused")):
        with torch.cuda.stream(st):
            for i in range(4000): ws[i % 8].forward_device(xd.data_ptr(), n_in, 1, yd.data_ptr(), n_out, fl, stream=st.cuda_stream)
            st.synchronize(); t0 = time.perf_counter()
            for i in range(4000): ws[i % 8].forward_device(xd.data_ptr(), n_in, 1, yd.data_ptr(), n_out, fl, stream=st.cuda_stream)
            st.synchronize(); res[nm] = (time.perf_counter() - t0) / 4000 * 1e6
        if nm == "direct": kind = hip.last_decode_kind()
    print(f"{n_in}->{n_out} {scheme} K={K} block={block}: kind {kind}  default {res['direct']:.2f} us  NO_DIRECT {res['fused']:.2f} us", flush=True)
