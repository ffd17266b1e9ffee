#!/usr/bin/env python3
"""Time scan candidates at Mistral-7B dims on the GPU (dev tool; prints per-candidate device ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import llama_cpp_amd  # noqa
from llama_cpp_amd import seeddelta, scan as SC
from llama_cpp_amd.dense import make_dense

seeddelta.init(0)
t0 = time.time()
Wg = make_dense(4096, 14336, "q8_0", seed=1, structure=0.3)
Wd = make_dense(14336, 4096, "q8_0", seed=3, structure=0.3)
print(f"gen {time.time()-t0:.1f}s", flush=True)
dg, dd = SC.DenseWeight(Wg), SC.DenseWeight(Wd)
cases = [("gate coo K=12 nobase", dg, "ffn_gate", SC.ScanParams(K=12)),
         ("gate coo K=128 base rs", dg, "ffn_gate", SC.ScanParams(K=128, base=True, row_scale=True)),
         ("gate coo K=768 base rs", dg, "ffn_gate", SC.ScanParams(K=768, base=True, row_scale=True)),
         ("down coo K=1536 base rs", dd, "ffn_down", SC.ScanParams(K=1536, base=True, row_scale=True)),
         ("down block32 K=1536 base rs", dd, "ffn_down", SC.ScanParams(scheme="block", K=1536, block=32, base=True, row_scale=True)),
         ("gate block16 K=128 base rs", dg, "ffn_gate", SC.ScanParams(scheme="block", K=128, block=16, base=True, row_scale=True))]
for name, w, kind, p in cases:
    for rep in range(2):
        s = SC.ScanSession(p.seed)
        t0 = time.time()
        r = s.tensor(w, p, kind, 0, (dg, dg, dd) if (p.scheme == "coo" and not p.base) else None)
        wall = (time.time() - t0) * 1e3
        s.close()
    print(f"{name:32s} device {r['gpu_ms']:8.2f} ms  wall {wall:8.2f} ms  cos={r['em'][2]:.3f} cos_x={r['em_x'][2]:.3f}", flush=True)
