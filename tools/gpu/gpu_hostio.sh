#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --host-io --no-scan --no-ffn --no-cpu-baseline --steps 2048 --warmup 256 > gpurun_out/bench_hostio.log 2>&1 || { tail -5 gpurun_out/bench_hostio.log; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/bench_hostio.log').read().strip().splitlines()[-1]);print(d['pcie_inclusive'])"
timeout -k 10 300 python -u - <<'PY'
import time, numpy as np
import llama_cpp_amd
from llama_cpp_amd import seeddelta as sd
from llama_cpp_amd.payload import make_payload
sd.init(0)
kw=dict(scheme="coo", K=128, base=True)
L=[]
for i in range(8):
    L.append([sd.SeedDeltaWeight(make_payload(4096,14336,seed=3*i,**kw)), sd.SeedDeltaWeight(make_payload(4096,14336,seed=3*i+1,**kw)), sd.SeedDeltaWeight(make_payload(14336,4096,seed=3*i+2,**kw))])
x=np.random.default_rng(0).standard_normal((1,4096)).astype(np.float32)
for g,u,d in L: sd.ffn_forward_host(g,u,d,x)
n=0; t=time.perf_counter()
while time.perf_counter()-t<2.0:
    for g,u,d in L: sd.ffn_forward_host(g,u,d,x)
    n+=len(L)
a=(time.perf_counter()-t)/n*1e6
def nodes(g,u,d,x):
    G=g.forward_host(x); U=u.forward_host(x); h=(G/(1+np.exp(-G))*U).astype(np.float32); return d.forward_host(h)
for g,u,d in L: nodes(g,u,d,x)
n=0; t=time.perf_counter()
while time.perf_counter()-t<2.0:
    for g,u,d in L: nodes(g,u,d,x)
    n+=len(L)
b=(time.perf_counter()-t)/n*1e6
print("FFN layer with host I/O (pageable, synchronous): one node %.1f us, three nodes + host activation %.1f us" % (a,b))
PY
