#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for d in 3 0 6 10 16; do
  for t in 2 4; do
    SDH_DECODE_DELAY=$d timeout -k 10 200 python bench.py --no-ffn --no-scan --no-cpu-baseline --steps 4096 --warmup 256 --tokens $t --workload cfg4_mistral_gate_coo_k128 > gpurun_out/ntd.json 2> gpurun_out/ntd.err || { tail -3 gpurun_out/ntd.err; exit 1; }
    python -c "import json;d=json.loads(open('gpurun_out/ntd.json').read().strip().splitlines()[-1]);print('delay $d T=$t', round(d['ms_per_step']*1e3,2),'us')"
  done
done
