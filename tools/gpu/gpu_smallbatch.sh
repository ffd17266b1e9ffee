#!/bin/bash
# decode-sized batches: us per step for T = 1, 2, 4, 8, 16 on the headline payload
set -o pipefail
mkdir -p gpurun_out
for t in ${TOKENS:-1 2 3 4 8}; do
  timeout -k 10 200 python bench.py --no-ffn --no-scan --no-cpu-baseline --steps 2048 --warmup 256 --tokens $t ${EXTRA} > gpurun_out/sb_$t.json 2> gpurun_out/sb_$t.err || { tail -3 gpurun_out/sb_$t.err; exit 1; }
  python -c "import json;d=json.loads(open('gpurun_out/sb_$t.json').read().strip().splitlines()[-1]);print('T=$t', round(d['ms_per_step']*1e3,2),'us/step', round(d['value']),'tok/s', (d.get('roofline') or {}).get('kernel'))"
done
