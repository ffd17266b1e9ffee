#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for q in 4 8 16 4 16; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python -u bench.py --no-ffn --no-cpu-baseline --steps 512 --warmup 64 > gpurun_out/bench_scan_q$q.log 2>&1 || { tail -5 gpurun_out/bench_scan_q$q.log; exit 1; }
  python -c "import json;d=json.loads(open('gpurun_out/bench_scan_q$q.log').read().strip().splitlines()[-1]);s=d['scan'];print('queues', $q, s['value'], s['with_payload_emission']['value'], d['value'])"
done
