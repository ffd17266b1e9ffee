#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for t in 16 8 24 32; do
  timeout -k 10 300 python -u bench.py --no-ffn --no-cpu-baseline --steps 512 --warmup 64 --scan-threads $t > gpurun_out/bench_scan_t$t.log 2>&1 || { tail -5 gpurun_out/bench_scan_t$t.log; exit 1; }
  python -c "import json;d=json.loads(open('gpurun_out/bench_scan_t$t.log').read().strip().splitlines()[-1]);s=d['scan'];print($t, s['value'], s['with_payload_emission']['value'])"
done
