#!/bin/bash
# same-box A/B of the scan leg: the in-tree library against tools/bin/variants/<name>/libseeddelta_hip.so, 3 runs each, interleaved
set -o pipefail
mkdir -p gpurun_out
V=${1:-scan_d60}
for i in 1 2 3; do
  for which in new $V; do
    if [ $which = new ]; then unset SEEDDELTA_HIP_LIB; else export SEEDDELTA_HIP_LIB=$PWD/tools/bin/variants/$which/libseeddelta_hip.so; fi
    timeout -k 10 300 python -u bench.py --no-ffn --no-cpu-baseline --no-roofline --steps 64 --warmup 32 > gpurun_out/ab_$which.log 2>&1 || { tail -5 gpurun_out/ab_$which.log; exit 1; }
    python -c "import json;d=json.loads(open('gpurun_out/ab_$which.log').read().strip().splitlines()[-1]);s=d['scan'];print('$which', round(s['value']), round(s['with_payload_emission']['value']))"
  done
done
