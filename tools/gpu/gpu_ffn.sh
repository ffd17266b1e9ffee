#!/bin/bash
# FFN-block leg of the bench on its own
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u bench.py --no-scan --no-cpu-baseline --steps 1024 --warmup 128 > gpurun_out/bench_ffn.log 2>&1 || { tail -5 gpurun_out/bench_ffn.log; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/bench_ffn.log').read().strip().splitlines()[-1]);print(json.dumps(d['ffn_block'], indent=1))"
