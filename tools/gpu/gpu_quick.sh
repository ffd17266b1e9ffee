#!/bin/bash
# all GPU tests, then the decode line of the bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8192 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -3 gpurun_out/bench_quick.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step']*1e3,3),'us', round(d['ffn_block']['us_per_layer'],2), round(d['scan']['value']))"
