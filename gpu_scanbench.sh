#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_scan.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_scan.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_scan.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -u bench.py --no-ffn --no-cpu-baseline --steps 512 --warmup 64 > gpurun_out/bench_scan.log 2>&1 || { tail -5 gpurun_out/bench_scan.log; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/bench_scan.log').read().strip().splitlines()[-1]);s=d['scan'];print({k:s[k] for k in ('value','candidates','seconds','with_payload_emission')})"
