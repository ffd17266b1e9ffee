#!/bin/bash
# K-grid scan: parity tests of the schedule call, the other scan tests, then the scan leg of the bench
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 500 python -u -m pytest $R/tests/test_gpu_scan_kgrid.py $R/tests/test_gpu_scan.py $R/tests/test_gpu_build.py -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/pytest_kgrid.log 2>&1
rc=$?; tail -15 $R/gpurun_out/pytest_kgrid.log; [ $rc -ne 0 ] && exit $rc
SDH_SCAN_TIMING=1 timeout -k 10 600 python -u $R/bench.py --no-ffn --no-stack --no-prefill --no-cpu-baseline > $R/gpurun_out/bench_scan_kgrid.log 2> $R/gpurun_out/bench_scan_kgrid.err || { tail -5 $R/gpurun_out/bench_scan_kgrid.err; exit 1; }
grep "scan handle" $R/gpurun_out/bench_scan_kgrid.err | tail -3
python3 -c "import json;d=json.loads(open('$R/gpurun_out/bench_scan_kgrid.log').read().strip().splitlines()[-1]);s=d['scan'];print({k:s[k] for k in ('value','candidates','seconds','with_payload_emission')})"
