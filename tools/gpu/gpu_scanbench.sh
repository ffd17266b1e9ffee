#!/bin/bash
# scan tests, then the scan leg of the bench (optionally A/B with env variants given as arguments: "NAME=VAL")
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_scan.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_scan.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_scan.log; [ $rc -ne 0 ] && exit $rc
run() {
  tag=$1; shift
  env "$@" timeout -k 10 900 python -u bench.py --no-ffn --no-cpu-baseline --steps 512 --warmup 64 > gpurun_out/bench_scan_$tag.log 2>&1 || { tail -5 gpurun_out/bench_scan_$tag.log; return 1; }
  python -c "import json;d=json.loads(open('gpurun_out/bench_scan_$tag.log').read().strip().splitlines()[-1]);s=d['scan'];print('$tag', {k:s[k] for k in ('value','candidates','seconds','with_payload_emission')})"
}
run default SDH_NOP=1 || exit 1
for v in "$@"; do run "${v%%=*}" "$v" || exit 1; done
