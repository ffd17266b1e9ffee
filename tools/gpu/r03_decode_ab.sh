#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
run() { python -u bench.py --no-scan --no-ffn --no-stack --no-prefill --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', round(d['roofline']['avg_kernel_us'],3))"; }
SDH_BENCH_DETAIL=/tmp/d.json run default || exit 1
for v in ${VARIANTS}; do
  SDH_BENCH_DETAIL=/tmp/d.json SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so run $v || exit 1
done
for e in ${ENVS}; do
  SDH_BENCH_DETAIL=/tmp/d.json env ${e//,/ } python -u bench.py --no-scan --no-ffn --no-stack --no-prefill --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$e', round(d['roofline']['avg_kernel_us'],3))" || exit 1
done
