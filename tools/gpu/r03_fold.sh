#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_prefill_full.py -m gpu -x -q --timeout 300 --timeout-method thread -k "fold" > gpurun_out/r03_fold_tests.log 2>&1 || { tail -40 gpurun_out/r03_fold_tests.log; exit 1; }
tail -3 gpurun_out/r03_fold_tests.log
SDH_BENCH_DETAIL=$R/gpurun_out/r03_fold_bench_detail.json timeout -k 10 300 python -u bench.py --no-scan --no-ffn --no-stack --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03_fold_bench.json 2> gpurun_out/r03_fold_bench.err || { tail -20 gpurun_out/r03_fold_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_fold_bench_detail.json'))
for k,v in d['prefill']['workloads'].items():
    print(k, {t: round(v[t]['ms'],4) for t in ('default','w0_mfma','rows_lds','mfma_fold')}, 'fold mfma frac', round(v['mfma_fold']['mfma']['frac'],3))
PY
