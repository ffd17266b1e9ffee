#!/bin/bash
# two-pass gate|up launch: FFN tests, then the FFN prefill layer with and without it
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_ffn.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pair_tests.log 2>&1 || { tail -30 gpurun_out/pair_tests.log; exit 1; }
tail -3 gpurun_out/pair_tests.log
for i in 1 2; do
SDH_BENCH_DETAIL=$R/gpurun_out/pair_on_$i.json timeout -k 10 300 python -u bench.py --no-scan --no-stack --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/pair_on_$i.line 2> gpurun_out/pair_on.err || { tail -20 gpurun_out/pair_on.err; exit 1; }
SDH_FOLD_NO_PAIR=1 SDH_BENCH_DETAIL=$R/gpurun_out/pair_off_$i.json timeout -k 10 300 python -u bench.py --no-scan --no-stack --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/pair_off_$i.line 2> gpurun_out/pair_off.err || { tail -20 gpurun_out/pair_off.err; exit 1; }
python - <<PY
import json
for n in ("on","off"):
    d=json.loads(open("gpurun_out/pair_%s_$i.line"%n).read().strip().splitlines()[-1])
    print(n, d["ffn_block"].get("prefill_layer"))
PY
done
