#!/bin/bash
# direct decode launch: correctness against the oracle, then the decode leg of bench.py with and without it
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout -k 10 300 python -u tools/direct_check.py > gpurun_out/direct_check.log 2>&1 || { tail -30 gpurun_out/direct_check.log; exit 1; }
cat gpurun_out/direct_check.log
for v in 1 0; do
SDH_DECODE_DIRECT=$v timeout -k 10 300 python -u bench.py --no-scan --no-stack --no-cpu-baseline --no-prefill --no-ffn --steps 200 --warmup 20 > gpurun_out/direct_$v.line 2> gpurun_out/direct_$v.err || { tail -20 gpurun_out/direct_$v.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/direct_$v.line").read().strip().splitlines()[-1])
print("direct=$v", d["value"], d["ms_per_step"], d["roofline"])
PY
done
