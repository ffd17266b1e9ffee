#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_forward.py tests/test_gpu_ffn.py -m gpu -x -q -k "graph or fused or gate_up" --timeout 120 --timeout-method thread > gpurun_out/pytest_capture.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_capture.log; [ $rc -eq 0 ] || exit $rc
B="python bench.py --no-scan --no-cpu-baseline --steps 8192"
timeout -k 10 200 $B --launch eager > gpurun_out/b_eager.json 2>gpurun_out/b_eager.err && \
SDH_DECODE_FORCE_RESET=1 timeout -k 10 200 $B --launch eager > gpurun_out/b_eager_reset.json 2>gpurun_out/b_eager_reset.err && \
timeout -k 10 200 $B --launch graph > gpurun_out/b_graph.json 2>gpurun_out/b_graph.err
rc=$?
for f in eager eager_reset graph; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], d.get("ffn_block"))
except Exception as e: print("$f", "ERR", e)
PY
done
exit $rc
