#!/bin/bash
# capture-safe fused decode: tests, then the decode bench eager vs graph
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_forward.py tests/test_gpu_ffn.py -m gpu -x -q -k "graph or fused or gate_up" --timeout 120 --timeout-method thread > gpurun_out/pytest_capture.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_capture.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --no-ffn --no-scan --no-cpu-baseline --steps 8192 --launch eager > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err && \
timeout -k 10 200 python bench.py --no-ffn --no-scan --no-cpu-baseline --steps 8192 --launch graph > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err && \
timeout -k 10 200 python bench.py --no-ffn --no-scan --no-cpu-baseline --steps 8192 --launch graph --no-fuse > gpurun_out/bench_graph_nofuse.json 2> gpurun_out/bench_graph_nofuse.err
rc=$?
for f in eager graph graph_nofuse; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"] if d.get("roofline") else None)
except Exception as e: print("$f", "ERR", e)
PY
done
exit $rc
