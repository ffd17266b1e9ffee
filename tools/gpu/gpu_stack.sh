#!/bin/bash
# stack kernel: parity tests, the in-kernel timeline (stamped build), then the bench leg
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest $R/tests/test_gpu_stack.py -x -q --timeout 100 --timeout-method thread > $R/gpurun_out/stack_tests.log 2>&1 || { tail -20 $R/gpurun_out/stack_tests.log; exit 1; }
tail -2 $R/gpurun_out/stack_tests.log
SEEDDELTA_HIP_LIB=$R/tools/bin/variants/stkstamps/libseeddelta_hip.so NL=6 timeout -k 10 200 python $R/tools/stack_stamps.py 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/stack_stamps.txt || exit 1
timeout -k 10 400 python $R/bench.py --steps 20 --warmup 5 --no-scan --no-cpu-baseline --no-prefill --no-ffn > $R/gpurun_out/bench_stack.json 2> $R/gpurun_out/bench_stack.err || { tail -c 600 $R/gpurun_out/bench_stack.err; exit 1; }
python3 - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/bench_stack.json"))
s = d["ffn_stack"]
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items() if k not in ("metric", "timing", "roofline")}, "frac", round(s["roofline"]["frac"], 4))
PY
