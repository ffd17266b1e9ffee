#!/bin/bash
# in-kernel stamps of the fused decode kernel, forward parity tests, decode timings per workload
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 ./tools/bin/fused_diag > gpurun_out/fused_diag.log 2>&1 || { tail -5 gpurun_out/fused_diag.log; exit 1; }
cat gpurun_out/fused_diag.log
timeout -k 10 600 python -u -m pytest tests/test_gpu_forward.py tests/test_gpu_shim.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_dev.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_dev.log; [ $rc -ne 0 ] && exit $rc
for w in cfg3_mistral_down_block32_k1536 cfg2_gemma4b_gate_coo_k128 cfg4_mistral_gate_coo_k128 cfg4_mistral_gate_block16_k128 cfg4_mistral_down_coo_k128 cfg1_gemma1b_gate_block16_k64 cfg1_gemma1b_down_block16_k128; do
  timeout -k 10 200 python bench.py --no-scan --no-ffn --no-cpu-baseline --workload $w --steps 2000 --warmup 200 > gpurun_out/bench_new_$w.json 2> gpurun_out/bench_new_$w.err || exit 1
  python -c "import json;d=json.loads(open('gpurun_out/bench_new_$w.json').read().strip().splitlines()[-1]);print('new $w', d['ms_per_step'])"
done
