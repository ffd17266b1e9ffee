#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/prefill.log
for wl in cfg4_mistral_gate_coo_k128 cfg4_mistral_gate_block16_k128 cfg4_mistral_down_coo_k128 cfg4_mistral_down_block16_k128 cfg3_mistral_down_block32_k1536; do
  timeout -k 10 300 python -u bench.py --workload $wl --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan 2>/dev/null | tail -1 >> gpurun_out/prefill.log || exit 1
  SDH_SPMM_MIN_T=100000 timeout -k 10 300 python -u bench.py --workload $wl --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan 2>/dev/null | tail -1 >> gpurun_out/prefill.log || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/prefill.log"):
    d=json.loads(l); r=d["roofline"]
    print(f'{d["config"]["workload"]:36s} T=512 ms/step={d["ms_per_step"]:.3f} tok/s={d["value"]:.0f} kernel_us={r["avg_kernel_us"]:.1f}')
PY

