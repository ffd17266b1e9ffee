#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python -u bench.py --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan --no-ffn"
for wl in cfg4_mistral_gate_coo_k128 cfg4_mistral_gate_block16_k128 cfg2_gemma4b_gate_coo_k128 cfg4_mistral_down_coo_k128; do
  for v in four one; do
    if [ $v = one ]; then export SDH_SPMM_ONE_TOKEN=1; else unset SDH_SPMM_ONE_TOKEN; fi
    timeout -k 10 300 $B --workload $wl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $v ms/step=%.3f kernel_us=%.1f' % (d['ms_per_step'], d['roofline']['avg_kernel_us']))" || exit 1
  done
done
