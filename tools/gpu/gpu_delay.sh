#!/bin/bash
# decode time per workload for several head-start values of the base role (runtime knob SDH_DECODE_DELAY)
set -o pipefail
mkdir -p gpurun_out
for w in cfg3_mistral_down_block32_k1536 cfg2_gemma4b_gate_coo_k128 cfg4_mistral_gate_coo_k128 cfg4_mistral_gate_block16_k128 cfg4_mistral_down_coo_k128 cfg1_gemma1b_gate_block16_k64 cfg1_gemma1b_down_block16_k128; do
  line="$w"
  for d in ${DELAYS:-0 3 5 7}; do
    SDH_DECODE_DELAY=$d timeout -k 10 200 python bench.py --no-scan --no-ffn --no-cpu-baseline --workload $w --steps 2000 --warmup 200 > gpurun_out/bench_delay.json 2> gpurun_out/bench_delay.err || exit 1
    line="$line $(python -c "import json;d=json.loads(open('gpurun_out/bench_delay.json').read().strip().splitlines()[-1]);print('d$d=%.2f' % (d['ms_per_step']*1e3))")"
  done
  echo "$line"
done
