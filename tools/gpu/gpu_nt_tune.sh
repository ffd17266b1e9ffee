#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline --steps 8192 ${ARGS} > gpurun_out/nt_$tag.json 2> gpurun_out/nt_$tag.err || { tail -3 gpurun_out/nt_$tag.err; return 1; }
  python -c "import json;d=json.loads(open('gpurun_out/nt_$tag.json').read().strip().splitlines()[-1]);f=d.get('ffn_block') or {};print('$tag', round(d['ms_per_step']*1e3,3),'us', 'ffn', f.get('us_per_layer'))"; }
ARGS="" run default SDH_X=1 || exit 1
ARGS="--no-ffn" run delay3 SDH_DECODE_DELAY=3 || exit 1
ARGS="--no-ffn" run delay4 SDH_DECODE_DELAY=4 || exit 1
ARGS="--no-ffn" run delay6 SDH_DECODE_DELAY=6 || exit 1
ARGS="--no-ffn" run delay8 SDH_DECODE_DELAY=8 || exit 1
for wl in cfg2_gemma4b_gate_coo_k128 cfg4_mistral_gate_coo_k128 cfg4_mistral_gate_block16_k128 cfg4_mistral_down_coo_k128 cfg1_gemma1b_down_block16_k128; do
  ARGS="--no-ffn --workload $wl" run $wl SDH_X=1 || exit 1
done
