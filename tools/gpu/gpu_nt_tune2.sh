#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline --steps 8192 ${ARGS} > gpurun_out/nt_$tag.json 2> gpurun_out/nt_$tag.err || { tail -3 gpurun_out/nt_$tag.err; return 1; }
  python -c "import json;d=json.loads(open('gpurun_out/nt_$tag.json').read().strip().splitlines()[-1]);f=d.get('ffn_block') or {};print('$tag', round(d['ms_per_step']*1e3,3),'us', 'ffn', f.get('us_per_layer'))"; }
for d in 0 1 2 3 2 3; do ARGS="" run delay$d SDH_DECODE_DELAY=$d || exit 1; done
for d in 1 2 3; do ARGS="--no-ffn --workload cfg4_mistral_gate_coo_k128" run gate_delay$d SDH_DECODE_DELAY=$d || exit 1; done
