#!/bin/bash
# direct decode launch against the fused one, launch period per workload (one token)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for wl in cfg3_mistral_down_block32_k1536 cfg4_mistral_gate_coo_k128 cfg4_mistral_down_coo_k128 cfg4_mistral_gate_block16_k128 cfg4_mistral_down_block16_k128 cfg2_gemma4b_gate_coo_k128; do
for v in 1 0; do
SDH_DECODE_DIRECT=$v timeout -k 10 200 python -u bench.py --workload $wl --no-scan --no-stack --no-cpu-baseline --no-prefill --no-ffn --no-roofline --steps 200 --warmup 20 > gpurun_out/dwl.line 2> gpurun_out/dwl.err || { tail -20 gpurun_out/dwl.err; exit 1; }
python - "$wl" "$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/dwl.line").read().strip().splitlines()[-1])
print("%-36s direct=%s us/launch %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"] * 1e3 / 32), flush=True)
PY
done
done
