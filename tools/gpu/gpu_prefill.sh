#!/bin/bash
# prefill A/B: token-tiled SpMM vs per-token kernels, FWHT base vs W0-MFMA base (T=512 ubatch)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/prefill.log
B="python -u bench.py --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan --no-ffn"
for wl in cfg4_mistral_gate_coo_k128 cfg4_mistral_gate_block16_k128 cfg4_mistral_down_coo_k128 cfg4_mistral_down_block16_k128 cfg3_mistral_down_block32_k1536; do
  for variant in "default:0:8" "w0_mfma:8:8" "per_token:0:100000"; do
    IFS=: read name flags mint <<< "$variant"
    line=$(SDH_BENCH_FLAGS=$flags SDH_SPMM_MIN_T=$mint timeout -k 10 300 $B --workload $wl 2>/dev/null | tail -1) || exit 1
    echo "{\"variant\": \"$name\", \"line\": $line}" >> gpurun_out/prefill.log
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/prefill.log"):
    v=json.loads(l); d=v["line"]
    print(f'{d["config"]["workload"]:36s} {v["variant"]:10s} T=512 ms/step={d["ms_per_step"]:.3f} tok/s={d["value"]:.0f}')
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for variant in "fwht:0" "mfma:8"; do
  IFS=: read name flags <<< "$variant"
  SDH_BENCH_FLAGS=$flags timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prefill_$name -- python3 $R/bench.py --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan --no-ffn --no-roofline --workload cfg4_mistral_gate_coo_k128 > $R/gpurun_out/rocprof_prefill_$name.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_prefill_$name -name "*kernel_stats.csv" | head -1); echo "== $name"; head -4 "$f" | cut -c1-200
done
