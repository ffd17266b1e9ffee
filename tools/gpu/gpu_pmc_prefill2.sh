#!/bin/bash
# SQ counters of the prefill kernels, every workload of the bench's prefill leg, default and W0-MFMA variants
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_prefill2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY"
for wl in cfg4_mistral_gate_coo_k128 cfg4_mistral_gate_block16_k128 cfg4_mistral_down_coo_k128 cfg4_mistral_down_block16_k128 cfg3_mistral_down_block32_k1536; do
  for variant in "default:0" "w0_mfma:8"; do
    IFS=: read name flags <<< "$variant"
    SDH_BENCH_FLAGS=$flags timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/${wl}__$name -- python3 $R/bench.py --tokens 512 --payloads 2 --steps 8 --warmup 2 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload $wl > $O/${wl}__$name.log 2>&1 || { echo "$wl $name failed"; tail -3 $O/${wl}__$name.log; exit 1; }
  done
done
python3 $R/tools/prefill_pmc_report.py $O $R/gpurun_out/prefill_pmc.json
