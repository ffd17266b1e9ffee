#!/bin/bash
# SQ counters for the prefill kernels (SpMM + FWHT base, and the W0-MFMA variant): one pass each
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
for variant in "fwht:0" "mfma:8"; do
  IFS=: read name flags <<< "$variant"
  SDH_BENCH_FLAGS=$flags timeout -s KILL 180 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_prefill_$name -- python3 $R/bench.py --tokens 512 --payloads 2 --steps 8 --warmup 2 --no-cpu-baseline --no-scan --no-ffn --no-roofline --workload cfg4_mistral_gate_coo_k128 > $R/gpurun_out/pmc_prefill_$name.log 2>&1
  echo "pmc_$name exit=$?"; tail -2 $R/gpurun_out/pmc_prefill_$name.log | cut -c1-300
done
find $R/gpurun_out/pmc_prefill_fwht $R/gpurun_out/pmc_prefill_mfma -name "*counter_collection.csv"
