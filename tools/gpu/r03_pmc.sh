#!/bin/bash
# SQ counters (one pass of 8) for the folded prefill contraction and for the scan kernels (fit, selection)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY"
for wl in cfg4_mistral_gate_coo_k128 cfg3_mistral_down_block32_k1536; do
  SDH_BENCH_FLAGS=64 timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/${wl}__mfma_fold -- python3 $R/bench.py --tokens 512 --payloads 2 --steps 8 --warmup 2 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload $wl > $O/${wl}__mfma_fold.log 2>&1 || { echo "$wl failed"; tail -3 $O/${wl}__mfma_fold.log; exit 1; }
done
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $O/scan_cfg5__payload -- python3 $R/tools/scan_idle.py --only payload --layers 2 --threads 4 --payload-groups 12 > $O/scan_cfg5__payload.log 2>&1 || { echo "scan failed"; tail -3 $O/scan_cfg5__payload.log; exit 1; }
python3 $R/tools/prefill_pmc_report.py $O $R/gpurun_out/r03_pmc.json
find $O -name "*.csv" -size +1M -delete
