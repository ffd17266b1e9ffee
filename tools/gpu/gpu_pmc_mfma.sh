#!/bin/bash
# SQ counters of the tiled W0-MFMA prefill (one pass, 8 SQ counters)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
C="${SDH_PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY}"
rm -rf $R/gpurun_out/pmc_mfma
SDH_BENCH_FLAGS=8 timeout -s KILL 180 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --tokens 512 --payloads 2 --steps 8 --warmup 2 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload ${WL:-cfg4_mistral_gate_coo_k128} > $R/gpurun_out/pmc_mfma.log 2>&1 || { tail -3 $R/gpurun_out/pmc_mfma.log; exit 1; }
f=$(find $R/gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_mfma.json $f
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/pmc_mfma.json"))
for run, ks in d.items():
    for k, m in ks.items():
        if "w0_mfma" in k: print({a: round(b, 1) for a, b in m.items() if a.startswith("SQ_")})
PY
