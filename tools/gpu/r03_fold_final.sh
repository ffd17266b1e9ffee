#!/bin/bash
# final evidence for the folded prefill: parity tests, call times (with / without the permuted form), kernel stats, SQ counters
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu/r03_perm.sh || exit 1
bash tools/gpu/r03_fold_prof.sh || exit 1
bash tools/gpu/r03_pmc_fold.sh > gpurun_out/r03_pmc_fold_gate.txt 2>&1 || { tail gpurun_out/r03_pmc_fold_gate.txt; exit 1; }
WL=cfg3_mistral_down_block32_k1536 bash tools/gpu/r03_pmc_fold.sh > gpurun_out/r03_pmc_fold_cfg3.txt 2>&1 || { tail gpurun_out/r03_pmc_fold_cfg3.txt; exit 1; }
tail -30 gpurun_out/r03_pmc_fold_gate.txt
