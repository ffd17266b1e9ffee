#!/bin/bash
# permuted-order fold kernel: parity tests, then call times with and without it
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_prefill_full.py -m gpu -x -q --timeout 300 --timeout-method thread -k "fold" > gpurun_out/r03_perm_tests.log 2>&1 || { tail -40 gpurun_out/r03_perm_tests.log; exit 1; }
tail -3 gpurun_out/r03_perm_tests.log
WL=cfg4_mistral_gate_coo_k128,cfg4_mistral_gate_block16_k128,cfg4_mistral_down_coo_k128,cfg4_mistral_down_block16_k128,cfg3_mistral_down_block32_k1536
WL=$WL timeout -k 10 120 python -u tools/fold_time.py > gpurun_out/r03_perm_time.txt 2>&1 || { tail gpurun_out/r03_perm_time.txt; exit 1; }
SDH_FOLD_NO_PERM=1 WL=$WL timeout -k 10 120 python -u tools/fold_time.py >> gpurun_out/r03_perm_time.txt 2>&1 || { tail gpurun_out/r03_perm_time.txt; exit 1; }
cat gpurun_out/r03_perm_time.txt
