#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_prefill_full.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_fold_tests.log 2>&1 || { tail -40 gpurun_out/r03_fold_tests.log; exit 1; }
tail -2 gpurun_out/r03_fold_tests.log
timeout -k 10 120 python -u tools/fold_time.py || exit 1
for v in ${VARIANTS}; do
  SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so timeout -k 10 120 python -u tools/fold_time.py || exit 1
done
