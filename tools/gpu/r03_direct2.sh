#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout -k 10 300 python -u tools/direct_check.py > gpurun_out/direct_check.log 2>&1 || { tail -30 gpurun_out/direct_check.log; exit 1; }
cat gpurun_out/direct_check.log
bash tools/gpu/r03_direct_ab.sh
