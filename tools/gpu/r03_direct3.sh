#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout -k 10 300 python -u tools/direct_check.py > gpurun_out/direct_check.log 2>&1 || { tail -30 gpurun_out/direct_check.log; exit 1; }
cat gpurun_out/direct_check.log
SEEDDELTA_HIP_LIB=$R/tools/bin/variants/dd_stamps/libseeddelta_hip.so timeout -k 10 200 python -u tools/direct_stamps.py > gpurun_out/direct_stamps.log 2>&1 || { tail gpurun_out/direct_stamps.log; exit 1; }
tail -14 gpurun_out/direct_stamps.log
bash tools/gpu/r03_direct_ab.sh
SDH_DECODE_DIRECT=0 VARIANTS="" bash tools/gpu/r03_direct_ab.sh
