#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for v in ${STAMPV}; do
echo "== $v"
SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so timeout -k 10 200 python -u tools/direct_stamps.py > gpurun_out/direct_stamps_$v.log 2>&1 || { tail gpurun_out/direct_stamps_$v.log; exit 1; }
tail -14 gpurun_out/direct_stamps_$v.log
done
bash tools/gpu/r03_direct_ab.sh
