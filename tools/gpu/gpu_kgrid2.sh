#!/bin/bash
# scan parity tests, then the idle-share measurement
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 500 python -u -m pytest $R/tests/test_gpu_scan_kgrid.py $R/tests/test_gpu_scan.py $R/tests/test_gpu_build.py -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/pytest_kgrid.log 2>&1
rc=$?; tail -15 $R/gpurun_out/pytest_kgrid.log; [ $rc -ne 0 ] && exit $rc
bash $R/tools/gpu/gpu_scan_idle.sh "$@"
