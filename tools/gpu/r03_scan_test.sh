#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_scan.py tests/test_gpu_scan_kgrid.py tests/test_gpu_scan_full.py tests/test_gpu_build.py tests/test_properties.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_scan_tests.log 2>&1 || { tail -40 gpurun_out/r03_scan_tests.log; exit 1; }
tail -2 gpurun_out/r03_scan_tests.log
timeout -k 10 200 python -u tools/select_time.py 2>/dev/null || exit 1
