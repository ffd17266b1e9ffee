#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_build.py tests/test_gpu_scan.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_build.log 2>&1
rc=$?; tail -25 gpurun_out/pytest_build.log; exit $rc
