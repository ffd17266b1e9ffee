#!/bin/bash
# development round: parity tests, stamps, microbench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest_exit=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 ./tools/bin/base_diag || exit 1
timeout -k 10 120 ./tools/bin/kbench 4096 1 | tail -5
