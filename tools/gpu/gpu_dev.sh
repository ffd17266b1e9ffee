#!/bin/bash
# quick GPU session: run the named test files (default: all gpu tests)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest ${@:-tests} -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_dev.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_dev.log; exit $rc
