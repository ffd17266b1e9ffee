#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
TOKENS="1 2 3 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg4 gate coo"; EXTRA="--workload cfg4_mistral_gate_coo_k128" TOKENS="1 2 3 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg4 gate coo two-launch"; EXTRA="--workload cfg4_mistral_gate_coo_k128 --no-fuse" TOKENS="3" ./tools/gpu/gpu_smallbatch.sh
