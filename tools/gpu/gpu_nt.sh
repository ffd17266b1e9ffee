#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
SDH_DEBUG=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "small_batch or fused" --timeout 120 --timeout-method thread > gpurun_out/pytest_nt.log 2>&1
rc=$?; grep -v "^\[sdh\]" gpurun_out/pytest_nt.log | tail -4; grep "LDS decode" gpurun_out/pytest_nt.log | sort | uniq -c | head; [ $rc -eq 0 ] || exit $rc
echo "--- cfg4 gate coo"; EXTRA="--workload cfg4_mistral_gate_coo_k128" TOKENS="2 3 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg4 gate coo, plain multi-token"; SDH_DECODE_XLDS=0 EXTRA="--workload cfg4_mistral_gate_coo_k128" TOKENS="2 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg2"; EXTRA="--workload cfg2_gemma4b_gate_coo_k128" TOKENS="2 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg2 plain"; SDH_DECODE_XLDS=0 EXTRA="--workload cfg2_gemma4b_gate_coo_k128" TOKENS="2 4" ./tools/gpu/gpu_smallbatch.sh
