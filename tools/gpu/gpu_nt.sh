#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
TOKENS="1 2 3 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg4 gate coo"; EXTRA="--workload cfg4_mistral_gate_coo_k128" TOKENS="1 2 3 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg2"; EXTRA="--workload cfg2_gemma4b_gate_coo_k128" TOKENS="2 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
echo "--- cfg2 two-launch"; EXTRA="--workload cfg2_gemma4b_gate_coo_k128 --no-fuse" TOKENS="2 4" ./tools/gpu/gpu_smallbatch.sh
