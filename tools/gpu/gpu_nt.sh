#!/bin/bash
# small-batch timings per shape: fused multi-token launch vs the two-kernel path
set -o pipefail
mkdir -p gpurun_out
for wl in cfg4_mistral_gate_block16_k128 cfg4_mistral_down_coo_k128 cfg1_gemma1b_down_block16_k128; do
  echo "--- $wl"; EXTRA="--workload $wl" TOKENS="1 2 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
  echo "--- $wl two kernels"; EXTRA="--workload $wl --no-fuse" TOKENS="2 4" ./tools/gpu/gpu_smallbatch.sh || exit 1
done
