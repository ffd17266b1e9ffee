#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 ./tools/bin/fused_diag > gpurun_out/fused_diag.log 2>&1 || { tail -5 gpurun_out/fused_diag.log; exit 1; }
cat gpurun_out/fused_diag.log
