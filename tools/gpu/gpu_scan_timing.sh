#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
SDH_SCAN_TIMING=1 timeout -k 10 600 python -u bench.py --no-ffn --no-cpu-baseline --steps 512 --warmup 64 > gpurun_out/scan_timing.log 2>&1
rc=$?
grep -E "scan handle|normal draws" gpurun_out/scan_timing.log | tail -6
exit $rc
