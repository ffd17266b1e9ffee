#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u tools/scan_time.py > gpurun_out/scan_time.log 2>&1 || { tail -20 gpurun_out/scan_time.log; exit 1; }
cat gpurun_out/scan_time.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scan -- python3 $R/tools/scan_time.py > $R/gpurun_out/rocprof_scan.log 2>&1
echo "rocprof_exit=$?"
f=$(find $R/gpurun_out/prof_scan -name "*kernel_stats.csv" | head -1); head -12 "$f"
