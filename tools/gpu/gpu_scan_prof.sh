#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $R/gpurun_out/prof_scan_hip -- python3 $R/bench.py --no-ffn --no-cpu-baseline --no-roofline --steps 64 --warmup 32 > $R/gpurun_out/prof_scan_hip.log 2>&1
rc=$?
tail -1 $R/gpurun_out/prof_scan_hip.log | cut -c1-200
find $R/gpurun_out/prof_scan_hip -name "*hip_api_stats.csv" | while read f; do cut -c1-120 "$f" | head -30; done
exit $rc
