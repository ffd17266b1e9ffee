#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o graph -- python3 $GRAFT_REPO_ROOT/bench.py --no-ffn --no-scan --no-cpu-baseline --no-roofline --steps 2048 --warmup 256 --launch graph > $GRAFT_REPO_ROOT/gpurun_out/prof_graph.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_graph -name "*kernel_stats.csv" | while read f; do cut -c1-200 "$f"; done
tail -3 gpurun_out/prof_graph.log
exit $rc
