#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 ./tools/bin/kbench 4096 1 > gpurun_out/kbench.log 2>&1 || { echo kbench failed; tail -5 gpurun_out/kbench.log; exit 1; }
timeout -k 10 120 ./tools/bin/kbench 4096 0 >> gpurun_out/kbench.log 2>&1
cat gpurun_out/kbench.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_kbench -- $GRAFT_REPO_ROOT/tools/bin/kbench 1024 1 > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof_kbench/*/*kernel_stats.csv | cut -c1-160
