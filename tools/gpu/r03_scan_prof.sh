#!/bin/bash
# kernel stats of the payload-emission sweep (8 layers, all 48 K schedules), and of the metrics-only sweep
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_scan_prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for phase in payload sweep; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$phase -- python3 $R/tools/scan_idle.py --only $phase --layers 8 --threads 16 --payload-groups 48 > $O/$phase.log 2>&1 || { tail -5 $O/$phase.log; exit 1; }
  grep '^{"sweep"' $O/$phase.log
  f=$(find $O/$phase -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${phase}_kernel_stats.csv
  head -12 "$f" | cut -c1-100,230-330
  find $O/$phase -name "*_trace.csv" -delete
done
