#!/bin/bash
# HBM traffic of the scan phases: FETCH_SIZE and WRITE_SIZE passes over a reduced sweep (8 layers)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/scan_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for phase in sweep payload; do
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    IFS=: read tag ctr <<< "$c"
    timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $O/${phase}_$tag -- python3 $R/tools/scan_idle.py --only $phase --layers 8 --threads 8 --payload-groups 12 > $O/${phase}_$tag.log 2>&1 || { echo "$phase $tag failed"; tail -3 $O/${phase}_$tag.log; exit 1; }
  done
done
python3 $R/tools/scan_pmc_report.py $O $R/gpurun_out/scan_pmc.json
find $O -name "*.csv" -size +2M -delete
