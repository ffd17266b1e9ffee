#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in cfg4_mistral_gate_coo_k128 cfg3_mistral_down_block32_k1536; do
  rm -rf $R/gpurun_out/foldprof_$wl
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/foldprof_$wl -- python3 $R/tools/fold_prof.py $wl 40 > $R/gpurun_out/foldprof_$wl.log 2>&1 || { tail -5 $R/gpurun_out/foldprof_$wl.log; exit 1; }
  f=$(find $R/gpurun_out/foldprof_$wl -name "*kernel_stats.csv" | head -1)
  echo "== $wl"; head -8 "$f" | cut -c1-200
  cp "$f" $R/gpurun_out/foldprof_${wl}_kernel_stats.csv
  find $R/gpurun_out/foldprof_$wl -name "*_trace.csv" -delete
done
