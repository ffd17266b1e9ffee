#!/bin/bash
# FFN layer kernel stats at one token count (default 8); SEEDDELTA_HIP_LIB selects an A/B build of the library
R=$GRAFT_REPO_ROOT
T=${1:-8}
tag=${2:-cur}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ffn_t
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ffn_t -- python3 $R/tools/ffn_tokens.py $T > $R/gpurun_out/prof_ffn_t.log 2>&1 || { tail -3 $R/gpurun_out/prof_ffn_t.log; exit 1; }
f=$(find $R/gpurun_out/prof_ffn_t -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/ffn_t${T}_${tag}_kernel_stats.csv; head -5 "$f" | cut -c1-200
find $R/gpurun_out/prof_ffn_t -name "*_trace.csv" -delete
