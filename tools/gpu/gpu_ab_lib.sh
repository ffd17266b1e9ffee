#!/bin/bash
# same-box A/B of the decode bench: in-tree library vs tools/bin/variants/<name>/libseeddelta_hip.so, interleaved
set -o pipefail
mkdir -p gpurun_out
V=${1:-nt}; shift
for i in 1 2 3; do
  for which in base $V; do
    if [ $which = base ]; then unset SEEDDELTA_HIP_LIB; else export SEEDDELTA_HIP_LIB=$PWD/tools/bin/variants/$which/libseeddelta_hip.so; fi
    timeout -k 10 200 python bench.py --no-ffn --no-scan --no-cpu-baseline --steps 8192 "$@" > gpurun_out/abl_$which.json 2> gpurun_out/abl_$which.err || { tail -3 gpurun_out/abl_$which.err; exit 1; }
    python -c "import json;d=json.loads(open('gpurun_out/abl_$which.json').read().strip().splitlines()[-1]);print('$which', round(d['ms_per_step']*1e3,3),'us')"
  done
done
