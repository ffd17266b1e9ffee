#!/bin/bash
# A/B builds of the direct decode kernel (tools/build_variants.sh SRC=sdh_direct): launch period on cfg3
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
run() {
  timeout -k 10 200 python -u bench.py --no-scan --no-stack --no-cpu-baseline --no-prefill --no-ffn --steps 200 --warmup 20 > gpurun_out/dab.line 2> gpurun_out/dab.err || { tail -20 gpurun_out/dab.err; exit 1; }
  python - "$1" <<PY
import json, sys
d=json.loads(open("gpurun_out/dab.line").read().strip().splitlines()[-1])
print("%-28s us/launch %.3f" % (sys.argv[1], d["ms_per_step"] * 1e3 / 32), flush=True)
PY
}
run default

for v in ${VARIANTS}; do
  SEEDDELTA_HIP_LIB=$R/tools/bin/variants/$v/libseeddelta_hip.so run $v
done
