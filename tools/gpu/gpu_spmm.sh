#!/bin/bash
# prefill: full-size and small-dims parity, then the prefill leg and a kernel-stats pass of the default path
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest $R/tests/test_gpu_prefill_full.py $R/tests/test_gpu_forward.py -m gpu -x -q --timeout 200 --timeout-method thread > $R/gpurun_out/pytest_spmm.log 2>&1
rc=$?; tail -6 $R/gpurun_out/pytest_spmm.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u $R/bench.py --no-ffn --no-stack --no-scan --no-cpu-baseline --steps 8 --warmup 2 > $R/gpurun_out/bench_prefill.log 2> $R/gpurun_out/bench_prefill.err || { tail -5 $R/gpurun_out/bench_prefill.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/bench_prefill.log").read().strip().splitlines()[-1])
for name, w in d["prefill"]["workloads"].items():
    print("%-34s default %.3f ms | w0_mfma %.3f ms" % (name, w["default"]["ms"], w["w0_mfma"]["ms"]))
PY
cd /tmp && export TMPDIR=/tmp
for wl in ${WLS:-cfg4_mistral_gate_coo_k128}; do
rm -rf $R/gpurun_out/prof_spmm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_spmm -- python3 $R/bench.py --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload $wl > $R/gpurun_out/rocprof_spmm.log 2>&1 || { tail -5 $R/gpurun_out/rocprof_spmm.log; exit 1; }
f=$(find $R/gpurun_out/prof_spmm -name "*kernel_stats.csv" | head -1); echo "== $wl"; head -4 "$f" | cut -c1-200
done
find $R/gpurun_out/prof_spmm -name "*_trace.csv" -delete
