#!/bin/bash
# tiled W0-MFMA: parity tests at small and full dims, then the prefill leg of the bench and a kernel-stats pass
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 500 python -u -m pytest $R/tests/test_gpu_prefill_full.py $R/tests/test_gpu_forward.py -m gpu -x -q -k "mfma or MFMA" --timeout 150 --timeout-method thread > $R/gpurun_out/pytest_mfma.log 2>&1
rc=$?; tail -12 $R/gpurun_out/pytest_mfma.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u $R/bench.py --no-ffn --no-stack --no-scan --no-cpu-baseline --steps 8 --warmup 2 > $R/gpurun_out/bench_prefill.log 2> $R/gpurun_out/bench_prefill.err || { tail -5 $R/gpurun_out/bench_prefill.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/bench_prefill.log").read().strip().splitlines()[-1])
for name, w in d["prefill"]["workloads"].items():
    print("%-34s default %.3f ms | w0_mfma %.3f ms (whole call %.0f TFLOP/s, frac %.3f)" % (name, w["default"]["ms"], w["w0_mfma"]["ms"], w["w0_mfma"]["mfma"]["achieved"], w["w0_mfma"]["mfma"]["frac"]))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_mfma
SDH_BENCH_FLAGS=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mfma -- python3 $R/bench.py --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload cfg4_mistral_gate_coo_k128 > $R/gpurun_out/rocprof_mfma.log 2>&1 || { tail -5 $R/gpurun_out/rocprof_mfma.log; exit 1; }
f=$(find $R/gpurun_out/prof_mfma -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-220
find $R/gpurun_out/prof_mfma -name "*_trace.csv" -delete
