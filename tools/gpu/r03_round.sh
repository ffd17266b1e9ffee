#!/bin/bash
# round-3 regular check: GPU tests, smoke, default bench line (+ detail file), rocprofv3 kernel stats of a short bench
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_a}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -u -m pytest ${TESTS:-tests} -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${TAG}_gpu_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_gpu_tests.log; exit 1; }
tail -3 gpurun_out/${TAG}_gpu_tests.log
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1 || { tail -5 gpurun_out/${TAG}_smoke.log; exit 1; }
tail -1 gpurun_out/${TAG}_smoke.log
SDH_BENCH_DETAIL=$R/gpurun_out/${TAG}_bench_detail.json timeout -k 10 600 python -u bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
wc -c gpurun_out/${TAG}_bench_line.json
tail -c 3000 gpurun_out/${TAG}_bench_line.json
