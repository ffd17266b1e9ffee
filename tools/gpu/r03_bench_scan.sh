#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
SDH_BENCH_DETAIL=$R/gpurun_out/r03_bench_scan_detail.json timeout -k 10 600 python -u bench.py --no-ffn --no-stack --no-prefill --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r03_bench_scan.json 2> gpurun_out/r03_bench_scan.err || { tail -20 gpurun_out/r03_bench_scan.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r03_bench_scan.json'));print(json.dumps(d['scan']))"
