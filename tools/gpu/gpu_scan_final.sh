#!/bin/bash
# scan leg of the bench (with CPU baseline), then a hip-runtime-trace + kernel-trace stats pass of the sweep
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 600 python -u $R/bench.py --no-ffn --no-stack --no-prefill --cpu-seconds 12 > $R/gpurun_out/bench_scan_final.log 2> $R/gpurun_out/bench_scan_final.err || { tail -5 $R/gpurun_out/bench_scan_final.err; exit 1; }
python3 -c "import json;d=json.loads(open('$R/gpurun_out/bench_scan_final.log').read().strip().splitlines()[-1]);s=d['scan'];print(json.dumps({k:s[k] for k in ('value','candidates','seconds','with_payload_emission','roofline','cpu_baseline')})[:1800])"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/scan_hiptrace
timeout -k 10 300 rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/scan_hiptrace -- python3 $R/tools/scan_idle.py > $R/gpurun_out/scan_hiptrace.log 2>&1 || { tail -5 $R/gpurun_out/scan_hiptrace.log; exit 1; }
grep '^{"sweep"' $R/gpurun_out/scan_hiptrace.log
python3 $R/tools/scan_idle_report.py $R/gpurun_out/scan_hiptrace > $R/gpurun_out/scan_hiptrace_idle.json
f=$(find $R/gpurun_out/scan_hiptrace -name "*hip_api_stats.csv" | head -1); head -12 "$f" | cut -c1-150
find $R/gpurun_out/scan_hiptrace -name "*_trace.csv" -delete
