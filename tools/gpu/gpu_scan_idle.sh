#!/bin/bash
# GPU busy / idle share of the scan sweep from a kernel trace; arguments are passed to tools/scan_idle.py
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/scan_idle_kt
SDH_SCAN_TIMING=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/scan_idle_kt -- python3 $R/tools/scan_idle.py "$@" > $R/gpurun_out/scan_idle.log 2>&1 || { tail -5 $R/gpurun_out/scan_idle.log; exit 1; }
grep "scan handle" $R/gpurun_out/scan_idle.log | tail -2
tail -1 $R/gpurun_out/scan_idle.log
python3 $R/tools/scan_idle_report.py $R/gpurun_out/scan_idle_kt > $R/gpurun_out/scan_idle_report.json && python3 - <<PY
import json
for p in json.load(open("$R/gpurun_out/scan_idle_report.json")):
    print(p["phase"], "span %.1f ms busy %.1f ms idle %.2f kernels %d" % (p["span_ms"], p["gpu_busy_ms"], p["idle_share"], p["kernels"]))
    for t in p["top"][:6]:
        print("    %-60s %8.2f ms %5d" % (t["kernel"], t["total_ms"], t["calls"]))
PY
find $R/gpurun_out/scan_idle_kt -name "*.csv" -size +3M -delete
