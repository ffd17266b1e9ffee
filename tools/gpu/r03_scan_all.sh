#!/bin/bash
# scan tests, select timing, kernel stats of both sweeps, then the FETCH_SIZE / WRITE_SIZE passes
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu/r03_scan_test.sh || exit 1
bash tools/gpu/r03_scan_prof.sh > gpurun_out/r03_scan_prof.txt 2>&1 || { tail -5 gpurun_out/r03_scan_prof.txt; exit 1; }
grep '^{"sweep"' gpurun_out/r03_scan_prof.txt
bash tools/gpu/gpu_scan_pmc.sh || exit 1
