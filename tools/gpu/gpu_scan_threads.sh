#!/bin/bash
# the sweep of tools/scan_idle.py at several host thread counts (no profiler)
R=$GRAFT_REPO_ROOT
for t in "$@"; do
  timeout -k 10 200 python3 $R/tools/scan_idle.py --threads $t 2>/dev/null | tail -1 || exit 1
done
