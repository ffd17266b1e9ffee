#!/bin/bash
# A/B of one workload at T=512 with environment variants: gpu_ab_env.sh WORKLOAD "NAME=VAL" ...  (first run: no variant)
R=$GRAFT_REPO_ROOT
wl=$1; shift
run() { env "$@" timeout -k 10 200 python3 $R/bench.py --tokens 512 --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload $wl 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']/32*1000,1) if False else d['ms_per_step'])"; }
run SDH_NOP=1 || exit 1
for v in "$@"; do run "$v" || exit 1; done
