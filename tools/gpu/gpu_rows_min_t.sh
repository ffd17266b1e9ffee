#!/bin/bash
# cfg3-like workloads at few tokens: LDS rows kernel from T=2 (SDH_ROWS_LDS_MIN_T=2) against the default threshold / off
R=$GRAFT_REPO_ROOT
wl=${1:-cfg3_mistral_down_block32_k1536}
for T in 3 4 5 6 7 8 12 16 32; do
  for v in SDH_ROWS_LDS_MIN_T=2 SDH_ROWS_NO_LDS=1; do
    env $v timeout -k 10 200 python3 $R/bench.py --tokens $T --payloads 4 --steps 32 --warmup 8 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload $wl 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T $v us/call', round(d['ms_per_step']*1000/4,2))" || exit 1
  done
done
