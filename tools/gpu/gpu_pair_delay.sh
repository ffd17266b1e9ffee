#!/bin/bash
# FFN-block leg with several head-start settings of the gate|up pair launch
R=$GRAFT_REPO_ROOT
for d in "$@"; do
  SDH_DECODE_PAIR_DELAY=$d timeout -k 10 300 python3 $R/bench.py --no-scan --no-stack --no-prefill --no-cpu-baseline --steps 64 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['ffn_block']; print('delay $d', round(f['us_per_layer'],2), round(f['us_per_layer_node_by_node'],2))" || exit 1
done
