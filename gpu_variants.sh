#!/bin/bash
set -o pipefail
B="python -u bench.py --steps 8192 --warmup 512 --no-cpu-baseline --no-scan --no-ffn"
run() { SEEDDELTA_HIP_LIB=$2 timeout -k 10 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 us/step=%.3f kernel_us=%.3f' % (d['ms_per_step']*1e3, d['roofline']['avg_kernel_us']))" || exit 1; }
run base ""
for v in poll1 poll16 delay2 delay8 delay24 prio0; do run $v $GRAFT_REPO_ROOT/tools/bin/variants/$v/libseeddelta_hip.so; done
run base ""
