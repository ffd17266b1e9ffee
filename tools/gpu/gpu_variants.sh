#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python -u bench.py --steps 8192 --warmup 512 --no-cpu-baseline --no-scan --no-ffn"
run() { SEEDDELTA_HIP_LIB=$2 timeout -k 10 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 us/step=%.3f kernel_us=%.3f' % (d['ms_per_step']*1e3, d['roofline']['avg_kernel_us']))" || exit 1; }
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests/test_gpu_forward.py tests/test_gpu_ffn.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_dev.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_dev.log; [ $rc -ne 0 ] && exit $rc
fi
run base ""
for v in ${VARIANTS:-t256 t1024 swizzle}; do run $v $GRAFT_REPO_ROOT/tools/bin/variants/$v/libseeddelta_hip.so; done
run base ""
