#!/bin/bash
# one GPU-box session: parity tests, smoke, bench (graph + eager), rocprof kernel trace
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest_exit=$rc"; tail -3 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; echo "smoke_exit=$rc"; tail -2 gpurun_out/smoke.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -u bench.py --launch graph > gpurun_out/bench_graph.log 2>&1
rc=$?; echo "bench_graph_exit=$rc"; tail -2 gpurun_out/bench_graph.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -u bench.py --launch eager --no-cpu-baseline > gpurun_out/bench_eager.log 2>&1
rc=$?; echo "bench_eager_exit=$rc"; tail -2 gpurun_out/bench_eager.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --launch eager --no-cpu-baseline --steps 2048 --warmup 256 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_run.log 2>&1
rc=$?; echo "rocprof_exit=$rc"; tail -2 $GRAFT_REPO_ROOT/gpurun_out/rocprof_run.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*stats*" | head
