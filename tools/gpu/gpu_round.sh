#!/bin/bash
# one GPU-box session: parity tests, smoke, bench (fused eager / two-launch / graph), rocprof kernel
# trace + HBM traffic counters. Usage: ./tools/gpu/gpu_round.sh <tag>   (outputs under gpurun_out/)
set -o pipefail
TAG=${1:-r01}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest_exit=$rc"; tail -3 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; echo "smoke_exit=$rc"; tail -2 gpurun_out/smoke.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -u bench.py > gpurun_out/bench_${TAG}.log 2>&1
rc=$?; echo "bench_exit=$rc"; tail -1 gpurun_out/bench_${TAG}.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -u bench.py --no-fuse --no-cpu-baseline --no-scan --no-ffn > gpurun_out/bench_${TAG}_nofuse.log 2>&1
rc=$?; echo "bench_nofuse_exit=$rc"; tail -1 gpurun_out/bench_${TAG}_nofuse.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -u bench.py --launch graph --no-cpu-baseline --no-scan --no-ffn > gpurun_out/bench_${TAG}_graph.log 2>&1
rc=$?; echo "bench_graph_exit=$rc"; tail -1 gpurun_out/bench_${TAG}_graph.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --no-cpu-baseline --steps 2048 --warmup 256 > $R/gpurun_out/rocprof_${TAG}.log 2>&1
rc=$?; echo "rocprof_exit=$rc"; tail -1 $R/gpurun_out/rocprof_${TAG}.log
[ $rc -ne 0 ] && exit $rc
# HBM traffic: separate PMC passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2)
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${TAG} -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-ffn --no-scan --steps 512 --warmup 64 > $R/gpurun_out/pmc_fetch_${TAG}.log 2>&1
echo "pmc_fetch_exit=$?"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${TAG} -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-ffn --no-scan --steps 512 --warmup 64 > $R/gpurun_out/pmc_write_${TAG}.log 2>&1
echo "pmc_write_exit=$?"
find $R/gpurun_out/prof_${TAG} $R/gpurun_out/pmc_fetch_${TAG} $R/gpurun_out/pmc_write_${TAG} -name "*.csv" | head -20
