#!/bin/bash
# round-2 GPU session: every gpu test, smoke, the bench (default and --steps 20 --warmup 5), a kernel-stats pass of the
# default bench command, and the HBM-traffic counter passes for the headline launch and for the FFN-stack kernel.
# Usage: tools/gpu/gpu_round2.sh TAG [skip-tests]
set -o pipefail
TAG=${1:-r02_x}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
if [ "$2" != "skip-tests" ]; then
  timeout -k 10 900 python -u -m pytest $R/tests -m gpu -x -q --timeout 200 --timeout-method thread > $R/gpurun_out/pytest_gpu.log 2>&1
  rc=$?; echo "pytest_exit=$rc"; tail -3 $R/gpurun_out/pytest_gpu.log
  [ $rc -ne 0 ] && exit $rc
  timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/smoke.log 2>&1
  rc=$?; echo "smoke_exit=$rc"; tail -2 $R/gpurun_out/smoke.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 700 python -u $R/bench.py > $R/gpurun_out/bench_${TAG}.log 2> $R/gpurun_out/bench_${TAG}.err
rc=$?; echo "bench_exit=$rc"; tail -1 $R/gpurun_out/bench_${TAG}.log | cut -c1-400
[ $rc -ne 0 ] && { tail -5 $R/gpurun_out/bench_${TAG}.err; exit $rc; }
timeout -k 10 500 python -u $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/bench_${TAG}_s20.log 2> $R/gpurun_out/bench_${TAG}_s20.err
rc=$?; echo "bench_s20_exit=$rc"; tail -1 $R/gpurun_out/bench_${TAG}_s20.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/rocprof_${TAG}.log 2>&1
rc=$?; echo "rocprof_exit=$rc"
[ $rc -ne 0 ] && { tail -3 $R/gpurun_out/rocprof_${TAG}.log; exit $rc; }
f=$(find $R/gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_bench_kernel_stats.csv; head -8 "$f" | cut -c1-160
find $R/gpurun_out/prof_${TAG} -name "*_trace.csv" -delete
# HBM traffic: separate passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2)
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
  IFS=: read tag ctr <<< "$c"
  rm -rf $R/gpurun_out/pmc_${tag}_hl $R/gpurun_out/pmc_${tag}_stk
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_${tag}_hl -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-ffn --no-stack --no-prefill --no-scan --steps 16 --warmup 4 > $R/gpurun_out/pmc_${tag}_hl.log 2>&1 || { echo "pmc $tag headline failed"; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_${tag}_stk -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-ffn --no-prefill --no-scan --steps 2 --warmup 1 --stack-reps 4 > $R/gpurun_out/pmc_${tag}_stk.log 2>&1 || { echo "pmc $tag stack failed"; exit 1; }
done
hl_f=$(find $R/gpurun_out/pmc_fetch_hl -name "*counter_collection.csv" | head -1); hl_w=$(find $R/gpurun_out/pmc_write_hl -name "*counter_collection.csv" | head -1)
st_f=$(find $R/gpurun_out/pmc_fetch_stk -name "*counter_collection.csv" | head -1); st_w=$(find $R/gpurun_out/pmc_write_stk -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_traffic.py $hl_f $hl_w decode_direct $R/gpurun_out/${TAG}_traffic_cfg3.json
python3 $R/tools/pmc_traffic.py $st_f $st_w ffn_stack_kernel $R/gpurun_out/${TAG}_traffic_stack.json
find $R/gpurun_out/pmc_fetch_hl $R/gpurun_out/pmc_write_hl $R/gpurun_out/pmc_fetch_stk $R/gpurun_out/pmc_write_stk -name "*.csv" -size +1M -delete
