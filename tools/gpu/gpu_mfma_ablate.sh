#!/bin/bash
# timing experiments on the tiled W0-MFMA kernel (results are wrong on purpose; only kernel time is read)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  rm -rf $R/gpurun_out/prof_abl
  SDH_MT_ABLATE=$a SDH_BENCH_FLAGS=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_abl -- python3 $R/bench.py --tokens 512 --payloads 4 --steps 16 --warmup 4 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload ${WL:-cfg4_mistral_gate_coo_k128} > $R/gpurun_out/abl.log 2>&1 || { tail -3 $R/gpurun_out/abl.log; exit 1; }
  f=$(find $R/gpurun_out/prof_abl -name "*kernel_stats.csv" | head -1)
  echo "ablate=$a $(grep w0_mfma_tile $f | cut -d, -f2-4)"
done
rm -rf $R/gpurun_out/prof_abl
