#!/bin/bash
# SQ counters of the folded prefill contraction, two passes of 8 (gate workload)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc_fold
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
wl=${WL:-cfg4_mistral_gate_coo_k128}
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY"
C2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
C3="SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
i=1
for C in "$C1" "$C2" "$C3"; do
  SDH_BENCH_FLAGS=64 timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pass$i -- python3 $R/bench.py --tokens 512 --payloads 2 --steps 8 --warmup 2 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline --workload $wl > $O/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pass$i.log; exit 1; }
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "mfma_kernel" not in k: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(p.split("/")[-3] if "pass" in p.split("/")[-3] else p)
    for c, (v, n) in sorted(acc.items()): print("  %-28s %14.0f per dispatch (%d)" % (c, v / n, n))
PY
find $O -name "*.csv" -size +1M -delete
