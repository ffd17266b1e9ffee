#!/bin/bash
# SQ counters of the direct one-token decode launch on cfg3, two passes of 8 (each pass a run of its own)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc_direct
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"
C2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"
i=1
for C in "$C1" "$C2"; do
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pass$i -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-scan --no-ffn --no-stack --no-prefill --no-roofline > $O/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pass$i.log; exit 1; }
  i=$((i+1))
done
python3 - <<PY > $R/gpurun_out/r03_y_pmc_direct.txt
import csv, glob, collections
print("SQ counters of decode_direct_kernel<true,3,4,true> (cfg3, one token), rocprofv3 --pmc, two passes of 8 counters (a run each);")
print("per dispatch, summed over the chip's shader engines as rocprofv3 reports them. 256 workgroups of 16 waves.")
tot = {}
for p in sorted(glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        if "decode_direct_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for c, (v, n) in sorted(acc.items()):
        tot[c] = v / n
        print("  %-24s %14.0f per dispatch (%d dispatches)" % (c, v / n, n))
w = tot.get("SQ_WAVE_CYCLES", 0)
if w:
    print("shares of the waves' resident cycles (SQ_WAVE_CYCLES):")
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY"):
        if c in tot: print("  %-24s %.3f" % (c, tot[c] / w))
if tot.get("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank-conflict share of LDS-active cycles: %.3f" % (tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"]))
PY
cat $R/gpurun_out/r03_y_pmc_direct.txt
find $O -name "*.csv" -size +1M -delete
