"""SQ counters of the prefill kernels per workload and variant -> profiles/r02_prefill_pmc.json (bench.py's prefill leg
attaches the entry of each workload). Input: OUT_ROOT/<workload>__<variant>/**/counter_collection.csv from
tools/gpu/gpu_pmc_prefill2.sh (one rocprofv3 --pmc pass of 8 SQ counters each).
Derived per kernel: valu_util = SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CYCLES-normalised SIMD cycles) is not available
without per-SIMD time, so utilisations are given against the kernel's own duration at the profiled clock:
  valu_busy_share  = SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) * 4 / (duration_cycles * 1024 SIMDs)
  mfma_busy_share  = SQ_VALU_MFMA_BUSY_CYCLES / (duration_cycles * 1024 SIMDs)
  lds_busy_share   = SQ_LDS_IDX_ACTIVE / (duration_cycles * 256 CUs);  lds_bank_conflict_share = CONFLICT / IDX_ACTIVE
duration_cycles = duration_ns * clock_ghz (default 2.1, the clock these passes ran at within a few percent)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root, out = sys.argv[1:3]
    ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.1
    doc = {"method": "rocprofv3 --pmc, one pass of 8 SQ counters per (workload, variant), bench.py --tokens 512; shares against "
                     "kernel duration x %.1f GHz x 1024 SIMDs (256 CUs for LDS)" % ghz}
    for d in sorted(glob.glob(root + "/*__*")):
        wl, variant = os.path.basename(d).split("__")
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "sdh" not in name:
                    continue
                name = name.replace("_ZN3sdh19w0_mfma_tile_kernelILb1ELb0EEEvPKDF16_lliiiPKfS4_S4_PKiPfl", "sdh::w0_mfma_tile_kernel<tall, unscaled>") \
                           .replace("_ZN3sdh19w0_mfma_tile_kernelILb0ELb0EEEvPKDF16_lliiiPKfS4_S4_PKiPfl", "sdh::w0_mfma_tile_kernel<wide, unscaled>") \
                           .replace("_ZN3sdh13w0_x16_kernelEPKflliPDF16_ll", "sdh::w0_x16_kernel")
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[name]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        ks = {}
        for k, cs in acc.items():
            m = {c: sum(v) / len(v) for c, v in cs.items()}
            cyc = m["_dur_ns"] * ghz
            r = {"duration_us": m["_dur_ns"] / 1e3,
                 "valu_busy_share": m.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (cyc * 1024),
                 "mfma_busy_share": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024),
                 "lds_busy_share": m.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * 256),
                 "lds_bank_conflict_share": (m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]) if m.get("SQ_LDS_IDX_ACTIVE") else 0.0,
                 "wave_wait_share": m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0.0,
                 "counters": {c: v for c, v in m.items() if c.startswith("SQ_")}}
            ks[k] = r
        doc.setdefault(wl, {})[variant] = ks
    json.dump(doc, open(out, "w"), indent=1)
    for wl, vs in doc.items():
        if wl == "method":
            continue
        for v, ks in vs.items():
            for k, r in ks.items():
                print("%-32s %-8s %-60s %7.1f us valu %.2f mfma %.2f lds %.2f conflict %.2f wait %.2f" % (
                    wl, v, k[:60], r["duration_us"], r["valu_busy_share"], r["mfma_busy_share"], r["lds_busy_share"],
                    r["lds_bank_conflict_share"], r["wave_wait_share"]))


if __name__ == "__main__":
    main()
