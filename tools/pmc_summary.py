#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 counter_collection.csv files, plus a few ratios.
usage: pmc_summary.py out.json file.csv [file.csv ...]"""
import csv
import json
import sys
from collections import defaultdict


def main():
    out, files = sys.argv[1], sys.argv[2:]
    doc = {}
    for path in files:
        acc = defaultdict(lambda: defaultdict(list))
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "sdh::" not in name and "3sdh" not in name:          # demangled or mangled
                    continue
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[name]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        for k, cs in acc.items():
            m = {c: sum(v) / len(v) for c, v in cs.items()}
            m["launches"] = len(cs["_dur_ns"]) // max(1, len(cs) - 1)
            wc = m.get("SQ_WAVE_CYCLES", 0) or 1
            m["valu_issue_share_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_VALU", 0) / wc
            m["lds_issue_share_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_LDS", 0) / wc
            if m.get("SQ_LDS_IDX_ACTIVE"):
                m["lds_bank_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
            if m.get("SQ_BUSY_CYCLES"):
                m["mfma_busy_share_of_sq_busy"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / m["SQ_BUSY_CYCLES"]
            doc.setdefault(path.split("/")[-3], {})[k] = m
    json.dump(doc, open(out, "w"), indent=1)
    for run, ks in doc.items():
        for k, m in ks.items():
            print(run, k[:60], {a: round(b, 4) for a, b in m.items() if "share" in a or a == "_dur_ns"})


if __name__ == "__main__":
    main()
