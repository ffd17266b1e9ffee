#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), with
the gfx950 corrections of MI355X_MICROARCH.md: both counters are in KB; FETCH_SIZE under-reports by 2x.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> <out.json>"""
import csv
import json
import sys


def per_launch(path, counter, needle):
    tot, n = 0.0, 0
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter and needle in row["Kernel_Name"]:
                tot += float(row["Counter_Value"])
                n += 1
    return tot / max(n, 1), n


def main():
    fetch_csv, write_csv, needle, out = sys.argv[1:5]
    f_kb, nf = per_launch(fetch_csv, "FETCH_SIZE", needle)
    w_kb, nw = per_launch(write_csv, "WRITE_SIZE", needle)
    doc = {"kernel": needle, "launches_sampled": [nf, nw], "fetch_size_kb_raw_per_launch": f_kb,
           "fetch_bytes_per_launch_corrected_x2": f_kb * 1024 * 2, "write_bytes_per_launch": w_kb * 1024,
           "hbm_bytes_per_launch": f_kb * 1024 * 2 + w_kb * 1024,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB units; FETCH_SIZE doubled (gfx950)"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
