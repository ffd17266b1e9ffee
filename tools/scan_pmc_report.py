"""HBM traffic of the scan phases from rocprofv3 --pmc passes over tools/scan_idle.py --only PHASE (FETCH_SIZE and
WRITE_SIZE in separate passes; both in KB; FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes): sums
over every sdh:: kernel dispatch of the process, divided by the candidates the phase evaluated.
    python3 tools/scan_pmc_report.py OUT_ROOT out.json      (expects OUT_ROOT/{sweep,payload}_{fetch,write}/ + .log)"""
import csv
import glob
import json
import sys


def total(d, counter):
    tot, n, per = 0.0, 0, {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "sdh::" in row["Kernel_Name"]:
                v = float(row["Counter_Value"])
                tot += v
                n += 1
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
                per[k] = per.get(k, 0.0) + v
    return tot, n, per


def main():
    root, out = sys.argv[1:3]
    doc = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, over tools/scan_idle.py --only PHASE "
                     "(no warm-up: every sdh:: dispatch belongs to the phase); KB units; FETCH_SIZE doubled (gfx950)"}
    for phase in ("sweep", "payload"):
        line = json.loads([ln for ln in open(f"{root}/{phase}_fetch.log") if ln.startswith('{"sweep"')][-1])
        n_c = line[phase]["candidates"]
        f_kb, nf, per_f = total(f"{root}/{phase}_fetch", "FETCH_SIZE")
        w_kb, nw, per_w = total(f"{root}/{phase}_write", "WRITE_SIZE")
        doc[phase] = {"candidates": n_c, "layers": line["layers"], "dispatches": [nf, nw],
                      "fetch_bytes_corrected_x2": f_kb * 2048, "write_bytes": w_kb * 1024,
                      "hbm_bytes_per_candidate": (f_kb * 2048 + w_kb * 1024) / n_c,
                      "fetch_bytes_by_kernel": {k: v * 2048 for k, v in sorted(per_f.items(), key=lambda kv: -kv[1])[:6]},
                      "write_bytes_by_kernel": {k: v * 1024 for k, v in sorted(per_w.items(), key=lambda kv: -kv[1])[:6]}}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: (v if k == "method" else {a: b for a, b in v.items() if "by_kernel" not in a}) for k, v in doc.items() if k != "method"}))


if __name__ == "__main__":
    main()
