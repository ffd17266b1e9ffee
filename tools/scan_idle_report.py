"""Cut a rocprofv3 kernel trace (csv) of tools/scan_idle.py into phases at idle gaps >= 0.3 s and report, per phase,
the span, the union of kernel intervals (GPU busy), the idle share and the kernels by summed duration.
    python3 tools/scan_idle_report.py OUT_DIR [hip_api_stats.csv]"""
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
    rows.sort()
    phases, cur = [], []
    for r in rows:
        if cur and r[0] - max(x[1] for x in cur[-64:]) > 300e6:
            phases.append(cur)
            cur = []
        cur.append(r)
    phases.append(cur)
    names = ["init+warm-up", "metrics-only sweep", "payload-emission sweep"]
    out = []
    for n, ph in enumerate(phases[-3:]):
        t0, t1 = ph[0][0], max(x[1] for x in ph)
        busy, end = 0, t0
        for s, e, _ in ph:
            if e > end:
                busy += e - max(s, end)
                end = e
        per = {}
        for s, e, k in ph:
            a = per.setdefault(k, [0, 0])
            a[0] += e - s
            a[1] += 1
        top = sorted(per.items(), key=lambda kv: -kv[1][0])[:8]
        out.append({"phase": names[n] if len(phases) >= 3 else f"phase{n}", "span_ms": (t1 - t0) / 1e6, "gpu_busy_ms": busy / 1e6,
                    "idle_share": 1.0 - busy / max(1, t1 - t0), "kernels": len(ph),
                    "top": [{"kernel": k, "total_ms": v[0] / 1e6, "calls": v[1]} for k, v in top]})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
