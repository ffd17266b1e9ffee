"""The scan sweep of bench.py's cfg5 leg in three phases separated by idle gaps (0.5 s), so that a kernel trace
of this process can be cut into phases by its gaps (tools/scan_idle_report.py): warm-up | metrics-only sweep |
payload-emission sweep. Prints the host wall time of each phase as one JSON line.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/scan_idle.py [--layers N] [--threads T]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                           # noqa: E402  (scan_model, scan_variants, SCAN_K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--payload-groups", type=int, default=24)
    ap.add_argument("--only", choices=["sweep", "payload"], help="run just this phase, without warm-up (counter passes: every "
                    "sdh:: dispatch of the process then belongs to the phase)")
    ap.add_argument("--keep-memo", action="store_true", help="do not clear the draw memo before the timed sweep")
    a = ap.parse_args()
    import torch
    from llama_cpp_amd import scan as SC, seeddelta as sd
    sd.init(0)
    variants, full = B.scan_variants(True), B.scan_variants(False)
    cands = []
    for v in variants:
        for c in SC.enumerate_candidates(range(a.layers), SC.KINDS, B.SCAN_K):
            c.variant = v
            cands.append(c)
    ev = SC.Evaluator(B.scan_model, variants[0], max_resident=3 * a.layers, dims_of=B.scan_dims)
    for c in cands:
        for k in (SC.KINDS if c.kind != "ffn_down" else (c.kind,)):
            ev.dense(c.layer, k)
    warm = cands[:: max(1, len(cands) // (8 * a.threads))][: 8 * a.threads]
    if not a.only:
        ev.run_many(warm, threads=a.threads, params_of=lambda c: variants[c.variant])
        ev.run_many(warm[:12], threads=a.threads, params_of=lambda c: full[c.variant])
    if not a.keep_memo:
        sd.load_library().sdh_scan_memo_clear()
    torch.cuda.synchronize()
    time.sleep(0.5)
    t0 = time.perf_counter()
    if a.only != "payload":
        ev.run_many(cands, threads=a.threads, params_of=lambda c: variants[c.variant])
    torch.cuda.synchronize()
    t_sweep = time.perf_counter() - t0
    time.sleep(0.5)
    groups = {}
    for c in cands:
        groups.setdefault((c.variant, c.layer, c.kind), []).append(c)
    keys = sorted(groups)
    keys = keys[:: max(1, len(keys) // a.payload_groups)][: a.payload_groups]
    sample = [c for k in keys for c in groups[k]]
    t0 = time.perf_counter()
    if a.only != "sweep":
        ev.run_many(sample, threads=a.threads, params_of=lambda c: full[c.variant])
    torch.cuda.synchronize()
    t_pay = time.perf_counter() - t0
    ev.close()
    print(json.dumps({"sweep": {"candidates": len(cands), "seconds": t_sweep, "per_s": len(cands) / t_sweep},
                      "payload": {"candidates": len(sample), "schedules": len(keys), "seconds": t_pay, "per_s": len(sample) / t_pay},
                      "threads": a.threads, "layers": a.layers, "only": a.only}))


if __name__ == "__main__":
    main()
