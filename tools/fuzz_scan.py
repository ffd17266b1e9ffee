"""Differential fuzzing of the K-schedule scan call (sdh_scan_tensor_kgrid, both draw modes) against lone
sdh_scan_tensor calls on random tensors, schemes and schedules; every result field and the fetched payload of a
random member must be identical. Test infrastructure; run on a GPU box.   python3 -u tools/fuzz_scan.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELDS = ("K_eff", "n_blocks", "L", "B", "has_x", "has_w", "proxy_available", "em", "em_x", "em_w", "em_x_w", "cost", "proxy",
          "delta_clamp_scale")


def same(a, b):
    for k in FIELDS:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray):
            if not np.array_equal(x.view(np.uint64), y.view(np.uint64)):
                return k
        elif x != y:
            return k
    return None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import llama_cpp_amd                                             # noqa: F401
    from llama_cpp_amd import scan as SC, seeddelta as sd
    from llama_cpp_amd.dense import make_dense
    sd.init(0)
    rng = np.random.default_rng(seed)
    t0, n, last = time.time(), 0, time.time()
    while time.time() - t0 < budget:
        n_embd = int(rng.choice([256, 320, 512, 1024, 2048]))
        n_ff = int(rng.choice([512, 800, 1408, 2560, 4096]))
        wtype = "q8_0" if rng.random() < 0.7 else "f16"
        kind = str(rng.choice(SC.KINDS))
        W = {k: SC.DenseWeight(make_dense(*((n_ff, n_embd) if k == "ffn_down" else (n_embd, n_ff)), wtype,
                                          seed=int(rng.integers(1 << 30)), structure=0.4)) for k in SC.KINDS}
        n_in = W[kind].n_in
        scheme = "coo" if rng.random() < 0.6 else "block"
        base = bool(rng.random() < 0.6)
        kw = dict(scheme=scheme, block=int(rng.choice([8, 16, 32])), base=base, row_scale=bool(rng.random() < 0.7),
                  base_max_samples=int(rng.choice([64, 128, 300])), base_perm_trials=int(rng.choice([1, 2, 4])),
                  eval_cols=int(rng.choice([8, 16, 32])), eval_x=int(rng.choice([0, 2, 5])), seed=int(rng.integers(1 << 20)),
                  metrics_only=bool(rng.random() < 0.5))
        ws = None
        if rng.random() < 0.3:
            imat = (rng.random(n_in).astype(np.float32) + 0.05) ** 2
            ws = SC.imatrix_scale(imat, n_in, eps=1e-8, power=1.0)
        p = SC.ScanParams(w_scale=ws, **kw)
        Ks = [int(k) for k in rng.choice([1, 4, 8, 16, 33, 64, 100, 128, 200, 300, 5000], size=int(rng.integers(1, 7)))]
        sequential = bool(rng.random() < 0.5)
        layer = int(rng.integers(0, 4))
        ffn = (W["ffn_gate"], W["ffn_up"], W["ffn_down"])
        s = SC.ScanSession(p.seed)
        grid = s.tensor_kgrid(W[kind], p, Ks, kind, layer, ffn, sequential=sequential)
        t = SC.ScanSession(p.seed)
        bad = None
        for j, K in enumerate(Ks):
            if not sequential:
                t.reseed(p.seed)
            pj = SC.ScanParams(**{**p.__dict__, "K": K, "base_mode": (1 if sequential else 2) if (j > 0 and base) else 0})
            lone = t.tensor(W[kind], pj, kind, layer, ffn)
            bad = same(grid[j], lone)
            if bad:
                break
            if not p.metrics_only and rng.random() < 0.4:
                s.pick(j)
                a, b = s.fetch_packed(), t.fetch_packed()
                for k in a:
                    if a[k] is not None and a[k].tobytes() != b[k].tobytes():
                        bad = "payload " + k
                if bad:
                    break
        if bad:
            print("MISMATCH", bad, dict(n_embd=n_embd, n_ff=n_ff, wtype=wtype, kind=kind, Ks=Ks, sequential=sequential, imatrix=ws is not None, **kw))
            sys.exit(1)
        s.close(); t.close()
        for w in W.values():
            w.close()
        n += 1
        if time.time() - last > 20:
            last = time.time()
            print(f"  ... {n} schedules, {time.time() - t0:.0f}s", flush=True)
    print(f"fuzz_scan: {n} schedules, {time.time() - t0:.0f}s, seed {seed}: K-schedule calls identical to lone calls")


if __name__ == "__main__":
    main()
