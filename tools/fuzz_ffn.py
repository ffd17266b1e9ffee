"""Differential fuzzing of sdh_ffn_forward (test infrastructure; run on a GPU box): random layer shapes, payload kinds,
activations and token counts for a time budget. Bars: every token's output equals the output of its own one-token call
bit for bit whenever the call runs on the fast-path forms (1..4 tokens; COO payloads at any count; block payloads up to
32 tokens — the gate/up nodes of larger block batches take the tiled SpMM, whose rows are the CPU op's bits instead),
and the whole call is within the propagated tolerance of the oracle's graph (tests/test_gpu_ffn.py: h_bounds, check_y).
      python3 tools/fuzz_ffn.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import llama_cpp_amd                                             # noqa: F401
    from llama_cpp_amd import seeddelta as hip
    from llama_cpp_amd.payload import make_payload, make_x
    import oracle_py as O
    from test_gpu_ffn import h_bounds, check_y
    hip.init(0)
    rng = np.random.default_rng(seed)
    t0, n, last, fused, by_t = time.time(), 0, time.time(), 0, {}
    while time.time() - t0 < budget:
        n_embd = int(rng.choice([1024, 2048, 4096]))
        n_ff = int(rng.choice([1000, 2816, 3000, 5632, 6912, 9000, 14336]))
        scheme = "coo" if rng.random() < 0.5 else "block"
        block = int(rng.choice([8, 16, 32]))
        K = int(rng.choice([16, 32, 64, 128, 256]))
        act = int(rng.choice([O.ACT_SILU, O.ACT_GELU]))
        T = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 16, 17, 31, 32, 33, 40, 70]))
        kw = dict(scheme=scheme, K=K, block=block, base=True, d13_random=bool(rng.random() < 0.5), row_scale=bool(rng.random() < 0.8))
        try:
            gate = make_payload(n_embd, n_ff, seed=int(rng.integers(1 << 30)), **kw)
            up = make_payload(n_embd, n_ff, seed=int(rng.integers(1 << 30)), **kw)
            down = make_payload(n_ff, n_embd, seed=int(rng.integers(1 << 30)), **kw)
        except Exception:
            continue
        wg, wu, wd = hip.SeedDeltaWeight(gate), hip.SeedDeltaWeight(up), hip.SeedDeltaWeight(down)
        x = make_x(n_embd, T, seed=int(rng.integers(1 << 30)))
        y = hip.ffn_forward_host(wg, wu, wd, x, act)
        fused += int(hip.ffn_last_fused())
        case = dict(n_embd=n_embd, n_ff=n_ff, scheme=scheme, K=K, block=block, act=act, T=T, d13=kw["d13_random"], rs=kw["row_scale"])
        smooth, full, h_ref = h_bounds(gate, up, x, act)
        y_ref = O.forward(down, O.glu(act, O.forward(gate, x), O.forward(up, x)))
        try:
            check_y(y, y_ref.astype(np.float64), down, h_ref.astype(np.float32), full)
        except AssertionError as e:
            print("MISMATCH vs oracle", case, e)
            sys.exit(1)
        if T <= 4 or scheme == "coo" or T <= 32:
            pick = sorted(set([0, T - 1, T // 2]))
            for t in pick:
                y1 = hip.ffn_forward_host(wg, wu, wd, x[t:t + 1], act)
                if not np.array_equal(y1.view(np.uint32), y[t:t + 1].view(np.uint32)):
                    print("MISMATCH vs one-token call", case, "token", t, float(np.abs(y1 - y[t:t + 1]).max()))
                    sys.exit(1)
        if wg.nonfinite_count() or wu.nonfinite_count() or wd.nonfinite_count():
            print("NONFINITE", case)
            sys.exit(1)
        for w in (wg, wu, wd):
            w.close()
        n += 1
        by_t[T] = by_t.get(T, 0) + 1
        if time.time() - last > 20:
            last = time.time()
            print(f"  ... {n} layers, {time.time() - t0:.0f}s", flush=True)
    print(f"fuzz_ffn: {n} layers ({fused} through a single gate|up launch), calls by token count {dict(sorted(by_t.items()))}, "
          f"{time.time() - t0:.0f}s, seed {seed}: all within their bars")


if __name__ == "__main__":
    main()
