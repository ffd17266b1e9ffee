"""Differential fuzzing of sdh_forward against the oracle (test infrastructure; run on a GPU box):
random shapes, schemes, dtypes, token counts and flag sets for a time budget; stops at the first mismatch and
prints the case so that it can be turned into a test. Bars are those of tests/test_gpu_forward.py: EXACT_ORDER and
the tiled SpMM path (T >= 6, no NO_SPMM) bit-identical to the oracle, the other fast paths within fast_tolerance,
W0_MFMA within its f16 bound, the direct one-token launch (counted as default_direct) within fast_tolerance, the folded contraction (SDH_FWD_MFMA_FOLD, and the default / NO_FUSE dispatch where the
token count is above the payload class's crossover) within tests/test_gpu_prefill_full.py's fold_bound.
    python3 tools/fuzz_forward.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def dense_abs_sum(O, p):
    """|rs| * (|W0| + |D|) as a dense (n_out, n_in) matrix: the folded contraction's bound multiplies it with |x|"""
    A = np.abs(O.w0_tile(p, 0, p.n_out, 0, p.n_in).astype(np.float64))
    rows = np.arange(p.n_out)
    if p.d_idx is not None:
        idx, val = p.d_idx.astype(np.int64), np.abs(p.d_val.astype(np.float64))
        ok = (idx >= 0) & (idx < p.n_in)
        np.add.at(A, (np.broadcast_to(rows[:, None], idx.shape)[ok], idx[ok]), val[ok])
    else:
        bi, bv = p.b_idx.astype(np.int64), np.abs(p.b_val.astype(np.float64))
        block = bv.shape[2]
        cols = bi[:, :, None] * block + np.arange(block)[None, None, :]
        ok = (bi[:, :, None] >= 0) & (cols < p.n_in)
        np.add.at(A, (np.broadcast_to(rows[:, None, None], cols.shape)[ok], cols[ok]), bv[ok])
    if p.row_scale is not None:
        A *= np.abs(p.row_scale.astype(np.float64))[:, None]
    return A


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import llama_cpp_amd                                             # noqa: F401
    from llama_cpp_amd import seeddelta as hip
    from llama_cpp_amd.payload import make_payload, make_x
    import oracle_py as O
    from test_gpu_forward import fast_tolerance
    hip.init(0)
    rng = np.random.default_rng(seed)
    t0, n, by = time.time(), 0, {}
    last = t0
    dims_pool = [64, 96, 128, 200, 256, 320, 512, 700, 1024, 1152, 1500, 2048, 2500, 4096, 5000]
    while time.time() - t0 < budget:
        n_in, n_out = int(rng.choice(dims_pool)), int(rng.choice(dims_pool))
        n_in = (n_in + 7) // 8 * 8 if rng.random() < 0.7 else n_in
        scheme = "coo" if rng.random() < 0.5 else "block"
        block = int(rng.choice([8, 12, 16, 32]))
        K = int(rng.choice([1, 8, 20, 32, 64, 96, 128, 256, 384, 512, 1024, 1536]))
        base = bool(rng.random() < 0.7)
        T = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 17, 40, 130, 170, 260, 300, 420]))
        direct_case = None
        if T == 1 and rng.random() < 0.6:                  # shapes the direct one-token launch takes: staged x = 256 * lanes per row
            direct_case = int(rng.integers(4))
            r8 = lambda lo, hi: int(rng.integers(lo // 8 + 1, hi // 8 + 1)) * 8
            r16 = lambda lo, hi: int(rng.integers(lo // 16 + 1, hi // 16 + 1)) * 16
            if direct_case == 0:   n_in, n_out, K = r8(1024, 2048), r16(2048, 6000), 64                             # tall, 8 lanes per row
            elif direct_case == 1: n_in, n_out, K = r8(2048, 4096), r16(4096, 9000), int(rng.choice([96, 128]))     # tall, 16 lanes
            elif direct_case == 2: n_in, n_out, K = r8(6144, 8192), r16(1024, 2048), 256                            # wide, 32 lanes, 4 blocks
            else:                  n_in, n_out, K = r8(14336, 16384), r16(1024, 2048), int(rng.choice([384, 512, 1024, 1536]))   # wide, 64 lanes, 8 blocks
            block = int(rng.choice([8, 16, 32]))
            base = True
        kw = dict(scheme=scheme, K=K, block=block, base=base, row_scale=bool(rng.random() < 0.8), d13_random=bool(base and rng.random() < 0.5),
                  seed=int(rng.integers(1 << 30)))
        if direct_case is not None:
            kw["d13_random"] = bool(rng.random() < 0.15)
        elif rng.random() < 0.25:
            kw.update(idx_dtype=np.int32 if rng.random() < 0.5 else np.int16, val_dtype=np.float32 if rng.random() < 0.5 else np.float16)
        if n_in > 32767:
            kw["idx_dtype"] = np.int32
        try:
            p = make_payload(n_in, n_out, **kw)
        except Exception:
            continue
        x = make_x(n_in, T, seed=int(rng.integers(1 << 30)))
        y_ref = O.forward(p, x, nth=8)
        try:
            w = hip.SeedDeltaWeight(p)
        except hip.SeedDeltaError:
            continue
        flagsets = [("default", 0), ("exact", hip.FWD_EXACT_ORDER), ("no_fuse", hip.FWD_NO_FUSE), ("no_spmm", hip.FWD_NO_SPMM)]
        if base:
            flagsets.append(("w0_mfma", hip.FWD_W0_MFMA))
            flagsets.append(("fold", hip.FWD_MFMA_FOLD))            # after default / no_fuse: fold_layout() != 0 there means THEY folded
        tol = None
        fold_A = None
        for name, fl in flagsets:
            y = w.forward_host(x, fl)
            folded = base and T >= 5 and name in ("default", "no_fuse", "fold") and w.fold_layout() != 0
            if base and T == 1 and name == "default" and hip.last_decode_kind() == 1:
                name = "default_direct"
            exact = not folded and (name == "exact" or (name in ("default", "no_fuse") and T >= 6))
            if folded:
                name = name if name == "fold" else name + "_folded"
            ok = True
            if exact:
                ok = np.array_equal(y.view(np.uint32), y_ref.view(np.uint32))
                if not ok and name != "exact":                           # a shape the tiled path declines runs the fast kernels
                    exact = False
            if not exact:
                if tol is None:
                    tol = fast_tolerance(p, x, y_ref)
                bound = tol
                if folded:
                    if fold_A is None:
                        fold_A = dense_abs_sum(O, p)
                    bound = 1.5e-3 * (np.abs(x.astype(np.float64)) @ fold_A.T) + 75.0 * tol
                if name == "w0_mfma":
                    W0 = np.abs(O.w0_tile(p, 0, n_out, 0, n_in).astype(np.float64))
                    rs = np.abs(p.row_scale.astype(np.float64))[None, :] if p.row_scale is not None else 1.0
                    bound = 1.5e-3 * rs * (np.abs(x.astype(np.float64)) @ W0.T) + tol
                err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
                ok = bool(np.all(err <= bound))
            by[name] = by.get(name, 0) + 1
            if not ok:
                print("MISMATCH", dict(n_in=n_in, n_out=n_out, T=T, flags=name, **{k: (v.__name__ if isinstance(v, type) else v) for k, v in kw.items()}))
                sys.exit(1)
        w.close()
        n += 1
        if time.time() - last > 20:
            last = time.time()
            print(f"  ... {n} payloads, {time.time() - t0:.0f}s", flush=True)
    print(f"fuzz_forward: {n} payloads, runs by flag set {by}, {time.time() - t0:.0f}s, seed {seed}: all within their bars")


if __name__ == "__main__":
    main()
