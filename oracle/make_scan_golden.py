#!/usr/bin/env python3
"""Freeze scan goldens from the REFERENCE's own builder code (oracle/_ref/libsdbuildref.so =
tools/seeddelta-build/{sd_eval,sd_cost,sd_ffn_proxy}.cpp compiled unmodified) into
tests/golden/scan_*.npz, for the GPU box where /root/reference does not exist.

Per case one synthetic FFN layer (gate, up [n_embd -> n_ff], down [n_ff -> n_embd]; Q8_0 or F32) and one
candidate on one of its tensors. Stored: the weights, the parameters, and what the reference computes
with an mt19937(seed) stream consumed in builder order — base fit arrays, weight-space metrics,
activation-space metrics, FFN proxy, cost. The per-row selection (run_trial's loop, which cannot be
compiled here) comes from the oracle restatement; the reference evals above consume exactly those
arrays. The selection itself is not stored (the tests recompute it with the oracle, which travels).

Run in the build container:  python oracle/make_scan_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    sys.path.insert(0, p)

import llama_cpp_amd  # noqa: E402,F401
import scan_py as S  # noqa: E402
from llama_cpp_amd.dense import make_dense  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CASES = [
    # name, n_embd, n_ff, wtype, kind, scheme, K, block, base, row_scale
    ("gate_coo_k12_nobase_q8", 256, 704, "q8_0", 0, "coo", 12, 32, False, False),
    ("up_coo_k32_nobase_rs_q8", 256, 704, "q8_0", 1, "coo", 32, 32, False, True),
    ("gate_coo_k64_base_rs_q8", 256, 704, "q8_0", 0, "coo", 64, 32, True, True),
    ("down_coo_k48_base_rs_q8", 256, 704, "q8_0", 2, "coo", 48, 32, True, True),
    ("down_block16_k64_base_rs_q8", 256, 704, "q8_0", 2, "block", 64, 16, True, True),
    ("gate_block32_k96_base_f32", 256, 704, "f32", 0, "block", 96, 32, True, False),
    ("up_block16_k40_nobase_rs_f32", 320, 800, "f32", 1, "block", 40, 16, False, True),
    ("down_coo_k12_nobase_q8", 256, 704, "q8_0", 2, "coo", 12, 32, False, False),
]
SEED, EVAL_COLS, EVAL_X, MAX_SAMPLES, TRIALS, LAYER = 1234, 24, 6, 96, 3, 5


def main():
    assert S.buildref_available(), "build oracle/_ref first (make -C oracle buildref)"
    for name, n_embd, n_ff, wtype, kind, scheme, K, block, base, row_scale in CASES:
        Wg = make_dense(n_embd, n_ff, wtype, seed=11)
        Wu = make_dense(n_embd, n_ff, wtype, seed=12)
        Wd = make_dense(n_ff, n_embd, wtype, seed=13)
        W = (Wg, Wu, Wd)[kind]
        n_in, n_out = S.w_dims(W)
        ref = S.Engine("ref", SEED)
        sch = 1 if scheme == "block" else 0
        Kc = max(1, min(K, n_in))
        keep = (Kc + block - 1) // block
        Ksel = keep if sch else Kc
        fit = ref.fit_base(W, MAX_SAMPLES, TRIALS) if base else None
        idx, val, rs = S.select_rows(W, fit, sch, Ksel, block, row_scale)
        em = ref.eval_w(W, fit, sch, idx, val, Ksel, block, rs, EVAL_COLS)
        em_x = ref.eval_x(W, fit, sch, idx, val, Ksel, block, rs, EVAL_COLS, EVAL_X)
        L, B = (fit.L, fit.B) if base else (0, 0)
        cost = ref.cost(L, B, n_in, n_out, keep * block if sch else Kc, row_scale)
        proxy_ok, proxy = False, np.zeros(8)
        if sch == 0 and not base:
            proxy_ok, proxy = ref.ffn_proxy(kind, Wg, Wu, Wd, idx, val, Ksel, EVAL_X, EVAL_COLS, SEED + 101 * LAYER + kind + 1)
        keep_w = dict(Wg=Wg, Wu=Wu, Wd=Wd) if (sch == 0 and not base) else {("Wg", "Wu", "Wd")[kind]: W}
        d = dict(**keep_w, kind=kind, scheme=sch, K=K, block=block, base=base, row_scale=row_scale,
                 seed=SEED, eval_cols=EVAL_COLS, eval_x=EVAL_X, max_samples=MAX_SAMPLES, trials=TRIALS, layer=LAYER,
                 em=em, em_x=em_x, cost=cost, proxy_ok=proxy_ok, proxy=proxy,
                 next_u32=np.uint32(ref.u32()))
        if base:
            d.update(fit_d2=fit.d2, fit_h2=fit.h2, fit_perm1=fit.perm1, fit_inv=fit.perm1_inv)
        np.savez_compressed(os.path.join(OUT, "scan_" + name + ".npz"), **d)
        print(f"{name:36s} em cos={em[2]:.4f} em_x cos={em_x[2]:.4f} proxy={proxy_ok} cost_ratio={cost[7]:.4f}")
        ref.close()


if __name__ == "__main__":
    main()
