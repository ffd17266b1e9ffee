#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own translation unit (oracle/_ref/libsdref.so,
i.e. /root/reference/src/llama-seeddelta.cpp compiled unmodified with its stock flags).

Run in the build container only (needs /root/reference to build oracle/_ref):

    python oracle/make_golden.py

Each fixture stores the inputs (payload tensors, x) and the reference's output y, nothing else.
The reference ships no golden vectors for this path, so these pin the oracle and
the HIP kernels to what the reference computes on identical inputs. Edge cases follow SURVEY.md
§7 step 1: -1 / out-of-range indices, ragged last block, n_in not a multiple of block, partially
filled last base block, out-of-range perm entries, I16/I32, F16/F32, with/without row_scale, a
row_scale that is exactly 1 on some rows, NaN repair with and without a dense weight.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import llama_cpp_amd  # noqa: E402,F401
from llama_cpp_amd.payload import make_payload, make_x  # noqa: E402
import oracle_py as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

FIELDS = ("d_idx", "d_val", "b_idx", "b_val", "row_scale", "base_d1", "base_d2", "base_d3",
          "base_perm1", "base_perm2", "w_ref")


def cases():
    f16, f32, i16, i32 = np.float16, np.float32, np.int16, np.int32
    c = []

    def add(name, n_in, n_out, T, mutate=None, **kw):
        c.append((name, n_in, n_out, T, mutate, kw))

    # plain COO / block, tall and wide, both dtypes
    add("coo_tall_f16_i16", 96, 256, 3, scheme="coo", K=24)
    add("coo_wide_f32_i32", 320, 64, 2, scheme="coo", K=40, idx_dtype=i32, val_dtype=f32, rs_dtype=f32)
    add("coo_k_not_mult8", 100, 130, 2, scheme="coo", K=13)
    add("coo_no_rowscale", 128, 128, 1, scheme="coo", K=16, row_scale=False)
    add("block16_tall", 128, 384, 3, scheme="block", K=64, block=16)
    add("block32_wide_f32", 512, 96, 2, scheme="block", K=96, block=32, val_dtype=f32, idx_dtype=i32)
    add("block16_ragged_nin", 120, 200, 2, scheme="block", K=48, block=16)      # n_in % block != 0
    add("block_odd_len", 60, 70, 2, scheme="block", K=20, block=5)               # block % 8 != 0
    # base, tall and wide, partially filled last base block, random d1/d3
    add("base_coo_tall", 64, 200, 2, scheme="coo", K=8, base=True, d13_random=True)      # L=64 B=4 (last partial)
    add("base_coo_wide", 200, 64, 3, scheme="coo", K=16, base=True, d13_random=True)     # L=64 B=4
    add("base_block_tall", 256, 640, 2, scheme="block", K=32, block=16, base=True)       # L=256 B=3 (partial)
    add("base_block_wide", 896, 256, 2, scheme="block", K=64, block=32, base=True)       # L=256 B=4 (3.5 chunks)
    add("base_block_wide_f32base", 448, 128, 1, scheme="block", K=32, block=16, base=True,
        base_dtype=f32, perm_dtype=i32, d13_random=True)
    add("base_coo_noperm", 128, 300, 2, scheme="coo", K=8, base=True, perm=False)
    add("base_nonpow2_dims", 100, 36, 2, scheme="coo", K=6, base=True, d13_random=True)  # L=64 B=2

    def bad_coo_idx(p, rng):
        p.d_idx[::3, 0] = -1
        p.d_idx[1::3, 1] = p.n_in          # == n_in -> skipped
        p.d_idx[2::3, 2] = 32000           # way out of range
    add("coo_bad_indices", 96, 128, 2, bad_coo_idx, scheme="coo", K=12)

    def bad_blk_idx(p, rng):
        p.b_idx[::2, 0] = -1
        p.b_idx[1::2, 1] = (p.n_in + 15) // 16      # first block past the end -> skipped
    add("block_bad_indices", 120, 96, 2, bad_blk_idx, scheme="block", K=48, block=16)

    def scale_one(p, rng):
        p.row_scale[::2] = 1.0                       # exercises (scale != 1) ? y*scale : y
    add("coo_rowscale_ones", 64, 96, 2, scale_one, scheme="coo", K=8)

    def bad_perm(p, rng):
        p.base_perm1[0, 3] = -5                      # out of range -> reads element 0
        p.base_perm1[1, 7] = p.base_perm1.shape[1]   # == L -> reads element 0
        p.base_perm1[2, 9] = p.base_perm1[2, 10]     # duplicate: perm need not be a bijection
    add("base_bad_perm", 64, 256, 2, bad_perm, scheme="coo", K=8, base=True)

    def depth1(p, rng):
        p.base_d2 = None                             # depth-1 shortcut (:537-544)
    add("base_depth1", 64, 128, 2, depth1, scheme="coo", K=8, base=True, d13_random=True)

    def with_perm2(p, rng):
        L = p.base_d1.shape[1]
        p.base_perm2 = np.stack([rng.permutation(L) for _ in range(p.base_d1.shape[0])]).astype(np.int16)
    add("base_perm2", 64, 128, 2, with_perm2, scheme="block", K=16, block=8, base=True, d13_random=True)

    def no_d3(p, rng):
        p.base_d3 = None
    add("base_no_d3", 128, 64, 2, no_d3, scheme="coo", K=8, base=True)

    def inf_vals_no_w(p, rng):
        p.d_val[5, 0] = np.float16(np.inf)           # -> Inf/NaN output, no dense weight -> 0.0
        p.d_val[9, 1] = np.float16(np.nan)
    add("coo_nonfinite_no_wref", 64, 96, 2, inf_vals_no_w, scheme="coo", K=8)

    def inf_vals_w(p, rng):
        p.d_val[5, 0] = np.float16(np.inf)
        p.d_val[9, 1] = np.float16(np.nan)
    add("coo_nonfinite_wref_f32", 64, 96, 2, inf_vals_w, scheme="coo", K=8, with_w_ref=True)
    return c


def main():
    if not O.ref_available():
        O.build_oracle()
    if not O.ref_available():
        raise SystemExit("oracle/_ref/libsdref.so missing and /root/reference not present")
    os.makedirs(OUT, exist_ok=True)
    n = 0
    for name, n_in, n_out, T, mutate, kw in cases():
        seed = 1000 + n
        p = make_payload(n_in, n_out, seed=seed, **kw)
        if mutate:
            mutate(p, np.random.default_rng(seed + 7))
        x = make_x(n_in, T, seed=seed + 1)
        y = O.forward(p, x, nth=1, use_ref=True)
        y2 = O.forward(p, x, nth=3, use_ref=True)
        assert np.array_equal(y.view(np.uint32), y2.view(np.uint32)), name
        d = {"n_in": np.int64(n_in), "n_out": np.int64(n_out), "x": x, "y": y}
        for f in FIELDS:
            a = getattr(p, f)
            if a is not None:
                d[f] = a
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        n += 1
        print(f"{name:32s} kind={p.kind} T={T} bytes={os.path.getsize(path)}")
    print(f"wrote {n} fixtures to {OUT}")


if __name__ == "__main__":
    main()
