#!/usr/bin/env python3
"""Generate tests/golden/ffn_*.npz: the gated FFN block as the REFERENCE computes it — one ggml graph
of the node sequence build_ffn emits for LLM_FFN_PAR (src/llama-graph.cpp:811-953): the reference's
SeedDelta constructors for ffn_up / ffn_gate / ffn_down (src/llama-seeddelta.cpp), its
ggml_swiglu_split / ggml_geglu_split between them, computed by its own CPU backend
(ggml/src/ggml-cpu/*), all compiled unmodified into oracle/_ref/libsdref.so.

Run in the build container only (needs /root/reference):   python oracle/make_ffn_golden.py

Each fixture stores inputs (three payloads, x, act) and the reference's h = act(gate)*up and y.
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
          "base_perm1", "base_perm2")

# name, n_embd, n_ff, T, act, x scale, payload kwargs
CASES = [
    # T = 1 with tall base gate/up of one shape and L = 1024: the gate|up single-launch path
    ("ffn_silu_base_coo_t1", 1024, 2560, 1, O.ACT_SILU, 1.0, dict(scheme="coo", K=16, base=True)),
    ("ffn_gelu_base_block_t1", 1024, 2304, 1, O.ACT_GELU, 4.0, dict(scheme="block", K=32, block=16, base=True, d13_random=True)),
    # several tokens / no base: node by node on the device
    ("ffn_silu_base_coo_t5", 256, 640, 5, O.ACT_SILU, 2.0, dict(scheme="coo", K=16, base=True, d13_random=True)),
    ("ffn_gelu_block_nobase_t3", 192, 512, 3, O.ACT_GELU, 60.0, dict(scheme="block", K=32, block=16)),
]


def main():
    assert O.ref_available(), "oracle/_ref/libsdref.so missing: run make -C oracle"
    for name, n_embd, n_ff, T, act, xs, kw in CASES:
        gate = make_payload(n_embd, n_ff, seed=11, **kw)
        up = make_payload(n_embd, n_ff, seed=12, **kw)
        down = make_payload(n_ff, n_embd, seed=13, **kw)
        x = (xs * make_x(n_embd, T, seed=5)).astype(np.float32)
        g = O.forward(gate, x, use_ref=True)
        h, y = O.ffn_forward_graph(gate, up, down, x, act)
        out = dict(n_embd=n_embd, n_ff=n_ff, act=act, x=x, h=h, y=y)
        for pre, p in (("gate_", gate), ("up_", up), ("down_", down)):
            for f in FIELDS:
                v = getattr(p, f)
                if v is not None:
                    out[pre + f] = v
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        frac = float(np.mean(np.abs(g) >= 10.0))
        print(f"{name}: |g|>=10 on {frac:.3f} of elements, max|y| {np.abs(y).max():.3g}, "
              f"{os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
