#!/usr/bin/env python3
"""Write tests/golden/gguf_toy.gguf with the REFERENCE's own gguf-py writer (imported from
/root/reference/gguf-py in this container) plus tests/golden/gguf_toy_expect.npz holding the arrays
that went in. The GGUF mimics what llama-seeddelta-build emits for one layer: Q8_0 dense FFN weights,
SeedDelta tensors under the builder's names, the seeddelta.* metadata of sd_write.cpp:44-74.

Run in the build container:  python oracle/make_gguf_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/gguf-py")

import gguf  # noqa: E402  (reference's gguf-py)
import llama_cpp_amd  # noqa: E402,F401
from llama_cpp_amd.dense import make_dense  # noqa: E402
from llama_cpp_amd.payload import make_payload  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N_EMBD, N_FF, LAYER = 64, 160, 2


def main():
    w = gguf.GGUFWriter(os.path.join(OUT, "gguf_toy.gguf"), "llama")
    expect = {}
    dense = {"ffn_gate": make_dense(N_EMBD, N_FF, "q8_0", seed=1), "ffn_up": make_dense(N_EMBD, N_FF, "q8_0", seed=2),
             "ffn_down": make_dense(N_FF, N_EMBD, "q8_0", seed=3)}
    for kind, raw in dense.items():
        name = f"blk.{LAYER}.{kind}.weight"
        n_in = raw.shape[1] // 34 * 32
        w.add_tensor(name, raw, raw_dtype=gguf.GGMLQuantizationType.Q8_0)
        expect[name] = raw
    gate = make_payload(N_EMBD, N_FF, scheme="coo", K=8, base=True, seed=5)
    down = make_payload(N_FF, N_EMBD, scheme="block", K=32, block=16, base=True, seed=6)
    for kind, p in (("ffn_gate", gate), ("ffn_down", down)):
        fields = {"d_idx": p.d_idx, "d_val": p.d_val, "b_idx": p.b_idx, "b_val": p.b_val, "d_row_scale": p.row_scale,
                  "base_d1": p.base_d1, "base_d2": p.base_d2, "base_d3": p.base_d3, "base_perm1": p.base_perm1}
        for suffix, arr in fields.items():
            if arr is None:
                continue
            name = f"blk.{LAYER}.{kind}.{suffix}"
            w.add_tensor(name, np.ascontiguousarray(arr))
            expect[name] = arr
    w.add_bool("seeddelta.enabled", True)
    w.add_uint32("seeddelta.version", 1)
    w.add_uint32("seeddelta.scheme", 0)
    w.add_bool("seeddelta.row_scale", True)
    w.add_uint32("seeddelta.resid.K", 8)
    w.add_bool("seeddelta.strip_dense", False)
    w.add_bool("seeddelta.resid.K_variable", False)
    w.add_bool("seeddelta.base.enabled", True)
    w.add_uint32("seeddelta.base.kind", 1)
    w.add_uint32("seeddelta.base.depth", 2)
    w.add_uint32(f"seeddelta.blk.{LAYER}.ffn_gate.K", 8)
    w.add_bool(f"seeddelta.blk.{LAYER}.ffn_gate.enabled", True)
    w.write_header_to_file()
    w.write_kv_data_to_file()
    w.write_tensors_to_file()
    w.close()
    np.savez_compressed(os.path.join(OUT, "gguf_toy_expect.npz"), **{k.replace(".", "__"): v for k, v in expect.items()})
    print("wrote gguf_toy.gguf", os.path.getsize(os.path.join(OUT, "gguf_toy.gguf")), "bytes,", len(expect), "tensors")


if __name__ == "__main__":
    main()
