import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import llama_cpp_amd  # noqa: E402,F401  (loads llama.cpp_amd/ under an importable name)
from llama_cpp_amd.payload import Payload  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
PAYLOAD_FIELDS = ("d_idx", "d_val", "b_idx", "b_val", "row_scale", "base_d1", "base_d2", "base_d3",
                  "base_perm1", "base_perm2", "w_ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """forward-op fixtures (reference TU outputs)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith(("scan_", "gguf_", "ffn_")))


def scan_golden_names():
    """builder-evaluation fixtures (reference sd_eval / sd_ffn_proxy / sd_cost outputs)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("scan_"))


def ffn_golden_names():
    """gated FFN block fixtures (the reference's ggml graph: three SeedDelta nodes + swiglu/geglu)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("ffn_"))


def load_ffn_golden(name):
    """-> (gate, up, down payloads, x, act, h_ref, y_ref)"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n_embd, n_ff = int(z["n_embd"]), int(z["n_ff"])
    ps = []
    for pre, (ni, no) in (("gate_", (n_embd, n_ff)), ("up_", (n_embd, n_ff)), ("down_", (n_ff, n_embd))):
        p = Payload(n_in=ni, n_out=no)
        for f in PAYLOAD_FIELDS:
            if pre + f in z.files:
                setattr(p, f, np.ascontiguousarray(z[pre + f]))
        ps.append(p)
    return ps[0], ps[1], ps[2], np.ascontiguousarray(z["x"]), int(z["act"]), z["h"], z["y"]


def load_scan_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}


def load_golden(name):
    """-> (Payload, x, y_ref). Fixtures are plain arrays (allow_pickle stays False)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    p = Payload(n_in=int(z["n_in"]), n_out=int(z["n_out"]))
    for f in PAYLOAD_FIELDS:
        if f in z.files:
            setattr(p, f, np.ascontiguousarray(z[f]))
    return p, np.ascontiguousarray(z["x"]), z["y"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def abs_term_sum(p, x, y_base=None):
    """sum_r |val*x| per output (T, n_out): the scale of the fp32 rounding-order tolerance."""
    xf = np.abs(x.astype(np.float64))
    T = x.shape[0]
    out = np.zeros((T, p.n_out))
    if p.kind in (1, 3):
        nb, block = p.b_idx.shape[1], p.b_val.shape[2]
        for o in range(p.n_out):
            for bi in range(nb):
                blk = int(p.b_idx[o, bi])
                in0 = blk * block
                if blk < 0 or in0 >= p.n_in:
                    continue
                n = min(block, p.n_in - in0)
                out[:, o] += xf[:, in0:in0 + n] @ np.abs(p.b_val[o, bi, :n].astype(np.float64))
    else:
        idx = p.d_idx.astype(np.int64)
        val = np.abs(p.d_val.astype(np.float64))
        ok = (idx >= 0) & (idx < p.n_in)
        idc = np.where(ok, idx, 0)
        for t in range(T):
            out[t] = np.where(ok, val * xf[t][idc], 0.0).sum(axis=1)
    return out


@pytest.fixture(scope="session")
def hip():
    """The HIP library bound to cuda:0. No fallback: missing library or GPU is an error."""
    from llama_cpp_amd import seeddelta
    seeddelta.init(0)
    return seeddelta
