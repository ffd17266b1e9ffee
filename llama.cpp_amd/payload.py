"""SeedDelta payloads on the host: the tensors a seeddelta-build GGUF carries for one FFN weight.

Mirrors ``struct llama_seeddelta_weight`` (reference ``src/llama-seeddelta.h:12-25``) with numpy
arrays. ggml's ``ne[0]`` is the fastest axis, so a ggml tensor ``[ne0, ne1, ne2]`` is a C-contiguous
numpy array of shape ``(ne2, ne1, ne0)``:

    d_idx  [K, n_out]          -> (n_out, K)            int16 / int32
    d_val  [K, n_out]          -> (n_out, K)            float16 / float32
    b_idx  [nb, n_out]         -> (n_out, nb)           int16 / int32
    b_val  [block, nb, n_out]  -> (n_out, nb, block)    float16 / float32
    row_scale [n_out]          -> (n_out,)              float16 / float32
    base_d1/d2/d3 [L, B]       -> (B, L)                float16 / float32
    base_perm1/2  [L, B]       -> (B, L)                int16 / int32
    x [n_in, T] -> (T, n_in) float32 ; y [n_out, T] -> (T, n_out) float32

``make_payload`` draws the synthetic payloads of SURVEY.md §8(d) (no model weights exist in this
environment): val ~ 0.02*N(0,1) as F16, indices uniform without replacement per row,
row_scale = 1 + 0.01*N(0,1) as F16, d1 = d3 = 1, d2 ~ N(0,1)/L as F16, perm1 a random permutation per
base block (what ``tools/seeddelta-build/sd_build.cpp:1196-1234`` emits).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

OP_COO, OP_BLOCK, OP_BASE_COO, OP_BASE_BLOCK = 0, 1, 2, 3
OP_NAMES = {OP_COO: "coo", OP_BLOCK: "block", OP_BASE_COO: "base+coo", OP_BASE_BLOCK: "base+block"}


@dataclass
class Payload:
    n_in: int
    n_out: int
    d_idx: Optional[np.ndarray] = None
    d_val: Optional[np.ndarray] = None
    b_idx: Optional[np.ndarray] = None
    b_val: Optional[np.ndarray] = None
    row_scale: Optional[np.ndarray] = None
    base_d1: Optional[np.ndarray] = None
    base_d2: Optional[np.ndarray] = None
    base_d3: Optional[np.ndarray] = None
    base_perm1: Optional[np.ndarray] = None
    base_perm2: Optional[np.ndarray] = None
    w_ref: Optional[np.ndarray] = None      # dense weight (n_out, n_in) float32/float16, for NaN repair
    meta: dict = field(default_factory=dict)

    @property
    def kind(self) -> int:
        """Dispatch priority of ``llm_graph_context::build_lora_mm`` (src/llama-graph.cpp:681-703):
        block payload wins over COO; base variant iff base_d1."""
        if self.b_idx is not None and self.b_val is not None:
            return OP_BASE_BLOCK if self.base_d1 is not None else OP_BLOCK
        if self.d_idx is not None and self.d_val is not None:
            return OP_BASE_COO if self.base_d1 is not None else OP_COO
        raise ValueError("payload has neither COO nor block residual")

    @property
    def is_tall(self) -> bool:
        return self.n_out >= self.n_in          # src/llama-seeddelta.cpp:1073

    def payload_bytes(self) -> int:
        """Bytes of the tensors the op selected by ``kind`` reads, as stored."""
        k = self.kind
        arrs = [self.row_scale]
        arrs += [self.b_idx, self.b_val] if k in (OP_BLOCK, OP_BASE_BLOCK) else [self.d_idx, self.d_val]
        if k in (OP_BASE_COO, OP_BASE_BLOCK):
            arrs += [self.base_d1, self.base_d2, self.base_d3, self.base_perm1, self.base_perm2]
        return int(sum(a.nbytes for a in arrs if a is not None))

    def algorithmic_bytes(self, n_tokens: int) -> int:
        """payload + T*4*(n_in+n_out): the per-launch byte count the roofline figure uses."""
        return self.payload_bytes() + n_tokens * 4 * (self.n_in + self.n_out)


def base_dims(n_in: int, n_out: int) -> tuple[int, int]:
    """L, B as the builder chooses them (tools/seeddelta-build/sd_eval.cpp:443-445):
    L = next pow2 >= min(n_in, n_out), B = ceil(max(n_in, n_out) / L)."""
    m = min(n_in, n_out)
    L = 1
    while L < m:
        L <<= 1
    B = (max(n_in, n_out) + L - 1) // L
    return L, B


def _choose_rows(rng: np.random.Generator, n_rows: int, n_choices: int, k: int) -> np.ndarray:
    """k distinct integers in [0, n_choices) per row, in random order."""
    k = min(k, n_choices)
    if n_choices <= 4 * k or n_choices <= 4096:
        keys = rng.random((n_rows, n_choices), dtype=np.float32)
        part = np.argpartition(keys, k - 1, axis=1)[:, :k] if k < n_choices else np.argsort(keys, axis=1)
        return part.astype(np.int64)
    # sparse draw: sample with replacement, fix up the (rare) duplicates row by row
    out = rng.integers(0, n_choices, size=(n_rows, k), dtype=np.int64)
    srt = np.sort(out, axis=1)
    bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
    for r in bad:
        out[r] = rng.choice(n_choices, size=k, replace=False)
    return out


def make_payload(n_in: int, n_out: int, *, scheme: str = "coo", K: int = 32, block: int = 16,
                 base: bool = False, row_scale: bool = True,
                 idx_dtype=np.int16, val_dtype=np.float16, rs_dtype=np.float16,
                 base_dtype=np.float16, perm_dtype=np.int16, perm: bool = True,
                 d13_random: bool = False, seed: int = 1234, with_w_ref: bool = False) -> Payload:
    rng = np.random.default_rng(seed)
    p = Payload(n_in=n_in, n_out=n_out)
    p.meta = dict(scheme=scheme, K=K, block=block, base=base, seed=seed)
    if scheme == "coo":
        K = max(1, min(K, n_in))
        p.d_idx = _choose_rows(rng, n_out, n_in, K).astype(idx_dtype)
        p.d_val = (0.02 * rng.standard_normal((n_out, K), dtype=np.float32)).astype(val_dtype)
    elif scheme == "block":
        n_blocks = (n_in + block - 1) // block
        nb = max(1, min((K + block - 1) // block, n_blocks))
        p.b_idx = _choose_rows(rng, n_out, n_blocks, nb).astype(idx_dtype)
        p.b_val = (0.02 * rng.standard_normal((n_out, nb, block), dtype=np.float32)).astype(val_dtype)
    else:
        raise ValueError(scheme)
    if row_scale:
        p.row_scale = (1.0 + 0.01 * rng.standard_normal(n_out, dtype=np.float32)).astype(rs_dtype)
    if base:
        L, B = base_dims(n_in, n_out)
        if d13_random:
            p.base_d1 = (1.0 + 0.1 * rng.standard_normal((B, L), dtype=np.float32)).astype(base_dtype)
            p.base_d3 = (1.0 + 0.1 * rng.standard_normal((B, L), dtype=np.float32)).astype(base_dtype)
        else:
            p.base_d1 = np.ones((B, L), dtype=base_dtype)
            p.base_d3 = np.ones((B, L), dtype=base_dtype)
        p.base_d2 = (rng.standard_normal((B, L), dtype=np.float32) / L).astype(base_dtype)
        if perm:
            pd = np.int32 if L > 32768 else perm_dtype
            p.base_perm1 = np.stack([rng.permutation(L) for _ in range(B)]).astype(pd)
    if with_w_ref:
        p.w_ref = (0.02 * rng.standard_normal((n_out, n_in), dtype=np.float32)).astype(np.float32)
    return p


def make_x(n_in: int, n_tokens: int, seed: int = 1) -> np.ndarray:
    return np.random.default_rng(seed).standard_normal((n_tokens, n_in), dtype=np.float32)


# Named shapes of BASELINE.json's configs (shapes paragraph + §8d)
MODEL_DIMS = {
    "mistral-7b": dict(n_embd=4096, n_ff=14336, n_layer=32),
    "gemma3-4b": dict(n_embd=2560, n_ff=10240, n_layer=34),
    "gemma3-1b": dict(n_embd=1152, n_ff=6912, n_layer=26),
}


def config_payload(name: str, seed: int = 1234) -> Payload:
    """Payloads of the named benchmark configs (SURVEY.md §8d 'Configs restated')."""
    m7, g4, g1 = MODEL_DIMS["mistral-7b"], MODEL_DIMS["gemma3-4b"], MODEL_DIMS["gemma3-1b"]
    table = {
        # cfg3: Mistral-7B blk.23.ffn_down, block32 K=1536 (nb=48), base L=4096 B=4, row_scale
        "cfg3_mistral_down_block32_k1536": dict(n_in=m7["n_ff"], n_out=m7["n_embd"], scheme="block", K=1536, block=32, base=True),
        # cfg2: Gemma-3 4B gate, COO K=128 + base L=4096 B=3
        "cfg2_gemma4b_gate_coo_k128": dict(n_in=g4["n_embd"], n_out=g4["n_ff"], scheme="coo", K=128, base=True),
        # cfg4: Mistral-7B prefill shapes, K=128, COO vs block16, gate + down, base on
        "cfg4_mistral_gate_coo_k128": dict(n_in=m7["n_embd"], n_out=m7["n_ff"], scheme="coo", K=128, base=True),
        "cfg4_mistral_gate_block16_k128": dict(n_in=m7["n_embd"], n_out=m7["n_ff"], scheme="block", K=128, block=16, base=True),
        "cfg4_mistral_down_coo_k128": dict(n_in=m7["n_ff"], n_out=m7["n_embd"], scheme="coo", K=128, base=True),
        "cfg4_mistral_down_block16_k128": dict(n_in=m7["n_ff"], n_out=m7["n_embd"], scheme="block", K=128, block=16, base=True),
        # cfg1: Gemma-3 1B block16, K_gate=K_up=64, K_down=128, base on
        "cfg1_gemma1b_gate_block16_k64": dict(n_in=g1["n_embd"], n_out=g1["n_ff"], scheme="block", K=64, block=16, base=True),
        "cfg1_gemma1b_down_block16_k128": dict(n_in=g1["n_ff"], n_out=g1["n_embd"], scheme="block", K=128, block=16, base=True),
    }
    return make_payload(seed=seed, **table[name])
