"""Synthetic dense FFN weights in the storage formats the builder reads (F32, F16, Q8_0).

A dense weight is a numpy array shaped (n_out, n_in) — one output row contiguous, the ggml layout
``ne[0] = n_in`` — as float32 / float16, or uint8 rows of raw ``block_q8_0`` (fp16 scale + 32 int8,
34 bytes; ggml/src/ggml-common.h:219-224) shaped (n_out, n_in // 32 * 34).
"""
from __future__ import annotations

import numpy as np

Q8_BLOCK = 32
Q8_BYTES = 34


def quantize_q8_0(w: np.ndarray) -> np.ndarray:
    """Reference rule of quantize_row_q8_0_ref (ggml/src/ggml-quants.c:199-): per block of 32,
    d = max|x| / 127 stored as fp16, q = round-half-away(x / d) computed with the fp32 d."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    n_out, n_in = w.shape
    assert n_in % Q8_BLOCK == 0
    blk = w.reshape(n_out, n_in // Q8_BLOCK, Q8_BLOCK)
    amax = np.abs(blk).max(axis=2)
    d = (amax / np.float32(127.0)).astype(np.float32)
    inv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, np.float32(1.0)), np.float32(0.0)).astype(np.float32)
    x0 = (blk * inv[:, :, None]).astype(np.float32)
    q = (np.sign(x0) * np.floor(np.abs(x0) + np.float32(0.5))).astype(np.int8)
    out = np.empty((n_out, n_in // Q8_BLOCK, Q8_BYTES), dtype=np.uint8)
    out[:, :, :2] = d.astype(np.float16).view(np.uint8).reshape(n_out, -1, 2)
    out[:, :, 2:] = q.view(np.uint8)
    return out.reshape(n_out, -1)


def dequantize_q8_0(raw: np.ndarray) -> np.ndarray:
    """y = q * fp32(d) (dequantize_row_q8_0, ggml/src/ggml-quants.c:401-414)."""
    n_out = raw.shape[0]
    blk = raw.reshape(n_out, -1, Q8_BYTES)
    d = np.ascontiguousarray(blk[:, :, :2]).view(np.float16).astype(np.float32)
    q = np.ascontiguousarray(blk[:, :, 2:]).view(np.int8).astype(np.float32)
    return (q * d).reshape(n_out, -1).astype(np.float32)


def dense_dims(W: np.ndarray):
    """-> (n_in, n_out)"""
    if W.dtype == np.uint8:
        return W.shape[1] // Q8_BYTES * Q8_BLOCK, W.shape[0]
    return W.shape[1], W.shape[0]


def dense_type(W: np.ndarray) -> int:
    """ggml type code of a dense weight array."""
    return {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.uint8): 8}[W.dtype]


def make_dense(n_in: int, n_out: int, kind: str = "q8_0", seed: int = 1234, structure: float = 0.5) -> np.ndarray:
    """0.02*N(0,1) weights plus, optionally, an XOR-circulant component so the base fit
    has something to find (structure = its share of the variance; 0 disables)."""
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((n_out, n_in), dtype=np.float32)
    if structure > 0:
        L = 1
        while L < min(n_in, n_out):
            L <<= 1
        h = rng.standard_normal(L, dtype=np.float32)
        o = np.arange(n_out)[:, None] % L
        i = np.arange(n_in)[None, :] % L
        w = np.sqrt(1 - structure).astype(np.float32) * w + np.sqrt(structure).astype(np.float32) * h[o ^ i]
    w = (np.float32(0.02) * w).astype(np.float32)
    if kind == "f32":
        return w
    if kind == "f16":
        return w.astype(np.float16)
    if kind == "q8_0":
        return quantize_q8_0(w)
    raise ValueError(kind)
