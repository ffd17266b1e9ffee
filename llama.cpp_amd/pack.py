"""Turning a scan candidate's staging arrays into the tensors a SeedDelta GGUF stores — the builder's
packing step (tools/seeddelta-build/sd_build.cpp:1104-1236) — and the names / metadata they are
stored under (src/llama-model.cpp:2775-2848, tools/seeddelta-build/sd_write.cpp:44-74).

Rules restated from the builder:
  * indices: int32 staging -> I16, anything outside [-32768, 32767] becomes -1 (:1132, :1162); I32 kept as is;
  * values, row_scale, base diagonals: fp32 -> F16 with round-to-nearest-even (ggml_fp32_to_fp16_row,
    ggml/src/ggml.c:457-462); row_scale is always F16 (:1188-1191);
  * base_perm1: I16 when L <= 32768 with out-of-range entries -> 0 (:1111, :1228), else I32;
  * base_d1 = base_d3 = 1 (the fit leaves them at identity, sd_eval.cpp:450-452); no base_perm2.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .payload import Payload

KV_PREFIX = "seeddelta."


def pack_indices(idx: np.ndarray, dtype=np.int16) -> np.ndarray:
    idx = np.asarray(idx, dtype=np.int32)
    if np.dtype(dtype) == np.int32:
        return idx.copy()
    out = idx.astype(np.int16)
    out[(idx < -32768) | (idx > 32767)] = -1
    return out


def pack_perm(perm: np.ndarray, L: int) -> np.ndarray:
    perm = np.asarray(perm, dtype=np.int32)
    if L > 32768:
        return perm.copy()
    out = perm.astype(np.int16)
    out[(perm < 0) | (perm > 32767)] = 0
    return out


def pack_f16(a: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return np.asarray(a, dtype=np.float32).astype(np.float16)


def pack_candidate(staging: dict, n_in: int, n_out: int, scheme: str, idx_dtype=np.int16, val_dtype=np.float16) -> Payload:
    """staging = ScanSession.fetch(): idx/val[/row_scale][/d2, perm1] -> the payload tensors."""
    p = Payload(n_in=n_in, n_out=n_out)
    idx = pack_indices(staging["idx"], idx_dtype)
    val = pack_f16(staging["val"]) if np.dtype(val_dtype) == np.float16 else np.asarray(staging["val"], np.float32).copy()
    if scheme == "block":
        p.b_idx, p.b_val = idx, val
    else:
        p.d_idx, p.d_val = idx, val
    if staging.get("row_scale") is not None:
        p.row_scale = pack_f16(staging["row_scale"])
    if staging.get("d2") is not None:
        B, L = staging["d2"].shape
        p.base_d1 = np.ones((B, L), dtype=np.float16)
        p.base_d2 = pack_f16(staging["d2"])
        p.base_d3 = np.ones((B, L), dtype=np.float16)
        p.base_perm1 = pack_perm(staging["perm1"], L)
    return p


# ---- names and metadata ---------------------------------------------------------------------------------
TENSOR_SUFFIX = {"d_idx": "d_idx", "d_val": "d_val", "b_idx": "b_idx", "b_val": "b_val", "row_scale": "d_row_scale",
                 "base_d1": "base_d1", "base_d2": "base_d2", "base_d3": "base_d3", "base_perm1": "base_perm1",
                 "base_perm2": "base_perm2"}


def tensor_names(layer: int, kind: str) -> Dict[str, str]:
    """payload field -> GGUF tensor name, e.g. blk.23.ffn_down.b_val (sd_build.cpp:367-371, :1197-1200)."""
    return {f: f"blk.{layer}.{kind}.{s}" for f, s in TENSOR_SUFFIX.items()}


def payload_tensors(layer: int, kind: str, p: Payload) -> Dict[str, np.ndarray]:
    names = tensor_names(layer, kind)
    return {names[f]: getattr(p, f) for f in TENSOR_SUFFIX if getattr(p, f) is not None}


def payload_from_tensors(layer: int, kind: str, tensors: Dict[str, np.ndarray], n_in: int, n_out: int) -> Optional[Payload]:
    """Collect one tensor's SeedDelta payload from a name -> array map (what llama-model.cpp does when it
    finds the optional tensors); None if neither residual pair is there."""
    names = tensor_names(layer, kind)
    p = Payload(n_in=n_in, n_out=n_out)
    for f in TENSOR_SUFFIX:
        if names[f] in tensors:
            setattr(p, f, tensors[names[f]])
    have_coo = p.d_idx is not None and p.d_val is not None
    have_blk = p.b_idx is not None and p.b_val is not None
    return p if (have_coo or have_blk) else None


def metadata(scheme: str, K: int, block: int, row_scale: bool, base: bool, strip_dense: bool = False,
             K_gate: Optional[int] = None, K_up: Optional[int] = None, K_down: Optional[int] = None,
             base_max_samples: int = 2048, base_perm_trials: int = 1, n_added: int = 1,
             entries: Optional[list] = None) -> Dict[str, object]:
    """The seeddelta.* key/values sd_write.cpp:44-74 puts into the GGUF header (all u32 / bool), in
    the order it writes them; `entries` are report entries for the per-tensor keys (:67-73)."""
    kv: Dict[str, object] = {
        KV_PREFIX + "enabled": n_added > 0, KV_PREFIX + "version": 1,
        KV_PREFIX + "scheme": 1 if scheme == "block" else 0,
        KV_PREFIX + "row_scale": bool(row_scale), KV_PREFIX + "resid.K": int(K),
        KV_PREFIX + "strip_dense": bool(strip_dense),
    }
    if scheme == "block":
        kv[KV_PREFIX + "resid.block"] = int(block)
    variable = any(v is not None and v != K for v in (K_gate, K_up, K_down))
    kv[KV_PREFIX + "resid.K_variable"] = variable
    if variable:
        kv.update({KV_PREFIX + "resid.K_gate": int(K_gate or K), KV_PREFIX + "resid.K_up": int(K_up or K),
                   KV_PREFIX + "resid.K_down": int(K_down or K)})
    kv[KV_PREFIX + "base.enabled"] = bool(base)
    if base:
        kv.update({KV_PREFIX + "base.kind": 1, KV_PREFIX + "base.depth": 2, KV_PREFIX + "base.R": 1,
                   KV_PREFIX + "base.max_samples": max(0, int(base_max_samples)),
                   KV_PREFIX + "base.perm_trials": max(1, int(base_perm_trials))})
    for e in entries or []:
        pre = f"{KV_PREFIX}blk.{e['layer']}.{e['kind']}"
        kv.update({pre + ".enabled": bool(e["emit"]), pre + ".gating_pass": bool(e["gating_pass"]),
                   pre + ".strip_dense": bool(e.get("strip_dense", False)), pre + ".K": max(0, int(e.get("K", 0)))})
    return kv
