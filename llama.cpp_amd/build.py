"""`llama-seeddelta-build` as one call: input GGUF -> SeedDelta GGUF (+ report.json).

Mirrors the builder's runner and writer (tools/seeddelta-build/sd_runner.cpp:38-224,
sd_write.cpp:12-135) around the GPU pipeline of scan.build_run:

  * layers discovered from the tensor names, `--layers A-B` clamped to them (sd_utils.cpp:18-39);
  * per-kind budgets `--K-gate/--K-up/--K-down` falling back to `--K` (sd_runner.cpp:151-155);
  * tensors that already carry delta tensors are left alone unless `overwrite_existing`
    (sd_build.cpp:373-377);
  * output = every source key/value and tensor copied through unchanged (whatever their types),
    minus the dense weights of stripped tensors, plus the new tensors in the builder's order
    (idx, val, [row_scale], [base_d1, base_d2, base_d3, base_perm1]) and the seeddelta.* keys of
    sd_write.cpp:44-74;
  * report.json through scan.report_document; the policy hash is the builder's `fnv1a64:<16 hex>` of
    the policy file's bytes (sd_policy_resolve.cpp:9-21, :49-52).

Dense FFN weights must be F32, F16 or Q8_0 (the types the scan kernels read); anything else is an
error here, where the reference would dequantise through ggml's type traits.
"""
from __future__ import annotations

import json
import math
import re
from typing import Dict, List, Optional

import numpy as np

from . import pack as PK
from . import policy as PL
from . import scan as SC
from .gguf_io import GGUFReader, write_gguf

DELTA_SUFFIXES = ("d_idx", "d_val", "b_idx", "b_val", "d_row_scale", "base_d1", "base_d2", "base_d3", "base_perm1", "base_perm2")


def policy_hash(policy_bytes: bytes) -> str:
    """sd_policy_resolve.cpp:9-21: the FNV-1a 64 recurrence over the file's bytes, 16 lower-case hex digits.
    The builder's offset basis is 1469598103934665603 (the published one minus its last digit): kept, so
    that hashes match the builder's reports rather than the published FNV vectors."""
    h = 1469598103934665603
    for c in policy_bytes:
        h ^= c
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"fnv1a64:{h:016x}" if policy_bytes else ""


def layer_range(spec: str, n_layer: int) -> List[int]:
    """sd_utils.cpp:18-39: empty = all layers, else exactly "A-B" (swapped if reversed, clamped)."""
    if not spec:
        return list(range(n_layer))
    m = re.fullmatch(r"\s*(\d+)\s*-\s*(\d+)\s*", spec)
    if not m:
        raise ValueError("invalid --layers, expected A-B")
    a, b = sorted((int(m.group(1)), int(m.group(2))))
    return list(range(max(0, a), min(n_layer - 1, b) + 1))


def build_gguf(in_path: str, out_path: str, params: SC.ScanParams, layers: str = "", K_gate: int = 0, K_up: int = 0,
               K_down: int = 0, idx_type: str = "i16", val_type: str = "f16", strip_dense: bool = False,
               policy_file: str = "", policy_strict: bool = False, imatrix_file: str = "", imatrix_eps: float = 1e-8,
               imatrix_power: float = 1.0, report_json: str = "", overwrite_existing: bool = False,
               stack_cost_cap: float = math.inf) -> dict:
    """Returns {"entries", "n_added", "stripped", "report"}; writes out_path (and report_json)."""
    if params.scheme not in ("coo", "block"):
        raise ValueError("invalid --scheme (expected: coo|block)")
    if params.scheme == "block" and not 0 < params.block <= 4096:
        raise ValueError("invalid --block (expected: 1..4096)")
    idx_dtype = {"i16": np.int16, "i32": np.int32}.get(idx_type)
    val_dtype = {"f16": np.float16, "f32": np.float32}.get(val_type)
    if idx_dtype is None:
        raise ValueError("invalid --idx-type")
    if val_dtype is None:
        raise ValueError("invalid --val-type")

    policy, phash = None, ""
    if policy_file:
        with open(policy_file, "rb") as fh:
            raw = fh.read()
        policy = PL.load_policy(raw.decode("utf-8"), strict=policy_strict)
        phash = policy_hash(raw)
    imatrix = SC.load_imatrix(imatrix_file) if imatrix_file else None

    with GGUFReader(in_path) as src:
        ids = [int(m.group(1)) for m in (re.search(r"blk\.(\d+)\.", n) for n in src.names()) if m]
        n_layer = max(ids) + 1 if ids else 0
        layer_list = layer_range(layers, n_layer)
        K_default = max(1, params.K)
        K_eff = {"ffn_gate": K_gate if K_gate > 0 else K_default, "ffn_up": K_up if K_up > 0 else K_default,
                 "ffn_down": K_down if K_down > 0 else K_default}

        def wname(layer, kind):
            return f"blk.{layer}.{kind}.weight"

        def has_delta(layer, kind):
            return any(f"blk.{layer}.{kind}.{s}" in src.infos for s in ("d_idx", "d_val", "b_idx", "b_val"))

        def have_tensor(layer, kind):
            n = wname(layer, kind)
            return n in src.infos and len(src.infos[n][0]) == 2

        def skip(layer, kind):
            return (not overwrite_existing) and has_delta(layer, kind)

        def model(layer, kind):
            n = wname(layer, kind)
            gtype = src.infos[n][1]
            if gtype not in (0, 1, 8):
                raise ValueError(f"{n}: ggml type {gtype} is not supported by the scan kernels (F32, F16, Q8_0)")
            return src.tensor(n)

        p = SC.ScanParams(**{**params.__dict__, "K": K_default})
        run = SC.build_run(model, layer_list, p, policy=policy, imatrix=imatrix, imatrix_eps=imatrix_eps,
                           imatrix_power=imatrix_power, stack_cost_cap=stack_cost_cap, strip_dense=strip_dense,
                           have_tensor=have_tensor, keep_staging=True, K_kind=K_eff, skip=skip,
                           fetch_dtypes=(idx_dtype, val_dtype))
        entries = run["entries"]

        stripped = {wname(e["layer"], e["kind"]) for e in entries if e.get("emit") and e.get("strip_dense")}
        rebuilt = {(l, k) for (l, k) in run["emitted"]} if overwrite_existing else set()
        tensors: Dict[str, object] = {}
        for name in src.names():
            if name in stripped:
                continue                                            # sd_write.cpp:27-29
            m = re.fullmatch(r"blk\.(\d+)\.(ffn_gate|ffn_up|ffn_down)\.(\w+)", name)
            if m and m.group(3) in DELTA_SUFFIXES and (int(m.group(1)), m.group(2)) in rebuilt:
                continue                                            # replaced below
            tensors[name] = src.raw(name)
        n_added = 0
        for (layer, kind), (pp, K, staging) in run["emitted"].items():
            n_in, n_out = src.infos[wname(layer, kind)][0][:2]
            payload = PK.pack_candidate(staging, n_in, n_out, pp.scheme, idx_dtype, val_dtype)
            new = PK.payload_tensors(layer, kind, payload)
            tensors.update(new)
            n_added += len(new)

        kv = PK.metadata(p.scheme, K_default, p.block, p.row_scale, p.base, strip_dense=bool(stripped),
                         K_gate=K_eff["ffn_gate"], K_up=K_eff["ffn_up"], K_down=K_eff["ffn_down"],
                         base_max_samples=p.base_max_samples, base_perm_trials=p.base_perm_trials, n_added=n_added,
                         entries=entries)
        kv_raw = tuple(src.kv_raw(k) for k in src.kv if k not in kv and k != "general.alignment")
        write_gguf(out_path, kv, tensors, alignment=src.alignment, kv_raw=kv_raw)

    report = SC.report_document(entries, p, input_name=in_path, output_name=out_path, policy_file=policy_file, policy_hash=phash)
    report["resid"].update(K_gate=K_eff["ffn_gate"], K_up=K_eff["ffn_up"], K_down=K_eff["ffn_down"], idx_type=idx_type,
                           val_type=val_type)
    report.update(stack_pass_gate_up=run["stack_pass_gate_up"], stack_pass_down=run["stack_pass_down"])
    if report_json:
        with open(report_json, "w") as fh:
            json.dump(report, fh, indent=2)
    return {"entries": entries, "n_added": n_added, "stripped": sorted(stripped), "report": report}
