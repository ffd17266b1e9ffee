"""SeedDelta build policies: per-layer / per-tensor overrides of K, block, strip, gating and autotune
(reference: tools/seeddelta-build/seeddelta_policy.{h,cpp}). Same JSON documents, same merge order

    CLI baseline -> defaults -> ranges (in file order) -> layers[N] -> tensors[kind]

and the same derived flags (gating is on unless a threshold is negative; which metrics need eval-x
and an imatrix). A rule is kept as a sparse dict of the fields it sets; merging is "later wins".
Pinned by the expectations of the reference's own policy self-test
(seeddelta_policy_selftest.cpp:66-220), restated in tests/test_policy.py.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional

METRICS = {"cos_x_w": "cos_x_w", "cos-x-w": "cos_x_w", "cos_x": "cos_x", "cos-x": "cos_x", "cos_w": "cos_w", "cos-w": "cos_w",
           "ffn_score": "ffn_score", "ffn-score": "ffn_score", "ffn": "ffn_score"}
KINDS = ("gate", "up", "down")


class PolicyError(ValueError):
    pass


@dataclass
class Autotune:
    enabled: bool = False
    schedule_gate_up: List[int] = field(default_factory=list)
    schedule_down: List[int] = field(default_factory=list)
    max_iters: int = 0


@dataclass
class Resolved:
    """sd_resolved_tensor (seeddelta_policy.h:70-84): the effective settings of one (layer, tensor)."""
    enabled: bool = True
    strip_dense: bool = False
    block: int = 32
    K_gate: int = 32
    K_up: int = 32
    K_down: int = 32
    metric: str = "cos_x_w"
    min_mean: float = 0.0
    min_p05: float = 0.0
    gating_enabled: bool = False
    require_eval_x: bool = False
    require_imatrix: bool = False
    autotune: Autotune = field(default_factory=Autotune)

    def K_of(self, kind: str) -> int:
        return {"ffn_gate": self.K_gate, "ffn_up": self.K_up}.get(kind, self.K_down)


def parse_layer_selector(s: str) -> List[int]:
    """"3-1,2,2,5" -> [1, 2, 3, 5] (seeddelta_policy.cpp:48-75)."""
    out = set()
    for tok in s.split(","):
        tok = tok.strip()
        if not tok:
            continue
        dash = tok.find("-")
        if dash < 0:
            out.add(int(tok))
        else:
            a, b = int(tok[:dash]), int(tok[dash + 1:])
            out.update(range(min(a, b), max(a, b) + 1))
    return sorted(out)


def _is_int(v) -> bool:
    return isinstance(v, int) and not isinstance(v, bool)


def _is_num(v) -> bool:
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _parse_rule(obj: dict, strict: bool, warnings: List[str]) -> dict:
    rule: dict = {}
    for key, val in obj.items():
        if key in ("enabled", "enable"):
            rule["enabled"] = bool(val)
        elif key in ("strip_dense", "strip"):
            rule["strip_dense"] = bool(val)
        elif key == "block":
            rule["block"] = int(val)
        elif key == "K":
            if isinstance(val, dict):
                for k in KINDS:
                    if k in val:
                        rule["K_" + k] = int(val[k])
            elif _is_int(val):
                rule.update(K_gate=val, K_up=val, K_down=val)
        elif key in ("metric", "gating_metric"):
            if isinstance(val, str):
                rule["metric"] = METRICS.get(val, "cos")
        elif key == "gating":
            if isinstance(val.get("metric"), str):
                rule["metric"] = METRICS.get(val["metric"], "cos")
            for name in ("min_mean", "min_p05"):
                if name not in val:
                    continue
                v = val[name]
                if _is_num(v):
                    for k in KINDS:
                        rule[f"{name}_{k}"] = float(v)
                elif isinstance(v, dict):
                    for k in KINDS:
                        if k in v and _is_num(v[k]):
                            rule[f"{name}_{k}"] = float(v[k])
                else:
                    warnings.append(f"gating.{name} is not number/object")
        elif key == "autotune":
            if isinstance(val, dict):
                ints = lambda a: [x for x in a if _is_int(x)] if isinstance(a, list) else []     # noqa: E731
                rule["autotune"] = Autotune(bool(val.get("enabled", False)), ints(val.get("schedule_gate_up", [])),
                                            ints(val.get("schedule_down", [])), int(val.get("max_iters", 0)))
        elif key == "tensors":
            if isinstance(val, dict):
                rule["tensors"] = {t: _parse_rule(r, strict, warnings) for t, r in val.items() if isinstance(r, dict)}
        elif key in ("layers", "version"):
            pass
        else:
            if strict:
                raise PolicyError("unknown key in policy: " + key)
            warnings.append("unknown key in policy: " + key)
    return rule


def _merge(dst: dict, src: dict) -> None:
    for k, v in src.items():
        if k == "tensors":
            for t, r in v.items():
                _merge(dst.setdefault("tensors", {}).setdefault(t, {}), r)
        else:
            dst[k] = v


@dataclass
class Policy:
    version: int = 1
    defaults: dict = field(default_factory=dict)
    ranges: List[tuple] = field(default_factory=list)          # [(sorted layer list, rule)]
    layers: Dict[int, dict] = field(default_factory=dict)
    warnings: List[str] = field(default_factory=list)


def load_policy(text_or_doc, strict: bool = False) -> Policy:
    """Parse a policy document (JSON text or an already decoded dict). Unknown keys warn, or raise in
    strict mode (seeddelta_policy.cpp:182-248)."""
    try:
        doc = json.loads(text_or_doc) if isinstance(text_or_doc, str) else text_or_doc
    except json.JSONDecodeError as e:
        raise PolicyError(f"failed to parse policy JSON: {e}") from e
    pol = Policy()
    for key in doc:
        if key not in ("version", "defaults", "ranges", "layers"):
            if strict:
                raise PolicyError("unknown top-level key in policy: " + key)
            pol.warnings.append("unknown top-level key in policy: " + key)
    if _is_int(doc.get("version")):
        pol.version = doc["version"]
    if isinstance(doc.get("defaults"), dict):
        pol.defaults = _parse_rule(doc["defaults"], strict, pol.warnings)
    for r in doc.get("ranges", []) if isinstance(doc.get("ranges"), list) else []:
        if isinstance(r, dict) and "layers" in r:
            pol.ranges.append((parse_layer_selector(str(r["layers"])), _parse_rule(r, strict, pol.warnings)))
    if isinstance(doc.get("layers"), dict):
        for key, r in doc["layers"].items():
            if isinstance(r, dict):
                pol.layers[int(key)] = _parse_rule(r, strict, pol.warnings)
    return pol


def resolve(policy: Optional[Policy], layer: int, kind: str, baseline: Resolved) -> Resolved:
    """Effective settings of (layer, tensor kind) (seeddelta_policy.cpp:250-313)."""
    out = replace(baseline, autotune=replace(baseline.autotune))
    if policy is None:
        out.gating_enabled = False
        return out
    acc: dict = {}
    _merge(acc, policy.defaults)
    for layers, rule in policy.ranges:
        if layer in layers:
            _merge(acc, rule)
    if layer in policy.layers:
        _merge(acc, policy.layers[layer])
    if kind in acc.get("tensors", {}):
        _merge(acc, acc["tensors"][kind])
    for name in ("enabled", "strip_dense", "block", "K_gate", "K_up", "K_down", "metric"):
        if name in acc:
            setattr(out, name, acc[name])
    if "autotune" in acc:
        out.autotune = replace(acc["autotune"])
    sfx = {"ffn_up": "up", "ffn_down": "down"}.get(kind, "gate")
    if "min_mean_" + sfx in acc:
        out.min_mean = acc["min_mean_" + sfx]
    if "min_p05_" + sfx in acc:
        out.min_p05 = acc["min_p05_" + sfx]
    requested = not (out.min_mean < 0.0 or out.min_p05 < 0.0)
    out.gating_enabled = requested
    out.require_eval_x = requested and out.metric in ("cos_x", "cos_x_w", "ffn_score")
    out.require_imatrix = requested and out.metric in ("cos_w", "cos_x_w", "ffn_score")
    return out
