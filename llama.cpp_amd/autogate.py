"""Layer / tensor selection from a sensitivity scan — the host logic of
``scripts/seeddelta-autogate.py`` (reference), restated as a library: scan document in, policy
document + layer scores (+ forbidden pairs and an evaluation trace) out. No GPU work here; the scan it
consumes is what ``llama_cpp_amd.scan.layer_scan`` produces on the MI355X.

Behaviour follows the reference script function by function (line numbers of
scripts/seeddelta-autogate.py):
  retention score / layer aggregate / rank adjustment      :14-57
  policy document                                           :84-126
  metric lookup with zero-means-missing fallback            :139-147
  functional (FFN-proxy) thresholds: calibrate, resolve, test   :219-315
  per-layer aggregate of the rank metric                    :317-332
  which tensors of a layer may be enabled                   :363-409
  ranking, greedy selection with gap / parity / threshold rules, incremental evaluation with
  rollback and forbidden pairs                              :672-931
The optional incremental evaluation calls back into user code (``evaluate``) instead of spawning
``seeddelta-policy-eval.sh``; everything decided from its result (ΔPPL limit, greedy check, rollback,
forbidden pairs, the step trace) is the script's.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Set, Tuple

TENSOR_KEYS = ("ffn_gate", "ffn_up", "ffn_down")


def _per_tensor(default=None):
    return field(default_factory=lambda: {"gate": default, "up": default, "down": default})


@dataclass
class AutogateConfig:
    """The script's command line (same names, '-' -> '_'; per-tensor overrides as dicts keyed gate/up/down)."""
    k: int = 12
    metric: str = "cos"
    rank_metric: str = "S_mean"
    aggregate: str = "mean"
    tensors: str = "ffn_gate,ffn_up"
    max_layers: int = 0
    percent: float = 0.0
    threshold: float = 0.0
    min_gap: int = 1
    prefer: str = "1,3,5,6"
    prefer_mult: float = 0.95
    avoid_even_only: bool = False
    enable_down: bool = False
    allow_up: bool = False
    gate_cos_metric: str = "cos_mean_x_w"
    gate_cos_p05_metric: str = "cos_p05_x_w"
    gate_cos_threshold: float = 0.7
    gate_cos_p05_threshold: float = 0.5
    gate_cos_threshold_by: Dict[str, Optional[float]] = _per_tensor()
    gate_cos_p05_threshold_by: Dict[str, Optional[float]] = _per_tensor()
    gate_allow_missing: bool = False
    retention_mode: str = "off"               # off | proxy | cos | auto
    retention_min: float = 0.0
    retention_topk: int = 0
    retention_aggregate: str = "max"
    retention_decay_alpha: float = 0.0
    retention_allow_missing: bool = False
    functional_gate: bool = False
    functional_good_layers: str = ""
    functional_cos_margin: float = 0.01
    functional_relax_cos: bool = False
    functional_l2_margin: float = 0.15
    functional_norm_margin: float = 0.15
    functional_cos_p05_threshold: Optional[float] = None
    functional_l2_threshold: Optional[float] = None
    functional_norm_threshold: Optional[float] = None
    functional_cos_p05_threshold_by: Dict[str, Optional[float]] = _per_tensor()
    functional_l2_threshold_by: Dict[str, Optional[float]] = _per_tensor()
    functional_norm_threshold_by: Dict[str, Optional[float]] = _per_tensor()
    functional_early_layer_cutoff: int = 12
    functional_allow_early: bool = False
    functional_early_cos_boost: float = 0.02
    functional_early_l2_scale: float = 0.85
    functional_early_norm_scale: float = 0.85
    functional_allow_missing: bool = False
    policy_strip_dense: bool = False
    layer_allowlist: str = ""
    layer_denylist: str = ""
    max_delta_ppl: float = 5.0


@dataclass
class AutogateResult:
    policy: dict
    scores: Dict[int, float]
    selected: List[int]
    selected_tensors: Dict[int, Set[str]]
    forbidden_pairs: List[List[int]]
    steps: List[dict]

    def report(self) -> Optional[dict]:
        """autogating_report.json (written only when evaluation steps exist, :922-931)."""
        if not self.steps:
            return None
        return {"selected_layers": self.selected,
                "selected_tensors": {str(l): sorted(self.selected_tensors.get(l, set())) for l in self.selected},
                "steps": self.steps}


class AutogateError(ValueError):
    pass


# ---- small helpers -----------------------------------------------------------------------------------------
def parse_layers(spec: str) -> List[int]:
    out: Set[int] = set()
    for tok in (t.strip() for t in spec.split(",")):
        if not tok:
            continue
        if "-" in tok:
            a, b = (int(v) for v in tok.split("-", 1))
            out.update(range(min(a, b), max(a, b) + 1))
        else:
            out.add(int(tok))
    return sorted(out)


def _as_float(v) -> Optional[float]:
    try:
        return None if v is None else float(v)
    except (TypeError, ValueError):
        return None


def _tensors_of(layer_metrics: dict) -> dict:
    inner = layer_metrics.get("tensors")
    return inner if isinstance(inner, dict) else layer_metrics


def metric_value(metrics: dict, primary: str, fallback: str):
    """primary, unless it is absent — or present but exactly 0 while the fallback is a non-zero value
    (a 0.0 written by the builder means 'not computed', e.g. the _x_w fields without an imatrix)."""
    val = metrics.get(primary)
    if val is None:
        return metrics.get(fallback)
    if fallback and isinstance(val, (int, float)) and val == 0.0:
        fb = metrics.get(fallback)
        if fb is not None and fb != 0.0:
            return fb
    return val


def _proxy_triplet(metrics: dict) -> Optional[Tuple[float, float, float]]:
    vals = tuple(_as_float(metrics.get(k)) for k in ("ffn_proxy_cos_p05", "ffn_proxy_l2_p95", "ffn_proxy_log_norm_ratio_p95"))
    return None if None in vals else vals  # type: ignore[return-value]


def write_policy(layers: Sequence[int], layer_tensors: Dict[int, Set[str]], k: int, metric: str,
                 enable_down: bool, strip_dense: bool) -> dict:
    ranges = []
    for layer in layers:
        allowed = set(layer_tensors.get(layer, set()))
        if not enable_down:
            allowed.discard("ffn_down")
        ranges.append({
            "layers": str(layer), "enabled": True, "strip_dense": bool(strip_dense),
            "K": {"gate": k, "up": k, "down": k},
            "gating": {"metric": metric, "min_mean": {"gate": 0.0, "up": 0.0, "down": 0.0},
                       "min_p05": {"gate": 0.0, "up": 0.0, "down": 0.0}},
            "tensors": {t: {"enabled": t in allowed} for t in TENSOR_KEYS},
            "autotune": {"enabled": False},
        })
    return {"version": 1,
            "defaults": {"enabled": False, "gating": {"metric": metric, "min_mean": -1.0, "min_p05": -1.0},
                         "autotune": {"enabled": False}},
            "ranges": ranges}


def validate_scan(layers_dict: dict) -> List[str]:
    problems = []
    for name, lm in layers_dict.items():
        tensors = _tensors_of(lm) if isinstance(lm, dict) else None
        if not isinstance(tensors, dict) or not tensors:
            problems.append(f"layer {name}: missing tensors")
            continue
        for t, m in tensors.items():
            if not isinstance(m, dict):
                problems.append(f"layer {name}:{t} invalid metrics")
            elif "S_mean" not in m and "rel_l2_mean" not in m:
                problems.append(f"layer {name}:{t} missing S_mean")
    return problems


# ---- scores ------------------------------------------------------------------------------------------------
def retention_score(metrics: dict, cfg: AutogateConfig) -> Optional[float]:
    mode = cfg.retention_mode
    if mode == "off":
        return None
    if mode in ("proxy", "auto") and metrics.get("ffn_proxy_available", False):
        trip = _proxy_triplet(metrics)
        if trip is None:
            return None
        cos, l2, norm = trip
        return max(0.0, max(0.0, cos) / (1.0 + l2 + abs(norm)))
    if mode in ("cos", "auto"):
        p05 = _as_float(metric_value(metrics, cfg.gate_cos_p05_metric, "cos_p05"))
        mean = _as_float(metric_value(metrics, cfg.gate_cos_metric, "cos_mean"))
        if p05 is None:
            return None
        if mean is None:
            mean = p05
        return max(0.0, max(0.0, p05) * max(0.0, mean))
    return None


def _layer_rank_value(lm: dict, tensors: Sequence[str], cfg: AutogateConfig) -> Optional[float]:
    if cfg.rank_metric == "S_layer" and "S_layer" in lm:
        return float(lm.get("S_layer", 0.0))
    per = _tensors_of(lm)
    vals = [float(m[cfg.rank_metric]) for m in (per.get(t) for t in tensors)
            if isinstance(m, dict) and m.get(cfg.rank_metric) is not None]
    if not vals:
        return None
    return max(vals) if cfg.aggregate == "max" else sum(vals) / float(len(vals))


def _retention_adjust(val: float, retention: Optional[float], higher_is_better: bool, alpha: float) -> float:
    if retention is None or alpha <= 0.0:
        return val
    penalty = alpha * (1.0 - retention)
    return val * (1.0 - penalty) if higher_is_better else val * (1.0 + penalty)


# ---- functional (FFN proxy) gate -----------------------------------------------------------------------------
def calibrate_functional_thresholds(layers_dict: dict, good_layers: Sequence[int], tensors: Sequence[str],
                                    cfg: AutogateConfig) -> Dict[str, Dict[str, float]]:
    out: Dict[str, Dict[str, float]] = {}
    for tensor in tensors:
        trips = []
        for layer in good_layers:
            lm = layers_dict.get(str(layer), {})
            m = (_tensors_of(lm) if isinstance(lm, dict) else {}).get(tensor)
            if not isinstance(m, dict) or not m.get("ffn_proxy_available", False):
                continue
            trip = _proxy_triplet(m)
            if trip is not None:
                trips.append(trip)
        if not trips:
            continue
        cos_thr = min(t[0] for t in trips)
        if cfg.functional_relax_cos:
            cos_thr -= cfg.functional_cos_margin
        out[tensor] = {"cos_p05": cos_thr,
                       "l2_p95": max(t[1] for t in trips) * (1.0 + cfg.functional_l2_margin),
                       "norm_p95": max(abs(t[2]) for t in trips) * (1.0 + cfg.functional_norm_margin)}
    return out


def resolve_functional_thresholds(cfg: AutogateConfig, tensors: Sequence[str],
                                  calibrated: Dict[str, Dict[str, float]]) -> Dict[str, Dict[str, float]]:
    out = {}
    for tensor in tensors:
        sfx = tensor.replace("ffn_", "")
        base = calibrated.get(tensor, {})

        def pick(override, key, default):
            return override if override is not None else base.get(key, default)
        cos = pick(cfg.functional_cos_p05_threshold_by.get(sfx), "cos_p05", cfg.functional_cos_p05_threshold)
        l2 = pick(cfg.functional_l2_threshold_by.get(sfx), "l2_p95", cfg.functional_l2_threshold)
        norm = pick(cfg.functional_norm_threshold_by.get(sfx), "norm_p95", cfg.functional_norm_threshold)
        if None in (cos, l2, norm):
            continue
        out[tensor] = {"cos_p05": float(cos), "l2_p95": float(l2), "norm_p95": float(norm)}
    return out


def passes_functional_gate(layer: int, metrics: dict, thr: Optional[Dict[str, float]], cfg: AutogateConfig) -> bool:
    if thr is None or not metrics.get("ffn_proxy_available", False):
        return cfg.functional_allow_missing
    trip = _proxy_triplet(metrics)
    if trip is None:
        return cfg.functional_allow_missing
    cos, l2, norm = trip
    cos_thr, l2_thr, norm_thr = thr["cos_p05"], thr["l2_p95"], thr["norm_p95"]
    if layer < cfg.functional_early_layer_cutoff:
        if not cfg.functional_allow_early:
            return False
        cos_thr += cfg.functional_early_cos_boost
        l2_thr *= cfg.functional_early_l2_scale
        norm_thr *= cfg.functional_early_norm_scale
    return cos >= cos_thr and l2 <= l2_thr and abs(norm) <= norm_thr


def decide_allowed_tensors(layer: int, lm: dict, cfg: AutogateConfig, functional: Dict[str, Dict[str, float]],
                           retention: Dict[int, Dict[str, float]], retention_allow: Optional[Dict[str, Set[int]]]) -> List[str]:
    out = []
    per = _tensors_of(lm)
    for tensor in TENSOR_KEYS:
        if (tensor == "ffn_down" and not cfg.enable_down) or (tensor == "ffn_up" and not cfg.allow_up):
            continue
        sfx = tensor.replace("ffn_", "")
        mean_thr = cfg.gate_cos_threshold_by.get(sfx)
        p05_thr = cfg.gate_cos_p05_threshold_by.get(sfx)
        mean_thr = cfg.gate_cos_threshold if mean_thr is None else mean_thr
        p05_thr = cfg.gate_cos_p05_threshold if p05_thr is None else p05_thr
        m = per.get(tensor)
        if not isinstance(m, dict):
            continue
        cos_mean = metric_value(m, cfg.gate_cos_metric, "cos_mean")
        cos_p05 = metric_value(m, cfg.gate_cos_p05_metric, "cos_p05")
        if cos_mean is None or cos_p05 is None:
            if cfg.gate_allow_missing:
                out.append(tensor)
            continue
        if cos_mean < mean_thr or cos_p05 < p05_thr:
            continue
        if cfg.functional_gate and not passes_functional_gate(layer, m, functional.get(tensor), cfg):
            continue
        if cfg.retention_mode != "off":
            score = retention.get(layer, {}).get(tensor)
            if score is None:
                if not cfg.retention_allow_missing:
                    continue
            else:
                if cfg.retention_min > 0.0 and score < cfg.retention_min:
                    continue
                if retention_allow is not None and layer not in retention_allow.get(tensor, set()):
                    continue
        out.append(tensor)
    return out


# ---- the selection -------------------------------------------------------------------------------------------
EvalFn = Callable[[List[int], Dict[int, Set[str]], dict], dict]   # -> {"base_ppl", "seeddelta_ppl", "greedy_ok"}


def autogate(scan: dict, cfg: AutogateConfig, evaluate: Optional[EvalFn] = None) -> AutogateResult:
    layers_dict = scan.get("layers", {})
    if not isinstance(layers_dict, dict):
        raise AutogateError("scan.layers missing or invalid")
    tensors = [t.strip() for t in cfg.tensors.split(",") if t.strip()]
    if not cfg.allow_up:
        tensors = [t for t in tensors if t != "ffn_up"]
    prefer = set(parse_layers(cfg.prefer)) if cfg.prefer else set()
    allow = set(parse_layers(cfg.layer_allowlist)) if cfg.layer_allowlist.strip() else None
    deny = set(parse_layers(cfg.layer_denylist)) if cfg.layer_denylist.strip() else set()

    functional: Dict[str, Dict[str, float]] = {}
    if cfg.functional_gate:
        calibrated: Dict[str, Dict[str, float]] = {}
        good = parse_layers(cfg.functional_good_layers) if cfg.functional_good_layers else []
        if good:
            calibrated = calibrate_functional_thresholds(layers_dict, good, tensors, cfg)
            if not calibrated:
                raise AutogateError("functional calibration produced no thresholds")
        functional = resolve_functional_thresholds(cfg, tensors, calibrated)
        missing = [t for t in tensors if t not in functional]
        if missing:
            raise AutogateError("functional thresholds missing for " + ", ".join(missing))

    def usable_layers():
        for name, lm in layers_dict.items():
            try:
                layer = int(name)
            except ValueError:
                continue
            if (allow is not None and layer not in allow) or layer in deny or not isinstance(lm, dict):
                continue
            yield layer, lm

    retention: Dict[int, Dict[str, float]] = {}
    for layer, lm in usable_layers():
        per = _tensors_of(lm)
        for tensor in TENSOR_KEYS:
            m = per.get(tensor)
            if isinstance(m, dict):
                sc = retention_score(m, cfg)
                if sc is not None:
                    retention.setdefault(layer, {})[tensor] = float(sc)

    retention_allow: Optional[Dict[str, Set[int]]] = None
    if cfg.retention_topk > 0 and retention:
        retention_allow = {}
        for tensor in TENSOR_KEYS:
            order = sorted(((sc[tensor], layer) for layer, sc in retention.items() if sc.get(tensor) is not None), reverse=True)
            retention_allow[tensor] = {layer for _, layer in order[:cfg.retention_topk]}

    higher = cfg.rank_metric.startswith("cos")
    scores: Dict[int, float] = {}
    for layer, lm in usable_layers():
        val = _layer_rank_value(lm, tensors, cfg)
        if val is None:
            continue
        if cfg.retention_decay_alpha > 0.0:
            vals = [v for v in retention.get(layer, {}).values() if v is not None]
            layer_ret = None if not vals else (sum(vals) / len(vals) if cfg.retention_aggregate == "mean" else max(vals))
            val = _retention_adjust(val, layer_ret, higher, cfg.retention_decay_alpha)
        scores[layer] = float(val)
    if not scores:
        raise AutogateError("no scores computed")

    ranked = sorted(((-v if higher else v) * (cfg.prefer_mult if layer in prefer else 1.0), layer) for layer, v in scores.items())
    cap = cfg.max_layers if cfg.max_layers > 0 else None
    if cfg.percent and cfg.percent > 0.0:
        cap = max(1, int(math.ceil(len(ranked) * (cfg.percent / 100.0))))

    selected: List[int] = []
    chosen: Dict[int, Set[str]] = {}
    forbidden: Set[Tuple[int, int]] = set()
    steps: List[dict] = []
    for _, layer in ranked:
        if cap is not None and len(selected) >= cap:
            break
        if cfg.threshold and cfg.threshold > 0.0:
            v = scores.get(layer, 0.0)
            if (not higher and v > cfg.threshold) or (higher and v < cfg.threshold):
                continue
        if cfg.min_gap > 0 and any(abs(layer - s) <= cfg.min_gap for s in selected):
            continue
        if cfg.avoid_even_only and layer % 2 == 0 and all(s % 2 == 0 for s in selected):
            continue
        allowed = decide_allowed_tensors(layer, layers_dict.get(str(layer), {}), cfg, functional, retention, retention_allow)
        if not allowed:
            continue
        cand = sorted(selected + [layer])
        cand_tensors = dict(chosen)
        cand_tensors[layer] = set(allowed)
        if evaluate is None:
            selected, chosen = cand, cand_tensors
            continue
        policy = write_policy(cand, cand_tensors, cfg.k, cfg.metric, cfg.enable_down, cfg.policy_strip_dense)
        r = evaluate(cand, cand_tensors, policy)
        base_ppl, sd_ppl, greedy_ok = r.get("base_ppl"), r.get("seeddelta_ppl"), r.get("greedy_ok")
        delta_pct, delta_ok = None, True
        if base_ppl and sd_ppl:
            delta_pct = (sd_ppl - base_ppl) / base_ppl * 100.0
            delta_ok = delta_pct <= cfg.max_delta_ppl
        ok = (greedy_ok is not False) and delta_ok
        steps.append({"layer": layer, "candidate": cand, "tensors": sorted(allowed), "run_dir": r.get("run_dir", ""),
                      "base_ppl": base_ppl, "seeddelta_ppl": sd_ppl, "delta_pct": delta_pct, "greedy_ok": greedy_ok, "pass": ok})
        if ok:
            selected, chosen = cand, cand_tensors
        else:
            forbidden.update(tuple(sorted((s, layer))) for s in selected)

    policy = write_policy(selected, chosen, cfg.k, cfg.metric, cfg.enable_down, cfg.policy_strip_dense)
    return AutogateResult(policy, scores, selected, chosen, sorted(list(p) for p in forbidden), steps)
