"""The candidate-evaluation loop behind ``scripts/seeddelta-layer-scan.py`` / ``seeddelta-autogate.py``
on the MI355X: host-side mirror of the reference's script and builder interfaces over the
``sdh_scan_*`` C ABI (include/seeddelta_hip.h).

What maps to what (reference paths relative to /root/reference):

* ``DenseWeight``      — a dense FFN tensor resident in HBM (``read_column_f32``,
                         tools/seeddelta-build/sd_eval.cpp:20-40, happens inside the kernels);
* ``ScanSession``      — one ``llama-seeddelta-build`` process: an ``std::mt19937(seed)``
                         (sd_build.cpp:260) consumed tensor after tensor in the builder's order;
* ``ScanParams``       — the builder flags that shape one trial (sd_cli.cpp / include/sd_cli.h:20-50);
* ``report_entry``     — the ``weights[]`` element of report.json (sd_report.cpp:163-281) with the
                         single-K gating decision of sd_build.cpp:843-913;
* ``write_policy``, ``collect_metrics``, ``layer_scan`` — scripts/seeddelta-layer-scan.py:31-59,
                         :62-122, :206-286: same policy document, same field mapping, same
                         ``layer_sensitivity_scan.json`` structure;
* ``shard`` / ``gather_records`` — the multi-GPU split of SURVEY §8e: independent (layer, tensor, K)
                         candidates dealt to ranks, one all_gather of fixed-size records at the end.

No CPU fallback: everything numeric runs in libseeddelta_hip.so.
"""
from __future__ import annotations

import ctypes as C
import json
import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import seeddelta as sd
from .dense import dense_dims, dense_type

KINDS = ("ffn_gate", "ffn_up", "ffn_down")
KIND_ID = {k: i for i, k in enumerate(KINDS)}
METRIC_NAMES = ("rel_l2_mean", "rel_l2_p95", "cos_mean", "cos_p05", "norm_ratio_mean")
COST_NAMES = ("base_L", "base_B", "ops_dense", "ops_base", "ops_delta", "ops_row_scale", "ops_total", "ops_ratio")
PROXY_NAMES = ("ffn_proxy_cos_mean", "ffn_proxy_cos_p05", "ffn_proxy_l2_mean", "ffn_proxy_l2_p95",
               "ffn_proxy_log_norm_ratio_mean", "ffn_proxy_log_norm_ratio_p95", "ffn_proxy_eval_x", "ffn_proxy_eval_out")


class DenseWeight:
    """Upload-once handle of a dense weight (numpy (n_out, n_in) f32 / f16, or raw Q8_0 rows)."""

    def __init__(self, W: np.ndarray):
        lib = sd.load_library()
        W = np.ascontiguousarray(W)
        self.n_in, self.n_out = dense_dims(W)
        t = sd.SdhTensor()
        t.data, t.type, t.n_dims = W.ctypes.data, dense_type(W), 2
        elem = {0: 4, 1: 2, 8: 34}[t.type]
        t.ne[:] = [self.n_in, self.n_out, 1, 1]
        row = W.strides[0]
        t.nb[:] = [elem, row, row * self.n_out, row * self.n_out]
        h = C.c_void_p()
        sd._check(lib.sdh_dense_create(C.byref(t), C.byref(h)), "sdh_dense_create")
        self._h = h
        self.stored_bytes = int(lib.sdh_dense_bytes(h))

    def close(self):
        if getattr(self, "_h", None):
            sd.load_library().sdh_dense_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class ScanParams:
    scheme: str = "coo"            # --scheme
    K: int = 12                    # --K  (layer-scan default, scripts/seeddelta-layer-scan.py:132)
    block: int = 32                # --block
    base: bool = False             # --base
    row_scale: bool = False        # --row-scale
    base_max_samples: int = 2048   # --base-max-samples
    base_perm_trials: int = 4      # --base-perm-trials (layer-scan default)
    eval_cols: int = 64            # --eval-cols
    eval_x: int = 32               # --eval-x
    seed: int = 1234               # --seed
    metric: str = "cos"            # policy gating metric
    w_scale: Optional[np.ndarray] = None   # imatrix scale for this tensor (imatrix_scale()); None = no imatrix
    base_mode: int = 0             # 0 fit; 1 reuse the previous fit, no draws; 2 reuse it, replay its draws
    delta_norm_clamp_down: float = 0.0   # --delta-norm-clamp-down (ffn_down, COO): cap ||delta||_2 at this multiple of ||W||_2
    metrics_only: bool = False     # select only the rows the metric passes sample: same report entry, no payload
                                   # to fetch (what layer-scan / autogate need; build_run keeps the full selection)

    def c_struct(self, kind: str, layer: int) -> sd.SdhScanParams:
        p = sd.SdhScanParams()
        p.scheme = 1 if self.scheme == "block" else 0
        p.kind = KIND_ID[kind]
        p.K, p.block = self.K, self.block
        p.write_base, p.write_row_scale = int(self.base), int(self.row_scale)
        p.base_max_samples, p.base_perm_trials = self.base_max_samples, self.base_perm_trials
        p.want_proxy = 1
        p.eval_cols, p.eval_x = self.eval_cols, self.eval_x
        p.proxy_seed = self.seed + layer * 101 + KIND_ID[kind] + 1        # sd_build.cpp:1032
        p.base_mode = self.base_mode if self.base else 0
        p.metrics_only = int(self.metrics_only)
        p.delta_norm_clamp_down = float(self.delta_norm_clamp_down)
        if self.w_scale is not None:
            p._ws = np.ascontiguousarray(self.w_scale, dtype=np.float32)      # kept alive by the struct that points at it
            p.w_scale = p._ws.ctypes.data
        return p


MAX_KGRID = 16       # SDH_SCAN_MAX_K of include/seeddelta_hip.h


class ScanSession:
    """One builder process: RNG stream + the last candidate's outputs on the device."""

    def __init__(self, seed: int):
        lib = sd.load_library()
        h = C.c_void_p()
        sd._check(lib.sdh_scan_create(C.c_uint32(seed & 0xFFFFFFFF), C.byref(h)), "sdh_scan_create")
        self._h = h
        self._last = None
        self._grid = None

    def close(self):
        if getattr(self, "_h", None):
            sd.load_library().sdh_scan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor(self, W: DenseWeight, params: ScanParams, kind: str = "ffn_gate", layer: int = 0,
               ffn: Optional[Tuple[DenseWeight, DenseWeight, DenseWeight]] = None) -> dict:
        """Evaluate one candidate; returns the raw result (metric vectors as numpy arrays)."""
        lib = sd.load_library()
        p = params.c_struct(kind, layer)
        r = sd.SdhScanResult()
        g, u, d = (x._h if x is not None else None for x in (ffn or (None, None, None)))
        sd._check(lib.sdh_scan_tensor(self._h, W._h, g, u, d, C.byref(p), C.byref(r)), "sdh_scan_tensor")
        self._last = (W.n_in, W.n_out, params, int(r.K_eff), int(r.n_blocks), int(r.L), int(r.B))
        self._grid = None
        return self._result(r)

    @staticmethod
    def _result(r) -> dict:
        return {"K_eff": int(r.K_eff), "n_blocks": int(r.n_blocks), "L": int(r.L), "B": int(r.B),
                "has_x": bool(r.has_x), "has_w": bool(r.has_w), "proxy_available": bool(r.proxy_available),
                "em": np.array(r.em[:]), "em_x": np.array(r.em_x[:]), "em_w": np.array(r.em_w[:]),
                "em_x_w": np.array(r.em_x_w[:]), "cost": np.array(r.cost[:]),
                "proxy": np.array(r.proxy[:]), "gpu_ms": float(r.gpu_ms),
                "delta_clamp_scale": float(r.delta_clamp_scale)}

    def tensor_kgrid(self, W: DenseWeight, params: ScanParams, Ks: Sequence[int], kind: str = "ffn_gate", layer: int = 0,
                     ffn: Optional[Tuple[DenseWeight, DenseWeight, DenseWeight]] = None, sequential: bool = False) -> List[dict]:
        """A K schedule of one tensor from ONE fit and ONE selection (the builder fits once per tensor and runs a
        trial per K, sd_build.cpp:444-451, :843-913): result j is what tensor() with K = Ks[j] returns from the
        same generator state; the generator advances once. pick(j) chooses the member fetch*() returns
        (default: the last). params.K is ignored. At most MAX_KGRID budgets per call.
        sequential=True: the K are the trials of ONE builder run's autotune schedule (sd_build.cpp:843-913) — trial j
        draws its eval samples after trial j-1's, so result j is what tensor() calls in sequence on this handle return
        (base_mode 1 from the second on); rewind(j) then puts the generator where a builder that stopped after trial
        j would be."""
        lib = sd.load_library()
        p = params.c_struct(kind, layer)
        p.kgrid_draws = 1 if sequential else 0
        n = len(Ks)
        rs = (sd.SdhScanResult * n)()
        ks = (C.c_int64 * n)(*[int(k) for k in Ks])
        g, u, d = (x._h if x is not None else None for x in (ffn or (None, None, None)))
        sd._check(lib.sdh_scan_tensor_kgrid(self._h, W._h, g, u, d, C.byref(p), ks, n, rs), "sdh_scan_tensor_kgrid")
        self._grid = [(W.n_in, W.n_out, params, int(r.K_eff), int(r.n_blocks), int(r.L), int(r.B)) for r in rs]
        self._last = self._grid[-1]
        return [self._result(r) for r in rs]

    def rewind(self, j: int):
        """After tensor_kgrid(sequential=True): generator state as after trial j."""
        sd._check(sd.load_library().sdh_scan_rewind(self._h, int(j)), "sdh_scan_rewind")

    def pick(self, j: int):
        """Choose which K of the last tensor_kgrid() the fetch calls return."""
        if not self._grid:
            raise ValueError("pick: the last call was not a K schedule")
        sd._check(sd.load_library().sdh_scan_pick(self._h, int(j)), "sdh_scan_pick")
        self._last = self._grid[j]

    def reseed(self, seed: int):
        """A new builder run on this handle (fresh mt19937), keeping device buffers and the last fit."""
        sd._check(sd.load_library().sdh_scan_reseed(self._h, C.c_uint32(seed & 0xFFFFFFFF)), "sdh_scan_reseed")

    def skip(self, n_in: int, n_out: int, params: ScanParams, kind: str = "ffn_gate", layer: int = 0):
        """Advance the RNG as evaluating a tensor of these dims would (no GPU work)."""
        p = params.c_struct(kind, layer)
        sd._check(sd.load_library().sdh_scan_skip(self._h, n_in, n_out, C.byref(p)), "sdh_scan_skip")

    def fetch(self) -> dict:
        """The emitted staging of the last candidate: idx/val[/row_scale][/base arrays]."""
        n_in, n_out, params, K_eff, n_blocks, L, B = self._last
        block = params.scheme == "block"
        idx = np.empty((n_out, n_blocks if block else K_eff), dtype=np.int32)
        val = np.empty((n_out, n_blocks, params.block) if block else (n_out, K_eff), dtype=np.float32)
        rs = np.empty(n_out, dtype=np.float32) if params.row_scale else None
        out = {"idx": idx, "val": val, "row_scale": rs}
        ptr = lambda a: None if a is None else C.c_void_p(a.ctypes.data)   # noqa: E731
        base = [None] * 4
        if params.base:
            out.update(d2=np.empty((B, L), np.float32), h2=np.empty((B, L), np.float32),
                       perm1=np.empty((B, L), np.int32), perm1_inv=np.empty((B, L), np.int32))
            base = [out["d2"], out["h2"], out["perm1"], out["perm1_inv"]]
        sd._check(sd.load_library().sdh_scan_fetch(self._h, ptr(idx), ptr(val), ptr(rs), *[ptr(a) for a in base]),
                  "sdh_scan_fetch")
        return out


    def fetch_packed(self, idx_dtype=np.int16, val_dtype=np.float16) -> dict:
        """The last candidate's outputs as the GGUF stores them: the builder's packing step
        (sd_build.cpp:1104-1236) runs on the device before the copy. Same keys as fetch() minus h2 /
        perm1_inv; pack.pack_candidate accepts the result unchanged (its conversions are idempotent)."""
        n_in, n_out, params, K_eff, n_blocks, L, B = self._last
        block = params.scheme == "block"
        idx = np.empty((n_out, n_blocks if block else K_eff), dtype=idx_dtype)
        val = np.empty((n_out, n_blocks, params.block) if block else (n_out, K_eff), dtype=val_dtype)
        rs = np.empty(n_out, dtype=np.float16) if params.row_scale else None
        tcode = sd.SDH_TYPE
        out = {"idx": idx, "val": val, "row_scale": rs}
        ptr = lambda a: None if a is None else C.c_void_p(a.ctypes.data)   # noqa: E731
        d2 = perm = None
        ptype = C.c_int(0)
        if params.base:
            d2, perm = np.empty((B, L), np.float16), np.empty((B, L), np.int32)
        sd._check(sd.load_library().sdh_scan_fetch_packed(self._h, tcode[np.dtype(idx_dtype)], ptr(idx), tcode[np.dtype(val_dtype)],
                                                          ptr(val), ptr(rs), ptr(d2), ptr(perm), C.byref(ptype)), "sdh_scan_fetch_packed")
        if params.base:
            out["d2"] = d2
            out["perm1"] = perm if ptype.value == tcode[np.dtype(np.int32)] else perm.view(np.int16).reshape(-1)[:B * L].reshape(B, L).copy()
        return out


# ---- report entry (sd_report.cpp:163-281; gating of sd_build.cpp:810-913 for a one-K schedule) -----------
def ffn_score(entry: dict) -> float:
    """sd_report.cpp:72-88: min(cos_mean, cos_p05) of the first available of x_w, x, w, plain."""
    for need_x, need_w, sfx in ((True, True, "_x_w"), (True, False, "_x"), (False, True, "_w")):
        if (not need_x or entry["has_x"]) and (not need_w or entry["has_w"]) and \
                (entry["cos_mean" + sfx] != 0.0 or entry["cos_p05" + sfx] != 0.0):
            return min(entry["cos_mean" + sfx], entry["cos_p05" + sfx])
    return min(entry["cos_mean"], entry["cos_p05"])


def report_entry(layer: int, kind: str, n_in: int, n_out: int, params: ScanParams, res: dict,
                 min_mean: float = 0.0, min_p05: float = 0.0, gating_enabled: bool = True) -> dict:
    e = {"layer": layer, "kind": kind, "n_in": n_in, "n_out": n_out, "K_budget": params.K, "K": res["K_eff"],
         "block": params.block, "n_blocks": res["n_blocks"], "has_w": bool(res.get("has_w", False)), "has_x": res["has_x"]}
    e.update({n: (int(v) if i < 2 else float(v)) for i, (n, v) in enumerate(zip(COST_NAMES, res["cost"]))})
    e.update({n: float(v) for n, v in zip(METRIC_NAMES, res["em"])})
    for suffix, vec in (("_w", res.get("em_w", np.zeros(5))), ("_x", res["em_x"]), ("_x_w", res.get("em_x_w", np.zeros(5)))):
        e.update({n + suffix: float(v) for n, v in zip(METRIC_NAMES, vec)})
    e["ffn_score"] = ffn_score(e)
    # gate/up prefer the FFN score when x metrics exist (sd_build.cpp:810-816)
    metric = "ffn_score" if (kind in ("ffn_gate", "ffn_up") and res["has_x"]) else params.metric
    needs_x = metric in ("cos_x", "cos_x_w", "ffn_score")
    needs_w = metric in ("cos_w", "cos_x_w")
    available = not (gating_enabled and ((needs_x and not res["has_x"]) or (needs_w and not e["has_w"])))
    pick = {"cos": ("cos_mean", "cos_p05"), "cos_x": ("cos_mean_x", "cos_p05_x"), "cos_w": ("cos_mean_w", "cos_p05_w"),
            "cos_x_w": ("cos_mean_x_w", "cos_p05_x_w")}
    val, p05 = (e["ffn_score"], e["ffn_score"]) if metric == "ffn_score" else (e[pick[metric][0]], e[pick[metric][1]])
    passed = (not gating_enabled) or (available and val >= min_mean and p05 >= min_p05)
    e.update(gating_enabled=gating_enabled, gating_metric=metric, gating_value=val, gating_p05=p05,
             gating_min_mean=min_mean, gating_min_p05=min_p05, gating_pass=bool(passed and available),
             metric_used=metric, targets_used={"tau_mean": min_mean, "tau_p05": min_p05}, emit=bool(passed))
    e["decision"] = ("no_gating" if not gating_enabled else "metric_unavailable" if not available
                     else "pass_gating" if passed else "gating_fail")
    e["reject_reason"] = e["decision"]
    e["ffn_proxy_available"] = res["proxy_available"]
    if res["proxy_available"]:
        e["ffn_proxy_reason"] = ""
        e["ffn_proxy_scope"] = "replace_only_current_tensor"
        e["ffn_proxy_base_used"] = False
        e["ffn_proxy_seed"] = params.seed + layer * 101 + KIND_ID[kind] + 1
        for i, (n, v) in enumerate(zip(PROXY_NAMES, res["proxy"])):
            e[n] = int(v) if i >= 6 else float(v)
    else:
        coo_nobase = params.scheme == "coo" and not params.base and params.eval_x > 0
        e["ffn_proxy_reason"] = (("proxy_kind_not_supported" if kind == "ffn_down" else "proxy_unavailable") if coo_nobase
                                 else "proxy_requires_eval_x" if params.eval_x <= 0 else "proxy_requires_coo")
    return e


def autotune(session: "ScanSession", W: DenseWeight, params: ScanParams, schedule: Sequence[int], kind: str, layer: int,
             ffn=None, min_mean: float = 0.0, min_p05: float = 0.0, max_iters: int = 0, gating_enabled: bool = True,
             kgrid: bool = True) -> dict:
    """The K schedule of one tensor inside one builder run (sd_build.cpp:802-948): the base is fitted
    once, then one trial per distinct positive K in schedule order, each drawing its eval samples from
    the running stream; the first K whose gating passes is emitted. If none passes, the trial with the
    best gating value is reported with emit = False and decision 'autotune_failed_keep_dense'."""
    seen, ks = set(), []
    for k in schedule:
        if k > 0 and k not in seen:
            seen.add(k)
            ks.append(k)
    if max_iters > 0:
        ks = ks[:max_iters]
    attempts, best, best_score, chosen = [], None, -1e30, None
    # The whole schedule goes to the device as one call: one fit, ONE selection at the largest K (every trial's
    # payload is a prefix of the sorted rows), each trial's metric passes on its own draws in the builder's order.
    # The builder breaks at the first trial that passes and draws nothing for the later ones: the generator is
    # rewound to that trial and its payload picked. The clamp rescales the stored values per K, and a schedule
    # longer than one call holds is rare: both keep the trial-by-trial form.
    clamp = params.delta_norm_clamp_down > 0.0 and kind == "ffn_down" and params.scheme == "coo"
    batched = None
    if kgrid and not clamp and 1 < len(ks) <= MAX_KGRID:
        batched = session.tensor_kgrid(W, ScanParams(**{**params.__dict__, "K": ks[0]}), ks, kind, layer, ffn, sequential=True)
    for n, k in enumerate(ks):
        p = ScanParams(**{**params.__dict__, "K": k, "base_mode": (1 if n > 0 else params.base_mode)})
        res = batched[n] if batched is not None else session.tensor(W, p, kind, layer, ffn)
        e = report_entry(layer, kind, W.n_in, W.n_out, p, res, min_mean, min_p05, gating_enabled)
        attempts.append({"K_budget": k, "K_eff": res["K_eff"], "n_blocks": res["n_blocks"], "metric_value": e["gating_value"],
                         "metric_p05": e["gating_p05"], "pass": e["emit"], "seconds": res["gpu_ms"] / 1e3})
        if e["decision"] != "metric_unavailable" and e["gating_value"] > best_score:
            best_score, best = e["gating_value"], e
        if e["emit"]:
            chosen = e
            break
    if batched is not None:
        last = len(attempts) - 1
        session.rewind(last)
        session.pick(last)                      # fetch*() returns the last trial run, as after trial-by-trial calls
    if chosen is None:
        chosen = dict(best) if best is not None else {"layer": layer, "kind": kind, "n_in": W.n_in, "n_out": W.n_out,
                                                      "decision": "metric_unavailable", "gating_pass": False}
        chosen["_trial_decision"] = chosen.get("decision")
        chosen.update(emit=False, strip_dense=False, decision="autotune_failed_keep_dense",
                      reject_reason="autotune_failed_keep_dense", autotune_selected_budget=0)
    else:
        chosen["autotune_selected_budget"] = chosen["K_budget"]
    chosen.update(autotune_enabled=True, autotune_attempts=attempts)
    return chosen


def imatrix_scale(imatrix_row: np.ndarray, n_in: int, eps: float = 1e-8, power: float = 1.0) -> Optional[np.ndarray]:
    """Per-input weighting from one importance-matrix entry (sd_make_imatrix_sqrt_scale,
    tools/seeddelta-build/sd_utils.cpp:254-295): max(v, eps)^(power/2) normalised to mean 1; None if the
    entry is shorter than n_in. Computed by the library with the same libm call the reference makes."""
    row = np.ascontiguousarray(imatrix_row, dtype=np.float32)
    if row.size < n_in:
        return None
    out = np.empty(n_in, dtype=np.float32)
    sd._check(sd.load_library().sdh_imatrix_scale(row.ctypes.data, n_in, eps, power, out.ctypes.data), "sdh_imatrix_scale")
    return out


def load_imatrix(path: str) -> Dict[str, np.ndarray]:
    """An imatrix GGUF -> {weight name: mean squared activation per input} (sd_utils.cpp:154-228: tensors
    <name>.in_sum2 divided by <name>.counts per row; rows with count 0 become 1)."""
    from .gguf_io import GGUFReader
    out = {}
    with GGUFReader(path) as r:
        for name in r.names():
            if not name.endswith(".in_sum2"):
                continue
            base = name[:-len(".in_sum2")]
            if base + ".counts" not in r.infos:
                raise ValueError(f"mismatched sums and counts for {base}")
            sums = np.array(r.tensor(name), dtype=np.float32)
            sums = sums.reshape(-1, sums.shape[-1])
            counts = np.array(r.tensor(base + ".counts"), dtype=np.float32).reshape(-1)
            vals = np.ones_like(sums)
            ok = counts > 0
            vals[ok] = sums[ok] / counts[ok, None]
            out[base] = vals.reshape(-1)
    return out


def round_like_report(v):
    """report.json is written through an ostream at default precision: 6 significant digits."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float(f"{v:.6g}")


# ---- scripts/seeddelta-layer-scan.py ------------------------------------------------------------------------
def parse_layers(spec: str) -> List[int]:
    """"0-31" / "0,2,4" / mixed -> sorted unique layer list (layer-scan.py:13-28; reversed ranges swap)."""
    layers = set()
    for token in (t.strip() for t in spec.split(",")):
        if not token:
            continue
        if "-" in token:
            lo, hi = (int(v) for v in token.split("-", 1))
            layers.update(range(min(lo, hi), max(lo, hi) + 1))
        else:
            layers.add(int(token))
    return sorted(layers)


def write_policy(layer: int, tensors: Sequence[str], k: int, metric: str) -> dict:
    """The per-layer policy document the scan hands to the builder (layer-scan.py:31-59)."""
    return {
        "version": 1,
        "defaults": {"enabled": False, "gating": {"metric": metric, "min_mean": -1.0, "min_p05": -1.0},
                     "autotune": {"enabled": False}},
        "ranges": [{
            "layers": str(layer), "enabled": True, "strip_dense": False,
            "K": {"gate": k, "up": k, "down": k},
            "gating": {"metric": metric, "min_mean": {"gate": 0.0, "up": 0.0, "down": 0.0},
                       "min_p05": {"gate": 0.0, "up": 0.0, "down": 0.0}},
            "tensors": {t: {"enabled": t in tensors} for t in KINDS},
            "autotune": {"enabled": False},
        }],
    }


def collect_metrics(entry: dict) -> dict:
    """report.json entry -> scan record (layer-scan.py:62-122), field for field."""
    def f(key):
        v = entry.get(key)
        try:
            return None if v is None else float(v)
        except (TypeError, ValueError):
            return None
    nr = f("norm_ratio_mean") or 0.0
    proxy = {k: f(k) for k in PROXY_NAMES[:6]}
    score = None
    if None not in (proxy["ffn_proxy_cos_p05"], proxy["ffn_proxy_l2_p95"], proxy["ffn_proxy_log_norm_ratio_p95"]):
        score = max(1.0 - proxy["ffn_proxy_cos_p05"], proxy["ffn_proxy_l2_p95"], abs(proxy["ffn_proxy_log_norm_ratio_p95"]))
    out = {"S_mean": f("rel_l2_mean") or 0.0, "S_p95": f("rel_l2_p95") or 0.0,
           "cos_mean": f("cos_mean") or 0.0, "cos_p05": f("cos_p05") or 0.0,
           "cos_mean_x_w": f("cos_mean_x_w") or 0.0, "cos_p05_x_w": f("cos_p05_x_w") or 0.0,
           "log_norm_ratio_mean": math.log(nr) if nr > 0 else 0.0, "log_norm_ratio_p95": None,
           "ffn_proxy_available": bool(entry.get("ffn_proxy_available", False))}
    out.update(proxy)
    out["ffn_proxy_score"] = score
    out["raw"] = {"rel_l2_mean": out["S_mean"], "rel_l2_p95": out["S_p95"], "norm_ratio_mean": nr}
    return out


ModelFn = Callable[[int, str], np.ndarray]        # (layer, kind) -> dense weight array


@dataclass
class Candidate:
    layer: int
    kind: str
    K: int

    def key(self):
        return (self.layer, KIND_ID[self.kind], self.K)


def enumerate_candidates(layers: Iterable[int], tensors: Sequence[str], k_grid: Dict[str, Sequence[int]]) -> List[Candidate]:
    """All (layer, tensor, K) trials of a sweep, in the order a serial run visits them."""
    return [Candidate(l, t, k) for l in layers for t in tensors for k in k_grid[t]]


def shard(cands: Sequence[Candidate], rank: int, world: int, cost: Callable[[Candidate], float]) -> List[Candidate]:
    """Deal candidates to ranks: heaviest first, each to the least-loaded rank so far (ties -> lowest
    rank). Deterministic and independent of who calls it."""
    load = [0.0] * world
    mine = []
    for c in sorted(cands, key=lambda c: (-cost(c), c.key())):
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += cost(c)
        if r == rank:
            mine.append(c)
    return mine


def shard_groups(cands: Sequence[Candidate], rank: int, world: int, cost: Callable[[Candidate], float],
                 group_key: Optional[Callable[[Candidate], tuple]] = None) -> Tuple[List[Candidate], List[int]]:
    """The deal the multi-GPU sweep uses: all trials of one (layer, tensor) go to ONE rank — its
    dense weight is uploaded once and the K grid shares fit, dequant and sort — and the groups are dealt
    heaviest first to the least-loaded rank (ties -> lowest rank). Every rank computes the same deal without
    communication. Returns (this rank's candidates in the order of `cands`, candidates per rank)."""
    key = group_key or (lambda c: (c.layer, KIND_ID[c.kind]))
    groups: Dict[tuple, List[Candidate]] = {}
    for c in cands:
        groups.setdefault(key(c), []).append(c)
    load, owner = [0.0] * world, {}
    for g in sorted(groups, key=lambda g: (-sum(cost(c) for c in groups[g]), g)):
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += sum(cost(c) for c in groups[g])
        owner[g] = r
    mine = [c for c in cands if owner[key(c)] == rank]
    counts = [sum(len(groups[g]) for g in groups if owner[g] == r) for r in range(world)]
    return mine, counts


RECORD_LEN = 40


def to_record(c: Candidate, res: dict) -> np.ndarray:
    r = np.zeros(RECORD_LEN)
    r[0:3] = c.key()
    r[3:9] = [res["K_eff"], res["n_blocks"], res["L"], res["B"], res["has_x"], res["proxy_available"]]
    r[9:14], r[14:19], r[19:27], r[27:35] = res["em"], res["em_x"], res["cost"], res["proxy"]
    r[35] = res["gpu_ms"]
    return r


def from_record(r: np.ndarray) -> Tuple[Candidate, dict]:
    c = Candidate(int(r[0]), KINDS[int(r[1])], int(r[2]))
    res = {"K_eff": int(r[3]), "n_blocks": int(r[4]), "L": int(r[5]), "B": int(r[6]), "has_x": bool(r[7]),
           "proxy_available": bool(r[8]), "em": r[9:14].copy(), "em_x": r[14:19].copy(), "cost": r[19:27].copy(),
           "proxy": r[27:35].copy(), "gpu_ms": float(r[35])}
    return c, res


def gather_records(local: np.ndarray, counts: Sequence[int], device=None) -> np.ndarray:
    """The scan's only collective: all ranks exchange their fixed-size result records
    (``torch.distributed.all_gather`` — RCCL over xGMI on the GPU box, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    n_max = max(counts)
    buf = torch.zeros((n_max, RECORD_LEN), dtype=torch.float64, device=device)
    if len(local):
        buf[:len(local)] = torch.from_numpy(np.ascontiguousarray(local)).to(buf.device)
    parts = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, buf)
    return np.concatenate([p[:n].cpu().numpy() for p, n in zip(parts, counts)], axis=0)


class Evaluator:
    """Runs candidates on this rank's GPU, keeping dense tensors resident across candidates."""

    def __init__(self, model: ModelFn, params: ScanParams, tensors: Sequence[str] = KINDS, max_resident: int = 12,
                 dims_of: Optional[Callable[[int, str], Tuple[int, int]]] = None):
        self.model, self.params = model, params
        self.dims_of = dims_of          # (layer, kind) -> (n_in, n_out) without touching the weights
        self.tensors = tuple(t for t in KINDS if t in tensors)      # the builder's visiting order
        self._cache: Dict[Tuple[int, str], DenseWeight] = {}
        self._order: List[Tuple[int, str]] = []
        self.max_resident = max(3, max_resident)
        # one scan handle (stream, events, device buffers) per worker thread, reused across groups:
        # creating and freeing them per group costs a dozen hipMalloc / hipFree each, and hipFree waits
        # for the whole device
        import threading
        self._tls = threading.local()
        self._sessions: List[ScanSession] = []
        self._sessions_lock = threading.Lock()
        self._pool = None               # worker threads live as long as the evaluator: their handles stay warm
        self._pool_threads = 0

    def _workers(self, threads: int):
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, threads)
        if self._pool is None or self._pool_threads != threads:
            if self._pool is not None:
                self._pool.shutdown(wait=True)
            self._pool = ThreadPoolExecutor(max_workers=threads)
            self._pool_threads = threads
        return self._pool

    def _session(self, seed: int) -> ScanSession:
        s = getattr(self._tls, "session", None)
        if s is None:
            s = ScanSession(seed)
            self._tls.session = s
            with self._sessions_lock:
                self._sessions.append(s)
        s.reseed(seed)
        return s

    def run_many(self, cands: Sequence[Candidate], threads: int = 8, params_of: Optional[Callable] = None) -> List[dict]:
        """Evaluate candidates layer by layer with a pool of host threads. Every candidate owns its RNG
        stream, and drawing its samples (std::shuffle, ~26 ns per Gaussian) is the host-side cost of a
        candidate, so several are prepared concurrently while the GPU works through their kernels in
        stream order. Dense tensors are uploaded by the calling thread before the pool touches them."""
        out: Dict[int, dict] = {}
        layers: Dict[int, List[Candidate]] = {}
        for c in cands:
            layers.setdefault(c.layer, []).append(c)

        def needed(layer, group):
            proxy = any(c.kind != "ffn_down" for c in group)          # the FFN proxy reads all three tensors
            return [(layer, k) for k in KINDS if proxy or any(c.kind == k for c in group) or self.dims_of is None]

        def groups_of(group):                                         # one handle per (tensor, parameter set)
            gs: Dict[tuple, List[Candidate]] = {}
            for c in group:
                gs.setdefault((c.layer, c.kind, id(params_of(c)) if params_of else 0), []).append(c)
            return list(gs.values())

        fn = lambda g: self.run_group(g, params_of(g[0]) if params_of else None)      # noqa: E731
        want = [key for layer, group in layers.items() for key in needed(layer, group)]
        pool = self._workers(threads)
        if True:
            if len(set(want)) <= self.max_resident:                   # everything fits: one wave of work
                for key in want:
                    self.dense(*key)
                batches = [groups_of(list(cands))]
            else:                                                     # layer by layer, evicting as we go
                batches = []
                for layer, group in layers.items():
                    batches.append((layer, group))
            for batch in batches:
                if isinstance(batch, tuple):
                    layer, group = batch
                    for key in needed(layer, group):
                        self.dense(*key)
                    gs = groups_of(group)
                else:
                    gs = batch
                gs.sort(key=lambda g: -sum(1.0 + c.K / 1536.0 for c in g))       # longest first
                for g, rs in zip(gs, pool.map(fn, gs)):
                    for c, r in zip(g, rs):
                        out[id(c)] = r
        return [out[id(c)] for c in cands]

    def dense(self, layer: int, kind: str) -> DenseWeight:
        key = (layer, kind)
        if key not in self._cache:
            while len(self._order) >= self.max_resident:
                self._cache.pop(self._order.pop(0)).close()
            self._cache[key] = DenseWeight(self.model(layer, kind))
            self._order.append(key)
        return self._cache[key]

    def preload(self, keys: Sequence[Tuple[int, str]], threads: int = 8):
        """Materialise and upload several dense tensors concurrently (reading / dequantising a tensor and the
        host-to-device copy both release the GIL): what a rank does with ITS layers at the start of a sharded
        scan. Needs room for all of them (no eviction while the pool runs)."""
        todo = [k for k in dict.fromkeys(keys) if k not in self._cache]
        if len(self._order) + len(todo) > self.max_resident:
            for k in todo:                                              # not enough room: one by one, evicting
                self.dense(*k)
            return
        pool = self._workers(threads)
        for k, w in zip(todo, pool.map(lambda k: DenseWeight(self.model(*k)), todo)):
            self._cache[k] = w
            self._order.append(k)

    def run(self, c: Candidate, params: Optional[ScanParams] = None) -> dict:
        """One (layer, tensor, K) trial with the RNG position a serial per-layer builder run has when
        it reaches this tensor: fresh stream, earlier tensors of the layer skipped."""
        p = ScanParams(**{**(params or self.params).__dict__, "K": c.K})
        s = ScanSession(p.seed)
        try:
            for t in self.tensors:
                if t == c.kind:
                    break
                if self.dims_of is not None:
                    n_in, n_out = self.dims_of(c.layer, t)
                else:
                    w = self.dense(c.layer, t)
                    n_in, n_out = w.n_in, w.n_out
                s.skip(n_in, n_out, p, t, c.layer)
            need_proxy = p.scheme == "coo" and not p.base and p.eval_x > 0 and c.kind != "ffn_down"
            ffn = tuple(self.dense(c.layer, k) for k in KINDS) if need_proxy else None
            return s.tensor(self.dense(c.layer, c.kind), p, c.kind, c.layer, ffn)
        finally:
            s.close()

    def _position(self, s: "ScanSession", c: Candidate, p: ScanParams):
        """Fresh builder run on the handle, generator moved past the layer's earlier tensors."""
        s.reseed(p.seed)
        for t in self.tensors:
            if t == c.kind:
                break
            n_in, n_out = self.dims_of(c.layer, t) if self.dims_of is not None else \
                (self.dense(c.layer, t).n_in, self.dense(c.layer, t).n_out)
            s.skip(n_in, n_out, p, t, c.layer)

    def run_group(self, group: Sequence[Candidate], params: Optional[ScanParams] = None, kgrid: bool = True,
                  session: Optional["ScanSession"] = None) -> List[dict]:
        """All K of one (layer, tensor) on one handle. Every K is its own builder run (fresh RNG) with the same
        seed, so all of them see the same draws and the same base fit: the whole schedule goes to the device as
        ONE call — one fit, one selection at the largest K, every K's metric passes behind it
        (ScanSession.tensor_kgrid; sd_build.cpp:444-451, :843-913). kgrid=False evaluates the budgets one by one
        (fit reused, its draws replayed): the form the clamp needs (--delta-norm-clamp-down rescales the stored
        values per K) and the one the parity tests compare against. `session`: evaluate on this handle and leave
        the schedule's payload on it (build_run fetches from it)."""
        base = params or self.params
        s = session or self._session(base.seed)
        c0 = group[0]
        assert all((c.layer, c.kind) == (c0.layer, c0.kind) for c in group)
        need_proxy = base.scheme == "coo" and not base.base and base.eval_x > 0 and c0.kind != "ffn_down"
        ffn = tuple(self.dense(c0.layer, k) for k in KINDS) if need_proxy else None
        clamp = base.delta_norm_clamp_down > 0.0 and c0.kind == "ffn_down" and base.scheme == "coo"
        out = []
        if kgrid and not clamp:
            for i in range(0, len(group), MAX_KGRID):
                part = group[i:i + MAX_KGRID]
                p = ScanParams(**{**base.__dict__, "K": part[0].K})
                p.base_mode = 2 if (i > 0 and p.base) else 0
                self._position(s, c0, p)
                out += s.tensor_kgrid(self.dense(c0.layer, c0.kind), p, [c.K for c in part], c0.kind, c0.layer, ffn)
            return out
        for n, c in enumerate(group):
            p = ScanParams(**{**base.__dict__, "K": c.K})
            p.base_mode = 2 if (n > 0 and p.base) else 0            # never across groups: the handle is reused
            self._position(s, c, p)
            out.append(s.tensor(self.dense(c.layer, c.kind), p, c.kind, c.layer, ffn))
        return out

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        with self._sessions_lock:
            for s in self._sessions:
                s.close()
            self._sessions.clear()
        self._tls = __import__("threading").local()
        for w in self._cache.values():
            w.close()
        self._cache.clear()
        self._order.clear()


def layer_scan(model: ModelFn, layers: Sequence[int], params: ScanParams, tensors: Sequence[str] = KINDS,
               input_name: str = "<memory>", rounded: bool = True) -> dict:
    """scripts/seeddelta-layer-scan.py main loop (:173-286) with the builder subprocess replaced by
    the GPU pipeline. Returns the layer_sensitivity_scan.json document."""
    scan = {
        "version": 1, "input": input_name, "scheme": params.scheme,
        "block": params.block if params.scheme == "block" else 0, "K_fixed": params.K,
        "eval_cols": params.eval_cols, "eval_x": params.eval_x, "seed": params.seed,
        "base_used": bool(params.base), "row_scale": bool(params.row_scale), "strip_dense": False,
        "meta": {"scheme": params.scheme, "block": params.block if params.scheme == "block" else 0,
                 "K_fixed": params.K, "eval_cols": params.eval_cols, "eval_x": params.eval_x, "seed": params.seed,
                 "metric": params.metric, "tensors_scanned": list(tensors), "seed_delta_bin": "libseeddelta_hip.so",
                 "threads": 0, "base_used": bool(params.base), "row_scale": bool(params.row_scale),
                 "strip_dense": False, "imatrix": ""},
        "layers": {}, "errors": [],
    }
    # the scan reads report entries only (the reference discards the builder's output GGUF here), so every
    # candidate selects just the rows its metric passes sample: identical entries, far less work
    ev = Evaluator(model, ScanParams(**{**params.__dict__, "metrics_only": True}), tensors)
    try:
        for layer in layers:
            out = {}
            for kind in [t for t in KINDS if t in tensors]:
                res = ev.run(Candidate(layer, kind, params.K))
                w = ev.dense(layer, kind)
                e = report_entry(layer, kind, w.n_in, w.n_out, params, res)
                if rounded:
                    e = {k: round_like_report(v) for k, v in e.items()}
                out[kind] = collect_metrics(e)
            if not out:
                scan["errors"].append({"layer": layer, "error": "no weights matched"})
            scan["layers"][str(layer)] = out
    finally:
        ev.close()
    return scan


def dumps(doc: dict) -> str:
    return json.dumps(doc, indent=2) + "\n"


# ---- report.json (sd_report.cpp:112-288) -------------------------------------------------------------------
# Field order of one weights[] element as the builder writes it. Fields this pipeline does not
# produce (tiles, K selector, autotune attempts) take the builder's defaults for a one-K run.
_ENTRY_ORDER = (
    ["layer", "kind", "n_in", "n_out", "K_budget", "K", "block", "n_blocks", "has_w", "has_x"] + list(COST_NAMES)
    + [n + s for s in ("", "_w", "_x", "_x_w") for n in METRIC_NAMES]
    + ["gating_enabled", "gating_metric", "gating_value", "gating_p05", "gating_min_mean", "gating_min_p05", "gating_pass",
       "metric_used", "targets_used", "stack_cost_delta", "stack_cost_total", "ffn_score", "ffn_proxy_available",
       "ffn_proxy_reason", "ffn_proxy_scope", "ffn_proxy_base_used", "ffn_proxy_eval_x", "ffn_proxy_eval_out",
       "ffn_proxy_seed"] + list(PROXY_NAMES[:6])
    + ["emit", "strip_dense", "decision", "reject_reason", "tile_rows", "tile_rows_align", "k_total_per_tensor", "k_levels",
       "unique_k_count", "tiles_rounded_count", "tiles_rounded_pct", "tiles_dropped_count", "tiles_dropped_pct",
       "k_per_tile", "kselector_mode", "kselector_rank", "kselector_tiles_sampled", "kselector_gap_vs_uniform",
       "k_custom_used", "k_requested_stats", "k_selected_stats", "autotune_enabled", "autotune_selected_budget",
       "autotune_attempts"])


def complete_entry(e: dict) -> dict:
    """Fill the fields of a report entry the builder derives after the trial for a plain one-K run
    (single tile, K selector 'cycle', no autotune; sd_build.cpp:959-1021, :1060-1076)."""
    out = dict(e)
    k_stats = {"has": False, "min": 0, "max": 0, "mean": 0}
    if "K" not in e:
        # a tensor that never reached a trial (disabled, missing, no imatrix): the builder's entry keeps its
        # zero-initialised fields (include/sd_report.h:11-70) next to the decision
        int_keys = {"K_budget", "K", "block", "n_blocks", "base_L", "base_B", "tile_rows", "tile_rows_align", "k_total_per_tensor",
                    "unique_k_count", "tiles_rounded_count", "tiles_dropped_count", "kselector_rank", "kselector_tiles_sampled",
                    "ffn_proxy_eval_x", "ffn_proxy_eval_out", "ffn_proxy_seed", "autotune_selected_budget"}
        bool_keys = {"has_w", "has_x", "ffn_proxy_available", "ffn_proxy_base_used", "k_custom_used", "autotune_enabled", "strip_dense"}
        str_keys = {"ffn_proxy_reason", "ffn_proxy_scope", "kselector_mode"}
        list_keys = {"k_levels", "k_per_tile", "autotune_attempts"}
        for k in _ENTRY_ORDER:
            if k in out:
                continue
            out[k] = (0 if k in int_keys else False if k in bool_keys else "" if k in str_keys else [] if k in list_keys
                      else dict(k_stats) if k in ("k_requested_stats", "k_selected_stats") else 0.0)
        return {k: out[k] for k in _ENTRY_ORDER}
    cost_delta = 0.0
    if e.get("emit"):
        cost_delta = max(0.0, e["gating_min_mean"] - e["gating_value"]) + max(0.0, e["gating_min_p05"] - e["gating_p05"])
    defaults = {
        "stack_cost_delta": cost_delta, "stack_cost_total": cost_delta, "ffn_proxy_reason": "", "ffn_proxy_scope": "",
        "ffn_proxy_base_used": False, "ffn_proxy_eval_x": 0, "ffn_proxy_eval_out": 0, "ffn_proxy_seed": 0,
        **{n: 0.0 for n in PROXY_NAMES[:6]}, "strip_dense": False,
        "tile_rows": e["n_out"], "tile_rows_align": 1, "k_total_per_tensor": e["K"], "k_levels": [e["K"]],
        "unique_k_count": 1, "tiles_rounded_count": 0, "tiles_rounded_pct": 0.0, "tiles_dropped_count": 0,
        "tiles_dropped_pct": 0.0, "k_per_tile": [e["K"]], "kselector_mode": "", "kselector_rank": 1,
        "kselector_tiles_sampled": 1, "kselector_gap_vs_uniform": 0.0, "k_custom_used": False,
        "k_requested_stats": dict(k_stats), "k_selected_stats": dict(k_stats), "autotune_enabled": False,
        "autotune_selected_budget": e["K_budget"] if e.get("emit") else 0, "autotune_attempts": [],
    }
    for k, v in defaults.items():
        out.setdefault(k, v)
    return {k: out[k] for k in _ENTRY_ORDER}


def finalize_report_entry(e: dict) -> dict:
    """sd_report::finalize_report_entry (sd_report.cpp:46-59): the defaults an entry takes just before it is
    written — metric_used from the gating metric, the tau targets from the gating thresholds, reject_reason from
    the decision. Works on an entry in this module's field names (`decision` is the builder's decision_reason,
    written as "decision", sd_report.cpp:231)."""
    out = dict(e)
    if not out.get("metric_used"):
        out["metric_used"] = out.get("gating_metric", "")
    t = dict(out.get("targets_used") or {})
    if not t.get("tau_mean"):
        t["tau_mean"] = out.get("gating_min_mean", 0.0)
    if not t.get("tau_p05"):
        t["tau_p05"] = out.get("gating_min_p05", 0.0)
    out["targets_used"] = t
    if not out.get("reject_reason"):
        out["reject_reason"] = out.get("decision", "")
    return out


def report_document(entries: Sequence[dict], params: ScanParams, input_name: str = "", output_name: str = "",
                    policy_file: str = "", policy_hash: str = "", imatrix: Optional[dict] = None) -> dict:
    """The report.json document (sd_report.cpp:122-161 header + weights[]), numbers at 6 significant
    digits as the builder's ostream prints them. `imatrix`: {"file", "eps", "power", "datasets"} when the run had
    an importance matrix (the header fields of sd_report.cpp:133-143)."""
    doc = {"input": input_name, "output": output_name, "base": bool(params.base), "policy_file": policy_file,
           "policy_hash": policy_hash}
    if params.base:
        doc.update(base_kind="xor_circulant", base_max_samples=params.base_max_samples, base_perm_trials=params.base_perm_trials)
    if imatrix is not None:
        doc.update(imatrix_file=imatrix.get("file", ""), imatrix_eps=round_like_report(float(imatrix.get("eps", 1e-8))),
                   imatrix_power=round_like_report(float(imatrix.get("power", 1.0))),
                   imatrix_datasets=list(imatrix.get("datasets", [])))
    resid = {"scheme": params.scheme}
    if params.scheme == "block":
        resid["block"] = params.block
    resid.update(K=params.K, K_gate=params.K, K_up=params.K, K_down=params.K, idx_type="i16", val_type="f16",
                 row_scale=bool(params.row_scale))
    doc.update(resid=resid, eval_cols=params.eval_cols, eval_x=params.eval_x, stack_pass_gate_up=0, stack_pass_down=0)

    def rnd(v):
        if isinstance(v, dict):
            return {k: rnd(x) for k, x in v.items()}
        if isinstance(v, list):
            return [rnd(x) for x in v]
        return round_like_report(v)
    doc["weights"] = [rnd(complete_entry(e)) for e in entries]
    return doc


# ---- one builder run over layers and tensors (sd_build.cpp:245-1102) ---------------------------------------
class StackSafetyTracker:
    """include/sd_build.h:22-50: thresholds tighten as more tensors of a kind have been emitted."""

    def __init__(self):
        self.gate_up_pass = 0
        self.down_pass = 0

    def adjust(self, kind: str, min_mean: float, min_p05: float) -> Tuple[float, float]:
        f32 = np.float32
        down = kind == "ffn_down"
        passed = self.down_pass if down else self.gate_up_pass
        if passed >= 8:
            lo = (0.35, 0.25) if down else (0.60, 0.45)
            return float(max(f32(min_mean), f32(lo[0]))), float(max(f32(min_p05), f32(lo[1])))
        if passed >= 3:
            return float(f32(min_mean) + f32(0.05)), float(f32(min_p05) + f32(0.05))
        return min_mean, min_p05

    def record_pass(self, kind: str):
        if kind == "ffn_down":
            self.down_pass += 1
        else:
            self.gate_up_pass += 1


def build_run(model: ModelFn, layers: Sequence[int], params: ScanParams, policy=None, imatrix: Optional[Dict[str, np.ndarray]] = None,
              imatrix_eps: float = 1e-8, imatrix_power: float = 1.0, stack_cost_cap: float = math.inf, strip_dense: bool = False,
              have_tensor: Optional[Callable[[int, str], bool]] = None, keep_staging: bool = True,
              K_kind: Optional[Dict[str, int]] = None, skip: Optional[Callable[[int, str], bool]] = None,
              fetch_dtypes=(np.int16, np.float16)) -> dict:
    """What one `llama-seeddelta-build --layers ... [--policy p.json]` process decides, tensor by tensor,
    on ONE running RNG stream: resolve the policy, adjust thresholds by the stack guard, fit / trial /
    autotune on the GPU, FFN proxy, stack cost accounting and cap. Returns the report entries and, for
    emitted tensors, their staging arrays (ScanSession.fetch) ready for pack.pack_candidate.

    `policy` is a llama_cpp_amd.policy.Policy or None (no policy: everything enabled, no gating)."""
    from . import policy as PL
    if policy is not None and params.eval_cols <= 0:
        raise ValueError("--policy requires --eval-cols > 0 for gating/autotune")        # sd_build.cpp:287-292
    session = ScanSession(params.seed)
    guard = StackSafetyTracker()
    entries, emitted, cost_running = [], {}, 0.0
    cache: Dict[Tuple[int, str], DenseWeight] = {}

    def dense(layer, kind):
        if (layer, kind) not in cache:
            cache[(layer, kind)] = DenseWeight(model(layer, kind))
        return cache[(layer, kind)]

    try:
        for layer in layers:
            for kind in KINDS:
                kk = K_kind or {}                    # --K-gate / --K-up / --K-down override --K per kind (sd_build.cpp:247-252)
                base_cfg = PL.Resolved(block=params.block, K_gate=kk.get("ffn_gate", params.K), K_up=kk.get("ffn_up", params.K),
                                       K_down=kk.get("ffn_down", params.K), strip_dense=strip_dense,
                                       enabled=True, metric="cos_x_w", gating_enabled=policy is not None)
                base_cfg.min_mean, base_cfg.min_p05 = (0.25, 0.15) if kind == "ffn_down" else (0.45, 0.30)     # :304-311
                cfg = PL.resolve(policy, layer, kind, base_cfg)
                min_mean, min_p05 = cfg.min_mean, cfg.min_p05
                if cfg.gating_enabled:
                    min_mean, min_p05 = guard.adjust(kind, min_mean, min_p05)
                if cfg.require_eval_x and (params.eval_x <= 0 or params.eval_cols <= 0):
                    raise ValueError(f"gating metric requires --eval-x and --eval-cols > 0 for blk.{layer}.{kind}.weight")
                if cfg.require_imatrix and imatrix is None:
                    raise ValueError(f"gating metric requires imatrix for blk.{layer}.{kind}.weight")

                def stub(decision, n_in=0, n_out=0):
                    return {"layer": layer, "kind": kind, "n_in": n_in, "n_out": n_out, "emit": False, "strip_dense": False,
                            "gating_enabled": policy is not None, "gating_pass": False, "decision": decision,
                            "reject_reason": decision, "gating_metric": cfg.metric, "metric_used": cfg.metric,
                            "gating_min_mean": min_mean, "gating_min_p05": min_p05,
                            "targets_used": {"tau_mean": min_mean, "tau_p05": min_p05}}
                if have_tensor is not None and not have_tensor(layer, kind):
                    entries.append(stub("missing_tensor"))
                    continue
                if skip is not None and skip(layer, kind):     # already has delta tensors: no entry, no draws (sd_build.cpp:373-377)
                    continue
                W = dense(layer, kind)
                if not cfg.enabled:
                    entries.append(stub("disabled", W.n_in, W.n_out))
                    continue
                ws = None
                if imatrix is not None and f"blk.{layer}.{kind}.weight" in imatrix:
                    ws = imatrix_scale(imatrix[f"blk.{layer}.{kind}.weight"], W.n_in, imatrix_eps, imatrix_power)
                if cfg.require_imatrix and ws is None:
                    entries.append(stub("missing_imatrix", W.n_in, W.n_out))
                    continue
                K_budget = max(1, min(cfg.K_of(kind), W.n_in))
                p = ScanParams(**{**params.__dict__, "K": K_budget, "block": cfg.block if params.scheme == "block" else params.block,
                                  "metric": cfg.metric, "w_scale": ws, "base_mode": 0})
                schedule = []
                if cfg.autotune.enabled:
                    schedule = cfg.autotune.schedule_down if kind == "ffn_down" else cfg.autotune.schedule_gate_up
                schedule = list(schedule) or [K_budget]
                proxy = p.scheme == "coo" and not p.base and p.eval_x > 0 and kind != "ffn_down"
                ffn = tuple(dense(layer, k) for k in KINDS) if proxy else None
                e = autotune(session, W, p, schedule, kind, layer, ffn, min_mean, min_p05,
                             cfg.autotune.max_iters if cfg.autotune.enabled else 0, gating_enabled=cfg.gating_enabled)
                if not cfg.autotune.enabled:                      # a one-K run reports like the plain trial
                    e["autotune_enabled"], e["autotune_attempts"] = False, []
                    if not e["emit"]:
                        e["decision"] = e["reject_reason"] = e.pop("_trial_decision", e["decision"])
                e.pop("_trial_decision", None)
                e["strip_dense"] = bool(e["emit"] and cfg.strip_dense)
                delta = 0.0
                if e["emit"] and math.isfinite(e["gating_value"]) and math.isfinite(e["gating_p05"]):
                    delta = max(0.0, min_mean - e["gating_value"]) + max(0.0, min_p05 - e["gating_p05"])        # :1060-1076
                e["stack_cost_delta"] = delta
                e["stack_cost_total"] = cost_running + (delta if e["emit"] else 0.0)
                if e["emit"]:
                    cost_running = e["stack_cost_total"]
                    if cost_running > stack_cost_cap:                                                          # :1084-1092; the CLI default is +inf: no cap (sd_cli.cpp:58)
                        e.update(emit=False, strip_dense=False, decision="stack_cost_cap", reject_reason="stack_cost_cap")
                    else:
                        guard.record_pass(kind)
                        if keep_staging:
                            emitted[(layer, kind)] = (p, e["K"], session.fetch_packed(*fetch_dtypes))       # packed on the device
                entries.append(e)
    finally:
        session.close()
        for w in cache.values():
            w.close()
    return {"entries": entries, "emitted": emitted, "stack_pass_gate_up": guard.gate_up_pass, "stack_pass_down": guard.down_pass}
