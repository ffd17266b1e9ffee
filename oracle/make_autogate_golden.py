#!/usr/bin/env python3
"""Run the REFERENCE's scripts/seeddelta-autogate.py (stdlib-only Python, executed from /root/reference
in this container) on synthetic scan documents with a spread of flag sets and freeze what it wrote
(policy.autogen.json, layer_scores.json, forbidden_pairs.json, autogating_report.json) into
tests/golden/autogate_cases.json. Only data is stored: the scan inputs, the flags, the outputs.

The --eval cases use a stand-in for seeddelta-policy-eval.sh that derives the perplexity / greedy logs
from the policy it is handed by a fixed rule (documented in EVAL_RULE below and re-implemented by the
test's callback), so the script's rollback / forbidden-pair logic is exercised without a model.

Run in the build container:  python oracle/make_autogate_golden.py
"""
import json
import os
import random
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scripts/seeddelta-autogate.py"

EVAL_RULE = "base_ppl = 10; seeddelta_ppl = 10 * (1 + 0.011 * n_layers) + (0.35 if 9 in layers else 0); greedy FAIL iff 21 in layers"

FAKE_EVAL = r'''#!/usr/bin/env python3
import json, sys, os
a = sys.argv[1:]
opt = {a[i]: a[i + 1] for i in range(0, len(a) - 1, 2)}
pol = json.load(open(opt["--policy"]))
layers = [int(r["layers"]) for r in pol["ranges"]]
out = opt["--outdir"]; os.makedirs(out, exist_ok=True)
sd = 10.0 * (1 + 0.011 * len(layers)) + (0.35 if 9 in layers else 0.0)
open(os.path.join(out, "perplexity_base.log"), "w").write("noise\nFinal estimate: PPL = 10.0000 +/- 0.1\n")
open(os.path.join(out, "perplexity_seeddelta.log"), "w").write("Final estimate: PPL = %.4f +/- 0.1\n" % sd)
open(os.path.join(out, "greedy_pack.log"), "w").write("RESULT: %s\n" % ("FAIL" if 21 in layers else "PASS"))
'''


def synth_scan(seed, n_layers=32, nested=False, with_xw=False):
    rnd = random.Random(seed)
    layers = {}
    for layer in range(n_layers):
        per = {}
        depth = layer / max(1, n_layers - 1)
        for t in ("ffn_gate", "ffn_up", "ffn_down"):
            cos = round(min(0.99, 0.45 + 0.5 * depth + rnd.uniform(-0.12, 0.12)), 6)
            m = {"S_mean": round(max(0.02, 1.0 - cos + rnd.uniform(-0.05, 0.05)), 6), "S_p95": round(1.1 - cos, 6),
                 "cos_mean": cos, "cos_p05": round(cos - rnd.uniform(0.05, 0.25), 6),
                 "cos_mean_x_w": round(cos + 0.01, 6) if with_xw else 0.0,
                 "cos_p05_x_w": round(cos - 0.1, 6) if with_xw else 0.0,
                 "log_norm_ratio_mean": round(rnd.uniform(-0.3, 0.0), 6), "log_norm_ratio_p95": None,
                 "ffn_proxy_available": t != "ffn_down"}
            if t != "ffn_down":
                m.update({"ffn_proxy_cos_mean": round(min(0.999, cos + 0.1), 6), "ffn_proxy_cos_p05": round(min(0.99, cos + 0.02), 6),
                          "ffn_proxy_l2_mean": round(rnd.uniform(0.01, 0.3), 6), "ffn_proxy_l2_p95": round(rnd.uniform(0.05, 0.6), 6),
                          "ffn_proxy_log_norm_ratio_mean": round(rnd.uniform(-0.2, 0.1), 6),
                          "ffn_proxy_log_norm_ratio_p95": round(rnd.uniform(-0.3, 0.2), 6)})
            per[t] = m
        if layer == 5:
            del per["ffn_up"]
        if layer == 11:
            per["ffn_gate"]["cos_p05"] = None
        layers[str(layer)] = {"tensors": per} if (nested and layer % 3 == 0) else per
    layers["not_a_layer"] = {"ffn_gate": {"S_mean": 0.1}}
    return {"version": 1, "scheme": "coo", "block": 0, "K_fixed": 12, "eval_cols": 64, "eval_x": 32, "layers": layers,
            "meta": {"scheme": "coo", "K_fixed": 12, "eval_cols": 64, "eval_x": 32}}


# config fields (llama_cpp_amd.autogate.AutogateConfig) per case; "eval": use the stand-in evaluator
CASES = [
    ("default", 1, {}, {}),
    ("allow_up_enable_down", 1, {}, {"allow_up": True, "enable_down": True, "tensors": "ffn_gate,ffn_up,ffn_down"}),
    ("max_layers_gap2", 2, {}, {"max_layers": 6, "min_gap": 2, "prefer": "", "gate_cos_threshold": 0.6, "gate_cos_p05_threshold": 0.35}),
    ("percent_even", 3, {"nested": True}, {"percent": 25.0, "avoid_even_only": True, "min_gap": 0, "gate_cos_threshold": 0.5, "gate_cos_p05_threshold": 0.3}),
    ("rank_cos_threshold", 4, {"with_xw": True}, {"rank_metric": "cos_mean", "threshold": 0.7, "aggregate": "max", "allow_up": True,
                                                 "gate_cos_threshold_by": {"up": 0.8}, "gate_cos_p05_threshold_by": {"gate": 0.45}}),
    ("functional_calibrated", 5, {}, {"functional_gate": True, "functional_good_layers": "20-26", "functional_relax_cos": True,
                                       "functional_allow_early": True, "gate_cos_threshold": 0.55, "gate_cos_p05_threshold": 0.3}),
    ("functional_fixed_thresholds", 5, {}, {"functional_gate": True, "functional_cos_p05_threshold": 0.7, "functional_l2_threshold": 0.5,
                                             "functional_norm_threshold": 0.25, "functional_cos_p05_threshold_by": {"gate": 0.75},
                                             "gate_cos_threshold": 0.5, "gate_cos_p05_threshold": 0.3, "functional_allow_missing": True}),
    ("retention_proxy_topk", 6, {}, {"retention_mode": "proxy", "retention_topk": 5, "retention_min": 0.3, "retention_decay_alpha": 0.5,
                                      "allow_up": True, "gate_cos_threshold": 0.5, "gate_cos_p05_threshold": 0.3}),
    ("retention_auto_mean", 7, {"with_xw": True}, {"retention_mode": "auto", "retention_aggregate": "mean", "retention_decay_alpha": 0.25,
                                                   "enable_down": True, "tensors": "ffn_gate,ffn_down", "retention_allow_missing": True,
                                                   "gate_cos_threshold": 0.5, "gate_cos_p05_threshold": 0.3}),
    ("allow_deny_lists", 8, {}, {"layer_allowlist": "8-30", "layer_denylist": "12,13,20-22", "gate_cos_threshold": 0.5,
                                  "gate_cos_p05_threshold": 0.25, "gate_allow_missing": True, "policy_strip_dense": True, "k": 64, "metric": "cos_x"}),
    ("eval_rollback", 9, {}, {"gate_cos_threshold": 0.5, "gate_cos_p05_threshold": 0.25, "max_delta_ppl": 6.0, "min_gap": 1, "eval": True}),
    ("eval_tight", 10, {}, {"gate_cos_threshold": 0.45, "gate_cos_p05_threshold": 0.2, "max_delta_ppl": 2.5, "min_gap": 0, "prefer": "9,21",
                             "prefer_mult": 0.5, "eval": True}),
]


def cli_of(cfg):
    out = []
    for key, val in cfg.items():
        if key == "eval":
            continue
        if key.endswith("_by"):
            for sfx, v in val.items():
                out += ["--" + key[:-3].replace("_", "-") + "-" + sfx, str(v)]
        elif isinstance(val, bool):
            if val:
                out.append("--" + key.replace("_", "-"))
        else:
            out += ["--" + key.replace("_", "-"), str(val)]
    return out


def main():
    cases = []
    with tempfile.TemporaryDirectory() as td:
        fake = Path(td) / "fake_eval.py"
        fake.write_text(FAKE_EVAL)
        fake.chmod(0o755)
        (Path(td) / "text.txt").write_text("x")
        (Path(td) / "base.gguf").write_text("x")
        for name, seed, scan_kw, cfg in CASES:
            scan = synth_scan(seed, **scan_kw)
            cdir = Path(td) / name
            cdir.mkdir()
            (cdir / "scan.json").write_text(json.dumps(scan))
            cmd = [sys.executable, REF, "--scan", str(cdir / "scan.json"), "--outdir", str(cdir / "out")] + cli_of(cfg)
            if cfg.get("eval"):
                cmd += ["--eval", "--eval-script", str(fake), "--base", str(Path(td) / "base.gguf"), "--text", str(Path(td) / "text.txt")]
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, (name, r.stderr[-2000:])
            out = {}
            for f in ("policy.autogen.json", "layer_scores.json", "forbidden_pairs.json", "autogating_report.json"):
                p = cdir / "out" / f
                out[f] = json.loads(p.read_text()) if p.exists() else None
            rep = out["autogating_report.json"]
            if rep:
                for s in rep["steps"]:
                    s["run_dir"] = ""            # temp paths are not data
            cases.append({"name": name, "scan": scan, "config": cfg, "outputs": out})
            print(f"{name:28s} selected={[r['layers'] for r in out['policy.autogen.json']['ranges']]}")
    with open(os.path.join(ROOT, "tests", "golden", "autogate_cases.json"), "w") as fh:
        json.dump({"eval_rule": EVAL_RULE, "cases": cases}, fh)
    print("wrote tests/golden/autogate_cases.json")


if __name__ == "__main__":
    main()
