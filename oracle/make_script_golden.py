#!/usr/bin/env python3
"""Freeze input/output pairs of the reference's scan script functions
(scripts/seeddelta-layer-scan.py: parse_layers, write_policy, collect_metrics) into
tests/golden/layer_scan_script.json. The script is stdlib-only Python and is imported from
/root/reference in this container; only data (inputs and the outputs it produced) is stored.

Run in the build container:  python oracle/make_script_golden.py
"""
import importlib.util
import json
import os
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scripts/seeddelta-layer-scan.py"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ls = load(REF, "ref_layer_scan")
    entries = [
        {"layer": 3, "kind": "ffn_gate", "rel_l2_mean": 0.812345, "rel_l2_p95": 0.9, "cos_mean": 0.58, "cos_p05": 0.41,
         "cos_mean_x_w": 0.0, "cos_p05_x_w": 0.0, "norm_ratio_mean": 0.61, "ffn_proxy_available": True,
         "ffn_proxy_cos_mean": 0.93, "ffn_proxy_cos_p05": 0.88, "ffn_proxy_l2_mean": 0.02, "ffn_proxy_l2_p95": 0.05,
         "ffn_proxy_log_norm_ratio_mean": -0.11, "ffn_proxy_log_norm_ratio_p95": -0.2},
        {"layer": 3, "kind": "ffn_down", "rel_l2_mean": 0.95, "rel_l2_p95": 0.97, "cos_mean": 0.3, "cos_p05": 0.2,
         "norm_ratio_mean": 0.0, "ffn_proxy_available": False},
        {"layer": 4, "kind": "ffn_up", "rel_l2_mean": "0.5", "rel_l2_p95": None, "cos_mean": 0.7, "cos_p05": "bad",
         "cos_mean_x_w": 0.66, "cos_p05_x_w": 0.5, "norm_ratio_mean": 1.25, "ffn_proxy_available": True,
         "ffn_proxy_cos_p05": 0.5, "ffn_proxy_l2_p95": 0.7, "ffn_proxy_log_norm_ratio_p95": 0.1},
        {},
    ]
    policies = []
    with tempfile.TemporaryDirectory() as td:
        for layer, tensors, k, metric in ((7, ["ffn_gate", "ffn_up", "ffn_down"], 12, "cos"), (0, ["ffn_gate"], 128, "cos_x")):
            path = Path(td) / "p.json"
            ls.write_policy(path, layer, tensors, k, metric)
            policies.append({"layer": layer, "tensors": tensors, "k": k, "metric": metric,
                             "policy": json.loads(path.read_text())})
    # field names of one report.json weights[] element, in the order sd_report.cpp:166-279 writes them
    import re
    src = open("/root/reference/tools/seeddelta-build/sd_report.cpp").read()
    body = src[src.index("for (size_t i = 0; i < report.size(); ++i)"):]
    fields = re.findall(r'out << "      \\"([A-Za-z0-9_]+)\\"', body)
    out = {
        "report_entry_fields": fields,
        "parse_layers": [{"spec": s, "layers": ls.parse_layers(s)} for s in ("0-31", "0,2,4", "5-3, 9", "", "7")],
        "collect_metrics": [{"entry": e, "metrics": ls.collect_metrics(e)} for e in entries],
        "write_policy": policies,
    }
    with open(os.path.join(ROOT, "tests", "golden", "layer_scan_script.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote tests/golden/layer_scan_script.json")


if __name__ == "__main__":
    main()
