"""llama_cpp_amd.autogate against the outputs of the reference's scripts/seeddelta-autogate.py on the
same scan documents and flags (tests/golden/autogate_cases.json, made by
oracle/make_autogate_golden.py): policy.autogen.json, layer_scores.json, forbidden_pairs.json and
autogating_report.json must be identical documents."""
import json
import os

import pytest

from conftest import GOLDEN_DIR
from llama_cpp_amd import autogate as AG

with open(os.path.join(GOLDEN_DIR, "autogate_cases.json")) as fh:
    GOLD = json.load(fh)


def stand_in_eval(layers, tensors, policy):
    """The rule of the stand-in evaluator the goldens were made with (GOLD['eval_rule'])."""
    assert [int(r["layers"]) for r in policy["ranges"]] == layers
    sd = 10.0 * (1 + 0.011 * len(layers)) + (0.35 if 9 in layers else 0.0)
    return {"base_ppl": 10.0, "seeddelta_ppl": float("%.4f" % sd), "greedy_ok": 21 not in layers}


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_matches_reference_script(case):
    cfg_kw = {k: v for k, v in case["config"].items() if k != "eval"}
    cfg = AG.AutogateConfig()
    for k, v in cfg_kw.items():
        if k.endswith("_by"):
            getattr(cfg, k).update(v)
        else:
            setattr(cfg, k, v)
    res = AG.autogate(case["scan"], cfg, evaluate=stand_in_eval if case["config"].get("eval") else None)
    out = case["outputs"]
    assert res.policy == out["policy.autogen.json"]
    scores = {str(k): v for k, v in res.scores.items()}
    assert scores == out["layer_scores.json"] and list(scores) == list(out["layer_scores.json"])
    if out["forbidden_pairs.json"] is not None:
        assert res.forbidden_pairs == out["forbidden_pairs.json"]
    else:
        assert res.forbidden_pairs == []
    assert res.report() == out["autogating_report.json"]


def test_errors_and_edges():
    scan = GOLD["cases"][0]["scan"]
    with pytest.raises(AG.AutogateError):
        AG.autogate({"layers": []}, AG.AutogateConfig())
    with pytest.raises(AG.AutogateError):
        AG.autogate(scan, AG.AutogateConfig(functional_gate=True))                       # no thresholds resolvable
    with pytest.raises(AG.AutogateError):
        AG.autogate(scan, AG.AutogateConfig(rank_metric="no_such_metric"))               # no scores
    assert AG.metric_value({"a": 0.0, "b": 0.4}, "a", "b") == 0.4 and AG.metric_value({"a": 0.0, "b": 0.0}, "a", "b") == 0.0
    assert AG.metric_value({"b": 0.4}, "a", "b") == 0.4 and AG.metric_value({"a": 0.2, "b": 0.4}, "a", "b") == 0.2
    assert AG.validate_scan({"0": {}, "1": {"ffn_gate": 3}, "2": {"ffn_gate": {"cos_mean": 1}}}) == [
        "layer 0: missing tensors", "layer 1:ffn_gate invalid metrics", "layer 2:ffn_gate missing S_mean"]
    assert AG.parse_layers("5-3, 9") == [3, 4, 5, 9]
