"""llama_cpp_amd.policy against the expectations of the reference's own policy self-test
(tools/seeddelta-build/seeddelta_policy_selftest.cpp:66-220: the same documents, the same asserted
values) and against the policy files the reference ships / its scripts emit."""
import json
import os

import pytest

from llama_cpp_amd import policy as P
from llama_cpp_amd import scan as SC


def test_layer_selector():                       # selftest :69-74
    assert P.parse_layer_selector("3-1,2,2,5") == [1, 2, 3, 5]
    assert P.parse_layer_selector(" 8-20, 24 ") == list(range(8, 21)) + [24]


def test_unknown_keys_lenient_vs_strict():       # selftest :76-103
    text = '{"version": 1, "defaults": {"enabled": true, "unknown_key": 123}}'
    pol = P.load_policy(text, strict=False)
    assert pol.warnings
    with pytest.raises(P.PolicyError):
        P.load_policy(text, strict=True)
    with pytest.raises(P.PolicyError):
        P.load_policy('{"version": 1, "extra": {}}', strict=True)


MERGE_DOC = """
{
  "version": 1,
  "defaults": {"enabled": false, "strip_dense": false, "block": 64, "K": {"gate": 100, "up": 200, "down": 300}},
  "ranges": [
    {"layers": "1-3", "enabled": true, "K": {"gate": 111}},
    {"layers": "2-4", "strip_dense": true, "K": {"gate": 222}}
  ],
  "layers": {
    "2": {"K": {"gate": 333}, "tensors": {"ffn_gate": {"K": 444}, "ffn_up": {"enabled": false}}},
    "3": {"enabled": false}
  }
}
"""


def test_merge_precedence():                     # selftest :105-186
    pol = P.load_policy(MERGE_DOC, strict=True)
    base = P.Resolved(enabled=True, strip_dense=False, block=11, K_gate=1, K_up=2, K_down=3, metric="cos_x_w")
    r = P.resolve(pol, 0, "ffn_gate", base)
    assert (r.enabled, r.block, r.K_gate) == (False, 64, 100)
    r = P.resolve(pol, 1, "ffn_gate", base)
    assert (r.enabled, r.K_gate, r.strip_dense) == (True, 111, False)
    r = P.resolve(pol, 2, "ffn_gate", base)
    assert (r.enabled, r.strip_dense, r.K_gate, r.block) == (True, True, 444, 64)
    assert P.resolve(pol, 2, "ffn_up", base).enabled is False
    r = P.resolve(pol, 3, "ffn_gate", base)
    assert (r.enabled, r.K_gate) == (False, 222)
    assert base.K_gate == 1                                   # the baseline is not modified


def test_metric_requirements():                  # selftest :188-214
    pol = P.load_policy('{"version": 1, "defaults": {"enabled": true, "gating": {"metric": "cos_x_w", "min_mean": 0.1, "min_p05": 0.1}}}',
                        strict=True)
    r = P.resolve(pol, 0, "ffn_gate", P.Resolved(metric="cos"))
    assert (r.metric, r.require_eval_x, r.require_imatrix, r.gating_enabled) == ("cos_x_w", True, True, True)
    r = P.resolve(None, 0, "ffn_gate", P.Resolved(min_mean=0.5))
    assert r.gating_enabled is False and r.min_mean == 0.5


def test_scan_and_shipped_policies_resolve():
    """The per-layer policy the scan script writes (layer-scan.py:31-59) and a policy of the shape the
    reference ships (policies/policy.mistral7b.base.json: defaults + ranges with per-kind thresholds
    and autotune schedules)."""
    pol = P.load_policy(SC.write_policy(7, ["ffn_gate", "ffn_down"], 12, "cos"), strict=True)
    base = P.Resolved(K_gate=32, K_up=32, K_down=32, metric="cos_x_w", min_mean=0.45, min_p05=0.30)
    r = P.resolve(pol, 7, "ffn_gate", base)
    assert (r.enabled, r.K_gate, r.metric, r.min_mean, r.gating_enabled, r.autotune.enabled) == (True, 12, "cos", 0.0, True, False)
    assert P.resolve(pol, 7, "ffn_up", base).enabled is False
    r = P.resolve(pol, 6, "ffn_gate", base)
    assert r.enabled is False and r.gating_enabled is False and r.min_mean == -1.0
    shipped = {
        "version": 1,
        "defaults": {"enabled": True, "strip_dense": True, "block": 32, "K": {"gate": 256, "up": 256, "down": 512},
                     "gating": {"metric": "cos_x_w", "min_mean": {"gate": 0.5, "up": 0.5, "down": 0.3},
                                "min_p05": {"gate": 0.35, "up": 0.35, "down": 0.2}},
                     "autotune": {"enabled": True, "schedule_gate_up": [128, 256, 384], "schedule_down": [256, 512], "max_iters": 5}},
        "ranges": [{"layers": "0-5", "enabled": False},
                   {"layers": "24-31", "K": {"gate": 384, "up": 384, "down": 640},
                    "autotune": {"enabled": True, "schedule_gate_up": [256, 384, 512, 768], "schedule_down": [512, 1536], "max_iters": 5}}]}
    pol = P.load_policy(json.dumps(shipped), strict=True)
    r = P.resolve(pol, 30, "ffn_down", P.Resolved())
    assert (r.K_down, r.min_mean, r.min_p05, r.autotune.schedule_down, r.require_imatrix) == (640, 0.3, 0.2, [512, 1536], True)
    assert P.resolve(pol, 3, "ffn_gate", P.Resolved()).enabled is False
    assert P.resolve(pol, 10, "ffn_up", P.Resolved()).autotune.schedule_gate_up == [128, 256, 384]
