"""Host logic of the scan (no GPU): the script mirrors against fixtures produced by the reference's own
scripts/seeddelta-layer-scan.py (tests/golden/layer_scan_script.json, made by
oracle/make_script_golden.py), report-entry gating, candidate sharding, and the one collective of the
multi-GPU scan on gloo with world_size 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from llama_cpp_amd import scan as SC


@pytest.fixture(scope="module")
def script_golden():
    with open(os.path.join(GOLDEN_DIR, "layer_scan_script.json")) as fh:
        return json.load(fh)


def test_parse_layers(script_golden):
    for case in script_golden["parse_layers"]:
        assert SC.parse_layers(case["spec"]) == case["layers"], case["spec"]


def test_collect_metrics_matches_reference_script(script_golden):
    for case in script_golden["collect_metrics"]:
        got = SC.collect_metrics(case["entry"])
        assert got == case["metrics"]
        assert list(got) == list(case["metrics"])          # same key order in the JSON document


def test_write_policy_matches_reference_script(script_golden):
    for case in script_golden["write_policy"]:
        assert SC.write_policy(case["layer"], case["tensors"], case["k"], case["metric"]) == case["policy"]


def fake_result(has_x=True, proxy=False, cos=(0.6, 0.4), cos_x=(0.7, 0.5)):
    return {"K_eff": 12, "n_blocks": 1, "L": 0, "B": 0, "has_x": has_x, "proxy_available": proxy,
            "em": np.array([0.8, 0.9, cos[0], cos[1], 0.6]), "em_x": np.array([0.7, 0.8, cos_x[0], cos_x[1], 0.7]),
            "cost": np.array([0, 0, 100.0, 0, 10.0, 0, 10.0, 0.1]), "proxy": np.arange(8.0), "gpu_ms": 1.0}


def test_report_entry_gating_rules():
    p = SC.ScanParams()
    # gate/up with x metrics gate on ffn_score = min(cos_mean_x, cos_p05_x) (sd_build.cpp:810-816, sd_report.cpp:72-88)
    e = SC.report_entry(3, "ffn_gate", 256, 704, p, fake_result())
    assert e["gating_metric"] == "ffn_score" and e["ffn_score"] == 0.5 and e["gating_pass"] and e["decision"] == "pass_gating"
    # down keeps the policy metric
    e = SC.report_entry(3, "ffn_down", 704, 256, p, fake_result())
    assert e["gating_metric"] == "cos" and e["gating_value"] == 0.6 and e["gating_p05"] == 0.4
    assert e["ffn_proxy_available"] is False and e["ffn_proxy_reason"] == "proxy_kind_not_supported"
    # thresholds
    e = SC.report_entry(3, "ffn_down", 704, 256, p, fake_result(), min_mean=0.7)
    assert not e["emit"] and e["decision"] == "gating_fail"
    # metric needing x without x
    q = SC.ScanParams(metric="cos_x", eval_x=0)
    e = SC.report_entry(3, "ffn_down", 704, 256, q, fake_result(has_x=False))
    assert e["decision"] == "metric_unavailable" and not e["gating_pass"] and e["ffn_proxy_reason"] == "proxy_requires_eval_x"
    e = SC.report_entry(3, "ffn_gate", 256, 704, SC.ScanParams(scheme="block"), fake_result())
    assert e["ffn_proxy_reason"] == "proxy_requires_coo"
    e = SC.report_entry(3, "ffn_up", 256, 704, p, fake_result(proxy=True))
    assert e["ffn_proxy_seed"] == 1234 + 303 + 2 and e["ffn_proxy_eval_x"] == 6 and e["ffn_proxy_cos_p05"] == 1.0
    assert SC.round_like_report(0.123456789) == 0.123457 and SC.round_like_report(True) is True


def test_shard_is_a_balanced_partition():
    grid = {"ffn_gate": [128, 256, 384], "ffn_up": [128, 256, 384], "ffn_down": [256, 512, 1536]}
    cands = SC.enumerate_candidates(range(32), SC.KINDS, grid)
    assert len(cands) == 32 * 9
    cost = lambda c: 1.0 + c.K / 1536.0        # noqa: E731
    for world in (1, 2, 4, 8):
        parts = [SC.shard(cands, r, world, cost) for r in range(world)]
        keys = sorted(k for p in parts for k in (c.key() for c in p))
        assert keys == sorted(c.key() for c in cands)                        # a partition: nothing lost or doubled
        loads = [sum(cost(c) for c in p) for p in parts]
        assert max(loads) - min(loads) <= 2.0                                # within one candidate of each other
    assert SC.shard(cands, 1, 4, cost) == SC.shard(list(reversed(cands)), 1, 4, cost)   # order-independent


def test_shard_groups_keeps_a_k_schedule_on_one_rank():
    """The deal bench.py's scan leg uses (SC.shard_groups): a partition of the candidates in which every K of a
    (layer, tensor) lands on ONE rank (its dense weight is uploaded once, fit / dequant / sort shared by the
    schedule), the same on every rank without communication, balanced to within one group, and with `counts`
    equal to what each rank really gets. Candidate independence: scripts/seeddelta-layer-scan.py:206-262."""
    grid = {"ffn_gate": [128, 256, 384, 512, 640, 768], "ffn_up": [128, 256, 384, 512, 640, 768],
            "ffn_down": [256, 512, 768, 1024, 1280, 1536]}
    cands = SC.enumerate_candidates(range(32), SC.KINDS, grid)
    cost = lambda c: 1.0 + c.K / 1536.0        # noqa: E731
    for world in (1, 2, 3, 4, 8):
        deals = [SC.shard_groups(cands, r, world, cost) for r in range(world)]
        parts = [d[0] for d in deals]
        assert all(d[1] == deals[0][1] for d in deals)                       # every rank computes the same counts
        assert [len(p) for p in parts] == deals[0][1]
        keys = sorted(c.key() for p in parts for c in p)
        assert keys == sorted(c.key() for c in cands)                        # nothing lost, nothing doubled
        owner = {}
        for r, p in enumerate(parts):
            for c in p:
                assert owner.setdefault((c.layer, c.kind), r) == r           # one rank per (layer, tensor)
            assert p == [c for c in cands if owner.get((c.layer, c.kind)) == r]   # serial visiting order kept
        loads = [sum(cost(c) for c in p) for p in parts]
        heaviest = max(sum(cost(c) for c in cands if (c.layer, c.kind) == g) for g in owner)
        assert max(loads) - min(loads) <= heaviest + 1e-9
    a = SC.shard_groups(cands, 1, 4, cost)
    assert a == SC.shard_groups(list(cands), 1, 4, cost)
    # a variant tag in the group key (bench.py deals the two parameter sets of a tensor to one rank as well)
    mine, counts = SC.shard_groups(cands, 0, 2, cost, group_key=lambda c: (c.layer,))
    assert sum(counts) == len(cands) and len({c.layer for c in mine}) == 16


def test_record_round_trip():
    c = SC.Candidate(17, "ffn_down", 1536)
    res = fake_result(proxy=True)
    c2, r2 = SC.from_record(SC.to_record(c, res))
    assert c2 == c
    for k in ("em", "em_x", "cost", "proxy"):
        assert np.array_equal(res[k], r2[k])
    assert r2["has_x"] and r2["proxy_available"] and r2["K_eff"] == 12


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
import llama_cpp_amd
from llama_cpp_amd import scan as SC
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
grid = {"ffn_gate": [128, 256], "ffn_up": [128], "ffn_down": [256, 512]}
cands = SC.enumerate_candidates(range(3), SC.KINDS, grid)
cost = lambda c: 1.0 + c.K / 512.0
mine, counts = SC.shard_groups(cands, rank, world, cost)          # the deal bench.py's scan leg uses
assert counts == SC.shard_groups(cands, 1 - rank, world, cost)[1] and counts[rank] == len(mine)
owners = {}
for r in range(world):
    for c in SC.shard_groups(cands, r, world, cost)[0]:
        assert owners.setdefault((c.layer, c.kind), r) == r       # all K of a (layer, tensor) on one rank
assert len(owners) == 9 and sum(counts) == len(cands)
def fake(c):
    v = float(c.layer * 1000 + SC.KIND_ID[c.kind] * 100) + c.K / 1000.0
    return {"K_eff": c.K, "n_blocks": 1, "L": 0, "B": 0, "has_x": True, "proxy_available": False,
            "em": np.full(5, v), "em_x": np.full(5, v + 0.5), "cost": np.zeros(8), "proxy": np.zeros(8), "gpu_ms": 1.0}
local = np.stack([SC.to_record(c, fake(c)) for c in mine]) if mine else np.zeros((0, SC.RECORD_LEN))
allr = SC.gather_records(local, counts)
assert sorted(tuple(r[:3]) for r in allr) == sorted(tuple(float(v) for v in c.key()) for c in cands)   # gathered set == serial set
got = {SC.from_record(r)[0].key(): SC.from_record(r)[1]["em"][0] for r in allr}
want = {c.key(): fake(c)["em"][0] for c in cands}
assert got == want, (rank, len(got), len(want))
dist.barrier()
if rank == 0:
    print("GATHER_OK", len(allr))
dist.destroy_process_group()
"""


def test_gather_records_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GATHER_OK 15" in out.stdout


def test_report_document_has_the_builder_fields(script_golden):
    """Field list and order of sd_report.cpp:163-281 (parity unpinned: sd_report.cpp includes the policy
    header GCC 11 rejects, so the reference writer cannot be run here); the scan script consumes it."""
    p = SC.ScanParams(K=12)
    entries = [SC.report_entry(3, k, 256, 704, p, fake_result(proxy=(k != "ffn_down"))) for k in SC.KINDS]
    doc = SC.report_document(entries, p, input_name="base.gguf")
    assert list(doc)[:5] == ["input", "output", "base", "policy_file", "policy_hash"] and "base_kind" not in doc
    w = doc["weights"][0]
    keys = list(w)
    assert keys == script_golden["report_entry_fields"]          # names and order as the reference source writes them
    assert keys.index("rel_l2_mean") < keys.index("rel_l2_mean_w") < keys.index("rel_l2_mean_x") < keys.index("rel_l2_mean_x_w")
    assert w["k_levels"] == [12] and w["targets_used"] == {"tau_mean": 0.0, "tau_p05": 0.0}
    # the reference's layer-scan script logic consumes the document unchanged
    for e in doc["weights"]:
        m = SC.collect_metrics(e)
        assert m["S_mean"] == 0.8 and m["ffn_proxy_available"] == (e["kind"] != "ffn_down")
    assert json.loads(SC.dumps(doc))["resid"]["K_down"] == 12
    assert SC.report_document(entries, SC.ScanParams(base=True))["base_kind"] == "xor_circulant"


def test_report_serialization_as_the_reference_selftest():
    """The reference's own expectations for the report writer: test_report_serialization
    (tools/seeddelta-build/selftest/sd_selftest.cpp:52-118) builds ONE entry (layer 3, ffn_gate, has_w, gating
    metric cos_x 0.7 / 0.4, decision pass_gating, emit, metric_used cleared), finalises it, writes it with a
    base + imatrix + COO K=32 header and looks for three substrings in the file. Same entry, same header, same
    substrings here — plus what finalize_report_entry (sd_report.cpp:46-59) must have filled in."""
    entry = {"layer": 3, "kind": "ffn_gate", "has_w": True, "gating_metric": "cos_x", "gating_value": 0.7, "gating_p05": 0.4,
             "decision": "pass_gating", "emit": True, "metric_used": ""}       # sd_constants.h:10
    entry = SC.finalize_report_entry(entry)
    assert entry["metric_used"] == "cos_x" and entry["reject_reason"] == "pass_gating"
    assert entry["targets_used"] == {"tau_mean": 0.0, "tau_p05": 0.0}
    cfg = SC.ScanParams(scheme="coo", K=32, base=True, row_scale=True, base_max_samples=1, base_perm_trials=1, eval_cols=1, eval_x=1)
    doc = SC.report_document([entry], cfg, input_name="in.gguf", output_name="out.gguf", policy_file="policy.json",
                             policy_hash="hash", imatrix={"file": "matrix.gguf", "eps": 1e-8, "power": 1.0, "datasets": ["dataset"]})
    contents = SC.dumps(doc)
    assert '"layer": 3' in contents                     # sd_selftest.cpp:107
    assert '"kind": "ffn_gate"' in contents             # :111
    assert "pass_gating" in contents                    # :115
    # the header the writer emits for that config, in the writer's order (sd_report.cpp:122-160)
    assert list(doc)[:15] == ["input", "output", "base", "policy_file", "policy_hash", "base_kind", "base_max_samples",
                              "base_perm_trials", "imatrix_file", "imatrix_eps", "imatrix_power", "imatrix_datasets", "resid",
                              "eval_cols", "eval_x"]
    back = json.loads(contents)
    assert back["imatrix_datasets"] == ["dataset"] and back["imatrix_eps"] == 1e-8 and back["resid"]["K_gate"] == 32
    w = back["weights"][0]
    assert w["decision"] == "pass_gating" and w["emit"] is True and w["has_w"] is True and w["gating_value"] == 0.7
    assert w["n_in"] == 0 and w["K"] == 0 and w["autotune_attempts"] == []     # zero-initialised fields of an entry that ran no trial


@pytest.mark.parametrize("seed,skip,n", [(1234, 0, 1), (1234, 3, 2), (7, 11, 7), (1234, 0, 32 * 4096), (99, 623, 100001), (5, 624, 32 * 14336)])
def test_fast_normals_equal_the_standard_library(seed, skip, n):
    """csrc/sdh_rng.h writes out libstdc++'s normal_distribution<float> over mt19937 (what eval_x and the FFN
    proxy draw, sd_eval.cpp:1084-1087, sd_ffn_proxy.cpp:77-80); the library's test hook draws the same n values
    both ways: identical bits, identical generator state afterwards. No GPU involved."""
    import ctypes as C
    from llama_cpp_amd import seeddelta as sd
    lib = sd.load_library()
    lib.sdh_test_normals.argtypes = [C.c_uint32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sdh_test_normals.restype = C.c_int
    a, b = np.empty(n, np.float32), np.empty(n, np.float32)
    assert lib.sdh_test_normals(seed, skip, n, a.ctypes.data, b.ctypes.data) == 1
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert abs(float(a.mean())) < 5.0 / np.sqrt(n) + (1.0 if n < 100 else 0.0) * 3
