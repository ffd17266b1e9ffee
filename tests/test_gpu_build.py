"""The whole builder as one call (llama_cpp_amd.build.build_gguf = llama-seeddelta-build's runner + writer,
tools/seeddelta-build/sd_runner.cpp, sd_write.cpp) on a small synthetic model file."""
import json

import numpy as np
import pytest

import oracle_py as O
from conftest import bits
from llama_cpp_amd import build as B, gguf_io as G, scan as SC
from llama_cpp_amd.dense import make_dense

pytestmark = pytest.mark.gpu


def write_model(path, n_layer=3, n_embd=256, n_ff=704):
    tensors, q8 = {}, []
    tensors["token_embd.weight"] = np.random.default_rng(1).standard_normal((16, n_embd)).astype(np.float16)
    for l in range(n_layer):
        for kind in SC.KINDS:
            n_in, n_out = (n_ff, n_embd) if kind == "ffn_down" else (n_embd, n_ff)
            name = f"blk.{l}.{kind}.weight"
            tensors[name] = make_dense(n_in, n_out, "q8_0", seed=11 * l + SC.KIND_ID[kind], structure=0.5)
            q8.append(name)
        tensors[f"blk.{l}.attn_norm.weight"] = np.ones(n_embd, np.float32)
    # a tensor of a type this package only lists (Q4_K: 256 elements per 144 bytes), copied through as bytes
    tensors["blk.0.attn_q.weight"] = (12, (256, 4), bytes(range(144)) * 4)
    kv = {"general.architecture": "llama", "llama.block_count": n_layer, "general.name": "tiny",
          "tokenizer.ggml.scores": np.arange(5, dtype=np.float32), "llama.rope.dimension_sections": np.array([1, 2, 3], np.int32)}
    G.write_gguf(path, kv, tensors, q8_0=tuple(q8))
    return tensors


@pytest.mark.parametrize("scheme", ["coo", "block"])
def test_build_gguf_end_to_end(hip, tmp_path, scheme):
    src_path, out_path, rep_path = (str(tmp_path / n) for n in ("in.gguf", "out.gguf", "report.json"))
    src_tensors = write_model(src_path)
    p = SC.ScanParams(scheme=scheme, K=32, block=16, base=True, row_scale=True, base_max_samples=128, base_perm_trials=2,
                      eval_cols=16, eval_x=4)
    res = B.build_gguf(src_path, out_path, p, layers="1-2", K_down=64, report_json=rep_path)
    es = res["entries"]
    assert [(e["layer"], e["kind"]) for e in es] == [(l, k) for l in (1, 2) for k in SC.KINDS]
    assert all(e["emit"] for e in es) and [e["K_budget"] for e in es[:3]] == [32, 32, 64]

    with G.GGUFReader(src_path) as a, G.GGUFReader(out_path) as b:
        # every source pair and tensor arrives unchanged, in order, whatever its type
        assert list(b.kv)[:len(a.kv)] == list(a.kv)
        assert all(b.kv_raw(k) == a.kv_raw(k) for k in a.kv)
        assert b.names()[:len(a.names())] == a.names()
        for n in a.names():
            ta, tb = a.raw(n), b.raw(n)
            assert ta[0] == tb[0] and ta[1] == tb[1] and bytes(ta[2]) == bytes(tb[2]), n
        # the builder's keys (sd_write.cpp:44-74)
        kv = b.kv
        assert kv["seeddelta.enabled"] is True and kv["seeddelta.version"] == 1 and kv["seeddelta.scheme"] == (1 if scheme == "block" else 0)
        assert kv["seeddelta.resid.K"] == 32 and kv["seeddelta.resid.K_variable"] is True and kv["seeddelta.resid.K_down"] == 64
        assert kv["seeddelta.base.enabled"] is True and kv["seeddelta.base.perm_trials"] == 2 and kv["seeddelta.strip_dense"] is False
        assert ("seeddelta.resid.block" in kv) == (scheme == "block")
        assert kv["seeddelta.blk.1.ffn_down.enabled"] is True and kv["seeddelta.blk.1.ffn_down.K"] == 64
        # new tensors in the builder's order and types
        new = b.names()[len(a.names()):]
        pre = "b" if scheme == "block" else "d"
        assert new[:7] == [f"blk.1.ffn_gate.{s}" for s in (f"{pre}_idx", f"{pre}_val", "d_row_scale", "base_d1", "base_d2", "base_d3", "base_perm1")]
        assert b.infos[f"blk.1.ffn_gate.{pre}_idx"][1] == 25 and b.infos[f"blk.1.ffn_gate.{pre}_val"][1] == 1
        assert len(new) == res["n_added"] == 7 * 6

    # the output loads as a SeedDelta context and the op agrees with the oracle on the stored payload
    reader, payloads, meta = G.load_seeddelta_context(out_path)
    assert sorted(payloads) == sorted(f"blk.{l}.{k}.weight" for l in (1, 2) for k in SC.KINDS)
    ctx = hip.SeedDeltaContext(enabled=True)
    key = "blk.2.ffn_down.weight"
    assert ctx.try_add(key, payloads[key])
    x = np.random.default_rng(4).standard_normal((9, payloads[key].n_in)).astype(np.float32)
    assert np.array_equal(bits(ctx.mul_mat(key, x)), bits(O.forward(payloads[key], x, nth=2)))

    rep = json.load(open(rep_path))
    assert rep["input"] == src_path and rep["output"] == out_path and rep["resid"]["K_down"] == 64 and rep["resid"]["idx_type"] == "i16"
    assert len(rep["weights"]) == 6 and rep["stack_pass_gate_up"] == 4 and rep["stack_pass_down"] == 2

    # a second run on the output leaves tensors that already have delta tensors alone (sd_build.cpp:373-377)
    again = B.build_gguf(out_path, str(tmp_path / "again.gguf"), p, layers="0-2", K_down=64)
    assert [(e["layer"], e["kind"]) for e in again["entries"]] == [(0, k) for k in SC.KINDS]
    with G.GGUFReader(str(tmp_path / "again.gguf")) as c:
        assert len(c.names()) == len(set(c.names())) and "blk.0.ffn_up.base_d2" in c.infos and c.kv["seeddelta.version"] == 1


def test_build_gguf_policy_strip_and_hash(hip, tmp_path):
    src_path, out_path = str(tmp_path / "in.gguf"), str(tmp_path / "out.gguf")
    write_model(src_path, n_layer=2)
    pol = {"version": 1, "defaults": {"enabled": True, "strip_dense": True, "K": {"gate": 16, "up": 16, "down": 24},
                                      "gating": {"metric": "cos", "min_mean": {"gate": 0.0, "up": 0.0, "down": 2.0}, "min_p05": 0.0}},
           "ranges": [{"layers": "0", "enabled": False}]}
    pol_path = tmp_path / "policy.json"
    pol_path.write_text(json.dumps(pol))
    p = SC.ScanParams(scheme="coo", K=8, base=False, row_scale=True, eval_cols=16, eval_x=4)
    res = B.build_gguf(src_path, out_path, p, policy_file=str(pol_path), strip_dense=True)
    assert res["report"]["policy_hash"] == B.policy_hash(pol_path.read_bytes()) and res["report"]["policy_hash"].startswith("fnv1a64:")
    assert len(res["report"]["policy_hash"]) == len("fnv1a64:") + 16
    by = {(e["layer"], e["kind"]): e for e in res["entries"]}
    assert by[(0, "ffn_gate")]["decision"] == "disabled" and by[(1, "ffn_down")]["decision"] == "gating_fail"
    assert res["stripped"] == ["blk.1.ffn_gate.weight", "blk.1.ffn_up.weight"]
    with G.GGUFReader(out_path) as b:
        assert "blk.1.ffn_gate.weight" not in b.infos and "blk.1.ffn_down.weight" in b.infos and "blk.0.ffn_gate.weight" in b.infos
        assert b.kv["seeddelta.strip_dense"] is True and b.kv["seeddelta.blk.1.ffn_gate.strip_dense"] is True
        assert b.kv["seeddelta.blk.1.ffn_down.enabled"] is False and b.kv["seeddelta.blk.0.ffn_up.enabled"] is False
        assert "blk.1.ffn_gate.d_idx" in b.infos and "blk.1.ffn_down.d_idx" not in b.infos
