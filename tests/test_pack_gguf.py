"""Packing (builder rules, tools/seeddelta-build/sd_build.cpp:1104-1236), tensor naming / metadata
(sd_write.cpp:44-74) and the GGUF reader / writer, on the CPU.

Fixtures: tests/golden/gguf_toy.gguf was written by the reference's own gguf-py (see
oracle/make_gguf_golden.py); gguf_toy_expect.npz holds the arrays that went in."""
import os

import numpy as np
import pytest

import oracle_py as O
from conftest import GOLDEN_DIR
from llama_cpp_amd import gguf_io as G
from llama_cpp_amd import pack as P
from llama_cpp_amd.dense import make_dense
from llama_cpp_amd.payload import make_payload


def test_fp16_packing_is_ggml_rounding():
    """numpy's float32 -> float16 is the RNE conversion ggml_fp32_to_fp16 performs (pinned through the
    oracle's sdo_fp32_to_fp16, itself pinned to the reference in test_oracle.py)."""
    lib = O.oracle_lib()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-8, 1e-5, 0.02, 1.0, 300.0, 7e4)]
                          + [np.array([0.0, -0.0, 65504.0, 65520.0, 1e9, 2 ** -24, 2 ** -25, 1.5 * 2 ** -24, 0.1], np.float32)])
    got = P.pack_f16(vals).view(np.uint16)
    want = np.array([lib.sdo_fp32_to_fp16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_index_and_perm_packing_rules():
    idx = np.array([[0, 5, 32767, 32768, -1, -32768, -40000, 70000]], dtype=np.int32)
    assert P.pack_indices(idx).tolist() == [[0, 5, 32767, -1, -1, -32768, -1, -1]]
    assert P.pack_indices(idx, np.int32).dtype == np.int32 and np.array_equal(P.pack_indices(idx, np.int32), idx)
    perm = np.array([[3, -2, 40000, 7]], dtype=np.int32)
    assert P.pack_perm(perm, 4096).tolist() == [[3, 0, 0, 7]] and P.pack_perm(perm, 4096).dtype == np.int16
    assert P.pack_perm(perm, 65536).dtype == np.int32


def test_pack_candidate_builds_the_builder_tensors():
    rng = np.random.default_rng(1)
    st = {"idx": rng.integers(0, 64, (10, 6)).astype(np.int32), "val": rng.standard_normal((10, 6)).astype(np.float32),
          "row_scale": np.ones(10, np.float32), "d2": rng.standard_normal((1, 64)).astype(np.float32),
          "perm1": np.tile(np.arange(64, dtype=np.int32), (1, 1))}
    p = P.pack_candidate(st, 64, 10, "coo")
    assert p.kind == 2 and p.d_idx.dtype == np.int16 and p.d_val.dtype == np.float16 and p.row_scale.dtype == np.float16
    assert np.all(p.base_d1 == 1) and np.all(p.base_d3 == 1) and p.base_perm2 is None and p.base_perm1.dtype == np.int16
    names = P.payload_tensors(23, "ffn_down", p)
    assert set(names) == {"blk.23.ffn_down." + s for s in ("d_idx", "d_val", "d_row_scale", "base_d1", "base_d2", "base_d3", "base_perm1")}
    st["val"] = rng.standard_normal((10, 6, 16)).astype(np.float32)
    q = P.pack_candidate(st, 64, 10, "block", val_dtype=np.float32)
    assert q.kind == 3 and q.b_val.dtype == np.float32 and q.b_val.shape == (10, 6, 16)


def test_metadata_keys():
    kv = P.metadata("block", 128, 16, True, True, base_perm_trials=4,
                    entries=[{"layer": 23, "kind": "ffn_down", "emit": True, "gating_pass": True, "K": 128}])
    assert kv["seeddelta.scheme"] == 1 and kv["seeddelta.resid.block"] == 16 and kv["seeddelta.base.perm_trials"] == 4
    assert kv["seeddelta.blk.23.ffn_down.K"] == 128 and kv["seeddelta.blk.23.ffn_down.enabled"] is True
    assert "seeddelta.resid.block" not in P.metadata("coo", 12, 32, False, False)
    assert P.metadata("coo", 64, 32, False, False, K_down=128)["seeddelta.resid.K_down"] == 128


def test_reads_gguf_written_by_reference_gguf_py():
    path = os.path.join(GOLDEN_DIR, "gguf_toy.gguf")
    expect = {k.replace("__", "."): v for k, v in np.load(os.path.join(GOLDEN_DIR, "gguf_toy_expect.npz")).items()}
    with G.GGUFReader(path) as r:
        assert r.version == 3 and r.kv["general.architecture"] == "llama"
        assert r.kv["seeddelta.enabled"] is True and r.kv["seeddelta.resid.K"] == 8 and r.kv["seeddelta.blk.2.ffn_gate.K"] == 8
        assert set(r.names()) == set(expect)
        for name, arr in expect.items():
            got = r.tensor(name)
            assert got.dtype == arr.dtype and got.shape == arr.shape, name
            assert np.array_equal(got, arr), name
        assert r.infos["blk.2.ffn_gate.weight"][:2] == ((64, 160), 8)          # ne = [n_in, n_out], Q8_0
    r, payloads, meta = G.load_seeddelta_context(path)
    try:
        assert set(payloads) == {"blk.2.ffn_gate.weight", "blk.2.ffn_down.weight"}
        g, d = payloads["blk.2.ffn_gate.weight"], payloads["blk.2.ffn_down.weight"]
        assert (g.n_in, g.n_out, g.kind) == (64, 160, 2) and (d.n_in, d.n_out, d.kind) == (160, 64, 3)
        assert meta["seeddelta.base.enabled"] is True
    finally:
        r.close()


def test_gguf_round_trip(tmp_path):
    p = make_payload(96, 40, scheme="block", K=32, block=16, base=True, seed=2)
    tensors = P.payload_tensors(7, "ffn_up", p)
    tensors["blk.7.ffn_up.weight"] = make_dense(96, 40, "q8_0", seed=3)
    kv = {"general.architecture": "llama", **P.metadata("block", 32, 16, True, True)}
    path = str(tmp_path / "rt.gguf")
    G.write_gguf(path, kv, tensors, q8_0=("blk.7.ffn_up.weight",))
    with G.GGUFReader(path) as r:
        assert r.kv == kv
        for n, a in tensors.items():
            assert np.array_equal(r.tensor(n), a), n
        assert r.infos["blk.7.ffn_up.weight"][:2] == ((96, 40), 8)
    ref_gguf_py = "/root/reference/gguf-py"
    if os.path.isdir(ref_gguf_py):                     # build container only: the reference's reader accepts the file
        import sys
        sys.path.insert(0, ref_gguf_py)
        import gguf
        rr = gguf.GGUFReader(path)
        assert {t.name for t in rr.tensors} == set(tensors)
        t = next(t for t in rr.tensors if t.name == "blk.7.ffn_up.b_val")
        assert t.shape.tolist() == [16, 2, 40] and np.array_equal(np.asarray(t.data).reshape(40, 2, 16), p.b_val)


def test_load_imatrix_gguf(tmp_path):
    """<name>.in_sum2 / <name>.counts pairs -> mean squared activation per input (sd_utils.cpp:176-226)."""
    from llama_cpp_amd.scan import load_imatrix
    sums = np.arange(12, dtype=np.float32).reshape(1, 12) + 1
    path = str(tmp_path / "imatrix.gguf")
    G.write_gguf(path, {"general.type": "imatrix", "imatrix.chunk_count": 4, "imatrix.chunk_size": 512, "imatrix.datasets": ["wiki"]},
                 {"blk.0.ffn_gate.weight.in_sum2": sums, "blk.0.ffn_gate.weight.counts": np.array([4.0], np.float32),
                  "blk.0.ffn_up.weight.in_sum2": sums, "blk.0.ffn_up.weight.counts": np.array([0.0], np.float32)})
    im = load_imatrix(path)
    assert np.array_equal(im["blk.0.ffn_gate.weight"], (sums / 4.0).reshape(-1))
    assert np.all(im["blk.0.ffn_up.weight"] == 1.0)
