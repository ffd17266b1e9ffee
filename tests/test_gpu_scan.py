"""GPU parity tests of the scan pipeline (sdh_scan_* through the C ABI) against
  * the committed goldens computed by the reference's own builder code (tests/golden/scan_*.npz,
    made by oracle/make_scan_golden.py), and
  * the scan oracle (oracle/sd_scan_oracle.c, pinned bit-for-bit to that reference code) on seeded
    inputs, consuming an identical mt19937 stream.

Bars:
  * base fit (h2, d2, perm1, perm1_inv): bit-exact;
  * selected indices and values: bit-exact (ties ordered by ascending index on both sides);
  * row_scale: fp32, |diff| <= 1 ulp (the LSQ sums are tree-reduced in double on the GPU);
  * metrics (double): relative 1e-9 when evaluated on identical idx/val/row_scale; 2e-6 end to end
    when a 1-ulp row_scale difference may enter.
"""
import numpy as np
import pytest

import scan_py as S
from conftest import load_scan_golden, scan_golden_names
from llama_cpp_amd import scan as SC
from llama_cpp_amd.dense import make_dense

pytestmark = pytest.mark.gpu

KIND = SC.KINDS


def ulp_close(a, b, ulps=1):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


def run_gpu(g):
    ws = {k: SC.DenseWeight(g[k]) for k in ("Wg", "Wu", "Wd") if k in g}
    W = ws[("Wg", "Wu", "Wd")[g["kind"]]]
    p = SC.ScanParams(scheme="block" if g["scheme"] else "coo", K=g["K"], block=g["block"], base=bool(g["base"]),
                      row_scale=bool(g["row_scale"]), base_max_samples=g["max_samples"], base_perm_trials=g["trials"],
                      eval_cols=g["eval_cols"], eval_x=g["eval_x"], seed=g["seed"])
    s = SC.ScanSession(g["seed"])
    ffn = (ws["Wg"], ws["Wu"], ws["Wd"]) if len(ws) == 3 else None
    res = s.tensor(W, p, KIND[g["kind"]], g["layer"], ffn)
    return s, p, res, W


@pytest.mark.parametrize("name", scan_golden_names())
def test_scan_matches_reference_goldens(hip, name):
    g = load_scan_golden(name)
    s, p, res, W = run_gpu(g)
    out = s.fetch()
    Wnp = g[("Wg", "Wu", "Wd")[g["kind"]]]
    fit = None
    if g["base"]:
        for a, b in (("d2", "fit_d2"), ("h2", "fit_h2")):
            assert np.array_equal(out[a].view(np.uint32), g[b].view(np.uint32)), a
        assert np.array_equal(out["perm1"], g["fit_perm1"]) and np.array_equal(out["perm1_inv"], g["fit_inv"])
        fit = S.Fit(res["L"], res["B"], g["fit_d2"], g["fit_h2"], g["fit_perm1"], g["fit_inv"])
    Ksel = res["n_blocks"] if g["scheme"] else res["K_eff"]
    idx, val, rs = S.select_rows(Wnp, fit, g["scheme"], Ksel, g["block"], bool(g["row_scale"]))
    assert np.array_equal(out["idx"], idx)
    assert np.array_equal(out["val"].view(np.uint32), val.view(np.uint32))
    if g["row_scale"]:
        assert ulp_close(out["row_scale"], rs)
    assert np.array_equal(res["cost"], g["cost"])
    assert np.allclose(res["em"], g["em"], rtol=2e-6, atol=1e-12), (res["em"], g["em"])
    assert np.allclose(res["em_x"], g["em_x"], rtol=2e-6, atol=1e-12), (res["em_x"], g["em_x"])
    assert res["proxy_available"] == bool(g["proxy_ok"])
    if g["proxy_ok"]:
        assert np.allclose(res["proxy"], g["proxy"], rtol=1e-9, atol=1e-12), (res["proxy"], g["proxy"])
    # same idx/val/row_scale on both sides: only the reduction order differs
    o = S.Engine("oracle", g["seed"])
    if g["base"]:
        o.fit_base(Wnp, g["max_samples"], g["trials"])
    em = o.eval_w(Wnp, fit, g["scheme"], out["idx"], out["val"], Ksel, g["block"], out["row_scale"], g["eval_cols"])
    em_x = o.eval_x(Wnp, fit, g["scheme"], out["idx"], out["val"], Ksel, g["block"], out["row_scale"], g["eval_cols"], g["eval_x"])
    assert np.allclose(res["em"], em, rtol=1e-9, atol=1e-14)
    assert np.allclose(res["em_x"], em_x, rtol=1e-9, atol=1e-14)
    # and the session's RNG stands where the reference's does: a follow-up draw-only tensor agrees
    assert o.u32() == g["next_u32"]


@pytest.mark.parametrize("dims", [(512, 1408), (1408, 512), (1024, 1024), (288, 800), (2048, 1024), (1024, 2560), (2560, 1024)])
@pytest.mark.parametrize("wtype", ["q8_0", "f16"])
def test_base_fit_bit_exact(hip, dims, wtype):
    n_in, n_out = dims
    W = make_dense(n_in, n_out, wtype, seed=n_in + 1)
    o = S.Engine("oracle", 77)
    fit = o.fit_base(W, 200, 3)
    s = SC.ScanSession(77)
    dw = SC.DenseWeight(W)
    res = s.tensor(dw, SC.ScanParams(K=8, base=True, base_max_samples=200, base_perm_trials=3, eval_cols=0, eval_x=0, seed=77))
    out = s.fetch()
    assert (res["L"], res["B"]) == (fit.L, fit.B)
    assert np.array_equal(out["h2"].view(np.uint32), fit.h2.view(np.uint32))
    assert np.array_equal(out["d2"].view(np.uint32), fit.d2.view(np.uint32))
    assert np.array_equal(out["perm1"], fit.perm1) and np.array_equal(out["perm1_inv"], fit.perm1_inv)


@pytest.mark.parametrize("scheme,K,block", [("coo", 1, 32), ("coo", 12, 32), ("coo", 200, 32), ("coo", 511, 32),
                                            ("block", 16, 16), ("block", 100, 32), ("block", 24, 8)])
@pytest.mark.parametrize("base", [False, True])
def test_selection_vs_oracle(hip, scheme, K, block, base):
    """int8-grid weights (Q8_0, no base) make equal scores common: the tie rule is exercised."""
    W = make_dense(544, 700, "q8_0", seed=K)
    o = S.Engine("oracle", 5)
    fit = o.fit_base(W, 64, 2) if base else None
    p = SC.ScanParams(scheme=scheme, K=K, block=block, base=base, row_scale=True, base_max_samples=64, base_perm_trials=2,
                      eval_cols=0, eval_x=0, seed=5)
    s = SC.ScanSession(5)
    res = s.tensor(SC.DenseWeight(W), p)
    out = s.fetch()
    sch = 1 if scheme == "block" else 0
    Ksel = res["n_blocks"] if sch else res["K_eff"]
    idx, val, rs = S.select_rows(W, fit, sch, Ksel, block, True)
    assert np.array_equal(out["idx"], idx)
    assert np.array_equal(out["val"].view(np.uint32), val.view(np.uint32))
    assert ulp_close(out["row_scale"], rs)
    if not base:
        assert np.all(out["row_scale"] == 1.0)


def test_skip_reproduces_stream_position(hip):
    """A rank that owns (layer, ffn_up) skips ffn_gate: same results as the serial gate -> up run."""
    Wg, Wu = make_dense(320, 800, "q8_0", seed=1), make_dense(320, 800, "q8_0", seed=2)
    p = SC.ScanParams(K=24, base=True, row_scale=True, base_max_samples=100, base_perm_trials=2, eval_cols=16, eval_x=4)
    dg, du = SC.DenseWeight(Wg), SC.DenseWeight(Wu)
    a = SC.ScanSession(p.seed)
    a.tensor(dg, p, "ffn_gate", 3)
    ra = a.tensor(du, p, "ffn_up", 3)
    b = SC.ScanSession(p.seed)
    b.skip(320, 800, p, "ffn_gate", 3)
    rb = b.tensor(du, p, "ffn_up", 3)
    for k in ("em", "em_x", "cost"):
        assert np.array_equal(ra[k], rb[k]), k
    fa, fb = a.fetch(), b.fetch()
    assert np.array_equal(fa["idx"], fb["idx"]) and np.array_equal(fa["perm1"], fb["perm1"])


def test_layer_scan_document(hip):
    """layer_scan() == collect_metrics over oracle-built report entries, for a 2-layer toy model."""
    def model(layer, kind):
        n_embd, n_ff = 256, 640
        shape = (n_ff, n_embd) if kind != "ffn_down" else (n_embd, n_ff)
        return make_dense(shape[1], shape[0], "q8_0", seed=100 * layer + SC.KIND_ID[kind])
    p = SC.ScanParams(K=12, eval_cols=16, eval_x=4)
    doc = SC.layer_scan(model, [0, 1], p)
    assert set(doc["layers"]) == {"0", "1"} and doc["K_fixed"] == 12 and doc["errors"] == []
    for layer in (0, 1):
        o = S.Engine("oracle", p.seed)
        ws = {k: model(layer, k) for k in SC.KINDS}
        for kind in SC.KINDS:
            W = ws[kind]
            idx, val, _ = S.select_rows(W, None, 0, 12, 32, False)
            em = o.eval_w(W, None, 0, idx, val, 12, 32, None, 16)
            em_x = o.eval_x(W, None, 0, idx, val, 12, 32, None, 16, 4)
            ok, proxy = o.ffn_proxy(SC.KIND_ID[kind], ws["ffn_gate"], ws["ffn_up"], ws["ffn_down"], idx, val, 12, 4, 16,
                                    p.seed + 101 * layer + SC.KIND_ID[kind] + 1)
            rec = doc["layers"][str(layer)][kind]
            assert rec["S_mean"] == pytest.approx(em[0], rel=2e-6)      # 6 significant digits, as report.json
            assert rec["cos_p05"] == pytest.approx(em[3], rel=2e-6)
            assert rec["ffn_proxy_available"] == ok == (kind != "ffn_down")
            if ok:
                assert rec["ffn_proxy_cos_p05"] == pytest.approx(proxy[1], rel=2e-6)
                assert rec["ffn_proxy_score"] == pytest.approx(max(1 - proxy[1], proxy[3], abs(proxy[5])), rel=2e-5)


def test_full_size_properties(hip):
    """Mistral-7B gate dims (4096 -> 14336, Q8_0), K=128 COO + base: properties that need no oracle run."""
    W = make_dense(4096, 14336, "q8_0", seed=9, structure=0.3)
    p = SC.ScanParams(K=128, base=True, row_scale=True, eval_cols=64, eval_x=8)
    s = SC.ScanSession(p.seed)
    dw = SC.DenseWeight(W)
    assert dw.stored_bytes == 14336 * 4096 // 32 * 34
    res = s.tensor(dw, p)
    out = s.fetch()
    assert (res["L"], res["B"], res["K_eff"]) == (4096, 4, 128)
    idx = out["idx"]
    assert idx.min() >= 0 and idx.max() < 4096
    assert all(len(set(r.tolist())) == 128 for r in idx[::997])              # no duplicates
    a = np.abs(out["val"])
    assert np.all(a[:, :-1] >= a[:, 1:])                                     # sorted by |residual| descending
    for b in range(4):
        assert sorted(out["perm1"][b].tolist()) == list(range(4096))          # a permutation
        assert np.array_equal(out["perm1"][b][out["perm1_inv"][b]], np.arange(4096))
    # threshold property on a few rows, against the oracle's row reader
    o = S.Engine("oracle", 0)
    fit = S.Fit(4096, 4, out["d2"], out["h2"], out["perm1"], out["perm1_inv"])
    for row in (0, 5000, 14335):
        w = o.read_row(W, row)
        b, pp = divmod(row, 4096)
        r = w - out["h2"][b][(pp ^ out["perm1_inv"][b]) & 4095]
        assert np.array_equal(out["val"][row], r[idx[row]])
        rest = np.delete(np.abs(r), idx[row])
        assert rest.max() <= a[row, -1]
    assert 0.0 < res["em"][0] < 1.0 and res["em"][2] > 0.5 and res["has_x"]
    print(f"\nfull-size candidate: {res['gpu_ms']:.2f} ms on device")


@pytest.mark.parametrize("scheme,K,block", [("coo", 96, 32), ("block", 96, 16)])
def test_scan_to_gguf_to_forward(hip, tmp_path, scheme, K, block):
    """End to end across both halves of the path: evaluate a candidate, pack its staging with the
    builder's rules, store it in a GGUF under the builder's names, load it back as a SeedDelta
    context and run the FFN op: the op's outputs track the dense W.x with the cosine the scan
    predicted (eval_x metrics), and equal the oracle's evaluation of the same payload."""
    import oracle_py as O
    from llama_cpp_amd import gguf_io as G, pack as P
    from llama_cpp_amd.dense import dequantize_q8_0
    from conftest import bits
    n_in, n_out, layer, kind = 512, 1408, 4, "ffn_gate"
    W = make_dense(n_in, n_out, "q8_0", seed=31, structure=0.4)
    p = SC.ScanParams(scheme=scheme, K=K, block=block, base=True, row_scale=True, base_max_samples=256, base_perm_trials=2,
                      eval_cols=64, eval_x=16)
    s = SC.ScanSession(p.seed)
    res = s.tensor(SC.DenseWeight(W), p, kind, layer)
    payload = P.pack_candidate(s.fetch(), n_in, n_out, scheme)
    tensors = P.payload_tensors(layer, kind, payload)
    tensors[f"blk.{layer}.{kind}.weight"] = W
    entry = SC.report_entry(layer, kind, n_in, n_out, p, res)
    kv = {"general.architecture": "llama", **P.metadata(scheme, K, block, True, True, base_perm_trials=2, entries=[entry])}
    path = str(tmp_path / "sd.gguf")
    G.write_gguf(path, kv, tensors, q8_0=(f"blk.{layer}.{kind}.weight",))

    reader, payloads, meta = G.load_seeddelta_context(path)
    assert meta[f"seeddelta.blk.{layer}.{kind}.enabled"] is True and meta["seeddelta.scheme"] == (1 if scheme == "block" else 0)
    ctx = hip.SeedDeltaContext(enabled=True)
    key = f"blk.{layer}.{kind}.weight"
    assert ctx.try_add(key, payloads[key])
    x = np.random.default_rng(3).standard_normal((24, n_in)).astype(np.float32)
    y = ctx.mul_mat(key, x)
    assert np.array_equal(bits(y), bits(O.forward(payloads[key], x, nth=4)))        # T >= 8: exact-order SpMM path
    y_dense = x.astype(np.float64) @ dequantize_q8_0(W).astype(np.float64).T
    cos = np.sum(y * y_dense, axis=1) / (np.linalg.norm(y, axis=1) * np.linalg.norm(y_dense, axis=1))
    assert abs(cos.mean() - res["em_x"][2]) < 0.03, (cos.mean(), res["em_x"][2])
    assert cos.mean() > 0.7
    reader.close()


@pytest.mark.parametrize("scheme,K,block", [("coo", 48, 32), ("block", 64, 16)])
@pytest.mark.parametrize("dims", [(512, 1408), (1408, 512)])
def test_imatrix_weighted_candidate(hip, scheme, K, block, dims):
    """With an imatrix scale the fit, the scores and the row scale are weighted and the _w / _x_w
    passes run (sd_build.cpp:421-422, :753-794); same bars as the unweighted candidate."""
    n_in, n_out = dims
    W = make_dense(n_in, n_out, "q8_0", seed=7, structure=0.4)
    imat = (np.random.default_rng(1).random(n_in).astype(np.float32) + 0.05) ** 2
    ws = SC.imatrix_scale(imat, n_in, eps=1e-8, power=1.0)
    if S.buildref_available():
        assert np.array_equal(ws.view(np.uint32), S.ref_imatrix_scale(imat, n_in, 1e-8, 1.0).view(np.uint32))
    assert abs(ws.mean() - 1.0) < 1e-5
    sch = 1 if scheme == "block" else 0
    p = SC.ScanParams(scheme=scheme, K=K, block=block, base=True, row_scale=True, base_max_samples=128, base_perm_trials=3,
                      eval_cols=32, eval_x=5, seed=11, w_scale=ws)
    s = SC.ScanSession(11)
    res = s.tensor(SC.DenseWeight(W), p)
    out = s.fetch()
    o = S.Engine("oracle", 11)
    fit = o.fit_base(W, 128, 3, ws)
    assert np.array_equal(out["h2"].view(np.uint32), fit.h2.view(np.uint32)) and np.array_equal(out["perm1"], fit.perm1)
    Ksel = res["n_blocks"] if sch else res["K_eff"]
    idx, val, rs = S.select_rows(W, fit, sch, Ksel, block, True, w_scale=ws)
    assert np.array_equal(out["idx"], idx) and np.array_equal(out["val"].view(np.uint32), val.view(np.uint32))
    assert ulp_close(out["row_scale"], rs)
    args = (W, fit, sch, out["idx"], out["val"], Ksel, block, out["row_scale"])
    em = o.eval_w(*args, 32)
    em_w = o.eval_w(*args, 32, ws)
    em_x = o.eval_x(*args, 32, 5)
    em_x_w = o.eval_x(*args, 32, 5, ws)
    assert res["has_w"] and res["has_x"]
    for k, want in (("em", em), ("em_w", em_w), ("em_x", em_x), ("em_x_w", em_x_w)):
        assert np.allclose(res[k], want, rtol=1e-9, atol=1e-14), k
    # skip() with the same params leaves the stream where the evaluation did
    t = SC.ScanSession(11)
    t.skip(n_in, n_out, p)
    r2 = t.tensor(SC.DenseWeight(W), SC.ScanParams(K=4, eval_cols=8, eval_x=0, seed=11))
    r3 = s.tensor(SC.DenseWeight(W), SC.ScanParams(K=4, eval_cols=8, eval_x=0, seed=11))
    assert np.array_equal(r2["em"], r3["em"])
    e = SC.report_entry(0, "ffn_gate", n_in, n_out, p, res)
    assert e["has_w"] and e["cos_mean_x_w"] == res["em_x_w"][2] and e["ffn_score"] == min(res["em_x_w"][2], res["em_x_w"][3])


def test_k_grid_reuses_the_fit_and_autotune_schedule(hip):
    """(a) run_group: each K is a fresh builder run, the fit is computed once and its draws replayed —
    results identical to evaluating every K from scratch. (b) autotune(): fit once, trials in schedule
    order on one running stream, first passing K wins (sd_build.cpp:802-948)."""
    def model(layer, kind):
        n_in, n_out = (800, 320) if kind == "ffn_down" else (320, 800)
        return make_dense(n_in, n_out, "q8_0", seed=10 * layer + SC.KIND_ID[kind], structure=0.4)
    p = SC.ScanParams(base=True, row_scale=True, base_max_samples=100, base_perm_trials=2, eval_cols=16, eval_x=4)
    ev = SC.Evaluator(model, p)
    group = [SC.Candidate(1, "ffn_up", k) for k in (8, 24, 64)]
    shared = ev.run_group(group)
    for c, r in zip(group, shared):
        fresh = ev.run(c)
        for key in ("em", "em_x", "cost"):
            assert np.array_equal(r[key], fresh[key]), (c.K, key)
    many = ev.run_many(group + [SC.Candidate(1, "ffn_down", 16)], threads=2)
    assert all(np.array_equal(a["em"], b["em"]) for a, b in zip(many[:3], shared))

    W = ev.dense(1, "ffn_up")
    s = SC.ScanSession(p.seed)
    e = SC.autotune(s, W, p, [8, 8, 0, 24, 64], "ffn_up", 1, min_mean=0.9, min_p05=0.0)
    ks = [a["K_budget"] for a in e["autotune_attempts"]]
    assert ks[0] == 8 and len(ks) == len(set(ks)) and e["autotune_enabled"]
    if e["emit"]:
        assert e["autotune_selected_budget"] == ks[-1] and e["gating_value"] >= 0.9
    else:
        assert e["decision"] == "autotune_failed_keep_dense" and e["autotune_selected_budget"] == 0 and ks == [8, 24, 64]
    # the oracle on one running stream: fit, then per K (eval_w draw, eval_x draws)
    o = S.Engine("oracle", p.seed)
    Wnp = model(1, "ffn_up")
    fit = o.fit_base(Wnp, 100, 2)
    for a in e["autotune_attempts"]:
        idx, val, rs = S.select_rows(Wnp, fit, 0, a["K_eff"], 32, True)
        o.eval_w(Wnp, fit, 0, idx, val, a["K_eff"], 32, rs, 16)
        em_x = o.eval_x(Wnp, fit, 0, idx, val, a["K_eff"], 32, rs, 16, 4)
        assert a["metric_value"] == pytest.approx(min(em_x[2], em_x[3]), rel=2e-6)       # gate/up gate on the FFN score
    ev.close()


@pytest.mark.parametrize("scheme,K,block,base", [("coo", 40, 32, True), ("block", 64, 16, True), ("coo", 24, 32, False)])
def test_fetch_packed_equals_the_builders_packing_step(hip, scheme, K, block, base):
    """sdh_scan_fetch_packed converts on the device; the bytes must equal the host packing of the
    fp32 / int32 staging (pack.py restates sd_build.cpp:1104-1236) and the reference's own
    ggml_fp32_to_fp16_row applied to the staging values, including overflow, subnormal and NaN inputs."""
    import ctypes as C
    import oracle_py as O
    from llama_cpp_amd import pack as P
    n_in, n_out = 640, 352
    W = make_dense(n_in, n_out, "f32", seed=5, structure=0.3).copy()
    W[3, :8] = [1e6, -1e6, 7e4, 65520.0, 3e-6, -2e-7, 6e-8, 1e-9]         # overflow, halfway to inf, f16 subnormals, underflow
    W[5, 0] = np.nan
    p = SC.ScanParams(scheme=scheme, K=K, block=block, base=base, row_scale=True, base_max_samples=128, base_perm_trials=2,
                      eval_cols=0, eval_x=0)
    s = SC.ScanSession(p.seed)
    s.tensor(SC.DenseWeight(W), p, "ffn_gate", 0)
    raw, packed = s.fetch(), s.fetch_packed()
    want = P.pack_candidate(raw, n_in, n_out, scheme)
    got = P.pack_candidate(packed, n_in, n_out, scheme)
    for name in ("d_idx", "d_val", "b_idx", "b_val", "row_scale", "base_d2", "base_perm1"):
        a, b = getattr(want, name), getattr(got, name)
        assert (a is None) == (b is None), name
        if a is not None:
            assert a.dtype == b.dtype and a.shape == b.shape, name
            assert np.array_equal(a.view(np.uint16) if a.dtype == np.float16 else a,
                                  b.view(np.uint16) if b.dtype == np.float16 else b), name
    assert packed["idx"].dtype == np.int16 and packed["val"].dtype == np.float16 and packed["row_scale"].dtype == np.float16
    # the reference's converter on the same staging values
    if not O.ref_available():
        pytest.skip("oracle/_ref/libsdref.so not built")
    lib = O.ref_lib()
    lib.ggml_fp32_to_fp16_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    v32 = np.ascontiguousarray(raw["val"], dtype=np.float32).reshape(-1)
    ref16 = np.empty(v32.size, dtype=np.uint16)
    lib.ggml_fp32_to_fp16_row(v32.ctypes.data, ref16.ctypes.data, v32.size)
    assert np.array_equal(ref16, packed["val"].view(np.uint16).reshape(-1))
    # types kept as they are on request
    wide = s.fetch_packed(np.int32, np.float32)
    assert np.array_equal(wide["idx"], raw["idx"]) and np.array_equal(wide["val"].view(np.uint32), raw["val"].view(np.uint32))


def test_fetch_packed_index_range_rule(hip):
    """n_in > 32768: indices that do not fit int16 become -1 (sd_build.cpp:1132, :1162)."""
    n_in, n_out = 40000, 24
    rng = np.random.default_rng(2)
    W = (0.01 * rng.standard_normal((n_out, n_in))).astype(np.float32)
    W[:, 39000:39010] = 5.0                                              # winners beyond the int16 range
    p = SC.ScanParams(scheme="coo", K=16, base=False, row_scale=False, eval_cols=0, eval_x=0)
    s = SC.ScanSession(p.seed)
    s.tensor(SC.DenseWeight(W), p, "ffn_down", 0)
    raw, packed = s.fetch(), s.fetch_packed()
    assert (raw["idx"] > 32767).sum() >= 10 * n_out
    assert np.array_equal(packed["idx"], np.where(raw["idx"] > 32767, -1, raw["idx"]).astype(np.int16))


def test_build_run_follows_the_builder_control_flow(hip):
    """One builder process over several layers on one RNG stream (sd_build.cpp:296-1102): policy
    resolution, disabled tensors, per-kind K, autotune, stack guard bookkeeping — and the metrics of
    every tensor equal the oracle's when it walks the same stream tensor after tensor."""
    from llama_cpp_amd import pack as PK, policy as PL

    def model(layer, kind):
        n_in, n_out = (704, 288) if kind == "ffn_down" else (288, 704)
        return make_dense(n_in, n_out, "q8_0", seed=7 * layer + SC.KIND_ID[kind], structure=0.5)
    pol = PL.load_policy({
        "version": 1,
        "defaults": {"enabled": True, "K": {"gate": 16, "up": 24, "down": 32},
                     "gating": {"metric": "cos", "min_mean": {"gate": 0.1, "up": 0.1, "down": 0.99}, "min_p05": 0.0}},
        "ranges": [{"layers": "0", "enabled": False}],
        "layers": {"2": {"tensors": {"ffn_up": {"autotune": {"enabled": True, "schedule_gate_up": [8, 40, 200], "max_iters": 3},
                                                "gating": {"min_mean": 0.8}}}}}}, strict=True)
    p = SC.ScanParams(base=True, row_scale=True, base_max_samples=64, base_perm_trials=2, eval_cols=16, eval_x=4, K=32)
    out = SC.build_run(model, [0, 1, 2], p, policy=pol)
    es = out["entries"]
    assert [(e["layer"], e["kind"]) for e in es] == [(l, k) for l in (0, 1, 2) for k in SC.KINDS]
    assert all(e["decision"] == "disabled" and not e["emit"] for e in es[:3])
    assert [e["K"] for e in es[3:6]] == [16, 24, 32]
    assert es[3]["emit"] and es[3]["gating_metric"] == "ffn_score" and es[3]["decision"] == "pass_gating"
    assert not es[5]["emit"] and es[5]["decision"] == "gating_fail" and es[5]["gating_metric"] == "cos"      # down needs 0.99
    up2 = es[7]
    assert up2["autotune_enabled"] and [a["K_budget"] for a in up2["autotune_attempts"]][0] == 8
    assert up2["emit"] == any(a["pass"] for a in up2["autotune_attempts"])
    assert out["stack_pass_gate_up"] == sum(e["emit"] for e in es if e["kind"] != "ffn_down") and out["stack_pass_down"] == 0
    # emitted tensors can be packed into payloads the FFN op accepts
    (layer, kind), (pp, K, staging) = next(iter(out["emitted"].items()))
    payload = PK.pack_candidate(staging, *SC.dense_dims(model(layer, kind)), "coo")
    assert hip.SeedDeltaWeight(payload).kind == 2

    # the oracle walking the same stream: disabled tensors draw nothing; each trial = [fit] select eval_w eval_x
    o = S.Engine("oracle", p.seed)
    for e in es[3:6]:
        Wnp = model(e["layer"], e["kind"])
        fit = o.fit_base(Wnp, 64, 2)
        idx, val, rs = S.select_rows(Wnp, fit, 0, e["K"], 32, True)
        em = o.eval_w(Wnp, fit, 0, idx, val, e["K"], 32, rs, 16)
        em_x = o.eval_x(Wnp, fit, 0, idx, val, e["K"], 32, rs, 16, 4)
        assert e["cos_mean"] == pytest.approx(em[2], rel=2e-6) and e["rel_l2_p95"] == pytest.approx(em[1], rel=2e-6)
        assert e["cos_p05_x"] == pytest.approx(em_x[3], rel=2e-6)


@pytest.mark.parametrize("name", scan_golden_names())
def test_metrics_only_gives_the_same_report_entry(hip, name):
    """metrics_only selects just the rows the metric passes sample: every number of the report entry is
    bit-identical to the full selection (same rows, same kernels, same order), the RNG stream ends in
    the same place, and no payload is kept."""
    from llama_cpp_amd.seeddelta import SeedDeltaError
    g = load_scan_golden(name)
    s_full, p, res_full, W = run_gpu(g)
    ws = {k: SC.DenseWeight(g[k]) for k in ("Wg", "Wu", "Wd") if k in g}
    Wm = ws[("Wg", "Wu", "Wd")[g["kind"]]]
    pm = SC.ScanParams(**{**p.__dict__, "metrics_only": True})
    s = SC.ScanSession(g["seed"])
    res = s.tensor(Wm, pm, KIND[g["kind"]], g["layer"], (ws["Wg"], ws["Wu"], ws["Wd"]) if len(ws) == 3 else None)
    for k in ("em", "em_x", "cost", "proxy"):
        assert np.array_equal(np.asarray(res[k]).view(np.uint64), np.asarray(res_full[k]).view(np.uint64)), k
    assert res["proxy_available"] == res_full["proxy_available"] and res["K_eff"] == res_full["K_eff"]
    # both sessions drew the same samples: a follow-up candidate agrees bit for bit as well
    nxt = SC.ScanParams(scheme="coo", K=8, eval_cols=8, eval_x=2, seed=g["seed"], metrics_only=True)
    a = s.tensor(Wm, nxt, KIND[g["kind"]], g["layer"], None)
    b = s_full.tensor(W, nxt, KIND[g["kind"]], g["layer"], None)
    assert np.array_equal(np.asarray(a["em_x"]).view(np.uint64), np.asarray(b["em_x"]).view(np.uint64))
    with pytest.raises(SeedDeltaError):
        s.fetch()


def test_metrics_only_with_imatrix_and_block_scheme(hip):
    rng = np.random.default_rng(5)
    W = make_dense(1024, 1536, "q8_0", seed=9, structure=0.3)
    im = (rng.random(1024) * 3 + 0.01).astype(np.float32)
    dw = SC.DenseWeight(W)
    for scheme, K, block in (("coo", 48, 16), ("block", 64, 16)):
        kw = dict(scheme=scheme, K=K, block=block, base=True, row_scale=True, eval_cols=32, eval_x=8, w_scale=SC.imatrix_scale(im, 1024))
        r1 = SC.ScanSession(7).tensor(dw, SC.ScanParams(**kw), "ffn_up", 3, None)
        r2 = SC.ScanSession(7).tensor(dw, SC.ScanParams(metrics_only=True, **kw), "ffn_up", 3, None)
        for k in ("em", "em_w", "em_x", "em_x_w"):
            assert np.array_equal(np.asarray(r1[k]).view(np.uint64), np.asarray(r2[k]).view(np.uint64)), (scheme, k)


def test_delta_norm_clamp_down(hip):
    """--delta-norm-clamp-down (sd_build.cpp:705-720): ffn_down, COO only; the residual values are scaled
    so that ||delta||_2 <= clamp * ||W||_2 before the metrics are taken. The two norms are tree-reduced in
    double on the GPU (the reference keeps one running sum): the scale agrees to 1e-12 relative and the
    values to 1 ulp. Parity of this step is unpinned (sd_build.cpp does not compile here): oracle
    restatement only."""
    W = make_dense(1536, 512, "q8_0", seed=21, structure=0.3)
    dw = SC.DenseWeight(W)
    base = dict(scheme="coo", K=96, eval_cols=32, eval_x=4, seed=11)
    s0 = SC.ScanSession(11)
    r0 = s0.tensor(dw, SC.ScanParams(**base), "ffn_down", 0, None)
    v0 = s0.fetch()["val"]
    assert r0["delta_clamp_scale"] == 1.0
    s1 = SC.ScanSession(11)
    r1 = s1.tensor(dw, SC.ScanParams(delta_norm_clamp_down=0.2, metrics_only=True, **base), "ffn_down", 0, None)
    out = s1.fetch()                                     # the clamp needs every row: the payload is kept
    val_or = v0.copy()
    scale = S.delta_clamp(W, val_or, 0.2)
    assert scale < 1.0 and abs(r1["delta_clamp_scale"] - scale) <= 1e-12 * scale
    assert ulp_close(out["val"], val_or)
    assert r1["em"][0] > r0["em"][0]                     # a scaled-down residual reconstructs worse
    o = S.Engine("oracle", 11)
    em = o.eval_w(W, None, 0, out["idx"], out["val"], 96, 1, None, 32)
    assert np.allclose(r1["em"], em, rtol=1e-9, atol=1e-14)
    # not ffn_down, or the block scheme: no clamp
    r2 = SC.ScanSession(11).tensor(dw, SC.ScanParams(delta_norm_clamp_down=0.2, **base), "ffn_up", 0, None)
    assert r2["delta_clamp_scale"] == 1.0
    r3 = SC.ScanSession(11).tensor(dw, SC.ScanParams(delta_norm_clamp_down=0.2, **{**base, "scheme": "block", "block": 16}), "ffn_down", 0, None)
    assert r3["delta_clamp_scale"] == 1.0
