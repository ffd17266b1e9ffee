"""Config 5 at its real size, on the path bench.py times: Mistral-7B FFN tensors in Q8_0 (ffn_gate 4096 -> 14336 with
the K schedule {128..768}, ffn_down 14336 -> 4096 with {256..1536}; policies/policy.mistral7b.base.json), evaluated by
sdh_scan_tensor_kgrid — one fit, one selection at the largest K, every K's metric passes — in the two variants of the
sweep (COO without base + FFN proxy; COO with base + row scale), eval_cols = 64, eval_x = 32, metrics_only and payload
emission (tools/seeddelta-build/sd_build.cpp:444-451, :843-913).

Every report field is compared with the scan oracle (oracle/sd_scan_oracle.c, pinned bit for bit to the compiled
reference evaluators) restating ONLY the rows the metric passes sample — the evaluators read nothing else
(sd_eval.cpp:138, :1052) — so a case takes seconds on the CPU. Each K of a layer-scan grid is a builder run of its own
with the same seed: the oracle walks fit -> eval_w -> eval_x from the same stream position for every K.

Bars: base fit bit-exact; indices and values bit-exact; row_scale <= 1 ulp; metrics 1e-9 relative on identical
idx / val / row_scale, 2e-6 when the oracle's own (<= 1 ulp different) row scales enter; metrics_only entries equal the
payload-emission entries bit for bit.
"""
import numpy as np
import pytest

import scan_py as S
from llama_cpp_amd import scan as SC
from llama_cpp_amd.dense import dequantize_q8_0, make_dense

pytestmark = pytest.mark.gpu

N_EMBD, N_FF = 4096, 14336
K_GATE = (128, 256, 384, 512, 640, 768)
K_DOWN = (256, 512, 768, 1024, 1280, 1536)
SEED = 1234


def ulp_close(a, b, ulps=1):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


_W = {}


def weight(kind):
    """bench.py's synthetic tensors: 0.02 * N(0,1) with an XOR-circulant component, Q8_0."""
    if kind not in _W:
        n_in, n_out = (N_FF, N_EMBD) if kind == "ffn_down" else (N_EMBD, N_FF)
        _W[kind] = make_dense(n_in, n_out, "q8_0", seed=1234 + SC.KIND_ID[kind], structure=0.3)
    return _W[kind]


_D = {}


def dense(kind):
    if kind not in _D:
        _D[kind] = SC.DenseWeight(weight(kind))
    return _D[kind]


def params(base, metrics_only):
    if base:
        return SC.ScanParams(scheme="coo", base=True, row_scale=True, eval_cols=64, eval_x=32, base_perm_trials=4,
                             seed=SEED, metrics_only=metrics_only)
    return SC.ScanParams(scheme="coo", base=False, row_scale=False, eval_cols=64, eval_x=32, seed=SEED, metrics_only=metrics_only)


def oracle_entries(kind, Ks, base, layer, proxy):
    """Per K: (em, em_x, proxy or None) from the oracle, restating only the sampled rows."""
    W = weight(kind)
    n_in, n_out = S.w_dims(W)
    o0 = S.Engine("oracle", SEED)
    fit = o0.fit_base(W, 2048, 4) if base else None
    out = []
    for K in Ks:
        o = o0.fork()                                   # every K is a builder run of its own with the same seed
        rows_w = o.fork().shuffle_iota_i64(n_out)[:64]
        sel = S.select_rows_list(W, fit, 0, K, 1, base, rows_w)
        em = o.eval_w(W, fit, 0, sel[0], sel[1], K, 1, sel[2], 64)
        rows_x = o.fork().shuffle_iota_i64(n_out)[:64]
        sel = S.select_rows_list(W, fit, 0, K, 1, base, rows_x, out=sel)
        em_x = o.eval_x(W, fit, 0, sel[0], sel[1], K, 1, sel[2], 64, 32)
        out.append((em, em_x))
    return fit, out


@pytest.mark.parametrize("kind,Ks", [("ffn_gate", K_GATE), ("ffn_down", K_DOWN)])
@pytest.mark.parametrize("base", [False, True])
def test_kgrid_metrics_only_at_mistral_dims_against_the_oracle(hip, kind, Ks, base):
    """The benched path: sdh_scan_tensor_kgrid with metrics_only = 1 over the policy's K schedule."""
    layer = 3
    W = dense(kind)
    ffn = None
    if not base and kind != "ffn_down":                 # the FFN proxy reads all three tensors of the layer
        ffn = (dense("ffn_gate"), dense("ffn_up"), dense("ffn_down"))
    s = SC.ScanSession(SEED)
    res = s.tensor_kgrid(W, params(base, True), Ks, kind, layer, ffn)
    full = SC.ScanSession(SEED).tensor_kgrid(W, params(base, False), Ks, kind, layer, ffn)
    fit, want = oracle_entries(kind, Ks, base, layer, ffn is not None)
    n_in, n_out = S.w_dims(weight(kind))
    o = S.Engine("oracle", SEED)
    for j, K in enumerate(Ks):
        r, (em, em_x) = res[j], want[j]
        assert r["K_eff"] == K and r["has_x"]
        assert (r["L"], r["B"]) == ((4096, 4) if base else (0, 0))
        # row scales may differ by an ulp between the two sides: 2e-6 end to end (exact for the no-base variant)
        tol = dict(rtol=2e-6, atol=1e-12) if base else dict(rtol=1e-9, atol=1e-14)
        assert np.allclose(r["em"], em, **tol), (K, r["em"], em)
        assert np.allclose(r["em_x"], em_x, **tol), (K, r["em_x"], em_x)
        assert np.array_equal(r["cost"], o.cost(4096 if base else 0, 4 if base else 0, n_in, n_out, K, base)), K
        # report entries only == payload emission, bit for bit
        for k in ("em", "em_x", "cost", "proxy"):
            assert np.array_equal(np.asarray(r[k]).view(np.uint64), np.asarray(full[j][k]).view(np.uint64)), (K, k)
        assert r["proxy_available"] == (ffn is not None)
    if ffn is not None:
        # the proxy samples hidden units with its own generator: restate every row once at the largest K (sorted rows:
        # every smaller K is a prefix) and let the oracle pick
        Wg, Wu, Wd = weight("ffn_gate"), weight("ffn_up"), weight("ffn_down")
        idx, val, _ = S.select_rows(weight(kind), None, 0, Ks[-1], 1, False, nth=16)
        for j, K in enumerate(Ks):
            ok, proxy = o.ffn_proxy(SC.KIND_ID[kind], Wg, Wu, Wd, np.ascontiguousarray(idx[:, :K]), np.ascontiguousarray(val[:, :K]),
                                    K, 32, 64, SEED + 101 * layer + SC.KIND_ID[kind] + 1)
            assert ok and np.allclose(res[j]["proxy"], proxy, rtol=1e-9, atol=1e-12), (K, res[j]["proxy"], proxy)


@pytest.mark.parametrize("kind,Ks", [("ffn_down", K_DOWN), ("ffn_gate", K_GATE)])
def test_kgrid_payload_emission_at_mistral_dims(hip, kind, Ks):
    """metrics_only = 0 (what build_run writes into the GGUF), base + row scale: the fit bit-exact; for every K of the
    schedule idx / val bit-equal and row_scale within an ulp on 32 sampled rows; the top-K threshold property and the
    value definition on ALL rows at the largest K; metrics 1e-9 against the oracle fed the GPU's own payload."""
    W = weight(kind)
    n_in, n_out = S.w_dims(W)
    s = SC.ScanSession(SEED)
    res = s.tensor_kgrid(dense(kind), params(True, False), Ks, kind, 7, None)
    o0 = S.Engine("oracle", SEED)
    fit = o0.fit_base(W, 2048, 4)
    rows = np.random.default_rng(5).choice(n_out, 32, replace=False).astype(np.int64)
    rows[:2] = (0, n_out - 1)
    for j, K in enumerate(Ks):
        s.pick(j)
        out = s.fetch()
        if j == 0:
            for a, b in (("h2", fit.h2), ("d2", fit.d2)):
                assert np.array_equal(out[a].view(np.uint32), b.view(np.uint32)), a
            assert np.array_equal(out["perm1"], fit.perm1) and np.array_equal(out["perm1_inv"], fit.perm1_inv)
        assert out["idx"].shape == (n_out, K)
        idx, val, rs = S.select_rows_list(W, fit, 0, K, 1, True, rows)
        assert np.array_equal(out["idx"][rows], idx[rows]), K
        assert np.array_equal(out["val"][rows].view(np.uint32), val[rows].view(np.uint32)), K
        assert ulp_close(out["row_scale"][rows], rs[rows]), K
        # identical payload on both sides: only the reduction order differs
        o = o0.fork()
        em = o.eval_w(W, fit, 0, out["idx"], out["val"], K, 1, out["row_scale"], 64)
        em_x = o.eval_x(W, fit, 0, out["idx"], out["val"], K, 1, out["row_scale"], 64, 32)
        assert np.allclose(res[j]["em"], em, rtol=1e-9, atol=1e-14), (K, res[j]["em"], em)
        assert np.allclose(res[j]["em_x"], em_x, rtol=1e-9, atol=1e-14), (K, res[j]["em_x"], em_x)
    # all rows at the largest K: residual r = w - h2[(p ^ inv[q]) & (L-1)] (sd_build.cpp:533-580)
    K = Ks[-1]
    Wf = dequantize_q8_0(W)
    L = 4096
    tall = n_out >= n_in
    if tall:
        b = np.arange(n_out) // L
        p = np.arange(n_out) - b * L
        base = np.stack([fit.h2[b[o]][(p[o] ^ fit.perm1_inv[b[o]]) & (L - 1)] for o in range(0, n_out)])      # [n_out][n_in]
    else:
        cols_b = np.arange(n_in) // L
        q = np.arange(n_in) - cols_b * L
        inv = fit.perm1_inv[cols_b, q]
        base = fit.h2[cols_b[None, :], (np.arange(n_out)[:, None] ^ inv[None, :]) & (L - 1)]
    r = Wf - base.astype(np.float32)
    idx, val = out["idx"], out["val"]
    assert idx.min() >= 0 and idx.max() < n_in
    assert np.array_equal(np.take_along_axis(r, idx, axis=1).view(np.uint32), val.view(np.uint32))       # values are the residual at idx
    a = np.abs(val)
    assert np.all(a[:, :-1] >= a[:, 1:])                                                                 # descending |r|
    ties = a[:, :-1] == a[:, 1:]
    assert np.all(idx[:, :-1][ties] < idx[:, 1:][ties])                                                  # ties by ascending index
    ar = np.abs(r)
    np.put_along_axis(ar, idx, -1.0, axis=1)
    assert np.all(ar.max(axis=1) <= a[:, -1])                                                            # nothing left out beats the K-th
    srt = np.sort(idx, axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])                                                             # no duplicates


def test_rejected_clamp_schedule_leaves_the_generator_untouched(hip):
    """delta_norm_clamp_down with several budgets is declined (SDH_ERR_UNSUPPORTED) before any draw or state write: the
    fallback the header names — one call per budget — then reproduces the builder's draws."""
    from llama_cpp_amd.seeddelta import SeedDeltaError
    W = make_dense(1536, 512, "q8_0", seed=21, structure=0.3)
    dw = SC.DenseWeight(W)
    base = dict(scheme="coo", base=True, row_scale=True, base_max_samples=128, base_perm_trials=2, eval_cols=32, eval_x=4, seed=11)
    ref = SC.ScanSession(11)
    want = [ref.tensor(dw, SC.ScanParams(K=K, delta_norm_clamp_down=0.2, base_mode=(0 if i == 0 else 1), **base), "ffn_down", 0, None)
            for i, K in enumerate((48, 96))]
    s = SC.ScanSession(11)
    with pytest.raises(SeedDeltaError):
        s.tensor_kgrid(dw, SC.ScanParams(delta_norm_clamp_down=0.2, **base), (48, 96), "ffn_down", 0, None)
    got = [s.tensor(dw, SC.ScanParams(K=K, delta_norm_clamp_down=0.2, base_mode=(0 if i == 0 else 1), **base), "ffn_down", 0, None)
           for i, K in enumerate((48, 96))]
    for a, b in zip(want, got):
        for k in ("em", "em_x"):
            assert np.array_equal(np.asarray(a[k]).view(np.uint64), np.asarray(b[k]).view(np.uint64)), k
        assert a["delta_clamp_scale"] == b["delta_clamp_scale"] < 1.0
