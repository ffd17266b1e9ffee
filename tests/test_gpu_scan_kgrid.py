"""sdh_scan_tensor_kgrid: a K schedule of one tensor from ONE fit and ONE selection (VERDICT r01 item 5;
tools/seeddelta-build/sd_build.cpp:444-451 fits once per tensor, :843-913 runs one trial per K).

The contract (include/seeddelta_hip.h): result j is, bit for bit, what a lone sdh_scan_tensor with K = Ks[j]
returns from the same generator state, and the payload sdh_scan_pick(j) exposes is the payload that lone call
leaves behind. The lone call is pinned to the reference elsewhere (goldens of the compiled sd_eval.cpp:
test_gpu_scan.py::test_scan_matches_reference_goldens), so equality with it carries that pin to the schedule.
Covered: COO / block, base on / off (the FFN proxy runs without base), row scale, imatrix weighting, the compact
metrics_only selection, schedules that are unsorted, contain duplicates, exceed n_in (clamped budgets) and a
one-member schedule; the clamp case must be declined.
"""
import numpy as np
import pytest

from llama_cpp_amd import scan as SC
from llama_cpp_amd import seeddelta as sd
from llama_cpp_amd.dense import make_dense

pytestmark = pytest.mark.gpu

FIELDS = ("K_eff", "n_blocks", "L", "B", "has_x", "has_w", "proxy_available", "em", "em_x", "em_w", "em_x_w", "cost", "proxy",
          "delta_clamp_scale")


def same(a, b):
    for k in FIELDS:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), (k, x, y)
        else:
            assert x == y, (k, x, y)


def dense_trio(n_embd, n_ff, wtype="q8_0", seed=3):
    return {"ffn_gate": make_dense(n_embd, n_ff, wtype, seed=seed, structure=0.4),
            "ffn_up": make_dense(n_embd, n_ff, wtype, seed=seed + 1, structure=0.4),
            "ffn_down": make_dense(n_ff, n_embd, wtype, seed=seed + 2, structure=0.4)}


CASES = {
    # name: (params, kind, K schedule)
    "coo_base_rs": (dict(scheme="coo", base=True, row_scale=True, base_max_samples=128, base_perm_trials=3, eval_cols=24, eval_x=5),
                    "ffn_gate", [8, 40, 100, 260, 300]),
    "coo_nobase_proxy": (dict(scheme="coo", base=False, row_scale=False, eval_cols=24, eval_x=6), "ffn_up", [12, 64, 128, 200]),
    "coo_nobase_rs_down": (dict(scheme="coo", base=False, row_scale=True, eval_cols=16, eval_x=4), "ffn_down", [16, 300, 33]),
    "block16_base_rs": (dict(scheme="block", block=16, base=True, row_scale=True, base_max_samples=96, base_perm_trials=2,
                             eval_cols=24, eval_x=5), "ffn_down", [16, 64, 65, 200]),
    "block32_nobase": (dict(scheme="block", block=32, base=False, row_scale=False, eval_cols=16, eval_x=3), "ffn_gate", [32, 96, 500]),
    "unsorted_dup_clamped": (dict(scheme="coo", base=True, row_scale=True, base_max_samples=64, base_perm_trials=2, eval_cols=16,
                                  eval_x=3), "ffn_gate", [100, 8, 100, 5000, 1, 511]),
    "single": (dict(scheme="coo", base=True, row_scale=True, base_max_samples=64, base_perm_trials=2, eval_cols=16, eval_x=3),
               "ffn_up", [48]),
    "no_eval_x": (dict(scheme="coo", base=False, row_scale=True, eval_cols=16, eval_x=0), "ffn_gate", [4, 90]),
}


@pytest.mark.parametrize("metrics_only", [False, True])
@pytest.mark.parametrize("imatrix", [False, True])
@pytest.mark.parametrize("name", list(CASES))
def test_schedule_equals_lone_calls(hip, name, imatrix, metrics_only):
    kw, kind, Ks = CASES[name]
    n_embd, n_ff = 512, 1408
    Wnp = dense_trio(n_embd, n_ff)
    W = {k: SC.DenseWeight(v) for k, v in Wnp.items()}
    n_in = W[kind].n_in
    ws = None
    if imatrix:
        imat = (np.random.default_rng(2).random(n_in).astype(np.float32) + 0.05) ** 2
        ws = SC.imatrix_scale(imat, n_in, eps=1e-8, power=1.0)
    p = SC.ScanParams(seed=21, w_scale=ws, metrics_only=metrics_only, **kw)
    ffn = (W["ffn_gate"], W["ffn_up"], W["ffn_down"])
    layer = 2

    s = SC.ScanSession(21)
    grid = s.tensor_kgrid(W[kind], p, Ks, kind, layer, ffn)
    assert len(grid) == len(Ks)
    after = SC.ScanSession(21)                       # the generator advanced once, as for one candidate
    after.skip(W[kind].n_in, W[kind].n_out, p, kind, layer)
    probe = SC.ScanParams(K=4, eval_cols=8, eval_x=0, seed=21)
    assert np.array_equal(s.tensor(W[kind], probe)["em"], after.tensor(W[kind], probe)["em"])

    s = SC.ScanSession(21)
    grid = s.tensor_kgrid(W[kind], p, Ks, kind, layer, ffn)
    for j, K in enumerate(Ks):
        t = SC.ScanSession(21)
        lone = t.tensor(W[kind], SC.ScanParams(**{**p.__dict__, "K": K}), kind, layer, ffn)
        same(grid[j], lone)
        if metrics_only:
            continue
        s.pick(j)
        a, b = s.fetch(), t.fetch()
        assert a.keys() == b.keys()
        for k in a:
            if a[k] is None:
                assert b[k] is None
                continue
            assert a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (K, k)
        a, b = s.fetch_packed(), t.fetch_packed()
        for k in a:
            if a[k] is not None:
                assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), (K, k)
        t.close()
    if metrics_only:
        with pytest.raises(sd.SeedDeltaError):
            s.fetch()
    s.close()
    for w in W.values():
        w.close()


def test_pick_and_limits(hip):
    W = SC.DenseWeight(make_dense(512, 1408, "q8_0", seed=5, structure=0.3))
    p = SC.ScanParams(scheme="coo", base=False, eval_cols=8, eval_x=0, seed=3)
    s = SC.ScanSession(3)
    with pytest.raises(ValueError):
        s.pick(0)
    s.tensor_kgrid(W, p, [4, 9])
    with pytest.raises(sd.SeedDeltaError):
        s.pick(2)
    s.pick(0)
    assert s.fetch()["idx"].shape == (1408, 4)
    with pytest.raises(sd.SeedDeltaError):
        s.tensor_kgrid(W, p, list(range(1, SC.MAX_KGRID + 2)))
    with pytest.raises(sd.SeedDeltaError):
        s.tensor_kgrid(W, p, [])
    # the clamp rescales the stored values per K: a schedule is declined, a lone budget still works
    Wd = SC.DenseWeight(make_dense(1408, 512, "q8_0", seed=6, structure=0.3))
    pc = SC.ScanParams(scheme="coo", base=True, row_scale=True, base_max_samples=64, base_perm_trials=2, eval_cols=8, eval_x=2,
                       seed=3, delta_norm_clamp_down=0.05)
    with pytest.raises(sd.SeedDeltaError, match="one by one"):
        SC.ScanSession(3).tensor_kgrid(Wd, pc, [16, 64], "ffn_down")
    assert SC.ScanSession(3).tensor_kgrid(Wd, pc, [64], "ffn_down")[0]["delta_clamp_scale"] <= 1.0
    s.close()


def test_run_group_and_run_many_use_the_schedule(hip):
    """Evaluator.run_group sends the whole K list as one call (chunks of MAX_KGRID), falls back to lone calls for
    the clamp; both forms give the same entries, also through the thread pool."""
    def model(layer, kind):
        n_in, n_out = (800, 320) if kind == "ffn_down" else (320, 800)
        return make_dense(n_in, n_out, "q8_0", seed=10 * layer + SC.KIND_ID[kind], structure=0.4)
    for kw in (dict(base=True, row_scale=True, base_max_samples=100, base_perm_trials=2, eval_cols=16, eval_x=4),
               dict(base=False, row_scale=False, eval_cols=16, eval_x=4),
               dict(base=True, row_scale=True, base_max_samples=100, base_perm_trials=2, eval_cols=16, eval_x=4, metrics_only=True)):
        p = SC.ScanParams(**kw)
        ev = SC.Evaluator(model, p)
        ks = list(range(4, 4 + 3 * 20, 3))                      # 20 budgets: two chunks
        for kind in SC.KINDS:
            group = [SC.Candidate(1, kind, k) for k in ks]
            a = ev.run_group(group)
            b = ev.run_group(group, kgrid=False)
            for x, y in zip(a, b):
                same(x, y)
            fresh = ev.run(group[7])
            same(a[7], fresh)
        cands = [SC.Candidate(layer, kind, k) for layer in (0, 1) for kind in SC.KINDS for k in (8, 24, 64)]
        many = ev.run_many(cands, threads=3)
        for c, r in zip(cands, many):
            same(r, ev.run(c))
        ev.close()
    pc = SC.ScanParams(base=True, row_scale=True, base_max_samples=100, base_perm_trials=2, eval_cols=16, eval_x=4,
                       delta_norm_clamp_down=0.02)
    ev = SC.Evaluator(model, pc)
    group = [SC.Candidate(0, "ffn_down", k) for k in (8, 32)]
    for x, c in zip(ev.run_group(group), group):
        same(x, ev.run(c))
    ev.close()


@pytest.mark.parametrize("case", ["coo_base_rs", "coo_nobase_proxy", "block16_base_rs", "coo_base_rs_imatrix", "metrics_only"])
@pytest.mark.parametrize("min_mean", [0.0, 0.6, 2.0])          # first trial passes / some trial may pass / none passes
def test_autotune_schedule_as_one_call(hip, case, min_mean):
    """scan.autotune sends the whole K schedule as ONE sequential-draw call (kgrid_draws = 1: one fit, one selection
    at the largest K, trial j's eval samples drawn after trial j-1's, sd_build.cpp:843-913) and rewinds the generator
    to the trial the builder would have stopped at. Against the trial-by-trial form (kgrid=False): the chosen entry,
    every attempt, the payload left on the handle and the stream position afterwards are identical."""
    kw = {"coo_base_rs": dict(scheme="coo", base=True, row_scale=True, base_max_samples=96, base_perm_trials=2, eval_cols=16, eval_x=4),
          "coo_nobase_proxy": dict(scheme="coo", base=False, row_scale=False, eval_cols=16, eval_x=4),
          "block16_base_rs": dict(scheme="block", block=16, base=True, row_scale=True, base_max_samples=96, base_perm_trials=2,
                                  eval_cols=16, eval_x=3),
          "coo_base_rs_imatrix": dict(scheme="coo", base=True, row_scale=True, base_max_samples=96, base_perm_trials=2, eval_cols=16,
                                      eval_x=4),
          "metrics_only": dict(scheme="coo", base=True, row_scale=True, base_max_samples=96, base_perm_trials=2, eval_cols=16,
                               eval_x=4, metrics_only=True)}[case]
    Wnp = dense_trio(512, 1408, seed=11)
    W = {k: SC.DenseWeight(v) for k, v in Wnp.items()}
    kind = "ffn_up"
    ws = None
    if case.endswith("imatrix"):
        imat = (np.random.default_rng(4).random(512).astype(np.float32) + 0.05) ** 2
        ws = SC.imatrix_scale(imat, 512, eps=1e-8, power=1.0)
    p = SC.ScanParams(seed=5, w_scale=ws, **kw)
    ffn = (W["ffn_gate"], W["ffn_up"], W["ffn_down"])
    schedule = [16, 16, 0, 48, 96, 200, 320]
    probe = SC.ScanParams(K=4, eval_cols=8, eval_x=2, seed=5)
    out = []
    for kgrid in (True, False):
        s = SC.ScanSession(5)
        e = SC.autotune(s, W[kind], p, schedule, kind, 3, ffn, min_mean=min_mean, min_p05=0.0, kgrid=kgrid)
        pay = None if p.metrics_only else s.fetch_packed()
        after = s.tensor(W[kind], probe, kind, 3, ffn)           # the next draws of the stream
        out.append((e, pay, after))
        s.close()
    (ea, pa, aa), (eb, pb, ab) = out
    strip = lambda e: {k: v for k, v in e.items() if k != "autotune_attempts"}           # noqa: E731
    assert strip(ea) == strip(eb)
    assert len(ea["autotune_attempts"]) == len(eb["autotune_attempts"])
    for x, y in zip(ea["autotune_attempts"], eb["autotune_attempts"]):
        assert {k: v for k, v in x.items() if k != "seconds"} == {k: v for k, v in y.items() if k != "seconds"}
    if min_mean == 2.0:
        assert not ea["emit"] and [a["K_budget"] for a in ea["autotune_attempts"]] == [16, 48, 96, 200, 320]
    if min_mean == 0.0:
        assert ea["emit"] and len(ea["autotune_attempts"]) == 1
    if pa is not None:
        for k in pa:
            if pa[k] is not None:
                assert pa[k].shape == pb[k].shape and pa[k].tobytes() == pb[k].tobytes(), k
    same(aa, ab)
    for w in W.values():
        w.close()
