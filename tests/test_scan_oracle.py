"""The scan oracle (oracle/sd_scan_oracle.c) against the reference's own builder code compiled
unmodified (oracle/_ref/libsdbuildref.so) and against libstdc++'s RNG. Bit-exact throughout: every
double accumulation in the restatement keeps the reference's order.

Ties: the reference orders equal scores however std::nth_element/std::sort leave them; the oracle
orders them by ascending index. Selections are therefore compared as (a) identical score sequences
and (b) identical index sets above the score of the last kept entry.
"""
import numpy as np
import pytest

import scan_py as S
from llama_cpp_amd.dense import dequantize_q8_0, make_dense

needs_ref = pytest.mark.skipif(not S.buildref_available(), reason="oracle/_ref/libsdbuildref.so not built")


def engines(seed):
    return S.Engine("oracle", seed), S.Engine("ref", seed)


def test_mt19937_known_answer():
    """C++11 [rand.predef]: the 10000th draw of a default-seeded mt19937 is 4123659995."""
    e = S.Engine("oracle", 5489)
    v = [e.u32() for _ in range(10000)]
    assert v[-1] == 4123659995


@needs_ref
def test_rng_matches_libstdcxx():
    o, r = engines(1234)
    assert [o.u32() for _ in range(2000)] == [r.u32() for _ in range(2000)]
    for n in (1, 2, 3, 10, 64, 1000, 4096, 14336, 65535, 65536, 70001):
        assert np.array_equal(o.shuffle_iota_i64(n, 5), r.shuffle_iota_i64(n, 5)), n
    for n in (2, 4096, 8192):
        assert np.array_equal(o.shuffle_iota_i32(n), r.shuffle_iota_i32(n)), n
    for n in (1, 7, 10001, 4096 * 3):
        a, b = o.normal(n), r.normal(n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), n
    assert o.u32() == r.u32()          # both streams consumed the same number of draws


@needs_ref
@pytest.mark.parametrize("kind", ["q8_0", "f16", "f32"])
def test_read_row(kind):
    W = make_dense(64, 10, kind, seed=3)
    o, r = engines(1)
    for row in (0, 9):
        assert np.array_equal(o.read_row(W, row), r.read_row(W, row))
    if kind == "q8_0":
        assert np.array_equal(o.read_row(W, 4), dequantize_q8_0(W)[4])


def same_selection(a, b, score):
    """identical score sequence; identical sets strictly above the boundary score"""
    sa, sb = score[a], score[b]
    assert np.array_equal(sa, sb)
    assert np.all(sa[:-1] >= sa[1:])
    if len(sa):
        cut = sa[-1]
        assert set(a[sa > cut].tolist()) == set(b[sb > cut].tolist())


@needs_ref
def test_topk_abs_and_blocks():
    rng = np.random.default_rng(0)
    o, r = engines(1)
    v = rng.standard_normal(1000).astype(np.float32)
    ws = (rng.random(1000) + 0.5).astype(np.float32)
    for K in (1, 17, 128, 999, 1000, 5000):
        assert np.array_equal(o.topk_abs(v, K), r.topk_abs(v, K)), K          # continuous scores: no ties
        assert np.array_equal(o.topk_abs(v, K, ws), r.topk_abs(v, K, ws)), K
    q = dequantize_q8_0(make_dense(1024, 1, "q8_0", seed=5))[0]                # int8 grid: many ties
    for K in (12, 128, 512):
        same_selection(o.topk_abs(q, K), r.topk_abs(q, K), np.abs(q))
    for block, keep in ((16, 8), (32, 48), (16, 1), (24, 5)):
        assert np.array_equal(o.topk_blocks(v, block, keep), r.topk_blocks(v, block, keep))
        assert np.array_equal(o.topk_blocks(v, block, keep, ws), r.topk_blocks(v, block, keep, ws))


DIMS = [(96, 200), (200, 96), (128, 512), (512, 128), (64, 64)]


@needs_ref
@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("kind", ["q8_0", "f32"])
def test_fit_base(dims, kind):
    n_in, n_out = dims
    n_in_q = (n_in + 31) // 32 * 32 if kind == "q8_0" else n_in
    W = make_dense(n_in_q, n_out, kind, seed=n_in)
    o, r = engines(1234)
    for max_samples, trials in ((2048, 4), (16, 2), (0, 1)):
        fo, fr = o.fit_base(W, max_samples, trials), r.fit_base(W, max_samples, trials)
        assert np.array_equal(fo.perm1, fr.perm1) and np.array_equal(fo.perm1_inv, fr.perm1_inv)
        assert np.array_equal(fo.h2.view(np.uint32), fr.h2.view(np.uint32))
        assert np.array_equal(fo.d2.view(np.uint32), fr.d2.view(np.uint32))
    assert o.u32() == r.u32()
    x = np.random.default_rng(2).standard_normal(fo.L).astype(np.float32)
    for b in range(fo.B):
        assert np.array_equal(o.apply_base_block(fo, b, x).view(np.uint32), r.apply_base_block(fr, b, x).view(np.uint32))


@needs_ref
@pytest.mark.parametrize("dims", [(96, 200), (200, 96), (256, 256)])
@pytest.mark.parametrize("base", [False, True])
@pytest.mark.parametrize("scheme", [0, 1])
@pytest.mark.parametrize("row_scale", [False, True])
def test_trial_metrics(dims, base, scheme, row_scale):
    """select (oracle) -> eval_w, eval_x on both engines, continuing one RNG stream like run_trial."""
    n_in, n_out = dims
    W = make_dense(n_in, n_out, "f32", seed=7 + n_in)
    o, r = engines(99)
    fo = o.fit_base(W, 64, 2) if base else None
    fr = r.fit_base(W, 64, 2) if base else None
    block = 16
    K = 3 if scheme == 1 else 24                 # keep / entries
    idx, val, rs = S.select_rows(W, fo, scheme, K, block, row_scale, nth=3)
    idx1, val1, rs1 = S.select_rows(W, fo, scheme, K, block, row_scale, nth=1)
    assert np.array_equal(idx, idx1) and np.array_equal(val, val1)
    mo = o.eval_w(W, fo, scheme, idx, val, K, block, rs, 20)
    mr = r.eval_w(W, fr, scheme, idx, val, K, block, rs, 20)
    assert np.array_equal(mo, mr), (mo, mr)
    xo = o.eval_x(W, fo, scheme, idx, val, K, block, rs, 20, 6)
    xr = r.eval_x(W, fr, scheme, idx, val, K, block, rs, 20, 6)
    assert np.array_equal(xo, xr), (xo, xr)
    assert o.u32() == r.u32()
    if row_scale and not base:
        assert np.all(rs == 1.0)                 # dot == nn term by term (sd_build.cpp:613-617)
    assert mo[2] > 0.2 and mo[0] < 1.0           # sane: the residual captures the largest entries


@needs_ref
def test_weighted_metrics_and_x_scale():
    """imatrix-style w_scale / x_scale variants (the _w and _x_w report fields)."""
    W = make_dense(128, 160, "f32", seed=21)
    ws = (np.random.default_rng(3).random(128) + 0.5).astype(np.float32)
    o, r = engines(5)
    fo, fr = o.fit_base(W, 64, 2, ws), r.fit_base(W, 64, 2, ws)
    assert np.array_equal(fo.h2.view(np.uint32), fr.h2.view(np.uint32))
    idx, val, rs = S.select_rows(W, fo, 0, 16, 16, True, w_scale=ws)
    assert np.array_equal(o.eval_w(W, fo, 0, idx, val, 16, 16, rs, 32, ws), r.eval_w(W, fr, 0, idx, val, 16, 16, rs, 32, ws))
    assert np.array_equal(o.eval_x(W, fo, 0, idx, val, 16, 16, rs, 32, 4, ws), r.eval_x(W, fr, 0, idx, val, 16, 16, rs, 32, 4, ws))


@needs_ref
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_ffn_proxy(kind):
    n_embd, n_ff = 64, 160
    Wg, Wu = make_dense(n_embd, n_ff, "q8_0", seed=1), make_dense(n_embd, n_ff, "q8_0", seed=2)
    Wd = make_dense(n_ff, n_embd, "q8_0", seed=3)
    W = (Wg, Wu, Wd)[kind]
    idx, val, _ = S.select_rows(W, None, 0, 12, 16, False)
    o, r = engines(0)
    ok_o, mo = o.ffn_proxy(kind, Wg, Wu, Wd, idx, val, 12, 8, 24, 1234 + 101 * 3 + kind + 1)
    ok_r, mr = r.ffn_proxy(kind, Wg, Wu, Wd, idx, val, 12, 8, 24, 1234 + 101 * 3 + kind + 1)
    assert ok_o == ok_r == (kind != 2)
    assert np.array_equal(mo, mr), (mo, mr)


@needs_ref
def test_cost_model():
    o, r = engines(0)
    for args in ((4096, 4, 4096, 14336, 128, True), (4096, 4, 14336, 4096, 1536, True), (0, 0, 4096, 14336, 12, False),
                 (2048, 4, 1152, 6912, 64, True)):
        assert np.array_equal(o.cost(*args), r.cost(*args))


def test_percentile_interpolates():
    v = [0.0, 1.0, 2.0, 3.0]
    assert S.percentile(v, 0.5) == 1.5 and S.percentile(v, 0.95) == pytest.approx(2.85) and S.percentile([], 0.5) == 0.0
