"""Prefill at the REAL dims of BASELINE.json's configs (VERDICT r01 'Next round' item 1): T = 512 per ubatch on
cfg4 (Mistral-7B gate / down, COO and block16, K=128), cfg2 (Gemma-3 4B gate COO; ctx 2048 = 4 x 512), cfg3
(Mistral down block32 K=1536, the PPL shape) and one COO payload with 4864 < n_in <= 9728.

Which kernel each shape takes (csrc/sdh_spmm.hip launch_vec, by bytes of one activation column in LDS):
    n_in <= 1216           TT=16 four-token form   (covered at small dims in test_gpu_forward.py)
    n_in <= 4864           TT=8  four-token form   cfg4 gate COO / block16, cfg2 gate COO
    n_in <= 9728           TT=4  four-token form   the 6912 -> 1152 payload below
    n_in <= 19456          TT=2  one-token form    cfg4 down COO, cfg4 down block16
    block, narrow tile and more than 256 values per row: per-token row kernels   cfg3
(a block payload's tile holds one spare row per block, n_in + n_in / block rows, so its limits sit that much lower)
The SpMM kernels accumulate each (row, token) in the CPU op's order with separately rounded multiply and add
(src/llama-seeddelta.cpp:1088-1175, :1178-1360), so the default path must equal the oracle BIT FOR BIT there;
the per-token row kernels (and SDH_FWD_NO_SPMM everywhere) keep the fast-path tolerance of
test_gpu_forward.py; SDH_FWD_W0_MFMA carries f16 operand rounding, bound stated in the test.
The oracle is the restatement pinned bit-for-bit to the reference TU (tests/test_oracle.py); 16 threads finish
one of these in 0.2-2.5 s.
"""
import numpy as np
import pytest

import oracle_py as O
from conftest import bits
from llama_cpp_amd.payload import config_payload, make_payload, make_x
from test_gpu_forward import check_fast, fast_tolerance

pytestmark = pytest.mark.gpu

T = 512
SPMM_EXACT = {      # workload -> True when the default path at T=512 is a bit-exact SpMM kernel
    "cfg4_mistral_gate_coo_k128": True, "cfg4_mistral_gate_block16_k128": True, "cfg4_mistral_down_coo_k128": True,
    "cfg4_mistral_down_block16_k128": True, "cfg2_gemma4b_gate_coo_k128": True, "cfg3_mistral_down_block32_k1536": False,
}
_cache = {}


def case(cfg):
    """payload, x (T tokens), oracle output — computed once per workload for the whole module"""
    if cfg not in _cache:
        _cache.clear()                                    # one workload resident at a time (cfg3: 13 MB payload + 2 x 29 MB)
        p = make_payload(6912, 1152, scheme="coo", K=128, base=True, seed=99) if cfg == "coo_tt4_6912x1152" else config_payload(cfg)
        x = make_x(p.n_in, T, seed=17)
        _cache[cfg] = (p, x, O.forward(p, x, nth=16))
    return _cache[cfg]


def token_sample(n=6):
    return np.unique(np.concatenate([[0, 1, T - 1], np.random.default_rng(5).integers(0, T, n)]))


@pytest.mark.parametrize("cfg", list(SPMM_EXACT))
def test_prefill_T512_default_path(hip, cfg):
    p, x, y_ref = case(cfg)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x)
    assert w.nonfinite_count() == 0
    # 512 tokens are above every Mistral payload class's crossover (fold_min_tokens, csrc/sdh_api.hip): sdh_forward takes the
    # folded f16 contraction by itself (its bound: test_prefill_T512_mfma_fold); SDH_FWD_NO_MFMA keeps the fp32 forms,
    # checked here: the tiled SpMM bit for bit, the LDS rows kernel (rows of 512 entries and more) within the fast bound
    # (a Gemma gate, 2560 inputs in blocks of 4096, has no permuted form — its blocks are 1.6 x its inputs: q-order kernel)
    folded = w.fold_layout() != 0
    assert folded, (cfg, w.fold_layout())
    assert bool(w.fold_layout() & 2) == cfg.startswith(("cfg4_mistral", "cfg3_mistral")), (cfg, w.fold_layout())
    if folded:
        assert np.array_equal(bits(y), bits(w.forward_host(x, hip.FWD_MFMA_FOLD)))
        y = w.forward_host(x, hip.FWD_NO_MFMA)
        assert not np.array_equal(bits(y), bits(w.forward_host(x, hip.FWD_MFMA_FOLD)))
    if SPMM_EXACT[cfg]:
        assert np.array_equal(bits(y), bits(y_ref)), f"{cfg}: max abs diff {np.abs(y - y_ref).max()}"
        # below the crossover the default IS the exact form
        assert np.array_equal(bits(w.forward_host(x[:64])), bits(y_ref[:64]))
    else:
        assert np.abs(y - y_ref).max() <= 1e-4 * np.abs(y_ref).max()          # every element, coarse
        sel = token_sample()
        check_fast(p, x[sel], y[sel], y_ref[sel])                             # sampled tokens, the stated bound
    # the per-token kernels at full size
    y2 = w.forward_host(x, hip.FWD_NO_SPMM)
    assert np.abs(y2 - y_ref).max() <= 1e-4 * np.abs(y_ref).max()
    sel = token_sample()
    check_fast(p, x[sel], y2[sel], y_ref[sel])
    w.close()


def test_prefill_T512_tt4_tile(hip):
    """COO with 4864 < n_in <= 9728 takes the TT=4 tile (a Gemma-3 1B ffn_down as a COO payload)."""
    p, x, y_ref = case("coo_tt4_6912x1152")
    w = hip.SeedDeltaWeight(p)
    assert np.array_equal(bits(w.forward_host(x, hip.FWD_NO_MFMA)), bits(y_ref))   # (the default at 512 tokens is the folded form)
    # ragged: a last tile with 1, 2, 3 tokens
    for t in (509, 510, 511):
        assert np.array_equal(bits(w.forward_host(x[:t], hip.FWD_NO_MFMA)), bits(y_ref[:t])), t
    assert np.array_equal(bits(w.forward_host(x[:150])), bits(y_ref[:150]))         # below the crossover: exact by default
    w.close()


def test_prefill_ctx2048_as_four_ubatches(hip):
    """cfg2 at ctx 2048: the caller hands the op 4 ubatches of 512 (n_ubatch); each must equal the oracle and
    the concatenation must not depend on how the tokens were batched."""
    p, x, y_ref = case("cfg2_gemma4b_gate_coo_k128")
    w = hip.SeedDeltaWeight(p)
    x4 = make_x(p.n_in, 2048, seed=23)
    ref4 = O.forward(p, x4, nth=16)
    parts = [w.forward_host(x4[i:i + 512], hip.FWD_NO_MFMA) for i in range(0, 2048, 512)]       # the fp32 form (512 tokens fold by default)
    assert np.array_equal(bits(np.concatenate(parts)), bits(ref4))
    assert np.array_equal(bits(w.forward_host(x4, hip.FWD_NO_MFMA)), bits(ref4))               # one call of 2048
    # the default dispatch: folded (q order), and batching still does not matter
    yd = w.forward_host(x4)
    assert w.fold_layout() & 3 == 1
    assert np.array_equal(bits(np.concatenate([w.forward_host(x4[i:i + 512]) for i in range(0, 2048, 512)])), bits(yd))
    assert np.abs(yd - ref4).max() <= 2e-2 * np.abs(ref4).max()
    w.close()


@pytest.mark.parametrize("cfg", ["cfg4_mistral_gate_coo_k128", "cfg4_mistral_gate_block16_k128", "cfg4_mistral_down_coo_k128",
                                 "cfg4_mistral_down_block16_k128", "cfg2_gemma4b_gate_coo_k128"])
def test_prefill_T512_w0_mfma(hip, cfg):
    """SDH_FWD_W0_MFMA at Mistral / Gemma dims: W0 tiles regenerated from (G, inv) and contracted on MFMA with
    f16 operands and fp32 accumulation. Each product carries <= 2 * 2^-11 relative error, so per output
        |y - y_ref| <= 1.5e-3 * |rs| * sum_i |W0[o,i] * x[i]|  +  the fast-path tolerance of the residual.
    Checked on every element against a coarse bound and on sampled (row, token) pairs against the stated one
    (the dense |W0| of 14336 x 4096 does not fit a test's memory budget)."""
    p, x, y_ref = case(cfg)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x, hip.FWD_W0_MFMA)
    assert np.isfinite(y).all()
    assert np.abs(y - y_ref).max() <= 2e-2 * np.abs(y_ref).max()
    rows = np.unique(np.random.default_rng(3).integers(0, p.n_out, 96))
    sel = token_sample(4)
    xs = np.abs(x[sel].astype(np.float64))
    tol = fast_tolerance(p, x[sel], y_ref[sel])
    worst = 0.0
    for o in rows:
        W0 = np.abs(O.w0_tile(p, int(o), int(o) + 1, 0, p.n_in).astype(np.float64))[0]
        rs = abs(float(p.row_scale[o])) if p.row_scale is not None else 1.0
        bound = 1.5e-3 * rs * (xs @ W0) + tol[:, o]
        err = np.abs(y[sel, o].astype(np.float64) - y_ref[sel, o].astype(np.float64))
        assert np.all(err <= bound), (cfg, int(o), float(np.max(err / bound)))
        worst = max(worst, float(np.max(err / bound)))
    assert worst > 1e-4          # the MFMA path really ran (it is not bit-exact)
    w.close()


def fold_bound(p, x, y_ref, rows, sel):
    """per sampled (token, row): 1.5e-3 * |rs| * sum_i (|W0[o,i]| + |D[o,i]|) * |x[i]| — f16 rounding of x, of W0 and of
    the sum W0 + D (each <= 2^-11 relative) — plus the fast path's fp32 ordering tolerance"""
    xs = np.abs(x[sel].astype(np.float64))
    tol = fast_tolerance(p, x[sel], y_ref[sel])                  # 2e-5 * |rs| * (sum |val x| + |W0 x|)
    out = {}
    for o in rows:
        W0 = np.abs(O.w0_tile(p, int(o), int(o) + 1, 0, p.n_in).astype(np.float64))[0]
        rs = abs(float(p.row_scale[o])) if p.row_scale is not None else 1.0
        out[int(o)] = 1.5e-3 * rs * (xs @ W0) + 75.0 * tol[:, o]
    return out


@pytest.mark.parametrize("cfg", list(SPMM_EXACT))
def test_prefill_T512_mfma_fold(hip, cfg):
    """SDH_FWD_MFMA_FOLD at the real dims: ONE dense contraction per tensor, the residual added to the W0 tile in LDS
    (src/llama-seeddelta.cpp:1088-1175 evaluated as (W0 + D) x). f16 operands, fp32 accumulation."""
    p, x, y_ref = case(cfg)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x, hip.FWD_MFMA_FOLD)
    assert w.nonfinite_count() == 0 and np.isfinite(y).all()
    assert np.abs(y - y_ref).max() <= 2e-2 * np.abs(y_ref).max()
    rows = np.unique(np.concatenate([[0, p.n_out - 1], np.random.default_rng(3).integers(0, p.n_out, 96)]))
    sel = token_sample(4)
    bound = fold_bound(p, x, y_ref, rows, sel)
    worst = 0.0
    for o in rows:
        err = np.abs(y[sel, o].astype(np.float64) - y_ref[sel, o].astype(np.float64))
        assert np.all(err <= bound[int(o)]), (cfg, int(o), float(np.max(err / bound[int(o)])))
        worst = max(worst, float(np.max(err / bound[int(o)])))
    assert worst > 1e-4          # the folded path really ran (it is not bit-exact)
    # a second call reuses the re-laid-out payload; ragged token counts (a partly filled last tile, fewer than a tile)
    for t in (300, 7):
        yt = w.forward_host(x[:t], hip.FWD_MFMA_FOLD)
        assert np.array_equal(bits(yt), bits(y[:t])), t          # a token's result does not depend on its batch
    w.close()


@pytest.mark.parametrize("dims", [(256, 640), (640, 256), (200, 1000), (1000, 200), (1024, 1024)])
@pytest.mark.parametrize("scheme,K,block", [("coo", 24, 16), ("block", 64, 16), ("block", 96, 32)])
def test_mfma_fold_small_shapes(hip, dims, scheme, K, block):
    """Odd dims (n_in not a multiple of the 64-wide slab, n_out not a multiple of the 128-row tile, a partly filled last
    base block), d1 / d3 present, bad indices, duplicates: every element against the oracle with the stated f16 bound."""
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme=scheme, K=K, block=block, base=True, seed=n_in + K, d13_random=True)
    rng = np.random.default_rng(4)
    if scheme == "coo":
        p.d_idx[::7, 0] = -1
        p.d_idx[1::9, 1] = n_in + 3                         # out of range: skipped (:693)
        p.d_idx[2::5, 3] = p.d_idx[2::5, 2]                 # duplicates sum
    else:
        p.b_idx[::6, 0] = -1
        p.b_idx[1::8, 1] = (n_in + block - 1) // block - 1  # the ragged last block (clipped to n_in) where n_in % block != 0
    T = 37
    x = make_x(n_in, T, seed=3)
    y_ref = O.forward(p, x, nth=4)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x, hip.FWD_MFMA_FOLD)
    rows = np.arange(n_out)
    sel = np.arange(T)
    bound = fold_bound(p, x, y_ref, rows, sel)
    for o in rows:
        err = np.abs(y[:, o].astype(np.float64) - y_ref[:, o].astype(np.float64))
        assert np.all(err <= bound[int(o)]), (dims, scheme, int(o), float(np.max(err / bound[int(o)])))
    assert w.nonfinite_count() == 0
    assert w.fold_layout() & 3 == 1                         # d1 / d3 present: the q-order kernel
    w.close()


@pytest.mark.parametrize("dims", [(256, 640), (640, 256), (200, 1000), (1000, 200), (1024, 1024), (128, 128), (2048, 384)])
@pytest.mark.parametrize("scheme,K,block", [("coo", 24, 16), ("block", 64, 16), ("block", 96, 32), ("coo", 120, 16)])
def test_mfma_fold_permuted_order_small_shapes(hip, dims, scheme, K, block):
    """The same odd dims with d1 = d3 = 1 (what the builder writes): the contraction runs in the permuted order j = inv_b[q]
    (w0p_mfma_kernel: x permuted and tiled by the pre-pass, W0 tile built from one segment of G_b, entries bucketed by j).
    Covers tall and wide with a partly filled last base block (columns of the block without an input multiply zeros), L =
    128 (two slabs: only the loop's tail runs), duplicates (the atomic variant), bad indices, and with K = 120 of 128-256
    inputs buckets of more than 2048 entries (the overflow variant). Every element against the oracle."""
    n_in, n_out = dims
    p = make_payload(n_in, n_out, scheme=scheme, K=K, block=block, base=True, seed=n_in + K + 1)
    if scheme == "coo":
        p.d_idx[::7, 0] = -1
        p.d_idx[1::9, 1] = n_in + 3                         # out of range: skipped (:693)
        if K < 100:
            p.d_idx[2::5, 3] = p.d_idx[2::5, 2]             # duplicates sum
    else:
        p.b_idx[::6, 0] = -1
        p.b_idx[1::8, 1] = (n_in + block - 1) // block - 1
    for T in (37, 300):
        x = make_x(n_in, T, seed=3)
        y_ref = O.forward(p, x, nth=4)
        w = hip.SeedDeltaWeight(p)
        y = w.forward_host(x, hip.FWD_MFMA_FOLD)
        rows = np.arange(n_out)
        sel = np.arange(0, T, 7)
        bound = fold_bound(p, x, y_ref, rows, sel)
        worst = 0.0
        for o in rows:
            err = np.abs(y[sel, o].astype(np.float64) - y_ref[sel, o].astype(np.float64))
            assert np.all(err <= bound[int(o)]), (dims, scheme, T, int(o), float(np.max(err / bound[int(o)])))
            worst = max(worst, float(np.max(err / bound[int(o)])))
        assert np.abs(y - y_ref).max() <= 2e-2 * np.abs(y_ref).max()
        assert worst > 1e-4                                 # not the bit-exact forms: the folded path ran
        assert w.nonfinite_count() == 0
        lay = w.fold_layout()
        assert lay & 2, lay                                 # ... in the permuted order
        if scheme == "coo" and K < 100:
            assert lay & 4, lay                             # the duplicates were seen
        if scheme == "coo" and K >= 100 and n_in <= 256:
            assert lay & 8, lay                             # buckets beyond the prefetch
        w.close()


def test_mfma_fold_token_counts_and_chunks(hip):
    """The folded contraction across its token regimes on one payload: token tiles of 128 (up to 256 tokens), of 256, and a
    call longer than one workspace chunk (2048 tokens): a token's result never depends on the batch it came in."""
    p = make_payload(1024, 2816, scheme="coo", K=32, base=True, seed=77)
    x = make_x(1024, 2500, seed=9)
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x, hip.FWD_MFMA_FOLD)                   # 2048 + 452
    assert w.fold_layout() & 2 and w.nonfinite_count() == 0
    for t0, t1 in ((0, 5), (5, 133), (100, 356), (300, 1324), (2047, 2500), (0, 2049)):
        yt = w.forward_host(x[t0:t1], hip.FWD_MFMA_FOLD)
        assert np.array_equal(bits(yt), bits(y[t0:t1])), (t0, t1)
    y_ref = O.forward(p, x[::50], nth=4)
    assert np.abs(y[::50] - y_ref).max() <= 2e-2 * np.abs(y_ref).max()
    # by default: exact below the class's crossover (COO: 160 tokens), folded from there on
    assert np.array_equal(bits(w.forward_host(x[:159])), bits(O.forward(p, x[:159], nth=4)))
    assert np.array_equal(bits(w.forward_host(x[:160])), bits(y[:160]))
    w.close()


def test_mfma_fold_counts_nonfinite_outputs(hip):
    p = make_payload(512, 768, scheme="coo", K=16, base=True, seed=5)
    x = make_x(512, 12, seed=1)
    x[3, 10] = np.inf
    w = hip.SeedDeltaWeight(p)
    y = w.forward_host(x, hip.FWD_MFMA_FOLD | hip.FWD_ZERO_NONFINITE)
    y_ref = O.forward(p, x, nth=4)                          # the reference guard writes 0 without a dense weight (:373-380)
    # an infinite input reaches every output of its token through the dense base: the whole row is non-finite
    assert w.nonfinite_count() == p.n_out
    assert np.all(y[3] == 0.0) and np.all(y_ref[3] == 0.0) and np.isfinite(y).all()
    ok = np.arange(12) != 3
    assert np.abs(y[ok] - y_ref[ok]).max() <= 2e-2 * np.abs(y_ref[ok]).max()
    w.close()
