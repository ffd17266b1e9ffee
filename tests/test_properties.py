"""Properties that need no compiled reference, for the pieces of the builder whose translation unit
(tools/seeddelta-build/sd_build.cpp, sd_policy_resolve.cpp) cannot be built here and so cannot pin parity
(DESIGN.md §1 'Oracle', §7 'Parity unpinned'):

  * row_scale (sd_build.cpp:586-626, :649-683) is the least-squares scale: s = <w, w_hat> / <w_hat, w_hat>
    minimises ||w - s*w_hat||_2 — checked against a float64 numpy solve, for the oracle here and for the HIP path
    in the gpu-marked tests;
  * delta_norm_clamp_down (sd_build.cpp:29-46, :705-720): values are scaled by clamp*||W||_2/||delta||_2 when the
    residual's norm exceeds clamp*||W||_2, and left alone otherwise;
  * the policy hash (sd_policy_resolve.cpp:9-21) is the FNV-1a 64 recurrence with the builder's offset basis
    1469598103934665603 (the published basis is ...656037): vectors worked out by hand from the recurrence
    h = ((h ^ byte) * 1099511628211) mod 2^64.
"""
import numpy as np
import pytest

import scan_py as S
from llama_cpp_amd import build as B, scan as SC
from llama_cpp_amd.dense import dequantize_q8_0, make_dense


def reconstruct(Wf, fit, scheme, idx, val, block):
    """w_hat (n_out, n_in) in float64 from the base table h2[(p ^ inv[i])] and the selected residual"""
    n_out, n_in = Wf.shape
    wh = np.zeros((n_out, n_in))
    if fit is not None:
        L = fit.L
        h2, inv = fit.h2.astype(np.float64), fit.perm1_inv
        if n_out >= n_in:                                   # tall: block = o // L, row p = o % L
            for b in range(fit.B):
                o0, o1 = b * L, min(n_out, (b + 1) * L)
                if o0 >= o1:
                    break
                p = np.arange(o0, o1)[:, None] - o0
                wh[o0:o1] = h2[b][(p ^ inv[b][None, :n_in]) & (L - 1)]
        else:                                               # wide: block = i // L
            for b in range(fit.B):
                i0, i1 = b * L, min(n_in, (b + 1) * L)
                if i0 >= i1:
                    break
                o = np.arange(n_out)[:, None]
                wh[:, i0:i1] = h2[b][(o ^ inv[b][None, :i1 - i0]) & (L - 1)]
    for o in range(n_out):
        if scheme == 1:
            for bi in range(idx.shape[1]):
                blk = int(idx[o, bi])
                if blk < 0 or blk * block >= n_in:
                    continue
                n = min(block, n_in - blk * block)
                wh[o, blk * block: blk * block + n] += val[o, bi, :n]
        else:
            ok = (idx[o] >= 0) & (idx[o] < n_in)
            np.add.at(wh[o], idx[o][ok], val[o][ok].astype(np.float64))
    return wh


def lsq_scale(Wf, wh):
    nn = (wh * wh).sum(axis=1)
    return np.where(nn > 1e-30, (Wf.astype(np.float64) * wh).sum(axis=1) / np.where(nn > 1e-30, nn, 1.0), 1.0)


def check_row_scale(Wf, wh, rs):
    want = lsq_scale(Wf, wh)
    assert np.all(np.abs(rs - want) <= 2.0 * np.spacing(np.abs(want).astype(np.float32))), float(np.abs(rs - want).max())
    # and it really is the minimiser: nudging the scale either way does not lower the residual
    r0 = ((Wf - rs[:, None].astype(np.float64) * wh) ** 2).sum(axis=1)
    for eps in (1e-3, -1e-3):
        r1 = ((Wf - (rs[:, None].astype(np.float64) * (1 + eps)) * wh) ** 2).sum(axis=1)
        assert np.all(r1 >= r0 * (1 - 1e-9))


CASES = [((256, 704), "f32", 0, 16, 16, True), ((704, 256), "q8_0", 0, 24, 16, True), ((256, 704), "q8_0", 1, 32, 16, True),
         ((512, 512), "f32", 1, 64, 32, False), ((704, 256), "f32", 0, 8, 16, False)]


@pytest.mark.parametrize("dims,kind,scheme,K,block,base", CASES)
def test_oracle_row_scale_is_the_least_squares_scale(dims, kind, scheme, K, block, base):
    n_in, n_out = dims
    W = make_dense(n_in, n_out, kind, seed=K + n_in)
    Wf = dequantize_q8_0(W) if kind == "q8_0" else W.astype(np.float32)
    fit = S.Engine("oracle", 7).fit_base(W, 256, 2) if base else None
    keep = (K + block - 1) // block if scheme else K
    idx, val, rs = S.select_rows(W, fit, scheme, keep, block, True, nth=4)
    check_row_scale(Wf.astype(np.float64), reconstruct(Wf, fit, scheme, idx, val, block), rs)


def test_policy_hash_vectors():
    assert B.policy_hash(b"") == ""
    assert B.policy_hash(b"a") == "fnv1a64:44bd8ad473cd9906"
    assert B.policy_hash(b"seeddelta") == "fnv1a64:768f2c833ad05968"
    assert B.policy_hash(b'{"version": 1}\n') == "fnv1a64:9a16775280afa51a"
    # first step by hand: (basis ^ 0x61) * prime mod 2^64
    assert ((1469598103934665603 ^ 0x61) * 1099511628211) % (1 << 64) == 0x44bd8ad473cd9906


def test_oracle_delta_clamp_scale():
    W = make_dense(704, 256, "f32", seed=3)
    idx, val, _ = S.select_rows(W, None, 0, 32, 16, False, nth=2)
    wn, dn = np.sqrt((W.astype(np.float64) ** 2).sum()), np.sqrt((val.astype(np.float64) ** 2).sum())
    for clamp in (0.05, 0.2, 10.0):
        out = val.copy()
        scale = S.delta_clamp(W, out, clamp)
        want = min(1.0, clamp * wn / dn)
        assert abs(scale - want) <= 1e-12 * want
        assert np.all(np.abs(out - (val.astype(np.float64) * want).astype(np.float32)) <= np.spacing(np.abs(out)))


# ---- the HIP path ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dims,kind,scheme,K,block,base", CASES)
def test_gpu_row_scale_is_the_least_squares_scale(hip, dims, kind, scheme, K, block, base):
    n_in, n_out = dims
    W = make_dense(n_in, n_out, kind, seed=K + n_in)
    Wf = dequantize_q8_0(W) if kind == "q8_0" else W.astype(np.float32)
    p = SC.ScanParams(scheme="block" if scheme else "coo", K=K, block=block, base=base, row_scale=True, base_max_samples=256,
                      base_perm_trials=2, eval_cols=8, eval_x=2, seed=7)
    s = SC.ScanSession(7)
    res = s.tensor(SC.DenseWeight(W), p, "ffn_down" if n_in > n_out else "ffn_gate", 0)
    out = s.fetch()
    fit = S.Fit(res["L"], res["B"], out["d2"], out["h2"], out["perm1"], out["perm1_inv"]) if base else None
    check_row_scale(Wf.astype(np.float64), reconstruct(Wf, fit, scheme, out["idx"], out["val"], block), out["row_scale"])


@pytest.mark.gpu
def test_gpu_delta_clamp_scale(hip):
    W = make_dense(704, 256, "f32", seed=3)
    base = dict(scheme="coo", K=32, base=False, row_scale=False, eval_cols=8, eval_x=2, seed=5)
    s = SC.ScanSession(5)
    s.tensor(SC.DenseWeight(W), SC.ScanParams(**base), "ffn_down", 0)
    val0 = s.fetch()["val"].astype(np.float64)
    wn, dn = np.sqrt((W.astype(np.float64) ** 2).sum()), np.sqrt((val0 ** 2).sum())
    for clamp in (0.05, 0.2, 10.0):
        s2 = SC.ScanSession(5)
        res = s2.tensor(SC.DenseWeight(W), SC.ScanParams(delta_norm_clamp_down=clamp, **base), "ffn_down", 0)
        want = min(1.0, clamp * wn / dn)
        assert abs(res["delta_clamp_scale"] - want) <= 1e-12 * want
        val = s2.fetch()["val"]
        assert np.all(np.abs(val - (val0 * want).astype(np.float32)) <= np.spacing(np.abs(val)))
