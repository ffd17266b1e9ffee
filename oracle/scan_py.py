"""ctypes bindings for the scan oracle (oracle/liboracle.so, sdso_*) and, when built, the reference's
own builder evaluation code (oracle/_ref/libsdbuildref.so, sdbref_*), behind one interface so tests
can run the same case through both.

TEST INFRASTRUCTURE ONLY — imported by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg. The product package never imports this module.

A dense weight is passed as ``(array, kind)``: float32 / float16 arrays of shape (n_out, n_in), or a
uint8 array of raw block_q8_0 rows of shape (n_out, n_in // 32 * 34).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from oracle_py import HERE, SdoTensor, oracle_lib, tensor_of

BUILDREF_SO = os.path.join(HERE, "_ref", "libsdbuildref.so")


class BaseArrays(C.Structure):
    _fields_ = [("L", C.c_int64), ("B", C.c_int64), ("d1", C.c_void_p), ("d2", C.c_void_p), ("d3", C.c_void_p),
                ("h2", C.c_void_p), ("perm1", C.c_void_p), ("perm1_inv", C.c_void_p)]


class SdsoRng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("pos", C.c_int32), ("have_saved", C.c_int32), ("saved", C.c_float)]


class Fit:
    """Result of the base fit: [B][L] arrays."""

    def __init__(self, L, B, d2, h2, perm1, perm1_inv):
        self.L, self.B, self.d2, self.h2, self.perm1, self.perm1_inv = L, B, d2, h2, perm1, perm1_inv

    def arrays(self) -> BaseArrays:
        a = BaseArrays()
        a.L, a.B = self.L, self.B
        a.d1 = None                               # D1 = D3 = I in the fit (sd_eval.cpp:450-452)
        a.d3 = None
        a.d2, a.h2 = self.d2.ctypes.data, self.h2.ctypes.data
        a.perm1, a.perm1_inv = self.perm1.ctypes.data, self.perm1_inv.ctypes.data
        return a


def w_tensor(W: np.ndarray) -> SdoTensor:
    if W.dtype == np.uint8:
        n_out, row = W.shape
        return tensor_of(W, q8_0_shape=(n_out, row // 34 * 32))
    return tensor_of(W)


def w_dims(W: np.ndarray):
    if W.dtype == np.uint8:
        return W.shape[1] // 34 * 32, W.shape[0]
    return W.shape[1], W.shape[0]


def fit_dims(n_in, n_out):
    L = 1
    while L < min(n_in, n_out):
        L <<= 1
    return L, (max(n_in, n_out) + L - 1) // L


def buildref_available() -> bool:
    return os.path.exists(BUILDREF_SO)


_ref = None


def _reflib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(BUILDREF_SO)
        _ref.sdbref_rng_new.restype = C.c_void_p
        _ref.sdbref_rng_new.argtypes = [C.c_uint32]
        _ref.sdbref_rng_free.argtypes = [C.c_void_p]
        _ref.sdbref_rng_u32.restype = C.c_uint32
        _ref.sdbref_rng_u32.argtypes = [C.c_void_p]
        _ref.sdbref_topk_abs.restype = C.c_int64
        _ref.sdbref_topk_blocks.restype = C.c_int64
    return _ref


def _p(a: Optional[np.ndarray]):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    """Either the oracle ('oracle') or the compiled reference ('ref'), with its own RNG stream."""

    def __init__(self, which: str, seed: int):
        self.which = which
        if which == "ref":
            self.lib = _reflib()
            self.rng = C.c_void_p(self.lib.sdbref_rng_new(C.c_uint32(seed)))
            self.pre = "sdbref_"
        else:
            self.lib = oracle_lib()
            self._state = SdsoRng()
            self.lib.sdso_rng_seed(C.byref(self._state), C.c_uint32(seed))
            self.rng = C.byref(self._state)
            self.pre = "sdso_"

    def close(self):
        if self.which == "ref" and self.rng:
            self.lib.sdbref_rng_free(self.rng)
            self.rng = None

    def _fn(self, name, restype=None):
        f = getattr(self.lib, self.pre + name)
        f.restype = restype
        return f

    # ---- RNG ----
    def fork(self) -> "Engine":
        """A second oracle engine at the same stream position (peek at draws without consuming them)."""
        assert self.which == "oracle"
        e = Engine("oracle", 0)
        C.memmove(C.byref(e._state), C.byref(self._state), C.sizeof(SdsoRng))
        return e

    def u32(self) -> int:
        return int(self._fn("rng_u32", C.c_uint32)(self.rng))

    def shuffle_iota_i64(self, n: int, first: int = 0) -> np.ndarray:
        out = np.arange(first, first + n, dtype=np.int64)
        if self.which == "ref":
            self.lib.sdbref_rng_shuffle_iota_i64(self.rng, C.c_int64(n), C.c_int64(first), _p(out))
        else:
            self.lib.sdso_shuffle_i64(self.rng, _p(out), C.c_int64(n))
        return out

    def shuffle_iota_i32(self, n: int) -> np.ndarray:
        out = np.arange(n, dtype=np.int32)
        if self.which == "ref":
            self.lib.sdbref_rng_shuffle_iota_i32(self.rng, C.c_int64(n), _p(out))
        else:
            self.lib.sdso_shuffle_i32(self.rng, _p(out), C.c_int64(n))
        return out

    def normal(self, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.float32)
        if self.which == "ref":
            self.lib.sdbref_rng_normal_f32(self.rng, C.c_int64(n), _p(out))
        else:
            self.lib.sdso_normal_f32(self.rng, C.c_int64(n), _p(out))
        return out

    # ---- pieces ----
    def read_row(self, W, row):
        n_in, _ = w_dims(W)
        out = np.empty(n_in, dtype=np.float32)
        t = w_tensor(W)
        self._fn("read_row")(C.byref(t), C.c_int64(row), _p(out))
        return out

    def topk_abs(self, v, K, w_scale=None):
        v, w_scale = _f32(v), _f32(w_scale)
        out = np.empty(min(K, len(v)), dtype=np.int32)
        n = self._fn("topk_abs", C.c_int64)(_p(v), _p(w_scale), C.c_int64(len(v)), C.c_int64(K), _p(out))
        return out[:n]

    def topk_blocks(self, v, block, keep, w_scale=None):
        v, w_scale = _f32(v), _f32(w_scale)
        out = np.empty(max(1, min(keep, (len(v) + block - 1) // block)), dtype=np.int32)
        n = self._fn("topk_blocks", C.c_int64)(_p(v), _p(w_scale), C.c_int64(len(v)), C.c_int64(block),
                                               C.c_int64(keep), _p(out))
        return out[:n]

    def fit_base(self, W, max_samples=2048, perm_trials=4, w_scale=None) -> Fit:
        n_in, n_out = w_dims(W)
        L, B = fit_dims(n_in, n_out)
        d2, h2 = np.zeros((B, L), np.float32), np.zeros((B, L), np.float32)
        p1, inv = np.zeros((B, L), np.int32), np.zeros((B, L), np.int32)
        t, w_scale = w_tensor(W), _f32(w_scale)
        if self.which == "ref":
            Lo, Bo = C.c_int64(), C.c_int64()
            self.lib.sdbref_fit_base(C.byref(t), C.c_int64(max_samples), _p(w_scale), C.c_int(perm_trials), self.rng,
                                     C.byref(Lo), C.byref(Bo), _p(d2), _p(h2), _p(p1), _p(inv))
            assert (Lo.value, Bo.value) == (L, B)
        else:
            self.lib.sdso_fit_base(C.byref(t), C.c_int64(max_samples), _p(w_scale), C.c_int(perm_trials), self.rng,
                                   _p(d2), _p(h2), _p(p1), _p(inv))
        return Fit(L, B, d2, h2, p1, inv)

    def apply_base_block(self, fit: Fit, b: int, x: np.ndarray) -> np.ndarray:
        y = np.empty(fit.L, dtype=np.float32)
        a = fit.arrays()
        self._fn("apply_base_block")(C.byref(a), C.c_int64(b), _p(_f32(x)), _p(y))
        return y

    def eval_w(self, W, fit, scheme, idx, val, K, block, row_scale, eval_cols, w_scale=None):
        out = np.zeros(5)
        t = w_tensor(W)
        a = fit.arrays() if fit is not None else None
        self._fn("eval_w")(C.byref(t), C.byref(a) if a is not None else None, C.c_int(scheme), _p(idx), _p(val),
                           C.c_int64(K), C.c_int64(block), _p(_f32(row_scale)), _p(_f32(w_scale)),
                           C.c_int64(eval_cols), self.rng, _p(out))
        return out

    def eval_x(self, W, fit, scheme, idx, val, K, block, row_scale, eval_cols, eval_x, x_scale=None):
        out = np.zeros(5)
        t = w_tensor(W)
        a = fit.arrays() if fit is not None else None
        self._fn("eval_x")(C.byref(t), C.byref(a) if a is not None else None, C.c_int(scheme), _p(idx), _p(val),
                           C.c_int64(K), C.c_int64(block), _p(_f32(row_scale)), _p(_f32(x_scale)),
                           C.c_int64(eval_cols), C.c_int64(eval_x), self.rng, _p(out))
        return out

    def ffn_proxy(self, kind, Wg, Wu, Wd, idx, val, K, eval_x, eval_cols, seed, fit=None):
        out = np.zeros(8)
        tg, tu, td = w_tensor(Wg), w_tensor(Wu), w_tensor(Wd)
        a = fit.arrays() if fit is not None else None
        ok = self._fn("ffn_proxy", C.c_int)(C.c_int(kind), C.byref(tg), C.byref(tu), C.byref(td), _p(idx), _p(val),
                                            C.byref(a) if a is not None else None, C.c_int64(K), C.c_int64(eval_x),
                                            C.c_int64(eval_cols), C.c_int(seed), _p(out))
        return bool(ok), out

    def cost(self, L, B, n_in, n_out, K, row_scale):
        out = np.zeros(8)
        self._fn("cost")(C.c_int64(L), C.c_int64(B), C.c_int64(n_in), C.c_int64(n_out), C.c_int64(K),
                         C.c_int(1 if row_scale else 0), _p(out))
        return out


def ref_imatrix_scale(imatrix: np.ndarray, n_in: int, eps: float, power: float):
    """The reference's sd_make_imatrix_sqrt_scale (needs libsdbuildref.so)."""
    lib = _reflib()
    row = np.ascontiguousarray(imatrix, dtype=np.float32)
    out = np.empty(n_in, dtype=np.float32)
    lib.sdbref_imatrix_scale.restype = C.c_int
    ok = lib.sdbref_imatrix_scale(_p(row), C.c_int64(row.size), C.c_int64(n_in), C.c_float(eps), C.c_float(power), _p(out))
    return out if ok else None


# ---- oracle-only entry points (no compiled-reference counterpart) --------------------------------------
def select_rows(W, fit: Optional[Fit], scheme: int, K_or_keep: int, block: int, row_scale: bool, w_scale=None,
                nth: int = 8):
    """run_trial column loop -> (idx int32, val float32, row_scale float32 or None)."""
    n_in, n_out = w_dims(W)
    lib = oracle_lib()
    idx = np.empty((n_out, K_or_keep), dtype=np.int32)
    val = np.empty((n_out, K_or_keep, block) if scheme == 1 else (n_out, K_or_keep), dtype=np.float32)
    rs = np.empty(n_out, dtype=np.float32) if row_scale else None
    t = w_tensor(W)
    a = fit.arrays() if fit is not None else None
    lib.sdso_select_rows(C.byref(t), C.byref(a) if a is not None else None, _p(_f32(w_scale)), C.c_int(scheme),
                         C.c_int64(K_or_keep), C.c_int64(block), _p(idx), _p(val), _p(rs), C.c_int(nth))
    return idx, val, rs


def select_rows_list(W, fit: Optional[Fit], scheme: int, K_or_keep: int, block: int, row_scale: bool, rows, w_scale=None,
                     out=None):
    """The run_trial column loop for the listed rows only; the [n_out] arrays (zero elsewhere, or `out` from an
    earlier call) receive them in the rows' own slots. For full-size tests: the evaluators read nothing else."""
    n_in, n_out = w_dims(W)
    lib = oracle_lib()
    if out is None:
        idx = np.zeros((n_out, K_or_keep), dtype=np.int32)
        val = np.zeros((n_out, K_or_keep, block) if scheme == 1 else (n_out, K_or_keep), dtype=np.float32)
        rs = np.ones(n_out, dtype=np.float32) if row_scale else None
    else:
        idx, val, rs = out
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    t = w_tensor(W)
    a = fit.arrays() if fit is not None else None
    lib.sdso_select_rows_list(C.byref(t), C.byref(a) if a is not None else None, _p(_f32(w_scale)), C.c_int(scheme),
                              C.c_int64(K_or_keep), C.c_int64(block), _p(idx), _p(val), _p(rs), _p(rows), C.c_int64(len(rows)))
    return idx, val, rs


def eval_w_rows(W, fit, scheme, idx, val, K, block, row_scale, rows, w_scale=None):
    lib = oracle_lib()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    per_row, out = np.zeros((len(rows), 3)), np.zeros(5)
    t = w_tensor(W)
    a = fit.arrays() if fit is not None else None
    lib.sdso_eval_w_rows(C.byref(t), C.byref(a) if a is not None else None, C.c_int(scheme), _p(idx), _p(val),
                         C.c_int64(K), C.c_int64(block), _p(_f32(row_scale)), _p(_f32(w_scale)), _p(rows),
                         C.c_int64(len(rows)), _p(per_row), _p(out))
    return per_row, out


def eval_x_rows(W, fit, scheme, idx, val, K, block, row_scale, rows, xs):
    lib = oracle_lib()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    xs = np.ascontiguousarray(xs, dtype=np.float32)
    y, out = np.zeros((xs.shape[0], len(rows), 2)), np.zeros(5)
    t = w_tensor(W)
    a = fit.arrays() if fit is not None else None
    lib.sdso_eval_x_rows(C.byref(t), C.byref(a) if a is not None else None, C.c_int(scheme), _p(idx), _p(val),
                         C.c_int64(K), C.c_int64(block), _p(_f32(row_scale)), _p(rows), C.c_int64(len(rows)),
                         _p(xs), C.c_int64(xs.shape[0]), _p(y), _p(out))
    return y, out


def percentile(v, p):
    lib = oracle_lib()
    lib.sdso_percentile.restype = C.c_double
    v = np.ascontiguousarray(v, dtype=np.float64)
    return float(lib.sdso_percentile(_p(v), C.c_int64(len(v)), C.c_double(p)))


def delta_clamp(W, val: np.ndarray, clamp: float):
    """--delta-norm-clamp-down on the residual values of one tensor (in place) -> scale applied."""
    lib = oracle_lib()
    lib.sdso_tensor_l2_norm.restype = C.c_double
    lib.sdso_delta_clamp.restype = C.c_double
    t = w_tensor(W)
    w_l2 = lib.sdso_tensor_l2_norm(C.byref(t))
    assert val.dtype == np.float32 and val.flags.c_contiguous
    return float(lib.sdso_delta_clamp(_p(val), C.c_int64(val.size), C.c_double(w_l2), C.c_double(clamp)))
