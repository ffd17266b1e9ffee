"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when built, the reference's own
translation unit (oracle/_ref/libsdref.so).

TEST INFRASTRUCTURE ONLY — imported by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg. The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsdref.so")
SHIM_SO = os.path.join(HERE, "_ref", "libsdshim.so")   # same harness + ggml, op TU = llama.cpp_amd/shim

TYPE_OF_DTYPE = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.int16): 25, np.dtype(np.int32): 26}
SDO_Q8_0 = 8


class SdoTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int32), ("n_dims", C.c_int32),
                ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4)]


class SdoOpArgs(C.Structure):
    _fields_ = [(n, SdoTensor) for n in
                ("x", "idx", "val", "row_scale", "d1", "d2", "d3", "p1", "p2", "w_ref", "dst")]


def tensor_of(arr: Optional[np.ndarray], q8_0_shape: Optional[tuple] = None) -> SdoTensor:
    """View a numpy array as a ggml-style tensor (ne[0] = fastest numpy axis)."""
    t = SdoTensor()
    if arr is None:
        return t
    t.data = arr.ctypes.data
    if q8_0_shape is not None:                      # raw uint8 rows of block_q8_0
        n_out, n_in = q8_0_shape
        t.type, t.n_dims = SDO_Q8_0, 2
        t.ne[:] = [n_in, n_out, 1, 1]
        row = (n_in // 32) * 34
        t.nb[:] = [34, row, row * n_out, row * n_out]
        return t
    t.type = TYPE_OF_DTYPE[arr.dtype]
    shape, strides = arr.shape[::-1], arr.strides[::-1]
    t.n_dims = len(shape)
    ne, nb = [1, 1, 1, 1], [0, 0, 0, 0]
    for i in range(len(shape)):
        ne[i], nb[i] = shape[i], strides[i]
    for i in range(len(shape), 4):
        nb[i] = nb[i - 1] * ne[i - 1]
    t.ne[:], t.nb[:] = ne, nb
    return t


def build_oracle(force: bool = False) -> None:
    """Compile liboracle.so (and _ref when /root/reference exists). Building the checker is not
    using it."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "sd_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, os.path.join(HERE, "liboracle.so")], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/src/llama-seeddelta.cpp") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE, "-j8", "ref"], stdout=subprocess.DEVNULL)


def build_shim() -> None:
    """Compile oracle/_ref/libsdshim.so (needs /root/reference for the ggml headers/objects and the
    built libseeddelta_hip.so to link against)."""
    if os.path.exists("/root/reference/src/llama-seeddelta.h"):
        subprocess.check_call(["make", "-C", HERE, "-j8", "shim"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def oracle_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _lib = C.CDLL(ORACLE_SO)
        _lib.sdo_run.argtypes = [C.c_int, C.POINTER(SdoOpArgs), C.c_int]
        _lib.sdo_run.restype = None
        _lib.sdo_fp16_to_fp32.argtypes = [C.c_uint16]
        _lib.sdo_fp16_to_fp32.restype = C.c_float
        _lib.sdo_fp32_to_fp16.argtypes = [C.c_float]
        _lib.sdo_fp32_to_fp16.restype = C.c_uint16
        _lib.sdo_nan_count.restype = C.c_int64
        _lib.sdo_hadamard.argtypes = [C.c_void_p, C.c_int64]
        _lib.sdo_w0_tables.argtypes = [C.POINTER(SdoTensor), C.POINTER(SdoTensor), C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_void_p]
        _lib.sdo_w0_elem.argtypes = [C.POINTER(SdoTensor), C.POINTER(SdoTensor), C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
        _lib.sdo_w0_elem.restype = C.c_float
        _lib.sdo_dequant_row_q8_0.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib.sdo_gelu_table.argtypes = [C.c_void_p]
        _lib.sdo_glu.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def ref_available() -> bool:
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.sdref_run.argtypes = [C.c_int, C.POINTER(SdoOpArgs), C.c_int]
        _ref.sdref_run.restype = C.c_int
        _ref.sdref_fp16_to_fp32.argtypes = [C.c_uint16]
        _ref.sdref_fp16_to_fp32.restype = C.c_float
        _ref.sdref_fp32_to_fp16.argtypes = [C.c_float]
        _ref.sdref_fp32_to_fp16.restype = C.c_uint16
        _bind_ffn(_ref)
    return _ref


REF_NATIVE_SO = os.path.join(HERE, "_ref", "libsdref_native.so")   # same library, ggml-cpu built -march=native (stock GGML_NATIVE)
_ref_native = None


def ref_native_available() -> bool:
    return os.path.exists(REF_NATIVE_SO)


def ref_native_lib():
    global _ref_native
    if _ref_native is None:
        _ref_native = C.CDLL(REF_NATIVE_SO)
        _bind_ffn(_ref_native)
    return _ref_native


def glu_native(act: int, g: np.ndarray, u: np.ndarray) -> np.ndarray:
    """act(g) * u by the reference's CPU backend built as a stock build builds it (SIMD bodies of vec.cpp / vec.h)."""
    g = np.ascontiguousarray(g, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.float32)
    y = np.empty_like(g)
    rc = ref_native_lib().sdref_glu(act, g.size, g.ctypes.data, u.ctypes.data, y.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"sdref_glu (native) failed: {rc}")
    return y


def _bind_ffn(lib):
    lib.sdref_glu.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sdref_glu.restype = C.c_int
    lib.sdref_ffn.argtypes = [C.POINTER(C.c_int), C.POINTER(SdoOpArgs), C.POINTER(SdoOpArgs), C.POINTER(SdoOpArgs),
                              C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.sdref_ffn.restype = C.c_int


_shim = None


def shim_available() -> bool:
    return os.path.exists(SHIM_SO)


def shim_lib():
    """The reference harness with the reference's op TU swapped for the HIP shim (needs a GPU at
    run time; libseeddelta_hip.so must already be loaded so torch's HIP runtime is the one used)."""
    global _shim
    if _shim is None:
        _shim = C.CDLL(SHIM_SO)
        _shim.sdref_run.argtypes = [C.c_int, C.POINTER(SdoOpArgs), C.c_int]
        _shim.sdref_run.restype = C.c_int
        _shim.llama_seeddelta_hip_release_all.restype = None
        _shim.llama_seeddelta_hip_nonfinite_total.restype = C.c_int64
        _bind_ffn(_shim)
        _shim.sdshim_ffn_node.argtypes = [C.POINTER(C.c_int), C.POINTER(SdoOpArgs), C.POINTER(SdoOpArgs), C.POINTER(SdoOpArgs),
                                          C.c_int, C.c_int, C.c_int, C.c_void_p]
        _shim.sdshim_ffn_node.restype = C.c_int
    return _shim


def forward_shim(payload, x: np.ndarray, nth: int = 1, w_ref_q8=None) -> np.ndarray:
    """Evaluate through the drop-in TU: reference constructors -> ggml custom op -> C ABI -> GPU."""
    y = np.full((x.shape[0], payload.n_out), np.nan, dtype=np.float32)
    a = _args(payload, x, y, w_ref_q8)
    lib = shim_lib()
    try:
        rc = lib.sdref_run(payload.kind, C.byref(a), nth)
    finally:
        lib.llama_seeddelta_hip_release_all()    # numpy buffers are reused across cases
    if rc != 0:
        raise RuntimeError(f"sdref_run (shim) failed: {rc}")
    return y


def _args(payload, x: np.ndarray, y: np.ndarray, w_ref_q8=None) -> SdoOpArgs:
    a = SdoOpArgs()
    kind = payload.kind
    a.x = tensor_of(x)
    if kind in (1, 3):
        a.idx, a.val = tensor_of(payload.b_idx), tensor_of(payload.b_val)
    else:
        a.idx, a.val = tensor_of(payload.d_idx), tensor_of(payload.d_val)
    a.row_scale = tensor_of(payload.row_scale)
    if kind in (2, 3):
        a.d1, a.d2, a.d3 = tensor_of(payload.base_d1), tensor_of(payload.base_d2), tensor_of(payload.base_d3)
        a.p1, a.p2 = tensor_of(payload.base_perm1), tensor_of(payload.base_perm2)
    if w_ref_q8 is not None:
        a.w_ref = tensor_of(w_ref_q8, q8_0_shape=(payload.n_out, payload.n_in))
    else:
        a.w_ref = tensor_of(payload.w_ref)
    a.dst = tensor_of(y)
    return a


def forward(payload, x: np.ndarray, nth: int = 1, use_ref: bool = False, w_ref_q8=None,
            out: Optional[np.ndarray] = None) -> np.ndarray:
    """Evaluate the op the payload dispatches to on x (T, n_in) -> y (T, n_out), on the CPU."""
    assert x.dtype in (np.float32, np.float16) and x.ndim == 2 and x.shape[1] == payload.n_in
    y = out if out is not None else np.full((x.shape[0], payload.n_out), np.nan, dtype=np.float32)
    a = _args(payload, x, y, w_ref_q8)
    if use_ref:
        rc = ref_lib().sdref_run(payload.kind, C.byref(a), nth)
        if rc != 0:
            raise RuntimeError(f"sdref_run failed: {rc}")
    else:
        oracle_lib().sdo_run(payload.kind, C.byref(a), nth)
    return y


def w0_tables(payload):
    L, B = payload.base_d1.shape[1], payload.base_d1.shape[0]
    G = np.empty((B, L), dtype=np.float32)
    inv = np.empty((B, L), dtype=np.int32)
    d2, p1 = tensor_of(payload.base_d2), tensor_of(payload.base_perm1)
    oracle_lib().sdo_w0_tables(C.byref(d2), C.byref(p1), L, B, G.ctypes.data, inv.ctypes.data)
    return G, inv


def w0_tile(payload, o0: int, o1: int, i0: int, i1: int) -> np.ndarray:
    """Dense W0[o0:o1, i0:i1] from the closed form (numpy, vectorised; same product order as
    sdo_w0_elem: (d3*G)*d1 in fp32)."""
    G, inv = w0_tables(payload)
    L = G.shape[1]
    o = np.arange(o0, o1)[:, None]
    i = np.arange(i0, i1)[None, :]
    d1 = payload.base_d1.astype(np.float32)
    d3 = payload.base_d3.astype(np.float32) if payload.base_d3 is not None else np.ones_like(d1)
    if payload.is_tall:
        b = o // L
        p = o - b * L
        q = i
    else:
        b = i // L
        q = i - b * L
        p = o
    b, p, q = np.broadcast_arrays(b, p, q)
    g = G[b, (p ^ inv[b, q]) & (L - 1)]
    return ((d3[b, p] * g).astype(np.float32) * d1[b, q]).astype(np.float32)


# ---- gated FFN block: the gate activation between the nodes and the three-node composition ----------
ACT_SILU, ACT_GELU = 0, 1
_gelu_table = None


def gelu_table() -> np.ndarray:
    """The 65536-entry f16 GELU table (oracle restatement of ggml-cpu.c:3657-3667)."""
    global _gelu_table
    if _gelu_table is None:
        t = np.empty(1 << 16, dtype=np.uint16)
        oracle_lib().sdo_gelu_table(t.ctypes.data)
        _gelu_table = t
    return _gelu_table


def glu(act: int, g: np.ndarray, u: np.ndarray, use_ref: bool = False) -> np.ndarray:
    """y = act(g) * u elementwise on float32 (ggml_swiglu_split / ggml_geglu_split). use_ref: the
    reference's own op computed by its CPU backend."""
    g = np.ascontiguousarray(g, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.float32)
    assert g.shape == u.shape
    y = np.empty_like(g)
    if use_ref:
        rc = ref_lib().sdref_glu(act, g.size, g.ctypes.data, u.ctypes.data, y.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"sdref_glu failed: {rc}")
    else:
        oracle_lib().sdo_glu(act, g.size, g.ctypes.data, u.ctypes.data, y.ctypes.data, gelu_table().ctypes.data)
    return y


def ffn_forward(gate, up, down, x: np.ndarray, act: int, nth: int = 1):
    """Oracle: down( act(gate(x)) * up(x) ) node by node on the CPU -> (h, y)."""
    g = forward(gate, x, nth=nth)
    u = forward(up, x, nth=nth)
    h = glu(act, g, u)
    return h, forward(down, h, nth=nth)


def ffn_forward_graph(gate, up, down, x: np.ndarray, act: int, nth: int = 1, shim: bool = False):
    """The same block as ONE ggml graph computed by the reference's CPU backend: the reference's
    constructors for the three nodes (its own op TU, or with shim=True the HIP drop-in TU) and its
    ggml_swiglu_split / ggml_geglu_split between them (src/llama-graph.cpp:811-953) -> (h, y)."""
    T = x.shape[0]
    h = np.empty((T, gate.n_out), dtype=np.float32)
    y = np.empty((T, down.n_out), dtype=np.float32)
    dummy_g = np.empty((T, gate.n_out), dtype=np.float32)     # only their shapes are read
    ga = _args(gate, x, dummy_g)
    ua = _args(up, x, dummy_g)
    da = _args(down, h, y)
    kinds = (C.c_int * 3)(gate.kind, up.kind, down.kind)
    lib = shim_lib() if shim else ref_lib()
    try:
        rc = lib.sdref_ffn(kinds, C.byref(ga), C.byref(ua), C.byref(da), act, nth, h.ctypes.data, y.ctypes.data)
    finally:
        if shim:
            lib.llama_seeddelta_hip_release_all()
    if rc != 0:
        raise RuntimeError(f"sdref_ffn failed: {rc}")
    return h, y


def ffn_node_shim(gate, up, down, x: np.ndarray, act: int, nth: int = 1, dense_ref=None):
    """The drop-in's optional whole-FFN node (llama_seeddelta_hip_ffn) inside a ggml graph computed by
    the reference's CPU backend -> y, or None when the constructor declines (keeps three nodes).
    dense_ref: a dense gate weight to present as still loaded."""
    T = x.shape[0]
    y = np.empty((T, down.n_out), dtype=np.float32)
    dummy = np.empty((T, gate.n_out), dtype=np.float32)
    if dense_ref is not None:
        gate = _copy_with_w_ref(gate, dense_ref)
    ga, ua, da = _args(gate, x, dummy), _args(up, x, dummy), _args(down, dummy, y)
    kinds = (C.c_int * 3)(gate.kind, up.kind, down.kind)
    lib = shim_lib()
    try:
        rc = lib.sdshim_ffn_node(kinds, C.byref(ga), C.byref(ua), C.byref(da), act, nth, 1 if dense_ref is not None else 0,
                                 y.ctypes.data)
    finally:
        lib.llama_seeddelta_hip_release_all()
    if rc == 1:
        return None
    if rc != 0:
        raise RuntimeError(f"sdshim_ffn_node failed: {rc}")
    return y


def _copy_with_w_ref(p, w_ref):
    import copy
    q = copy.copy(p)
    q.w_ref = w_ref
    return q
