"""ctypes binding of libseeddelta_hip.so (include/seeddelta_hip.h) — the host-side mirror of the
reference's SeedDelta operator interface.

The reference exposes four graph-node constructors (``src/llama-seeddelta.h:41-84``):
``llama_seeddelta_mul_mat``, ``_block``, ``_base``, ``_base_block``. The functions of the same name
here take the same arguments (numpy arrays instead of ggml tensors) and run the op on the MI355X
through the C ABI. ``SeedDeltaWeight`` is the upload-once handle (``struct llama_seeddelta_weight``,
``src/llama-seeddelta.h:12-25``); ``SeedDeltaContext`` mirrors ``llama_seeddelta_context`` (:27-38):
a map from a dense-weight key to its payload, with ``find`` / ``mul_mat`` dispatch as in
``llm_graph_context::build_lora_mm`` (``src/llama-graph.cpp:679-715``).

There is no CPU fallback: if the HIP library or a GPU is missing every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .payload import Payload

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEEDDELTA_HIP_LIB") or os.path.join(HERE, "csrc", "libseeddelta_hip.so")   # env: dev A/B builds

SDH_TYPE = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.int16): 25, np.dtype(np.int32): 26}

FWD_DEVICE_IO, FWD_HOST_IO, FWD_ZERO_NONFINITE, FWD_EXACT_ORDER, FWD_W0_MFMA, FWD_NO_FUSE, FWD_NO_SPMM = 0x0, 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
FWD_NO_MFMA = 0x80            # keep the fp32 forms where sdh_forward would choose the folded contraction by itself
FWD_NO_DIRECT = 0x100         # one token: keep the fused launch (base term with the bits of the FWHT form), not the direct one
FWD_MFMA_FOLD = 0x40          # prefill as one dense contraction, residual folded into the W0 tile (f16 operands)

# every symbol include/seeddelta_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "sdh_init", "sdh_abi_version", "sdh_last_error", "sdh_device_synchronize", "sdh_library_stream",
    "sdh_weight_create", "sdh_weight_destroy", "sdh_weight_op_kind", "sdh_weight_payload_bytes",
    "sdh_forward", "sdh_forward_noalloc", "sdh_noalloc_scratch_bytes", "sdh_debug_alloc_calls", "sdh_debug_fold_layout", "sdh_debug_last_decode_kind", "sdh_probe_stream_floor", "sdh_nonfinite_count", "sdh_w0_tile", "sdh_ffn_forward", "sdh_ffn_gate_up", "sdh_ffn_last_fused", "sdh_glu",
    "sdh_ffn_stack_create", "sdh_ffn_stack_destroy", "sdh_ffn_stack_forward", "sdh_ffn_stack_status", "sdh_ffn_stack_payload_bytes",
    "sdh_profile_enable", "sdh_profile_read", "sdh_profile_span_begin", "sdh_profile_span_end",
    "sdh_dense_create", "sdh_dense_destroy", "sdh_dense_bytes",
    "sdh_scan_create", "sdh_scan_destroy", "sdh_scan_reseed", "sdh_scan_tensor", "sdh_scan_tensor_kgrid", "sdh_scan_pick", "sdh_scan_rewind", "sdh_scan_skip", "sdh_scan_fetch", "sdh_scan_fetch_packed", "sdh_scan_memo_clear", "sdh_imatrix_scale",
]


class SdhTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int32), ("n_dims", C.c_int32),
                ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4)]


class SdhWeightDesc(C.Structure):
    _fields_ = [("n_in", C.c_int64), ("n_out", C.c_int64)] + [(n, SdhTensor) for n in (
        "d_idx", "d_val", "b_idx", "b_val", "row_scale",
        "base_d1", "base_d2", "base_d3", "base_perm1", "base_perm2")]


class SdhScanParams(C.Structure):
    _fields_ = [("scheme", C.c_int32), ("kind", C.c_int32), ("K", C.c_int64), ("block", C.c_int64),
                ("write_base", C.c_int32), ("write_row_scale", C.c_int32), ("base_max_samples", C.c_int64),
                ("base_perm_trials", C.c_int32), ("want_proxy", C.c_int32), ("eval_cols", C.c_int64),
                ("eval_x", C.c_int64), ("proxy_seed", C.c_int32), ("base_mode", C.c_int32), ("w_scale", C.c_void_p),
                ("metrics_only", C.c_int32), ("kgrid_draws", C.c_int32), ("delta_norm_clamp_down", C.c_double)]


class SdhScanResult(C.Structure):
    _fields_ = [("K_eff", C.c_int64), ("n_blocks", C.c_int64), ("L", C.c_int64), ("B", C.c_int64),
                ("has_x", C.c_int32), ("proxy_available", C.c_int32), ("em", C.c_double * 5),
                ("em_x", C.c_double * 5), ("cost", C.c_double * 8), ("proxy", C.c_double * 8),
                ("gpu_ms", C.c_double), ("has_w", C.c_int32), ("reserved", C.c_int32),
                ("em_w", C.c_double * 5), ("em_x_w", C.c_double * 5), ("delta_clamp_scale", C.c_double)]


class SeedDeltaError(RuntimeError):
    pass


def tensor_of(arr: Optional[np.ndarray]) -> SdhTensor:
    """View a numpy array as a ggml-style tensor (ne[0] = fastest numpy axis)."""
    t = SdhTensor()
    if arr is None:
        return t
    if arr.dtype not in SDH_TYPE:
        raise SeedDeltaError(f"unsupported dtype {arr.dtype}")
    t.data = arr.ctypes.data
    t.type = SDH_TYPE[arr.dtype]
    shape, strides = arr.shape[::-1], arr.strides[::-1]
    t.n_dims = len(shape)
    ne, nb = [1, 1, 1, 1], [0, 0, 0, 0]
    for i in range(len(shape)):
        ne[i], nb[i] = shape[i], strides[i]
    for i in range(len(shape), 4):
        nb[i] = nb[i - 1] * ne[i - 1]
    t.ne[:], t.nb[:] = ne, nb
    return t


_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library. Raises if it is not built — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise SeedDeltaError(f"{path} not built: run `make -C llama.cpp_amd/csrc` (or __graft_entry__.build())")
    try:
        # torch ships its own libamdhip64/libhsa-runtime64. If this library (linked against
        # libamdhip64.so.7) is loaded first, the process ends up with two HIP runtimes and torch
        # then reports "No HIP GPUs are available". Loading torch first makes both share one.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.sdh_init.argtypes = [C.c_int]
    lib.sdh_abi_version.restype = C.c_int
    lib.sdh_last_error.restype = C.c_char_p
    lib.sdh_library_stream.restype = C.c_void_p
    lib.sdh_weight_create.argtypes = [C.POINTER(SdhWeightDesc), C.POINTER(C.c_void_p)]
    lib.sdh_weight_destroy.argtypes = [C.c_void_p]
    lib.sdh_weight_destroy.restype = None
    lib.sdh_weight_op_kind.argtypes = [C.c_void_p]
    lib.sdh_weight_payload_bytes.argtypes = [C.c_void_p]
    lib.sdh_weight_payload_bytes.restype = C.c_int64
    lib.sdh_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                C.c_int, C.c_void_p]
    lib.sdh_forward_noalloc.argtypes = lib.sdh_forward.argtypes
    lib.sdh_noalloc_scratch_bytes.restype = C.c_int64
    lib.sdh_debug_alloc_calls.restype = C.c_int64
    lib.sdh_debug_fold_layout.restype = C.c_int
    lib.sdh_debug_last_decode_kind.restype = C.c_int
    lib.sdh_debug_fold_layout.argtypes = [C.c_void_p]
    lib.sdh_ffn_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.sdh_ffn_gate_up.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.sdh_ffn_stack_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
    lib.sdh_ffn_stack_destroy.argtypes = [C.c_void_p]
    lib.sdh_ffn_stack_destroy.restype = None
    lib.sdh_ffn_stack_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.sdh_ffn_stack_status.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.sdh_ffn_stack_payload_bytes.argtypes = [C.c_void_p]
    lib.sdh_ffn_stack_payload_bytes.restype = C.c_int64
    lib.sdh_glu.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.sdh_nonfinite_count.argtypes = [C.c_void_p]
    lib.sdh_nonfinite_count.restype = C.c_int64
    lib.sdh_w0_tile.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                C.c_int, C.c_void_p]
    lib.sdh_profile_enable.argtypes = [C.c_int]
    lib.sdh_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.sdh_profile_span_begin.argtypes = [C.c_void_p]
    lib.sdh_profile_span_end.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.sdh_dense_create.argtypes = [C.POINTER(SdhTensor), C.POINTER(C.c_void_p)]
    lib.sdh_dense_destroy.argtypes = [C.c_void_p]
    lib.sdh_dense_destroy.restype = None
    lib.sdh_dense_bytes.argtypes = [C.c_void_p]
    lib.sdh_dense_bytes.restype = C.c_int64
    lib.sdh_scan_create.argtypes = [C.c_uint32, C.POINTER(C.c_void_p)]
    lib.sdh_scan_reseed.argtypes = [C.c_void_p, C.c_uint32]
    lib.sdh_scan_destroy.argtypes = [C.c_void_p]
    lib.sdh_scan_destroy.restype = None
    lib.sdh_scan_tensor.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(SdhScanParams), C.POINTER(SdhScanResult)]
    lib.sdh_scan_skip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(SdhScanParams)]
    lib.sdh_scan_tensor_kgrid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(SdhScanParams), C.POINTER(C.c_int64), C.c_int, C.POINTER(SdhScanResult)]
    lib.sdh_scan_pick.argtypes = [C.c_void_p, C.c_int]
    lib.sdh_scan_rewind.argtypes = [C.c_void_p, C.c_int]
    lib.sdh_scan_fetch.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    lib.sdh_scan_fetch_packed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_int)]
    lib.sdh_scan_memo_clear.restype = None
    lib.sdh_imatrix_scale.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().sdh_last_error().decode(errors="replace")
        raise SeedDeltaError(f"{what} failed ({rc}): {msg}")


_inited_device = None


def init(device: int = 0):
    global _inited_device
    lib = load_library()
    _check(lib.sdh_init(device), "sdh_init")
    _inited_device = device


def library_stream() -> int:
    """The library's private non-blocking stream (what host-I/O calls with stream NULL run on)."""
    return int(load_library().sdh_library_stream() or 0)


def profile_enable(on: bool):
    _check(load_library().sdh_profile_enable(1 if on else 0), "sdh_profile_enable")


def profile_read():
    """-> (total_ms, launches) of the dominant kernel since the last read (HIP events)."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    _check(load_library().sdh_profile_read(C.byref(ms), C.byref(n)), "sdh_profile_read")
    return ms.value, n.value


def profile_span_begin(stream: int = 0):
    _check(load_library().sdh_profile_span_begin(C.c_void_p(stream) if stream else None), "sdh_profile_span_begin")


def profile_span_end(stream: int = 0):
    """-> (elapsed_ms, forwards) between the two HIP events recorded on `stream`."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    _check(load_library().sdh_profile_span_end(C.c_void_p(stream) if stream else None, C.byref(ms), C.byref(n)),
           "sdh_profile_span_end")
    return ms.value, n.value


def probe_stream_floor(nbytes: int, n_buffers: int = 32, reps: int = 64, stream: int = 0):
    """-> (us per launch of a kernel that only streams `nbytes` once, us per empty launch of that shape)."""
    a, b = C.c_double(0.0), C.c_double(0.0)
    _check(load_library().sdh_probe_stream_floor(C.c_int64(nbytes), n_buffers, reps, C.c_void_p(stream) if stream else None,
                                                 C.byref(a), C.byref(b)), "sdh_probe_stream_floor")
    return a.value, b.value


class SeedDeltaWeight:
    """Device-resident payload of one FFN weight (upload once, evaluate many times)."""

    def __init__(self, payload: Payload):
        if _inited_device is None:
            init(0)
        lib = load_library()
        d = SdhWeightDesc()
        d.n_in, d.n_out = payload.n_in, payload.n_out
        # keep references so the host arrays outlive the create call
        self._keep = payload
        for name in ("d_idx", "d_val", "b_idx", "b_val", "row_scale",
                     "base_d1", "base_d2", "base_d3", "base_perm1", "base_perm2"):
            setattr(d, name, tensor_of(getattr(payload, name)))
        h = C.c_void_p()
        _check(lib.sdh_weight_create(C.byref(d), C.byref(h)), "sdh_weight_create")
        self._h = h
        self.n_in, self.n_out = payload.n_in, payload.n_out
        self.kind = lib.sdh_weight_op_kind(h)
        self.payload_bytes = lib.sdh_weight_payload_bytes(h)
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            load_library().sdh_weight_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host buffers (what the ggml custom-op callback sees) ---------------------------------
    def forward_host(self, x: np.ndarray, flags: int = 0) -> np.ndarray:
        """x (T, n_in) float32 host -> y (T, n_out) float32 host."""
        assert x.dtype == np.float32 and x.ndim == 2 and x.shape[1] == self.n_in
        assert x.strides[1] == 4 and x.strides[0] % 4 == 0
        T = x.shape[0]
        y = np.empty((T, self.n_out), dtype=np.float32)
        rc = load_library().sdh_forward(self._h, x.ctypes.data, x.strides[0] // 4, T,
                                        y.ctypes.data, self.n_out, flags | FWD_HOST_IO, None)
        _check(rc, "sdh_forward")
        return y

    # -- device buffers (raw pointers; torch is only the allocator) ---------------------------
    def forward_device(self, x_ptr: int, ldx: int, T: int, y_ptr: int, ldy: int,
                       flags: int = 0, stream: int = 0):
        rc = load_library().sdh_forward(self._h, x_ptr, ldx, T, y_ptr, ldy, flags & ~FWD_HOST_IO,
                                        C.c_void_p(stream) if stream else None)
        _check(rc, "sdh_forward")

    def forward_device_noalloc(self, x_ptr: int, ldx: int, T: int, y_ptr: int, ldy: int,
                               flags: int = 0, stream: int = 0) -> int:
        """sdh_forward_noalloc; returns the status code (0, or SDH_ERR_TOO_LARGE = -5 ...) instead of raising."""
        return int(load_library().sdh_forward_noalloc(self._h, x_ptr, ldx, T, y_ptr, ldy, flags,
                                                      C.c_void_p(stream) if stream else None))

    def fold_layout(self) -> int:
        """test hook: sdh_debug_fold_layout (bit 1 = permuted order, bit 2 = duplicates, bit 3 = overflow buckets)"""
        return int(load_library().sdh_debug_fold_layout(self._h))

    def nonfinite_count(self) -> int:
        n = load_library().sdh_nonfinite_count(self._h)
        if n < 0:
            _check(int(n), "sdh_nonfinite_count")
        return int(n)

    def w0_tile(self, o0: int, o1: int, i0: int, i1: int) -> np.ndarray:
        """Dense W0[o0:o1, i0:i1] regenerated on the device; returned as (o1-o0, i1-i0)."""
        out = np.empty((i1 - i0, o1 - o0), dtype=np.float32)   # device layout: o fastest
        rc = load_library().sdh_w0_tile(self._h, o0, o1, i0, i1, out.ctypes.data, FWD_HOST_IO, None)
        _check(rc, "sdh_w0_tile")
        return np.ascontiguousarray(out.T)


# ---- gated FFN block: build_ffn's three SeedDelta nodes + gate activation (src/llama-graph.cpp:811-953)
ACT_SILU, ACT_GELU = 0, 1          # LLM_FFN_SILU / LLM_FFN_GELU with LLM_FFN_PAR gating


def ffn_forward_host(gate: SeedDeltaWeight, up: SeedDeltaWeight, down: SeedDeltaWeight, x: np.ndarray,
                     act: int = ACT_SILU, flags: int = 0) -> np.ndarray:
    """y = down( act(gate(x)) * up(x) ) with x (T, n_embd) and y (T, n_embd) float32 on the host; the
    intermediate activations stay on the device."""
    assert x.dtype == np.float32 and x.ndim == 2 and x.shape[1] == gate.n_in
    assert x.strides[1] == 4 and x.strides[0] % 4 == 0
    T = x.shape[0]
    y = np.empty((T, down.n_out), dtype=np.float32)
    rc = load_library().sdh_ffn_forward(gate._h, up._h, down._h, act, x.ctypes.data, x.strides[0] // 4, T,
                                        y.ctypes.data, down.n_out, flags | FWD_HOST_IO, None)
    _check(rc, "sdh_ffn_forward")
    return y


def ffn_gate_up_host(gate: SeedDeltaWeight, up: SeedDeltaWeight, x: np.ndarray, act: int = ACT_SILU,
                     flags: int = 0) -> np.ndarray:
    """h = act(gate(x)) * up(x): x (T, n_embd) -> h (T, n_ff), float32 on the host."""
    assert x.dtype == np.float32 and x.ndim == 2 and x.shape[1] == gate.n_in
    assert x.strides[1] == 4 and x.strides[0] % 4 == 0
    T = x.shape[0]
    h = np.empty((T, gate.n_out), dtype=np.float32)
    rc = load_library().sdh_ffn_gate_up(gate._h, up._h, act, x.ctypes.data, x.strides[0] // 4, T,
                                        h.ctypes.data, gate.n_out, flags | FWD_HOST_IO, None)
    _check(rc, "sdh_ffn_gate_up")
    return h


def ffn_gate_up_device(gate: SeedDeltaWeight, up: SeedDeltaWeight, act: int, x_ptr: int, ldx: int, T: int,
                       h_ptr: int, ldh: int, flags: int = 0, stream: int = 0):
    rc = load_library().sdh_ffn_gate_up(gate._h, up._h, act, x_ptr, ldx, T, h_ptr, ldh,
                                        flags & ~FWD_HOST_IO, C.c_void_p(stream) if stream else None)
    _check(rc, "sdh_ffn_gate_up")


def ffn_forward_device(gate: SeedDeltaWeight, up: SeedDeltaWeight, down: SeedDeltaWeight, act: int,
                       x_ptr: int, ldx: int, T: int, y_ptr: int, ldy: int, flags: int = 0, stream: int = 0):
    rc = load_library().sdh_ffn_forward(gate._h, up._h, down._h, act, x_ptr, ldx, T, y_ptr, ldy,
                                        flags & ~FWD_HOST_IO, C.c_void_p(stream) if stream else None)
    _check(rc, "sdh_ffn_forward")


class FfnStack:
    """n gated FFN blocks evaluated by one persistent launch at T = 1 (sdh_ffn_stack_*). `layers` is a list of
    (gate, up, down) SeedDeltaWeight triples, kept alive by this object. Raises SeedDeltaError with status -4
    (SDH_ERR_UNSUPPORTED) in the text when the layers do not have the shape the kernel is built for."""

    def __init__(self, layers):
        lib = load_library()
        self.layers = [tuple(t) for t in layers]
        n = len(self.layers)
        arrs = [(C.c_void_p * n)(*[t[k]._h for t in self.layers]) for k in range(3)]
        h = C.c_void_p()
        _check(lib.sdh_ffn_stack_create(arrs[0], arrs[1], arrs[2], n, C.byref(h)), "sdh_ffn_stack_create")
        self._h = h
        self.n_layers = n
        self.payload_bytes = lib.sdh_ffn_stack_payload_bytes(h)

    def forward_device(self, act: int, x_ptr: int, y_ptr: int, flags: int = 0, stream: int = 0):
        _check(load_library().sdh_ffn_stack_forward(self._h, act, x_ptr, y_ptr, flags, C.c_void_p(stream) if stream else None),
               "sdh_ffn_stack_forward")

    def status(self):
        """-> (non-finite node outputs of the last call, hand-off timeouts so far); waits for the last call."""
        nf, to = C.c_int64(0), C.c_int64(0)
        _check(load_library().sdh_ffn_stack_status(self._h, C.byref(nf), C.byref(to)), "sdh_ffn_stack_status")
        return nf.value, to.value

    def close(self):
        if getattr(self, "_h", None):
            load_library().sdh_ffn_stack_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def last_decode_kind() -> int:
    """which launch the last one-token forward with a base took: 1 direct, 2 fused, 3 two kernels, 0 none yet"""
    return int(load_library().sdh_debug_last_decode_kind())


def ffn_last_fused() -> bool:
    return bool(load_library().sdh_ffn_last_fused())


def glu_device(act: int, g_ptr: int, u_ptr: int, y_ptr: int, n: int, stream: int = 0):
    """y = act(g) * u over n device floats (ggml_swiglu_split / ggml_geglu_split)."""
    _check(load_library().sdh_glu(act, g_ptr, u_ptr, y_ptr, n, C.c_void_p(stream) if stream else None), "sdh_glu")


# ---- the reference's four constructors, same argument order (src/llama-seeddelta.h:41-84) ----
def llama_seeddelta_mul_mat(x, w_ref, d_idx, d_val, row_scale, flags: int = 0):
    n_in, n_out = x.shape[1], d_idx.shape[0]
    w = SeedDeltaWeight(Payload(n_in, n_out, d_idx=d_idx, d_val=d_val, row_scale=row_scale, w_ref=w_ref))
    return w.forward_host(x, flags)


def llama_seeddelta_mul_mat_block(x, w_ref, b_idx, b_val, row_scale, flags: int = 0):
    n_in, n_out = x.shape[1], b_idx.shape[0]
    w = SeedDeltaWeight(Payload(n_in, n_out, b_idx=b_idx, b_val=b_val, row_scale=row_scale, w_ref=w_ref))
    return w.forward_host(x, flags)


def llama_seeddelta_mul_mat_base(x, w_ref, base_d1, base_d2, base_d3, base_perm1, base_perm2,
                                 d_idx, d_val, row_scale, flags: int = 0):
    n_in, n_out = x.shape[1], d_idx.shape[0]
    w = SeedDeltaWeight(Payload(n_in, n_out, d_idx=d_idx, d_val=d_val, row_scale=row_scale,
                                base_d1=base_d1, base_d2=base_d2, base_d3=base_d3,
                                base_perm1=base_perm1, base_perm2=base_perm2, w_ref=w_ref))
    return w.forward_host(x, flags)


def llama_seeddelta_mul_mat_base_block(x, w_ref, base_d1, base_d2, base_d3, base_perm1, base_perm2,
                                       b_idx, b_val, row_scale, flags: int = 0):
    n_in, n_out = x.shape[1], b_idx.shape[0]
    w = SeedDeltaWeight(Payload(n_in, n_out, b_idx=b_idx, b_val=b_val, row_scale=row_scale,
                                base_d1=base_d1, base_d2=base_d2, base_d3=base_d3,
                                base_perm1=base_perm1, base_perm2=base_perm2, w_ref=w_ref))
    return w.forward_host(x, flags)


class SeedDeltaContext:
    """Mirror of ``llama_seeddelta_context`` (src/llama-seeddelta.h:27-38): dense-weight key ->
    device payload, filled with the acceptance rules of ``try_add_seeddelta``
    (src/llama-context.cpp:145-286; enforced inside sdh_weight_create)."""

    def __init__(self, enabled: bool = True, gap_tol: float = 0.02):
        self.enabled = enabled
        self.gap_tol = gap_tol          # stored and never read, as in the reference
        self.weights: dict = {}

    def try_add(self, key, payload: Payload) -> bool:
        try:
            self.weights[key] = SeedDeltaWeight(payload)
            return True
        except SeedDeltaError:
            return False               # reference: LLAMA_LOG_WARN + keep the dense path

    def find(self, key) -> Optional[SeedDeltaWeight]:
        return self.weights.get(key)

    def mul_mat(self, key, x: np.ndarray) -> Optional[np.ndarray]:
        """build_lora_mm's SeedDelta branch: None means 'use the dense matmul'."""
        if not self.enabled:
            return None
        w = self.find(key)
        return None if w is None else w.forward_host(x)
