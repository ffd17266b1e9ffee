"""Minimal GGUF v3 reader / writer for the tensors and metadata of the SeedDelta path
(file layout: ggml/src/gguf.cpp, ggml/include/gguf.h; element types: ggml/include/ggml.h:385-411).

Only what this path needs: little-endian files, scalar / string / array metadata, and the element
types F32, F16, Q8_0, I8, I16, I32, I64, F64. Quantised tensors other than Q8_0 are listed (name,
shape, type, file offset) but their data is returned as raw bytes. Tensors are memory-mapped, so
opening a 7B GGUF costs nothing until a tensor is touched.

numpy shape convention: reversed ggml dims (ne[0] is the fastest numpy axis); Q8_0 rows come back as
uint8 (n_rows, n_in // 32 * 34), the layout llama_cpp_amd.dense and sdh_dense_create take.
"""
from __future__ import annotations

import mmap
import struct
from typing import Dict, List, Tuple

import numpy as np

MAGIC = 0x46554747          # "GGUF"
DEFAULT_ALIGNMENT = 32
# gguf metadata value types (ggml/include/gguf.h: enum gguf_type)
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f",
           T_BOOL: "<?", T_U64: "<Q", T_I64: "<q", T_F64: "<d"}
# ggml element types: code -> (numpy dtype, elements per block, bytes per block)
GGML_TYPES = {0: (np.float32, 1, 4), 1: (np.float16, 1, 2), 8: (np.uint8, 32, 34), 24: (np.int8, 1, 1),
              25: (np.int16, 1, 2), 26: (np.int32, 1, 4), 27: (np.int64, 1, 8), 28: (np.float64, 1, 8)}
GGML_OF_DTYPE = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.int8): 24, np.dtype(np.int16): 25,
                 np.dtype(np.int32): 26, np.dtype(np.int64): 27, np.dtype(np.float64): 28}
# bytes per block of the quantised types this reader only lists (ggml/src/ggml.c type traits)
_OTHER_BLOCKS = {2: (32, 18), 3: (32, 20), 6: (32, 22), 7: (32, 24), 9: (32, 36), 10: (256, 84), 11: (256, 110),
                 12: (256, 144), 13: (256, 176), 14: (256, 210), 15: (256, 292), 30: (1, 2)}


class GGUFReader:
    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._pos = 0
        magic, self.version = self._unpack("<II")
        if magic != MAGIC:
            raise ValueError(f"{path}: not a GGUF file")
        if self.version not in (2, 3):
            raise ValueError(f"{path}: GGUF version {self.version} not supported")
        n_tensors, n_kv = self._unpack("<QQ")
        self.kv: Dict[str, object] = {}
        self.kv_spans: Dict[str, Tuple[int, int]] = {}                   # key -> byte range of the encoded pair
        for _ in range(n_kv):
            start = self._pos
            key = self._string()
            (vt,) = self._unpack("<I")
            self.kv[key] = self._value(vt)
            self.kv_spans[key] = (start, self._pos)
        self.alignment = int(self.kv.get("general.alignment", DEFAULT_ALIGNMENT))
        self.infos: Dict[str, Tuple[Tuple[int, ...], int, int]] = {}     # name -> (ne, ggml type, offset)
        for _ in range(n_tensors):
            name = self._string()
            (nd,) = self._unpack("<I")
            ne = self._unpack("<" + "Q" * nd)
            gtype, off = self._unpack("<IQ")
            self.infos[name] = (tuple(int(v) for v in ne), int(gtype), int(off))
        self.data_start = (self._pos + self.alignment - 1) // self.alignment * self.alignment

    def close(self):
        try:
            self._mm.close()
        except BufferError:          # tensor views still alive: the mapping goes when they do
            pass
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _unpack(self, fmt):
        vals = struct.unpack_from(fmt, self._mm, self._pos)
        self._pos += struct.calcsize(fmt)
        return vals

    def _string(self) -> str:
        (n,) = self._unpack("<Q")
        s = self._mm[self._pos:self._pos + n].decode("utf-8", errors="replace")
        self._pos += n
        return s

    def _value(self, vt):
        if vt in _SCALAR:
            return self._unpack(_SCALAR[vt])[0]
        if vt == T_STR:
            return self._string()
        if vt == T_ARR:
            et, n = self._unpack("<IQ")
            if et in _SCALAR:
                fmt = _SCALAR[et]
                out = np.frombuffer(self._mm, dtype=np.dtype(fmt), count=n, offset=self._pos).copy()
                self._pos += n * struct.calcsize(fmt)
                return out
            return [self._value(et) for _ in range(n)]
        raise ValueError(f"unknown GGUF value type {vt}")

    def names(self) -> List[str]:
        return list(self.infos)

    def nbytes(self, name: str) -> int:
        ne, gtype, _ = self.infos[name]
        per, size = (GGML_TYPES[gtype][1:] if gtype in GGML_TYPES else _OTHER_BLOCKS[gtype])
        n = 1
        for v in ne:
            n *= v
        return n // per * size

    def kv_raw(self, key: str) -> bytes:
        """The key/value pair exactly as the file encodes it (copied through by the builder's writer)."""
        a, b = self.kv_spans[key]
        return bytes(self._mm[a:b])

    def raw(self, name: str) -> Tuple[int, Tuple[int, ...], memoryview]:
        """(ggml type, ne, bytes) of a tensor of any type, for copying it into another GGUF unchanged."""
        ne, gtype, off = self.infos[name]
        start = self.data_start + off
        return gtype, ne, memoryview(self._mm)[start:start + self.nbytes(name)]

    def tensor(self, name: str) -> np.ndarray:
        """Zero-copy view of a tensor (numpy shape = reversed ne)."""
        ne, gtype, off = self.infos[name]
        start = self.data_start + off
        if gtype not in GGML_TYPES:
            return np.frombuffer(self._mm, dtype=np.uint8, count=self.nbytes(name), offset=start)
        dt, per, size = GGML_TYPES[gtype]
        if per == 1:
            n = 1
            for v in ne:
                n *= v
            return np.frombuffer(self._mm, dtype=dt, count=n, offset=start).reshape(ne[::-1])
        row = ne[0] // per * size                                  # Q8_0: raw rows
        rows = 1
        for v in ne[1:]:
            rows *= v
        return np.frombuffer(self._mm, dtype=np.uint8, count=rows * row, offset=start).reshape(rows, row)


def _kv_bytes(key: str, value) -> bytes:
    def s(x: str) -> bytes:
        b = x.encode("utf-8")
        return struct.pack("<Q", len(b)) + b
    out = s(key)
    if isinstance(value, bool):
        return out + struct.pack("<I?", T_BOOL, value)
    if isinstance(value, int):
        return out + (struct.pack("<II", T_U32, value) if 0 <= value < 2 ** 32 else struct.pack("<Iq", T_I64, value))
    if isinstance(value, float):
        return out + struct.pack("<If", T_F32, value)
    if isinstance(value, str):
        return out + struct.pack("<I", T_STR) + s(value)
    if isinstance(value, (list, tuple)) and all(isinstance(v, str) for v in value):
        return out + struct.pack("<IIQ", T_ARR, T_STR, len(value)) + b"".join(s(v) for v in value)
    if isinstance(value, np.ndarray) and value.ndim == 1:
        et = {np.dtype(np.int32): T_I32, np.dtype(np.uint32): T_U32, np.dtype(np.float32): T_F32,
              np.dtype(np.int64): T_I64, np.dtype(np.uint64): T_U64}[value.dtype]
        return out + struct.pack("<IIQ", T_ARR, et, len(value)) + value.tobytes()
    raise TypeError(f"unsupported metadata value for {key}: {type(value)}")


def write_gguf(path: str, kv: Dict[str, object], tensors: Dict[str, object], q8_0: Tuple[str, ...] = (),
               alignment: int = DEFAULT_ALIGNMENT, kv_raw: Tuple[bytes, ...] = ()) -> None:
    """Write a GGUF v3 file. A tensor is a numpy array (uint8 arrays named in `q8_0` are raw Q8_0 rows
    (n_rows, n_in//32*34)) or a (ggml type, ne, bytes) triple from GGUFReader.raw, written unchanged.
    `kv_raw` are already encoded key/value pairs (GGUFReader.kv_raw), written before `kv`."""
    infos, blobs, off = [], [], 0
    for name, a in tensors.items():
        if isinstance(a, tuple):
            gtype, ne, raw = a
        else:
            a = np.ascontiguousarray(a)
            if name in q8_0:
                gtype, ne = 8, (a.shape[1] // 34 * 32, a.shape[0])
            else:
                gtype, ne = GGML_OF_DTYPE[a.dtype], a.shape[::-1]
            raw = a.tobytes()
        nb = name.encode("utf-8")
        infos.append(struct.pack("<Q", len(nb)) + nb + struct.pack("<I", len(ne)) + struct.pack("<" + "Q" * len(ne), *ne)
                     + struct.pack("<IQ", gtype, off))
        pad = (-len(raw)) % alignment
        blobs.append((raw, pad))
        off += len(raw) + pad
    kv = dict(kv)
    if alignment != DEFAULT_ALIGNMENT:
        kv["general.alignment"] = alignment
    head = struct.pack("<IIQQ", MAGIC, 3, len(tensors), len(kv) + len(kv_raw)) + b"".join(kv_raw) + \
        b"".join(_kv_bytes(k, v) for k, v in kv.items()) + b"".join(infos)
    with open(path, "wb") as f:
        f.write(head)
        f.write(b"\x00" * ((-len(head)) % alignment))
        for raw, pad in blobs:
            f.write(raw)
            f.write(b"\x00" * pad)


def load_seeddelta_context(path: str, n_layer: int = None):
    """Open a SeedDelta GGUF and collect, per FFN tensor, its payload — what llama_model::load_tensors does
    for the optional blk.N.ffn_*.{d_idx,...} tensors (src/llama-model.cpp:2775-2848). Returns
    (reader, {weight name: Payload}, seeddelta.* metadata)."""
    from .pack import KV_PREFIX, payload_from_tensors
    r = GGUFReader(path)
    meta = {k: v for k, v in r.kv.items() if k.startswith(KV_PREFIX)}
    payloads = {}
    layers = sorted({int(n.split(".")[1]) for n in r.infos if n.startswith("blk.") and n.split(".")[1].isdigit()})
    for layer in layers if n_layer is None else range(n_layer):
        for kind in ("ffn_gate", "ffn_up", "ffn_down"):
            have = {n: None for n in r.infos if n.startswith(f"blk.{layer}.{kind}.")}
            if not have:
                continue
            wname = f"blk.{layer}.{kind}.weight"
            idx_name = next((n for n in have if n.endswith((".d_idx", ".b_idx"))), None)
            if idx_name is None:
                continue
            n_out = r.infos[idx_name][0][1]
            if wname in r.infos:
                n_in = r.infos[wname][0][0]
            else:                                   # dense weight stripped: the base dims bound n_in
                d1 = f"blk.{layer}.{kind}.base_d1"
                if d1 not in r.infos:
                    continue
                L, B = r.infos[d1][0][:2]
                n_in = L if n_out >= L else None
                if n_in is None:
                    continue
            tensors = {n: r.tensor(n) for n in have if not n.endswith(".weight")}
            p = payload_from_tensors(layer, kind, tensors, n_in, n_out)
            if p is not None:
                payloads[wname] = p
    return r, payloads, meta
