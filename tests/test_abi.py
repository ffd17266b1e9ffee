"""CPU checks of the drop-in boundary: the library loads without a GPU and exports every symbol
include/seeddelta_hip.h declares; struct layouts match the header; host-side dispatch logic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from llama_cpp_amd import seeddelta
from llama_cpp_amd.payload import (OP_BASE_BLOCK, OP_BASE_COO, OP_BLOCK, OP_COO, base_dims, config_payload,
                                   make_payload)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "seeddelta_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdh_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_python_binds():
    assert declared_symbols() == sorted(seeddelta.ABI_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(seeddelta.LIB_PATH), "libseeddelta_hip.so not built (run __graft_entry__.build())"
    lib = C.CDLL(seeddelta.LIB_PATH)
    for sym in declared_symbols():
        assert hasattr(lib, sym), f"{sym} declared in include/seeddelta_hip.h but not exported"
    lib.sdh_abi_version.restype = C.c_int
    assert lib.sdh_abi_version() == 2


def test_struct_layouts_match_header():
    # sdh_tensor: void*, int32, int32, int64[4], int64[4]
    assert C.sizeof(seeddelta.SdhTensor) == 8 + 4 + 4 + 32 + 32
    assert C.sizeof(seeddelta.SdhWeightDesc) == 16 + 10 * C.sizeof(seeddelta.SdhTensor)


def test_tensor_view_follows_ggml_axis_order():
    a = np.zeros((5, 3, 16), dtype=np.float16)        # ggml [16, 3, 5]
    t = seeddelta.tensor_of(a)
    assert list(t.ne) == [16, 3, 5, 1] and list(t.nb)[:3] == [2, 32, 96] and t.type == 1 and t.n_dims == 3
    assert seeddelta.tensor_of(None).data is None


def test_dispatch_priority_matches_build_lora_mm():
    # src/llama-graph.cpp:681-703: block wins over COO, base iff base_d1
    p = make_payload(64, 128, scheme="coo", K=8)
    assert p.kind == OP_COO
    q = make_payload(64, 128, scheme="block", K=16, block=16, base=True)
    assert q.kind == OP_BASE_BLOCK
    q.d_idx, q.d_val = p.d_idx, p.d_val
    assert q.kind == OP_BASE_BLOCK
    q.b_idx = None
    assert q.kind == OP_BASE_COO
    q.base_d1 = None
    assert q.kind == OP_COO
    assert make_payload(64, 128, scheme="block", K=16).kind == OP_BLOCK


def test_base_dims_and_algorithmic_bytes_of_named_configs():
    assert base_dims(14336, 4096) == (4096, 4)      # Mistral-7B
    assert base_dims(2560, 10240) == (4096, 3)      # Gemma-3 4B
    assert base_dims(1152, 6912) == (2048, 4)       # Gemma-3 1B
    # cfg3 = 13,189,120 B per token per tensor
    p = config_payload("cfg3_mistral_down_block32_k1536")
    assert p.algorithmic_bytes(1) == 13_189_120
    # cfg2 Gemma-4B gate COO K=128 + base: 5,412,864 B/token
    assert config_payload("cfg2_gemma4b_gate_coo_k128").algorithmic_bytes(1) == 5_412_864


def test_no_silent_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, not fall back to a CPU computation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(seeddelta.SeedDeltaError):
        seeddelta.init(0)
    with pytest.raises(seeddelta.SeedDeltaError):
        seeddelta.SeedDeltaWeight(make_payload(64, 64))


def test_shim_library_defines_the_reference_constructors():
    """oracle/_ref/libsdshim.so (built where /root/reference exists) must define the four
    constructors of src/llama-seeddelta.h:41-84 and bind the C ABI rather than carry CPU loops."""
    import subprocess
    import oracle_py as O
    if not O.shim_available():
        pytest.skip("shim test library not built (needs /root/reference at build time)")
    defined = subprocess.run(["nm", "-D", "--defined-only", "-C", O.SHIM_SO], capture_output=True, text=True).stdout
    for name in ("llama_seeddelta_mul_mat(", "llama_seeddelta_mul_mat_block(", "llama_seeddelta_mul_mat_base(",
                 "llama_seeddelta_mul_mat_base_block(", "llama_seeddelta_hip_release_all"):
        assert name in defined, name
    undefined = subprocess.run(["nm", "-D", "--undefined-only", O.SHIM_SO], capture_output=True, text=True).stdout
    for name in ("sdh_init", "sdh_weight_create", "sdh_forward", "sdh_nonfinite_count"):
        assert name in undefined, name


def test_python_flag_constants_match_the_header():
    """Every SDH_FWD_* flag of include/seeddelta_hip.h that the Python layer names has the header's value."""
    import re
    import llama_cpp_amd.seeddelta as sd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "seeddelta_hip.h")).read()
    flags = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define\s+SDH_(FWD_[A-Z_0-9]+)\s+(0x[0-9A-Fa-f]+)", text)}
    assert flags["FWD_NO_DIRECT"] == 0x100 and flags["FWD_MFMA_FOLD"] == 0x40
    checked = 0
    for name, value in flags.items():
        if hasattr(sd, name):
            assert getattr(sd, name) == value, name
            checked += 1
    assert checked >= 8
    assert len(set(flags.values()) - {0}) == len([v for v in flags.values() if v != 0])      # no two flags share a bit pattern
