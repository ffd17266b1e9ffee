"""llama.cpp_amd — MI355X-native SeedDelta FFN hot path (host side).

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/seeddelta_hip.h),
``shim/`` (the replacement translation unit for the reference's src/llama-seeddelta.cpp),
``payload`` (host-side payload containers / synthetic shapes) and ``seeddelta`` (ctypes binding
mirroring the reference's operator interface). Import as ``llama_cpp_amd``.
"""
from . import payload  # noqa: F401
