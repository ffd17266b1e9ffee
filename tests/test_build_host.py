"""Host-side helpers of the one-call builder (llama_cpp_amd/build.py)."""
import pytest

from llama_cpp_amd import build as B


def test_policy_hash_follows_the_builders_constants():
    """sd_policy_resolve.cpp:9-21 is FNV-1a 64 with the offset basis 1469598103934665603 (the published
    basis 14695981039346656037 without its last digit), so its values are NOT the published FNV test
    vectors: "a" gives 44bd8ad473cd9906 there, af63dc4c8601ec8c with the published basis. The mirror has
    to reproduce the builder's value. Parity unpinned: the routine sits in an anonymous namespace of a
    file that does not compile here (seeddelta_policy.h under GCC 11); the expected values below are
    the stated recurrence evaluated by hand-written arithmetic."""
    assert B.policy_hash(b"") == ""
    h = ((1469598103934665603 ^ ord("a")) * 1099511628211) % 2 ** 64
    assert B.policy_hash(b"a") == f"fnv1a64:{h:016x}" == "fnv1a64:44bd8ad473cd9906"
    h2 = 1469598103934665603
    for c in b'{"version": 1}':
        h2 = ((h2 ^ c) * 1099511628211) % 2 ** 64
    assert B.policy_hash(b'{"version": 1}') == f"fnv1a64:{h2:016x}"
    assert B.layer_range("", 3) == [0, 1, 2] and B.layer_range(" 5 - 1 ", 4) == [1, 2, 3]
    with pytest.raises(ValueError):
        B.layer_range("1,2", 4)
