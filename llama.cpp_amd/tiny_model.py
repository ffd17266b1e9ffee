"""A synthetic llama-architecture GGUF small enough to run through a CPU build of the reference in
seconds, for the end-to-end drop-in check (policy_eval.py, tests/test_gpu_e2e.py).

No model weights or datasets exist in this environment, so the end-to-end run cannot be Mistral-7B on
wikitext-2; what CAN be checked is that a GGUF written by build.py loads through the reference's own
loader (src/llama-model.cpp:2775-2848, strip stubs :2606-2627), is accepted by try_add_seeddelta
(src/llama-context.cpp:145-320), is dispatched by build_lora_mm (src/llama-graph.cpp:672-746), and
that logits / perplexity / greedy tokens of the reference CPU op and of the HIP path agree.

The file carries `tokenizer.ggml.model = "no_vocab"` (src/llama-vocab.cpp:1692: vocabulary type NONE,
`llama.vocab_size` dummy tokens), so token ids are fed directly and no tokenizer data is needed.
Weights are drawn so that activations stay O(1) through the stack (1/sqrt(fan_in) scaling) and the FFN
weights carry an XOR-circulant component, which is what the builder's base fit looks for.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .gguf_io import write_gguf


def tiny_llama_tensors(n_layer: int, n_embd: int, n_ff: int, n_head: int, n_vocab: int, seed: int = 11,
                       structure: float = 0.5, ffn_dtype=np.float32) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)

    def mat(n_out, n_in, s=1.0):
        return (s * rng.standard_normal((n_out, n_in), dtype=np.float32) / np.sqrt(n_in)).astype(np.float32)

    def ffn(n_out, n_in):
        w = mat(n_out, n_in, 1.0 - structure)
        if structure > 0:            # W[o,i] += h[(o ^ perm_inv[i]) & (L-1)]: what fit_base_xor_circulant models
            m = min(n_in, n_out)
            L = 1
            while L < m:
                L <<= 1
            h = (structure * rng.standard_normal(L, dtype=np.float32) / np.sqrt(n_in)).astype(np.float32)
            o = np.arange(n_out)[:, None] & (L - 1)
            i = np.arange(n_in)[None, :] & (L - 1)
            w += h[o ^ i]
        return w.astype(ffn_dtype)

    t: Dict[str, np.ndarray] = {"token_embd.weight": mat(n_vocab, n_embd, np.sqrt(n_embd))}
    for il in range(n_layer):
        p = f"blk.{il}."
        t[p + "attn_norm.weight"] = np.ones(n_embd, dtype=np.float32)
        for nm in ("attn_q", "attn_k", "attn_v", "attn_output"):
            t[p + nm + ".weight"] = mat(n_embd, n_embd)
        t[p + "ffn_norm.weight"] = np.ones(n_embd, dtype=np.float32)
        t[p + "ffn_gate.weight"] = ffn(n_ff, n_embd)
        t[p + "ffn_up.weight"] = ffn(n_ff, n_embd)
        t[p + "ffn_down.weight"] = ffn(n_embd, n_ff)
    t["output_norm.weight"] = np.ones(n_embd, dtype=np.float32)
    t["output.weight"] = mat(n_vocab, n_embd, 4.0)
    return t


def tiny_llama_kv(n_layer: int, n_embd: int, n_ff: int, n_head: int, n_vocab: int, n_ctx_train: int = 4096) -> Dict[str, object]:
    return {
        "general.architecture": "llama", "general.name": "tiny-synthetic-llama",
        "llama.context_length": n_ctx_train, "llama.embedding_length": n_embd, "llama.block_count": n_layer,
        "llama.feed_forward_length": n_ff, "llama.attention.head_count": n_head,
        "llama.attention.head_count_kv": n_head, "llama.attention.layer_norm_rms_epsilon": 1e-5,
        "llama.rope.dimension_count": n_embd // n_head, "llama.vocab_size": n_vocab,
        "tokenizer.ggml.model": "no_vocab",
    }


def write_tiny_llama(path: str, n_layer: int = 2, n_embd: int = 1024, n_ff: int = 2816, n_head: int = 8,
                     n_vocab: int = 512, seed: int = 11, structure: float = 0.5, ffn_dtype=np.float32) -> dict:
    tensors = tiny_llama_tensors(n_layer, n_embd, n_ff, n_head, n_vocab, seed, structure, ffn_dtype)
    write_gguf(path, tiny_llama_kv(n_layer, n_embd, n_ff, n_head, n_vocab), tensors)
    return dict(n_layer=n_layer, n_embd=n_embd, n_ff=n_ff, n_head=n_head, n_vocab=n_vocab)


def random_tokens(path: str, n: int, n_vocab: int, seed: int = 3) -> np.ndarray:
    """A token file for the e2e driver: int32 ids from a first-order Markov chain (so that a model could
    in principle predict them; the check is parity between two builds, not quality)."""
    rng = np.random.default_rng(seed)
    step = rng.integers(1, 7, size=n)
    toks = (np.cumsum(step) * 37 + rng.integers(0, 3, size=n)) % n_vocab
    toks = toks.astype(np.int32)
    toks.tofile(path)
    return toks
