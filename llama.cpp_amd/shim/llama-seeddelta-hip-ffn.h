// llama-seeddelta-hip-ffn.h — OPTIONAL fifth constructor of the HIP drop-in (no counterpart in
// src/llama-seeddelta.h): the whole parallel-gated FFN of one layer as ONE ggml node, so that the two
// intermediate activations stay in HBM and gate|up run as one launch at n_tokens == 1.
//
// It replaces, inside build_ffn (src/llama-graph.cpp:811-953), the node sequence
//   ffn_up (:826) -> ffn_gate (:848) -> ggml_swiglu_split (:870) | ggml_geglu_split (:879) -> ffn_down (:932)
// when all three weights have SeedDelta payloads. INTEGRATION.md §1b shows the call site.
#pragma once

#include "llama-seeddelta.h"

// gate/up/down: the payloads llama_seeddelta_context::find() returned for the three dense weights;
// w_gate/w_up/w_down: those dense weights (may be stripped stubs). gelu: false = LLM_FFN_SILU, true =
// LLM_FFN_GELU (both with LLM_FFN_PAR gating). Returns an F32 [n_embd, n_tokens] node, or nullptr when
// the block must stay three nodes to keep the reference's behaviour: more than one token AND a dense
// weight still present (its NaN guard would repair from the dense row, src/llama-seeddelta.cpp:294-361,
// which needs each node's output on the host). With one token, or stripped weights, the reference's
// guard writes 0.0 for a non-finite output (:323, :373-380) and so does this node, on the device.
ggml_tensor * llama_seeddelta_hip_ffn(
        ggml_context * ctx, ggml_tensor * x,
        const llama_seeddelta_weight * gate, const llama_seeddelta_weight * up, const llama_seeddelta_weight * down,
        const ggml_tensor * w_gate, const ggml_tensor * w_up, const ggml_tensor * w_down,
        bool gelu);
