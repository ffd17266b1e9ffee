// ref_ffn_harness.cpp — test infrastructure: the gated FFN block as the REFERENCE computes it. The
// node sequence build_ffn emits for LLM_FFN_PAR (src/llama-graph.cpp:811-953) is built with the
// reference's own constructors (src/llama-seeddelta.h:41-84) and ggml_swiglu_split / ggml_geglu_split,
// and computed by the reference's own CPU backend (ggml/src/ggml-cpu/*, compiled unmodified from
// /root/reference; see oracle/Makefile). Linked with src/llama-seeddelta.cpp it is the FFN oracle pin
// and golden generator; linked with llama.cpp_amd/shim/llama-seeddelta-hip.cpp (libsdshim.so) it is
// the drop-in test of the HIP op inside a real ggml graph. Not part of the product.
#include "sd_oracle.h"          // sdo_op_args / sdo_tensor (argument layout only)

#include "ggml.h"
#include "ggml-cpu.h"
#include "llama-seeddelta.h"

#include <cstring>

namespace {

ggml_tensor * wrap(ggml_context * ctx, const sdo_tensor & t) {
    if (!t.data) return nullptr;
    int nd = t.n_dims > 0 ? t.n_dims : 1;
    int64_t ne[4] = {1, 1, 1, 1};
    for (int i = 0; i < nd && i < 4; ++i) ne[i] = t.ne[i];
    ggml_tensor * g = ggml_new_tensor(ctx, (ggml_type) t.type, nd, ne);
    g->data = t.data;
    for (int i = 0; i < nd && i < 4; ++i) g->nb[i] = (size_t) t.nb[i];
    for (int i = nd; i < 4; ++i) g->nb[i] = g->nb[i - 1] * (size_t) g->ne[i - 1];
    return g;
}

ggml_tensor * node(ggml_context * ctx, ggml_context * wctx, int kind, const sdo_op_args * a, ggml_tensor * x) {
    ggml_tensor * idx = wrap(wctx, a->idx), * val = wrap(wctx, a->val), * rs = wrap(wctx, a->row_scale);
    ggml_tensor * d1 = wrap(wctx, a->d1), * d2 = wrap(wctx, a->d2), * d3 = wrap(wctx, a->d3);
    ggml_tensor * p1 = wrap(wctx, a->p1), * p2 = wrap(wctx, a->p2);
    switch (kind) {
        case 0: return llama_seeddelta_mul_mat(ctx, x, nullptr, idx, val, rs);
        case 1: return llama_seeddelta_mul_mat_block(ctx, x, nullptr, idx, val, rs);
        case 2: return llama_seeddelta_mul_mat_base(ctx, x, nullptr, d1, d2, d3, p1, p2, idx, val, rs);
        case 3: return llama_seeddelta_mul_mat_base_block(ctx, x, nullptr, d1, d2, d3, p1, p2, idx, val, rs);
        default: return nullptr;
    }
}

} // namespace

// kinds[3]: op kind of gate / up / down; ga->x is the block input [n_embd, T]; act 0 = SiLU, 1 = GELU.
// h_out (optional): [n_ff, T] contiguous, y_out: [n_embd_out, T] contiguous.
extern "C" int sdref_ffn(const int * kinds, const sdo_op_args * ga, const sdo_op_args * ua, const sdo_op_args * da,
                         int act, int nth, float * h_out, float * y_out) {
    ggml_cpu_init();
    ggml_init_params wp = { 64 * ggml_tensor_overhead() + 4096, nullptr, /*no_alloc*/ true };
    ggml_context * wctx = ggml_init(wp);
    const int64_t T = ga->x.ne[1] > 0 ? ga->x.ne[1] : 1;
    const size_t n_ff = (size_t) ga->dst.ne[0], n_out = (size_t) da->dst.ne[0];
    const size_t mem = (size_t) T * (3 * n_ff + n_out) * 4 + 16 * ggml_tensor_overhead() + ggml_graph_overhead() + (1u << 20);
    ggml_init_params cp = { mem, nullptr, /*no_alloc*/ false };
    ggml_context * ctx = ggml_init(cp);
    int rc = 0;
    if (!wctx || !ctx) rc = -1;
    if (rc == 0) {
        ggml_tensor * x = wrap(wctx, ga->x);
        ggml_tensor * up   = node(ctx, wctx, kinds[1], ua, x);      // build_ffn order: ffn_up first (:826)
        ggml_tensor * gate = node(ctx, wctx, kinds[0], ga, x);      // then ffn_gate on the same input (:848)
        ggml_tensor * h = nullptr, * y = nullptr;
        if (up && gate) h = act == 0 ? ggml_swiglu_split(ctx, gate, up) : ggml_geglu_split(ctx, gate, up);   // :870 / :879
        if (h) y = node(ctx, wctx, kinds[2], da, h);                 // ffn_down (:932)
        if (!y) rc = -3;
        else {
            ggml_cgraph * gf = ggml_new_graph(ctx);
            ggml_build_forward_expand(gf, y);
            if (ggml_graph_compute_with_ctx(ctx, gf, nth) != GGML_STATUS_SUCCESS) rc = -4;
            else {
                if (h_out) std::memcpy(h_out, h->data, ggml_nbytes(h));
                std::memcpy(y_out, y->data, ggml_nbytes(y));
            }
        }
    }
    if (ctx) ggml_free(ctx);
    if (wctx) ggml_free(wctx);
    return rc;
}

// the gate activation op on its own, through the same backend
extern "C" int sdref_glu(int act, int64_t n, const float * g, const float * u, float * y) {
    ggml_cpu_init();
    const size_t mem = (size_t) n * 4 * 3 + 8 * ggml_tensor_overhead() + ggml_graph_overhead() + (1u << 16);
    ggml_init_params cp = { mem, nullptr, false };
    ggml_context * ctx = ggml_init(cp);
    if (!ctx) return -1;
    ggml_tensor * tg = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n), * tu = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n);
    std::memcpy(tg->data, g, (size_t) n * 4);
    std::memcpy(tu->data, u, (size_t) n * 4);
    ggml_tensor * h = act == 0 ? ggml_swiglu_split(ctx, tg, tu) : ggml_geglu_split(ctx, tg, tu);
    ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, h);
    const int rc = ggml_graph_compute_with_ctx(ctx, gf, 1) == GGML_STATUS_SUCCESS ? 0 : -4;
    if (rc == 0) std::memcpy(y, h->data, (size_t) n * 4);
    ggml_free(ctx);
    return rc;
}
