// shim_ffn_harness.cpp — test infrastructure (libsdshim.so only): evaluates the HIP drop-in's optional
// whole-FFN node (llama.cpp_amd/shim/llama-seeddelta-hip-ffn.h) inside a ggml graph computed by the
// reference's CPU backend, the way build_ffn would after the INTEGRATION.md §1b change.
#include "sd_oracle.h"

#include "ggml.h"
#include "ggml-cpu.h"
#include "llama-seeddelta-hip-ffn.h"

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
llama_seeddelta_weight weight_of(ggml_context * ctx, int kind, const sdo_op_args * a) {
    llama_seeddelta_weight w;
    const bool block = kind == 1 || kind == 3, base = kind >= 2;
    (block ? w.b_idx : w.d_idx) = wrap(ctx, a->idx);
    (block ? w.b_val : w.d_val) = wrap(ctx, a->val);
    w.row_scale = wrap(ctx, a->row_scale);
    if (base) {
        w.base_d1 = wrap(ctx, a->d1); w.base_d2 = wrap(ctx, a->d2); w.base_d3 = wrap(ctx, a->d3);
        w.base_perm1 = wrap(ctx, a->p1); w.base_perm2 = wrap(ctx, a->p2);
    }
    return w;
}
} // namespace

// returns 1 if the constructor declined (nullptr), 0 on success, <0 on error
extern "C" int sdshim_ffn_node(const int * kinds, const sdo_op_args * ga, const sdo_op_args * ua, const sdo_op_args * da,
                               int act, int nth, int dense_refs, float * y_out) {
    ggml_cpu_init();
    ggml_init_params wp = { 96 * ggml_tensor_overhead() + 4096, nullptr, true };
    ggml_context * wctx = ggml_init(wp);
    const int64_t T = ga->x.ne[1] > 0 ? ga->x.ne[1] : 1;
    const size_t n_out = (size_t) da->dst.ne[0];
    ggml_init_params cp = { (size_t) T * n_out * 4 + 8 * ggml_tensor_overhead() + ggml_graph_overhead() + (1u << 20), nullptr, false };
    ggml_context * ctx = ggml_init(cp);
    int rc = 0;
    if (!wctx || !ctx) rc = -1;
    if (rc == 0) {
        ggml_tensor * x = wrap(wctx, ga->x);
        const llama_seeddelta_weight g = weight_of(wctx, kinds[0], ga), u = weight_of(wctx, kinds[1], ua), d = weight_of(wctx, kinds[2], da);
        // dense_refs: pretend the dense weights are still loaded (the constructor must then decline for T > 1)
        ggml_tensor * wg = dense_refs ? wrap(wctx, ga->w_ref) : nullptr;
        ggml_tensor * y = llama_seeddelta_hip_ffn(ctx, x, &g, &u, &d, wg, wg, wg, act != 0);
        if (!y) rc = 1;
        else {
            ggml_cgraph * gf = ggml_new_graph(ctx);
            ggml_build_forward_expand(gf, y);
            if (ggml_graph_compute_with_ctx(ctx, gf, nth) != GGML_STATUS_SUCCESS) rc = -4;
            else std::memcpy(y_out, y->data, ggml_nbytes(y));
        }
    }
    if (ctx) ggml_free(ctx);
    if (wctx) ggml_free(wctx);
    return rc;
}
