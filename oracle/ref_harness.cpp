// ref_harness.cpp — drives the REFERENCE's own SeedDelta translation unit
// (/root/reference/src/llama-seeddelta.cpp, compiled unmodified from where it lies; see
// oracle/Makefile) on caller-supplied buffers. Test infrastructure only: it validates
// oracle/sd_oracle.c, generates tests/golden/, and is the "reference" CPU baseline.
//
// The reference's op bodies are file-static; the only way in is the public constructors of
// src/llama-seeddelta.h:41-84, which return a ggml custom-op node whose op_params hold
// {fun, n_tasks, userdata} (ggml/src/ggml-impl.h:189-193). We build the node on tensors that
// alias the caller's memory and invoke fun(dst, ith, nth, userdata) for every slice, which
// is precisely what ggml-cpu does (ggml/src/ggml-cpu/ops.cpp:10141-10149).
#include "sd_oracle.h"          // sdo_op_args / sdo_tensor (argument layout only)

#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-impl.h"
#include "llama-seeddelta.h"

#include <cstring>
#include <thread>
#include <vector>

namespace {

struct holder {
    ggml_context * ctx = nullptr;
    std::vector<ggml_backend_buffer_t> bufs;
    ~holder() {
        for (auto b : bufs) ggml_backend_buffer_free(b);
        if (ctx) ggml_free(ctx);
    }
};

ggml_tensor * wrap(holder & h, const sdo_tensor & t, bool with_buffer) {
    if (!t.data) return nullptr;
    int nd = t.n_dims > 0 ? t.n_dims : 1;
    int64_t ne[4] = {1, 1, 1, 1};
    for (int i = 0; i < nd && i < 4; ++i) ne[i] = t.ne[i];
    ggml_tensor * g = ggml_new_tensor(h.ctx, (ggml_type) t.type, nd, ne);
    g->data = t.data;
    // honour caller strides (strided/non-contiguous cases)
    for (int i = 0; i < nd && i < 4; ++i) g->nb[i] = (size_t) t.nb[i];
    for (int i = nd; i < 4; ++i) g->nb[i] = g->nb[i - 1] * (size_t) g->ne[i - 1];
    if (with_buffer) {
        // the NaN-repair path insists on a host backend buffer (src/llama-seeddelta.cpp:219-226)
        size_t span = 0;
        for (int i = 0; i < 4; ++i) span += (size_t) (g->ne[i] - 1) * g->nb[i];
        span += ggml_type_size(g->type);
        // the buffer only answers "is this host memory?"; its base must be 32-byte aligned
        const uintptr_t p0 = (uintptr_t) t.data & ~(uintptr_t) 63;
        span += (uintptr_t) t.data - p0;
        ggml_backend_buffer_t b = ggml_backend_cpu_buffer_from_ptr((void *) p0, span);
        h.bufs.push_back(b);
        g->buffer = b;
    }
    return g;
}

} // namespace

extern "C" int sdref_run(int kind, const sdo_op_args * a, int nth) {
    holder h;
    ggml_init_params ip = { /*mem_size*/ 64 * ggml_tensor_overhead() + 4096, /*mem_buffer*/ nullptr, /*no_alloc*/ true };
    h.ctx = ggml_init(ip);
    if (!h.ctx) return -1;

    ggml_tensor * x   = wrap(h, a->x, true);
    ggml_tensor * idx = wrap(h, a->idx, false);
    ggml_tensor * val = wrap(h, a->val, false);
    ggml_tensor * rs  = wrap(h, a->row_scale, false);
    ggml_tensor * d1  = wrap(h, a->d1, false);
    ggml_tensor * d2  = wrap(h, a->d2, false);
    ggml_tensor * d3  = wrap(h, a->d3, false);
    ggml_tensor * p1  = wrap(h, a->p1, false);
    ggml_tensor * p2  = wrap(h, a->p2, false);
    ggml_tensor * w   = wrap(h, a->w_ref, true);
    if (w) ggml_set_name(w, "w_ref");

    ggml_tensor * dst = nullptr;
    switch (kind) {
        case 0: dst = llama_seeddelta_mul_mat(h.ctx, x, w, idx, val, rs); break;
        case 1: dst = llama_seeddelta_mul_mat_block(h.ctx, x, w, idx, val, rs); break;
        case 2: dst = llama_seeddelta_mul_mat_base(h.ctx, x, w, d1, d2, d3, p1, p2, idx, val, rs); break;
        case 3: dst = llama_seeddelta_mul_mat_base_block(h.ctx, x, w, d1, d2, d3, p1, p2, idx, val, rs); break;
        default: return -2;
    }
    if (!dst) return -3;
    dst->data = a->dst.data;
    dst->nb[0] = (size_t) a->dst.nb[0];
    dst->nb[1] = (size_t) a->dst.nb[1];

    ggml_custom_op_params p;
    std::memcpy(&p, dst->op_params, sizeof(p));

    if (nth <= 1) {
        p.fun(dst, 0, 1, p.userdata);
    } else {
        std::vector<std::thread> th;
        th.reserve((size_t) nth);
        for (int i = 0; i < nth; ++i) th.emplace_back([&, i] { p.fun(dst, i, nth, p.userdata); });
        for (auto & t : th) t.join();
    }
    return 0;
}

// exported conversions of the reference, to pin sdo_fp16_to_fp32 / sdo_fp32_to_fp16
extern "C" float    sdref_fp16_to_fp32(uint16_t h) { return ggml_fp16_to_fp32(h); }
extern "C" uint16_t sdref_fp32_to_fp16(float f)    { return ggml_fp32_to_fp16(f); }
