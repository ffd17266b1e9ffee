/*
 * seeddelta_hip.h — C ABI of libseeddelta_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the SeedDelta FFN hot path of frudas24/llama.cpp.
 * Everything crossing it is plain C: raw pointers, sizes, ggml type codes. No torch
 * types, no C++ types, no exceptions. Every entry point returns 0 on success or a
 * negative sdh_status; sdh_last_error() gives the text for the calling thread.
 *
 * What the reference binds here (reference paths relative to /root/reference):
 *   - the four custom-op callbacks of src/llama-seeddelta.cpp (:582 coo, :714 block,
 *     :1014 base+coo, :1178 base+block) — replaced by sdh_forward();
 *   - the payload struct llama_seeddelta_weight (src/llama-seeddelta.h:12-25) — mirrored
 *     by sdh_weight_desc, validated with the rules of src/llama-context.cpp:145-268;
 *   - the builder's candidate-evaluation loop (tools/seeddelta-build/sd_build.cpp:463-800,
 *     sd_eval.cpp, sd_ffn_proxy.cpp) — replaced by the sdh_scan_* calls.
 *
 * The host-side C++ translation unit that forwards the reference's ggml custom ops to
 * this ABI lives in llama.cpp_amd/shim/llama-seeddelta-hip.cpp; INTEGRATION.md shows the
 * two-line change a maintainer makes to use it.
 */
#ifndef SEEDDELTA_HIP_H
#define SEEDDELTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: stream == NULL on device-I/O calls is HIP's legacy default stream (version 1: the library's private stream);
 *    SDH_FWD_NO_SPMM selects the row kernels' arithmetic; sdh_scan_params.reserved0 became kgrid_draws;
 *    sdh_debug_alloc_calls, sdh_debug_fold_layout, sdh_debug_last_decode_kind, SDH_FWD_NO_DIRECT added. INTEGRATION.md "Migration notes". */
#define SDH_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------------------ */
typedef enum sdh_status {
    SDH_OK              =  0,
    SDH_ERR_INVALID     = -1,  /* bad argument / payload rejected by validation          */
    SDH_ERR_NO_DEVICE   = -2,  /* no HIP device, or sdh_init() not called                 */
    SDH_ERR_HIP         = -3,  /* a HIP runtime call failed (text in sdh_last_error)      */
    SDH_ERR_UNSUPPORTED = -4,  /* valid payload, but a shape/type this build cannot run   */
    SDH_ERR_TOO_LARGE   = -5   /* exceeds a stated static limit (no-alloc entry points)   */
} sdh_status;

/* ---- element types: numerically identical to enum ggml_type
 *      (ggml/include/ggml.h:385-411) so a shim can pass tensor->type through unchanged.  */
typedef enum sdh_type {
    SDH_TYPE_F32  = 0,   /* GGML_TYPE_F32  */
    SDH_TYPE_F16  = 1,   /* GGML_TYPE_F16  */
    SDH_TYPE_Q8_0 = 8,   /* GGML_TYPE_Q8_0 (dense weights for the scan only) */
    SDH_TYPE_I16  = 25,  /* GGML_TYPE_I16  */
    SDH_TYPE_I32  = 26   /* GGML_TYPE_I32  */
} sdh_type;

/* A borrowed view of one ggml tensor: the fields of struct ggml_tensor the reference op
 * reads (data, type, ne[], nb[]; ggml/include/ggml.h:640-676). data == NULL means "tensor
 * absent" (the reference passes nullptr for optional tensors). Host memory. */
typedef struct sdh_tensor {
    const void * data;
    int32_t      type;     /* sdh_type */
    int32_t      n_dims;   /* ggml_n_dims() of the tensor */
    int64_t      ne[4];    /* elements per dimension, ne[0] fastest */
    int64_t      nb[4];    /* stride in bytes per dimension */
} sdh_tensor;

/* Mirror of struct llama_seeddelta_weight (src/llama-seeddelta.h:12-25), plus the dims of
 * the dense weight it replaces (w->ne[0] = n_in, w->ne[1] = n_out). */
typedef struct sdh_weight_desc {
    int64_t    n_in;
    int64_t    n_out;
    sdh_tensor d_idx;       /* [K, n_out]        I16/I32 */
    sdh_tensor d_val;       /* [K, n_out]        F16/F32 */
    sdh_tensor b_idx;       /* [nb, n_out]       I16/I32 */
    sdh_tensor b_val;       /* [block,nb,n_out]  F16/F32 */
    sdh_tensor row_scale;   /* [n_out]           F16/F32, optional */
    sdh_tensor base_d1;     /* [L, B]            F16/F32, optional (presence selects the base ops) */
    sdh_tensor base_d2;     /* [L, B]            F16/F32, optional */
    sdh_tensor base_d3;     /* [L, B]            F16/F32, optional */
    sdh_tensor base_perm1;  /* [L, B]            I16/I32, optional */
    sdh_tensor base_perm2;  /* [L, B]            I16/I32, optional */
} sdh_weight_desc;

typedef struct sdh_weight sdh_weight;   /* opaque: device-resident payload */

/* Which of the reference's four ops a handle dispatches to (src/llama-graph.cpp:681-703:
 * block payload wins over COO; base variant iff base_d1). */
typedef enum sdh_op_kind {
    SDH_OP_COO        = 0,  /* llama_seeddelta_coo_op        src/llama-seeddelta.cpp:582  */
    SDH_OP_BLOCK      = 1,  /* llama_seeddelta_block_op      :714  */
    SDH_OP_BASE_COO   = 2,  /* llama_seeddelta_base_op       :1014 */
    SDH_OP_BASE_BLOCK = 3   /* llama_seeddelta_base_block_op :1178 */
} sdh_op_kind;

/* ---- forward flags ------------------------------------------------------------------ */
#define SDH_FWD_DEVICE_IO     0x0  /* x and y are device pointers; async on `stream`          */
#define SDH_FWD_HOST_IO       0x1  /* x and y are host pointers; copies in/out and waits       */
#define SDH_FWD_ZERO_NONFINITE 0x2 /* non-finite outputs are written as 0.0f (the reference's
                                      behaviour when the dense weight is stripped,
                                      src/llama-seeddelta.cpp:373-380); without this flag they
                                      are left as computed so the caller can repair them       */
#define SDH_FWD_EXACT_ORDER   0x4  /* verification mode: accumulate every output in the
                                      reference's sequential order (one lane per output) so the
                                      result is bit-identical to the CPU op. Slow.             */
#define SDH_FWD_NO_FUSE      0x10  /* batch-1 decode with a base: run the base and the residual as two
                                      launches instead of the single fused launch (A/B measurement)  */
#define SDH_FWD_NO_SPMM      0x20  /* many tokens: the row kernels (fast-path arithmetic; from 5 tokens with the
                                      activations of up to 8 tokens in LDS) instead of the token-tiled sparse x
                                      dense kernel, whose rows are bit-identical to the CPU op. Faster for COO
                                      payloads and for long inputs (Mistral down, T=512: 0.11 vs 0.19 ms);
                                      each token's result is then its one-token result. The gated FFN entry
                                      points make this choice per tensor themselves.                          */
#define SDH_FWD_W0_MFMA       0x8  /* prefill: evaluate the base as W0 tiles regenerated in LDS
                                      and contracted on MFMA instead of the FWHT form          */
#define SDH_FWD_MFMA_FOLD    0x40  /* prefill (5 tokens and more, payloads with a base): the whole op as ONE dense
                                      contraction on the matrix cores — the W0 tile of every k-slab is regenerated in
                                      LDS and the residual entries of that slab are added to it there, so the residual
                                      costs one LDS add per entry and workgroup instead of a gather-multiply per entry
                                      and token. f16 operands (x, W0 + delta), fp32 accumulation, row_scale in fp32:
                                      results within 2e-3 * sum|terms| of the CPU op (tests state it), not bit-identical
                                      to it. The first call on a handle re-lays the residual out by (row tile, k-slab)
                                      on the device (once; it doubles the payload's footprint in HBM). Without any
                                      flag sdh_forward takes this form by itself from the measured crossover of the
                                      payload's class on. Payloads as the builder writes them (d1 = d3 = 1, blocks at
                                      most a third longer than their inputs): COO from 160 tokens, block payloads from
                                      224 (tall) / 320 (wide), rows of 512 entries and more from 80; other payloads with
                                      a closed-form base: COO from 320, tall block payloads from 448, long rows from 160; sdh_ffn_gate_up /
                                      sdh_ffn_forward from 224 tokens (Mistral-7B tensors at 512 tokens: 0.074-0.086 ms
                                      against 0.13-0.36 ms for the fp32 forms). Fewer tokens: the fp32 forms.        */
#define SDH_FWD_NO_MFMA      0x80  /* never choose SDH_FWD_MFMA_FOLD by default: keep the fp32 forms (the tiled SpMM,
                                      bit-identical to the CPU op, or the rows kernels) at every token count; so do
                                      SDH_FWD_NO_SPMM and SDH_FWD_EXACT_ORDER                                    */
#define SDH_FWD_NO_DIRECT   0x100  /* one token, payloads with a base: never the launch in which every workgroup forms
                                      the base term of its own rows (sdh_direct.hip; fp32, within the fast path's bound,
                                      not the reference's butterfly order): keep the fused launch whose base term has the
                                      bits of the FWHT form                                                              */

/* ---- library / device ----------------------------------------------------------------
 * sdh_init: bind the calling process to HIP device `device` and create the library's
 * stream and scratch. Idempotent per device. Replaces nothing in the reference (its op is
 * CPU-only); a shim calls it lazily from the first custom-op callback. */
int          sdh_init(int device);
int          sdh_abi_version(void);
const char * sdh_last_error(void);
int          sdh_device_synchronize(void);
/* Streams. Every entry point that takes a `stream` takes a hipStream_t and uses exactly that stream; NULL is
 * HIP's legacy default stream, as in every HIP API, so a caller that produces x or consumes y on the default
 * stream is ordered with the call. Only calls with SDH_FWD_HOST_IO treat NULL differently: they are synchronous
 * (copy in, run, copy out, wait) and then run on the library's private non-blocking stream, which this getter
 * returns (NULL before sdh_init). No counterpart in the reference (its op is CPU-only). */
void *       sdh_library_stream(void);

/* ---- weights -------------------------------------------------------------------------
 * sdh_weight_create: validate a payload exactly as try_add_seeddelta does
 * (src/llama-context.cpp:145-268: dims, types, L pow2, L >= min(n_in,n_out),
 * B >= ceil(max(n_in,n_out)/L)), upload it once, precompute what the kernels need.
 * Rejected sub-payloads degrade the same way the reference does (a bad COO pair is
 * dropped, a bad row_scale is ignored, a bad base drops all base tensors); if nothing
 * usable remains the call fails with SDH_ERR_INVALID. The host tensors are not referenced
 * after the call returns. */
int  sdh_weight_create(const sdh_weight_desc * desc, sdh_weight ** out);
void sdh_weight_destroy(sdh_weight * w);
int  sdh_weight_op_kind(const sdh_weight * w);            /* sdh_op_kind, or <0 */
/* bytes of the payload as stored by the GGUF (the "algorithmic bytes" of SURVEY §8d,
 * without the T*4*(n_in+n_out) activation term). */
int64_t sdh_weight_payload_bytes(const sdh_weight * w);

/* sdh_forward — replaces the body of the reference's custom-op callback for this payload:
 *   y[o + t*ldy] = row_scale[o] * ( W0(x[:,t])[o] + sum_r val[r,o] * x[idx[r,o] + t*ldx] )
 * x: F32 [n_in, n_tokens] with column stride ldx (elements); y: F32 [n_out, n_tokens] with
 * column stride ldy. `stream` is a hipStream_t (NULL = the legacy default stream; with SDH_FWD_HOST_IO
 * NULL = the library's private stream, see sdh_library_stream).
 * Semantics follow src/llama-seeddelta.cpp: invalid indices skipped (:693, :929-936),
 * out-of-range perm entries read element 0 (:524-527), scale applied only if != 1 (:645),
 * tall/wide base split on n_out >= n_in (:1073).
 * HIP graphs: with device pointers the call may be made on a capturing stream. Call it once
 * eagerly first (workspaces and the fused decode launch's one-time residency check need the
 * host); a decode launch captured before that records the two-kernel path instead. Captured
 * and eager calls on one handle may be mixed freely on one stream; calls on one handle from
 * different streams or graph branches must be ordered by the caller (they share the handle's
 * hand-off buffer). The same holds for sdh_ffn_forward / sdh_ffn_gate_up. */
int sdh_forward(sdh_weight * w,
                const float * x, int64_t ldx, int64_t n_tokens,
                float * y, int64_t ldy,
                int flags, void * stream);

/* sdh_forward_noalloc — sdh_forward for callers that cannot allocate. No counterpart in the reference (its op
 * is CPU code with thread_local scratch, src/llama-seeddelta.cpp:78-80,1003-1012); it is for callers that may not
 * call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation. Same
 * result as sdh_forward(w, ..., flags | SDH_FWD_NO_FUSE, stream), bit for bit, but it calls no hipMalloc, hipFree
 * or allocating library routine: x and y must be device pointers (SDH_FWD_HOST_IO is rejected, as is
 * SDH_FWD_W0_MFMA), and the base term that sdh_forward keeps in a per-handle workspace lives in a file-scope
 * static device array of SDH_NOALLOC_SCRATCH_BYTES (each GPU the library is initialised on holds it until the
 * process ends). Tokens are processed in chunks that fit; a payload whose single-token base term
 * (B * L * 4 bytes) exceeds the array is rejected with SDH_ERR_TOO_LARGE. The array is shared by every call on
 * that GPU and keeps the previous call's contents (nothing in it needs clearing: the base kernel overwrites
 * every element the residual kernel reads): calls that overlap on one GPU — two streams, two host threads —
 * race on it; issue them on one stream. */
#define SDH_NOALLOC_SCRATCH_BYTES (8 * 1024 * 1024)
int     sdh_forward_noalloc(sdh_weight * w,
                            const float * x, int64_t ldx, int64_t n_tokens,
                            float * y, int64_t ldy,
                            int flags, void * stream);
int64_t sdh_noalloc_scratch_bytes(void);   /* SDH_NOALLOC_SCRATCH_BYTES of the loaded library */
/* Test hook (no counterpart in the reference): hipMalloc + hipHostMalloc + hipFree calls the library has made in this
 * process so far. tests/test_zz_no_alloc.py reads it around sdh_forward_noalloc calls and asserts a difference of 0. */
int64_t sdh_debug_alloc_calls(void);
/* Test hook (no counterpart in the reference): how the handle's residual is laid out for SDH_FWD_MFMA_FOLD. 0 = not built
 * yet (no folded call so far); else bit 0 set, bit 1 = permuted order (j = inv_b[q]; w0p_mfma_kernel), bit 2 = some (row,
 * input) pair repeats (atomic adds), bit 3 = some bucket holds more than 2048 entries (overflow variant). */
int     sdh_debug_fold_layout(sdh_weight * w);
/* Test / measurement hook (no counterpart in the reference): which launch the most recent one-token sdh_forward with a base
 * took in this process: 1 = direct (every workgroup forms its rows' base term, sdh_direct.hip), 2 = fused (base workgroups
 * hand over, sdh_decode.hip), 3 = two kernels, 0 = none yet. */
int     sdh_debug_last_decode_kind(void);

/* Number of non-finite outputs produced by the most recent sdh_forward on this handle
 * (the reference's llama_seeddelta_nan_count, src/llama-seeddelta.cpp:77,382, per call).
 * Waits for that forward to finish. */
int64_t sdh_nonfinite_count(sdh_weight * w);

/* ---- gated FFN block (the caller either side of the op) ----------------------
 * sdh_ffn_forward — the three SeedDelta nodes and the gate activation that build_ffn emits for a
 * parallel-gated FFN (src/llama-graph.cpp:811-953 with type_gate == LLM_FFN_PAR: ffn_up :826,
 * ffn_gate :848, ggml_swiglu_split :870 / ggml_geglu_split :879, ffn_down :932), kept on the device:
 *   y = down( act(gate(x)) * up(x) )
 * gate/up: [n_embd -> n_ff] handles, down: [n_ff -> n_embd]; x: F32 [n_embd, n_tokens], y: F32
 * [n_embd, n_tokens]. The two intermediate activations never leave HBM (with SDH_FWD_HOST_IO only x
 * goes up and y comes down). At n_tokens == 1, when gate and up are tall base payloads of one shape,
 * they run as ONE launch whose residual waves compute row o of both and write act(g)*u directly.
 * act: SDH_ACT_SILU = ggml_silu_f32 (ggml/src/ggml-cpu/vec.h:1061), x/(1+expf(-x));
 *      SDH_ACT_GELU = ggml's table form (vec.h:1430-1443): 0 below -10, g*u above 10, else the f16
 *      table entry f16(gelu_f32(f16(g))) (ggml-cpu.c:3665), built on the host with the same libm.
 * Non-finite outputs of each node are counted on that node's handle (sdh_nonfinite_count) and, with
 * SDH_FWD_ZERO_NONFINITE, zeroed before the activation, as the reference's guard does per node. A
 * caller that wants the dense repair (src/llama-seeddelta.cpp:294-361) re-runs the layer node by
 * node when a count is non-zero. Other flags as sdh_forward (SDH_FWD_NO_FUSE also disables the
 * gate|up launch). The intermediate activations live in one workspace per device: FFN calls on a
 * device must not overlap (issue them on one stream, as a decode loop does); like sdh_forward, calls
 * that overlap on one weight handle race on its hand-off buffer. */
#define SDH_ACT_SILU 0
#define SDH_ACT_GELU 1
int sdh_ffn_forward(sdh_weight * gate, sdh_weight * up, sdh_weight * down, int act,
                    const float * x, int64_t ldx, int64_t n_tokens,
                    float * y, int64_t ldy,
                    int flags, void * stream);

/* The first half of the block on its own: h[:, t] = act(gate(x[:, t])) * up(x[:, t]), h: F32
 * [n_ff, n_tokens] with column stride ldh (what build_ffn hands to ffn_down, :926-932). */
int sdh_ffn_gate_up(sdh_weight * gate, sdh_weight * up, int act,
                    const float * x, int64_t ldx, int64_t n_tokens,
                    float * h, int64_t ldh,
                    int flags, void * stream);

/* 1 if the most recent sdh_ffn_gate_up / sdh_ffn_forward evaluated gate and up as the single
 * gate|up launch, 0 if node by node (no counterpart in the reference; tests and bench.py use it to
 * state which path they measured). */
int sdh_ffn_last_fused(void);

/* The gate activation alone: y[i] = act(g[i]) * u[i] over n device elements (the op of
 * ggml_swiglu_split / ggml_geglu_split on F32, ggml/src/ggml-cpu/vec.cpp:416, vec.h:1430). */
int sdh_glu(int act, const float * g, const float * u, float * y, int64_t n, void * stream);

/* ---- a stack of gated FFN blocks in ONE launch (decode, one token) -----------------------------------------
 * y = FFN_{n-1}( ... FFN_1( FFN_0(x) ) ),  FFN_l(v) = down_l( act(gate_l(v)) * up_l(v) ): the node sequence of
 * build_ffn (src/llama-graph.cpp:826-932) for n_layers consecutive blocks, evaluated by one persistent grid that
 * hands its intermediate vectors from phase to phase on the device (no launch per tensor, no launch per layer;
 * DESIGN.md §3 "FFN stack"). With n_layers = 1 it is sdh_ffn_forward at one token as a single launch. It is the
 * decode form for runs of SeedDelta layers whose blocks follow one another directly, and the vehicle of
 * bench.py's `ffn_stack` leg; a model with attention between its FFN blocks calls it with n_layers = 1 per layer.
 * Results are bit-identical to the same layers evaluated one by one with sdh_ffn_forward on the fast path.
 * All layers must share one shape: tall gate/up [n_embd -> n_ff], wide down, base present with L in {1024, 2048,
 * 4096} and at most 4 blocks, I16/F16 payload with K (or nb*block) a multiple of 8 up to 1024, one scheme;
 * otherwise create returns SDH_ERR_UNSUPPORTED and the caller evaluates layer by layer. x, y: device pointers,
 * n_embd floats. The handles must outlive the stack. One call at a time per stack (it owns its hand-off buffers);
 * not capturable into a HIP graph (SDH_ERR_UNSUPPORTED on a capturing stream).
 * ONE TENANT: the launch is a persistent grid of one workgroup per CU whose phases wait for one another, so it is
 * only correct while all its workgroups are resident at once. That is checked once per stack (occupancy query + a
 * census launch on the first call), not per call: while a stack call runs, nothing else may occupy the GPU — no
 * kernel on another stream, no second stack in flight, no other process on the device. If co-residency is lost
 * anyway, every hand-off wait is bounded (500 us), NaN is published in place of the missing value, the call still
 * returns SDH_OK (it is asynchronous) and y is not valid. sdh_ffn_stack_status waits for the last call and returns
 * the number of non-finite node outputs (counted and, with SDH_FWD_ZERO_NONFINITE, zeroed per node as in
 * sdh_ffn_forward) and the number of hand-off waits of THAT call that timed out: a caller that cannot guarantee the
 * single tenant checks it (non-zero: discard y and evaluate the layers with sdh_ffn_forward). */
typedef struct sdh_ffn_stack sdh_ffn_stack;
int     sdh_ffn_stack_create(sdh_weight * const * gate, sdh_weight * const * up, sdh_weight * const * down, int n_layers,
                             sdh_ffn_stack ** out);
void    sdh_ffn_stack_destroy(sdh_ffn_stack * s);
int     sdh_ffn_stack_forward(sdh_ffn_stack * s, int act, const float * x, float * y, int flags, void * stream);
int     sdh_ffn_stack_status(sdh_ffn_stack * s, int64_t * nonfinite, int64_t * timeouts);
int64_t sdh_ffn_stack_payload_bytes(const sdh_ffn_stack * s);

/* Regenerate a tile of the dense base W0 (bit-exact integer XOR indexing into the per-block
 * table G_b = FWHT(d2_b)): out[(o-o0) + (i-i0)*(o1-o0)] = W0[o,i] for o in [o0,o1), i in [i0,i1).
 * `out` is a device or host pointer according to `flags` (SDH_FWD_HOST_IO). Used by the
 * parity tests and by the MFMA prefill path. Follows the W0 definition implied by
 * apply_base_block (src/llama-seeddelta.cpp:497-574) and used explicitly by the builder
 * (tools/seeddelta-build/sd_eval.cpp:506,528,714-716). Only valid for depth-2 payloads
 * without base_perm2 and with a bijective perm1 (what the builder writes). */
int sdh_w0_tile(sdh_weight * w, int64_t o0, int64_t o1, int64_t i0, int64_t i1,
                float * out, int flags, void * stream);

/* ---- scan: the builder's candidate evaluation (tools/seeddelta-build) ------------------------------
 * One sdh_scan is one `llama-seeddelta-build` process as scripts/seeddelta-layer-scan.py:215-262
 * launches it: a std::mt19937 seeded with --seed (tools/seeddelta-build/sd_build.cpp:260) that every
 * tensor evaluated afterwards draws from, in the builder's order. A candidate is one (tensor, scheme,
 * K[, base][, row_scale]) trial = one call of the run_trial lambda (sd_build.cpp:463-800).          */
typedef struct sdh_dense sdh_dense;   /* opaque: a dense FFN weight resident in HBM in its stored type */
typedef struct sdh_scan  sdh_scan;    /* opaque: RNG stream + the last candidate's outputs on device   */

/* Upload a dense weight W [n_in, n_out] (F32, F16 or Q8_0; ne[0] = n_in; rows may be strided by
 * nb[1]). Replaces the per-row read_column_f32 of tools/seeddelta-build/sd_eval.cpp:20-40 (Q8_0 rule:
 * ggml/src/ggml-quants.c:401-414): kernels dequantise on the fly, bit-exactly. */
int     sdh_dense_create(const sdh_tensor * W, sdh_dense ** out);
void    sdh_dense_destroy(sdh_dense * W);
int64_t sdh_dense_bytes(const sdh_dense * W);      /* bytes as stored (the scan's algorithmic bytes) */

typedef struct sdh_scan_params {
    int32_t scheme;             /* 0 = COO, 1 = block                     (--scheme)                    */
    int32_t kind;               /* 0 ffn_gate, 1 ffn_up, 2 ffn_down       (selects the proxy variant)   */
    int64_t K;                  /* residual budget per output row         (--K / policy K)              */
    int64_t block;              /* block length for scheme 1              (--block)                     */
    int32_t write_base;         /* fit and use the XOR-circulant base     (--base)                      */
    int32_t write_row_scale;    /* least-squares row scale                (--row-scale)                 */
    int64_t base_max_samples;   /* (--base-max-samples, default 2048)                                   */
    int32_t base_perm_trials;   /* (--base-perm-trials, builder default 1, layer-scan passes 4)         */
    int32_t want_proxy;         /* evaluate the FFN proxy when the builder would (COO, no base, eval_x>0) */
    int64_t eval_cols;          /* (--eval-cols)                                                        */
    int64_t eval_x;             /* (--eval-x)                                                           */
    int32_t proxy_seed;         /* seed + 101*layer + {1,2,3}             (sd_build.cpp:1032)            */
    int32_t base_mode;          /* with write_base: 0 = draw and fit (a builder run reaching this tensor);
                                   1 = keep the fit of the previous call, draw nothing (the next K of an
                                       autotune schedule: the builder fits once per tensor, sd_build.cpp:444-451,
                                       then runs trials, :843-913);
                                   2 = replay the fit's draws but keep the previous fit (another builder run
                                       with the same seed on the same tensor: the fit would come out identical) */
    const float * w_scale;      /* host, n_in floats or NULL: the imatrix scale of sdh_imatrix_scale(); when
                                   set the fit, the selection scores and the row scale are weighted and the
                                   _w / _x_w metric passes run as well (sd_build.cpp:421-422, :753-794)      */
    int32_t metrics_only;       /* 0 = select every output row (the payload the builder writes; sdh_scan_fetch
                                       returns it); 1 = select only the rows the metric passes sample: the report
                                       entry is identical (eval_* and the proxy read nothing else, sd_eval.cpp:138,
                                       :1052, sd_ffn_proxy.cpp:60-75) and no payload is kept. This is the layer-scan /
                                       autogate use, where the builder's output GGUF is discarded.              */
    int32_t kgrid_draws;        /* sdh_scan_tensor_kgrid only: 0 = every K is a builder run of its own with the same seed
                                       (a layer-scan K grid: all K see the same draws, the generator advances once);
                                       1 = the K are the trials of ONE builder run's autotune schedule
                                       (sd_build.cpp:843-913): trial j draws its eval samples after trial j-1's. outs[j]
                                       then equals sdh_scan_tensor calls in sequence on the handle (base_mode 1 from the
                                       second on); the generator ends after the last trial and sdh_scan_rewind(j) puts it
                                       where a builder that stopped after trial j would be.                      */
    double  delta_norm_clamp_down; /* (--delta-norm-clamp-down, default 0 = off) ffn_down, COO only: if the L2 norm
                                       of all residual values exceeds clamp * ||W||_2 they are scaled down to it
                                       before the metrics are taken (sd_build.cpp:29-46, :384-386, :705-720).
                                       Needs every row, so it overrides metrics_only for that candidate.        */
} sdh_scan_params;

/* What one report.json entry is computed from (tools/seeddelta-build/sd_report.cpp:112-288). Metric
 * vectors are {rel_l2_mean, rel_l2_p95, cos_mean, cos_p05, norm_ratio_mean} (struct eval_metrics,
 * include/sd_eval.h:23-29); cost = {L, B, ops_dense, ops_base, ops_delta, ops_row_scale, ops_total,
 * ops_ratio} (sd_cost.cpp:7-40); proxy = {cos_mean, cos_p05, l2_mean, l2_p95, log_norm_ratio_mean,
 * log_norm_ratio_p95, eval_x, eval_out} (sd_ffn_proxy.cpp:238-245). */
typedef struct sdh_scan_result {
    int64_t K_eff;              /* entries per row actually kept (block scheme: keep*block)              */
    int64_t n_blocks;           /* ceil(K / block)                                                       */
    int64_t L, B;               /* base dims, 0 when no base                                             */
    int32_t has_x;
    int32_t proxy_available;
    double  em[5];              /* weight-space metrics   (eval_*_residual, sd_eval.cpp:118-995)         */
    double  em_x[5];            /* activation-space       (eval_seeddelta_x[_block], sd_eval.cpp:997-1419) */
    double  cost[8];
    double  proxy[8];
    double  gpu_ms;             /* device time of this candidate (HIP events on the library stream)      */
    int32_t has_w;              /* w_scale was given: em_w / em_x_w are valid                            */
    int32_t reserved;
    double  em_w[5];            /* weight-space metrics with the imatrix weighting                       */
    double  em_x_w[5];          /* activation-space metrics with x drawn as N(0,1) * w_scale             */
    double  delta_clamp_scale;  /* factor applied by delta_norm_clamp_down, 1.0 when it did not act       */
} sdh_scan_result;

/* The builder's per-input weighting from an importance matrix row (sd_make_imatrix_sqrt_scale,
 * tools/seeddelta-build/sd_utils.cpp:254-295): s_i = max(imatrix_i, eps)^(power/2), NaN/Inf -> 1,
 * normalised to mean 1. Host arithmetic (powf), identical to the reference's on the same libm. */
int  sdh_imatrix_scale(const float * imatrix, int64_t n_in, float eps, float power, float * scale_out);

int  sdh_scan_create(uint32_t seed, sdh_scan ** out);
void sdh_scan_destroy(sdh_scan * s);
/* Start a new builder run on the same handle: re-seed the RNG, keep the device buffers and the last fit. */
int  sdh_scan_reseed(sdh_scan * s, uint32_t seed);
/* Forget the memoised draws. The library keeps the results of the host routines that consume the
 * generator stream — the eval_x * n_in normals of a metric pass (with one device copy shared by all
 * handles), the shuffled row samples and the fit's row / permutation plan — keyed on the full generator
 * state and their arguments, because across a sweep the same (state, arguments) recurs for every layer
 * and every K. No counterpart in the reference; bench.py calls it before the timed sweep
 * so that the sweep pays for every distinct draw itself. */
void sdh_scan_memo_clear(void);
/* Evaluate one candidate on W. Wg/Wu/Wd (the layer's three dense weights) are only read by the FFN
 * proxy and may be NULL. Draws from the scan's RNG exactly as the builder does (base fit, then the
 * eval row sample, then the eval-x row sample and Gaussian activations). Blocks until done. */
int  sdh_scan_tensor(sdh_scan * s, const sdh_dense * W, const sdh_dense * Wg, const sdh_dense * Wu,
                     const sdh_dense * Wd, const sdh_scan_params * p, sdh_scan_result * out);
/* A K schedule of one tensor from one fit and one selection. The builder fits the base once per tensor
 * (tools/seeddelta-build/sd_build.cpp:444-451) and then runs one trial per K of the schedule (:843-913), each
 * of which dequantises, forms the residual and sorts again; here the rows are selected and sorted once at the
 * largest budget — descending score, ties by ascending index (sd_eval.cpp:53-77), so the top-K of every
 * smaller K is a prefix of each stored row — and the row scales and metric passes of all n_k budgets are
 * queued behind that one selection, with one copy and one synchronisation for the whole schedule.
 * outs[j] equals, bit for bit, what sdh_scan_tensor with p->K = Ks[j] returns from the same generator state
 * (p->K itself is ignored); the generator advances once, as for one candidate: every K of a layer-scan grid is
 * a builder run of its own with the same seed, so all of them see the same draws. gpu_ms is an equal share of
 * the schedule's device time. 1 <= n_k <= SDH_SCAN_MAX_K. With delta_norm_clamp_down acting (ffn_down, COO)
 * the stored values depend on K: SDH_ERR_UNSUPPORTED for n_k > 1 — evaluate those budgets one by one. */
#define SDH_SCAN_MAX_K 16
int  sdh_scan_tensor_kgrid(sdh_scan * s, const sdh_dense * W, const sdh_dense * Wg, const sdh_dense * Wu,
                           const sdh_dense * Wd, const sdh_scan_params * p, const int64_t * Ks, int n_k,
                           sdh_scan_result * outs);
/* Choose the member j of the last schedule that sdh_scan_fetch / sdh_scan_fetch_packed return (after
 * sdh_scan_tensor_kgrid: the last one; after sdh_scan_tensor there is only member 0). */
int  sdh_scan_pick(sdh_scan * s, int j);
/* After sdh_scan_tensor_kgrid with kgrid_draws = 1: set the generator to its state after trial j (the autotune loop
 * breaks at the first trial that passes its gate and draws nothing for the later ones, sd_build.cpp:905-913). */
int  sdh_scan_rewind(sdh_scan * s, int j);
/* Advance the RNG as sdh_scan_tensor would for a tensor of these dims, without touching the GPU:
 * lets a rank that owns (layer, ffn_up) reproduce the stream position after ffn_gate (sharding). */
int  sdh_scan_skip(sdh_scan * s, int64_t n_in, int64_t n_out, const sdh_scan_params * p);
/* Copy out what the last sdh_scan_tensor produced (any pointer may be NULL): idx int32 [n_out][K_eff]
 * or [n_out][n_blocks]; val f32 [n_out][K_eff] or [n_out][n_blocks][block]; row_scale f32 [n_out];
 * base arrays [B][L] (d2, h2 f32; perm1, perm1_inv int32) — the staging the builder packs into the
 * GGUF (sd_build.cpp:1104-1236). */
int  sdh_scan_fetch(sdh_scan * s, int32_t * idx, float * val, float * row_scale,
                    float * d2, float * h2, int32_t * perm1, int32_t * perm1_inv);
/* The same outputs as the GGUF stores them: the builder's packing step (sd_build.cpp:1104-1236) run on
 * the device before the copy, so half the bytes cross PCIe and no host pass is needed. idx_type
 * SDH_TYPE_I16: int32 -> int16 with anything outside [-32768, 32767] -> -1 (:1132, :1162), SDH_TYPE_I32:
 * as is; val_type SDH_TYPE_F16: ggml_fp32_to_fp16_row (ggml/src/ggml.c:457-462; round to nearest even,
 * NaN -> sign | 0x7E00), SDH_TYPE_F32: as is; row_scale always F16 (:1188-1191); d2 F16 [B][L];
 * perm1 int16 with out-of-range entries -> 0 when L <= 32768 (:1111, :1228), int32 otherwise
 * (*perm1_type reports which; the caller's buffer must hold B*L int32 in the worst case).
 * base_d1 / base_d3 are all ones in the builder's output (sd_eval.cpp:450-452) and are not returned.
 * Any pointer may be NULL. */
int  sdh_scan_fetch_packed(sdh_scan * s, int idx_type, void * idx, int val_type, void * val,
                           uint16_t * row_scale_f16, uint16_t * d2_f16, void * perm1, int * perm1_type);

/* ---- measurement hooks (no counterpart in the reference; used by bench.py) -------------------
 * sdh_profile_enable(1): from now on every sdh_forward brackets its dominant kernel (the
 * residual/epilogue kernel that streams the payload) with a pair of hipEvents recorded on the
 * stream the kernel is launched on. sdh_profile_read waits for the recorded events, returns the
 * summed elapsed milliseconds and the number of bracketed launches since the last read, and
 * resets both. sdh_profile_enable(0) stops recording. Not for use under graph capture. */
int sdh_profile_enable(int on);
int sdh_profile_read(double * total_ms, int64_t * launches);
/* Span form, free of per-launch overhead: sdh_profile_span_begin records one hipEvent on `stream`;
 * sdh_profile_span_end records a second one, waits for it, and returns the elapsed milliseconds
 * and the number of sdh_forward calls issued in between. For an op that is one kernel per call
 * (the fused decode launch) elapsed / launches is that kernel's back-to-back launch period. */
int sdh_profile_span_begin(void * stream);
int sdh_profile_span_end(void * stream, double * elapsed_ms, int64_t * forwards);
/* The device's floor for a launch that reads `bytes` once and does nothing else (a wave per 3072-byte run, 16-byte
 * non-temporal loads, one store per wave), launched back to back on `stream` over n_buffers distinct buffers the call
 * allocates and frees (pick n_buffers * bytes above the 256 MB Infinity Cache): microseconds per launch in *us_stream,
 * and of an empty launch of the same shape in *us_empty (may be NULL). bench.py reports it as roofline.attainable:
 * what a one-launch-per-token decode of that payload size could reach at best on this device. */
int sdh_probe_stream_floor(int64_t bytes, int n_buffers, int reps, void * stream, double * us_stream, double * us_empty);

#ifdef __cplusplus
}
#endif
#endif /* SEEDDELTA_HIP_H */
