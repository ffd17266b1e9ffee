// sdh_internal.h — private declarations shared by the translation units of libseeddelta_hip.so.
// gfx950 (MI355X) only. Nothing here crosses the C ABI.
#pragma once
#include <atomic>
#include <cstring>
#include <exception>
#include <new>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/seeddelta_hip.h"

namespace sdh {

// ---- allocation census: every hipMalloc / hipHostMalloc / hipFree of the library goes through these, so that
// sdh_debug_alloc_calls() can state how many a call sequence made (tests/test_zz_no_alloc.py asserts zero)
extern std::atomic<long long> g_alloc_calls;
inline hipError_t counted_malloc(void ** p, size_t n) { g_alloc_calls.fetch_add(1, std::memory_order_relaxed); return hipMalloc(p, n); }
inline hipError_t counted_host_malloc(void ** p, size_t n, unsigned f) { g_alloc_calls.fetch_add(1, std::memory_order_relaxed); return hipHostMalloc(p, n, f); }
inline hipError_t counted_free(void * p) { if (p) g_alloc_calls.fetch_add(1, std::memory_order_relaxed); return hipFree(p); }
#define hipMalloc(p, n) ::sdh::counted_malloc((void **) (p), (n))
#define hipHostMalloc(p, n, f) ::sdh::counted_host_malloc((void **) (p), (n), (f))
#define hipFree(p) ::sdh::counted_free((void *) (p))

// ---- error plumbing ----------------------------------------------------------------------
void set_error(const char * fmt, ...);
#define SDH_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            sdh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SDH_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

// Where the per-block base parameters live inside the packed parameter slab that the hot-path
// kernels read (bytes from the start of the block's slab; -1 = tensor absent, or all ones and
// elided because x * 1.0f == x exactly). One contiguous region per base block: a cold block
// costs one page walk and one HBM round trip.
struct BasePackDesc {
    const uint8_t * pack = nullptr;   // device
    int blk_stride = 0;               // bytes per base block
    int off_p1 = -1, off_d2 = -1, off_d1 = -1, off_d3 = -1, off_p2 = -1;
    int wide32 = 0;                   // 1: f32 / i32 parameters, 0: f16 / i16 (as the builder writes them)
};

// ---- device-resident payload ---------------------------------------------------------------
// All tensors are repacked contiguous at upload; element types of idx/val are kept as stored
// in the GGUF (they are the HBM stream), small per-row / per-base-block vectors are widened
// exactly (f16 -> f32, i16 -> i32, out-of-range perm entries -> 0 as the reference reads them).
struct DeviceWeight {
    int     device = 0;
    int     kind = 0;            // sdh_op_kind
    int64_t n_in = 0, n_out = 0;
    // residual
    int     idx_type = 0;        // SDH_TYPE_I16 / I32
    int     val_type = 0;        // SDH_TYPE_F16 / F32
    int64_t K = 0;               // COO: entries per row
    int64_t nb = 0, block = 0;   // block scheme: blocks per row, block length
    void *  idx = nullptr;       // [n_out][K] or [n_out][nb]
    void *  val = nullptr;       // [n_out][K] or [n_out][nb][block]
    float * rs = nullptr;        // [n_out] or null
    // base
    int64_t L = 0, B = 0;
    int     logL = 0;
    bool    tall = false;
    float * d1 = nullptr;        // [B][L]
    float * d2 = nullptr;        // [B][L] or null
    float * d3 = nullptr;        // [B][L] or null
    int *   p1 = nullptr;        // [B][L] or null
    int *   p2 = nullptr;        // [B][L] or null
    float * G = nullptr;         // [B][L]  FWHT(d2_b) in the reference butterfly order
    int *   inv = nullptr;       // [B][L]  perm1_b^-1
    bool    w0_closed_form = false;   // depth 2, no perm2, perm1 bijective
    uint8_t *    pack_dev = nullptr;  // packed 16/32-bit base parameters (hot path)
    BasePackDesc pack;
    // fused decode hand-off: [B][L] granules {value, epoch};
    // epoch bumped per fused launch
    uint8_t * handoff = nullptr;
    unsigned epoch = 0;
    // the hand-off buffer holds MAX_FUSED_TOKENS x B x L granules (small decode batches run as one launch)
    // accounting
    int64_t payload_bytes = 0;   // as stored in the GGUF
    // non-finite accounting: ring of per-call counters (no per-call memset on the hot path)
    static constexpr int NF_SLOTS = 256;
    int *    nf_ring = nullptr;  // device, NF_SLOTS ints
    uint64_t seq = 0;            // forwards issued
    hipStream_t last_stream = nullptr;
    // workspace for the base term v[T_chunk][B][L]
    float * ws = nullptr;
    const float * x4_hint = nullptr;  // token-interleaved copy of the next call's x (see DeltaLaunch::x4); set and cleared by the FFN path
    bool    base_ready = false;  // ws already holds the base terms of the next call's tokens (gate|up base launch of the FFN path)
    size_t  ws_bytes = 0;
    // the residual re-laid out for the folded MFMA contraction (sdh_w0fold.hip), built on first use: entries bucketed by
    // (128-row tile, 64-wide k-slab), fold_off [tiles][fold_ng] = {begin, end} of a bucket in fold_ent
    unsigned * fold_ent = nullptr; int2 * fold_off = nullptr; int fold_ng = 0, fold_spb = 0;
    bool fold_dup = false;       // some (row, input) pair occurs twice: the adds must be atomic
    bool fold_over = false;      // some bucket holds more than 2048 entries
    bool fold_perm = false;      // the buckets are in the permuted input order j = inv_b[q] (x is permuted by the pre-pass, the W0 tile is
    short * fold_p16 = nullptr;  // [B][L] the input each permuted column multiplies (index into the row of x), -1 = none
    _Float16 * fold_gp = nullptr;// then a plain XOR-circulant of G_b: copied from fold_gp [B][8][L], fold_gp[b][c][i] = f16(G_b[i ^ c]))
    // batch-1 decode with the base term formed by the row workgroups themselves (sdh_direct.hip), built at creation (one allocation)
    uint8_t * dir_space = nullptr;        // room for the two tables behind the packed base parameters (inside pack_dev), or null
    float * dir_d2h = nullptr;            // [B][L/16][16] d2 transformed over the upper index bits
    unsigned short * dir_p1b = nullptr;   // [B][L] 4 * perm1
};

struct LibState {
    bool        inited = false;
    int         device = -1;
    hipStream_t stream = nullptr;
    int         num_cu = 256;
    // staging for host-IO forwards
    float * hx = nullptr; size_t hx_bytes = 0;
    float * hy = nullptr; size_t hy_bytes = 0;
    // pinned bounce buffers for small host-IO transfers (a decode step): a pageable hipMemcpy stages and
    // synchronises inside the driver, a pinned one is a plain DMA
    static constexpr size_t PIN_BYTES = 512 * 1024;
    float * pin_x = nullptr; float * pin_y = nullptr;
    unsigned * census_ctr = nullptr;    // residency census of the fused decode kernel (sdh_decode.hip)
    // gated FFN block (sdh_ffn.hip): f16 GELU table as ggml builds it, intermediate activations
    uint16_t * gelu_tab = nullptr;      // [65536] device
    float * ffn_ws = nullptr; size_t ffn_ws_bytes = 0;
    int ffn_last_fused = 0;             // the last gate/up evaluation ran as the single gate|up launch
    int last_decode_kind = 0;           // sdh_debug_last_decode_kind: 1 direct, 2 fused, 3 two kernels
    // measurement hooks (sdh_profile_*): event pairs around the dominant kernel
    bool   prof_on = false;
    static constexpr int PROF_PAIRS = 4096;
    hipEvent_t * prof_ev = nullptr;     // 2*PROF_PAIRS events, created on first enable
    int     prof_used = 0;              // pairs recorded and not yet read
    double  prof_ms = 0.0;              // accumulated from flushed pairs
    int64_t prof_n = 0;
    hipEvent_t span_ev[2] = {nullptr, nullptr};   // sdh_profile_span_*
    int64_t span_forwards = 0;          // sdh_forward calls since span_begin
};
LibState & lib();

// ---- kernel launchers (sdh_base.hip / sdh_delta.hip) -------------------------------------------
// v[t][b][i] = (D3 H [D2 [P2] H] P1 D1 x_chunk)(i) for t in [0,T), b in [0,B)
int launch_base_apply_pair(const DeviceWeight & w0, const DeviceWeight & w1, const float * x, int64_t ldx, int64_t T,
                           float * v0, float * v1, hipStream_t s);
int launch_base_apply(const DeviceWeight & w, const float * x, int64_t ldx, int64_t T,
                      float * v, hipStream_t s);

int launch_base_apply_generic(const float * x, int64_t ldx, int n_in, int L, int B, bool tall,
                              const float * d1, const float * d2, const float * d3, const int * p1, const int * p2,
                              int64_t T, float * v, hipStream_t s);

constexpr int MAX_FUSED_TOKENS = 4;
// internal flag of sdh_forward_impl (gated FFN node launches): COO payloads with few tokens take the LDS rows kernel
// (fast-path arithmetic) rather than the tiled SpMM
constexpr int SDH_FWD_INTERNAL_ROWS = 0x40000000;

struct DeltaLaunch {
    const float * x; int64_t ldx; int64_t T;
    float * y; int64_t ldy;
    const float * v;        // base term scratch [T][B][L] or null
    int * nf_counter;       // device counter for this call
    bool zero_nonfinite;
    bool exact_order;
    int  v_blocks = 0;      // 0: v is [T][B][L] (FWHT form); 1: v is [T][L], base blocks already summed (MFMA form, wide)
    int64_t v_ld = 0;       // 0: tokens are v_blocks*L (or B*L) apart; else the stride (the MFMA form, tall: the base term
                            // is written straight into y and the residual kernel updates y in place, v == y, v_ld == ldy)
    // gated FFN, ffn_up node: the epilogue writes act(gate[t][o]) * y instead of y (the node's own guard runs first)
    const float * glu_gate = nullptr; int64_t glu_ldg = 0; int glu_act = 0; const uint16_t * glu_tab = nullptr;
    // 2..4 tokens: a token-interleaved copy of x, x4[4 * i + t] (slots past T zero), when the producer wrote one (the FFN
    // path's gate|up launch): a COO row entry is then one 16-byte gather for all tokens instead of one 4-byte gather per token
    const float * x4 = nullptr;
};
int launch_delta(const DeviceWeight & w, const DeltaLaunch & a, hipStream_t s);

// Prefill: base term as W0 tiles regenerated in registers and contracted on MFMA (sdh_w0mfma.hip).
// Returns SDH_OK if launched, 1 if the payload has no closed-form W0 (caller uses the FWHT form).
int launch_w0_mfma(const DeviceWeight & w, const float * x, int64_t ldx, int64_t T, float * v_out, hipStream_t s);
size_t w0_mfma_x16_bytes(const DeviceWeight & w, int64_t T);
int launch_w0_mfma_tiled(const DeviceWeight & w, const float * x, int64_t ldx, int64_t T, _Float16 * x16_ws,
                         float * out, int64_t ldo, hipStream_t s);

int launch_w0_x16(const float * x, int64_t ldx, int64_t T, int n_in, _Float16 * x16, int64_t ld16, int64_t T_pad, hipStream_t s);
// Prefill as one dense contraction: residual entries added to the W0 tile in LDS (sdh_w0fold.hip). ws: w0_fold_ws_bytes.
// Returns SDH_OK if launched (y complete: row scale applied, non-finite outputs counted), 1 if the payload does not fit.
size_t w0_fold_ws_bytes(const DeviceWeight & w, int64_t T);
// glu_gate: as ffn_up inside a gated FFN the stored value is act(gate) * value (tall tensors, permuted kernel; else 1)
int  launch_w0_fold(DeviceWeight & w, const float * x, int64_t ldx, int64_t T, void * ws, float * y, int64_t ldy,
                    int * nf_counter, bool zero_nf, hipStream_t s,
                    const float * glu_gate = nullptr, int64_t glu_ldg = 0, int glu_act = 0, const uint16_t * glu_tab = nullptr);
// gate and up of a gated FFN as one two-pass launch (tall, one shape, permuted order); each handle's ws holds its x image
int  launch_w0_fold_pair(DeviceWeight & g, DeviceWeight & u, const float * x, int64_t ldx, int64_t T, float * h, int64_t ldh,
                         int * nf_g, int * nf_u, bool zero_nf, int act, const uint16_t * tab, hipStream_t s);
void fold_free(DeviceWeight & w);
bool w0_fold_permuted(const DeviceWeight & w);   // the handle's folded contraction runs (or would run) in the permuted order

// Prefill: token-tiled SpMM (sdh_spmm.hip). Returns SDH_OK if launched, 1 if the shape does not fit.
int launch_delta_spmm(const DeviceWeight & w, const DeltaLaunch & a, hipStream_t s);

// Batch-1 decode with a base as ONE launch (sdh_decode.hip). Returns SDH_OK if launched, 1 if the
// payload / call does not qualify (caller runs base + delta kernels instead), <0 on error.
int launch_decode_fused(DeviceWeight & w, const DeltaLaunch & a, hipStream_t s);
// One token, closed-form base without d1: every 16-row workgroup forms its rows' base term itself (no hand-off).
// Returns SDH_OK if launched, 1 if the payload / call does not qualify, <0 on error.
int launch_decode_direct(DeviceWeight & w, const DeltaLaunch & a, hipStream_t s);
int  direct_build_tables(DeviceWeight & w, hipStream_t s);   // at weight creation; a payload without tables never takes the direct launch
void direct_free(DeviceWeight & w);
// 1..MAX_FUSED_TOKENS tokens, gate and up tall base payloads of one shape: h[t][o] = act(gate(x_t)[o]) * up(x_t)[o] in
// one launch. Returns SDH_OK, 1 if the pair does not qualify, <0 on error. nf_g / nf_u: the two nodes' counters.
int launch_decode_pair(DeviceWeight & g, DeviceWeight & u, int act, const uint16_t * gelu_tab,
                       const float * x, int64_t ldx, int n_tok, float * h, int64_t ldh, float * h4,
                       int * nf_g, int * nf_u, bool zero_nonfinite, hipStream_t s);
int launch_delta_rows_pair(const DeviceWeight & g, const DeviceWeight & u, const DeltaLaunch & ag, const DeltaLaunch & au,
                           int act, const uint16_t * gelu_tab, float * h, int64_t ldh, hipStream_t s);
int launch_glu(int act, const uint16_t * gelu_tab, const float * g, const float * u, float * y, int64_t n, hipStream_t s);

// Persistent FFN stack (sdh_stack.hip): n_layers gated FFN blocks at T = 1 in one launch.
struct FfnStack;
int  stack_create(const sdh_weight * const * gate, const sdh_weight * const * up, const sdh_weight * const * down, int n_layers, FfnStack ** out);
void stack_free(FfnStack * st);
int  stack_forward(FfnStack * st, int act, const uint16_t * gelu_tab, const float * x, float * y, bool zero_nf, hipStream_t s);
int  stack_status(FfnStack * st, int64_t * nonfinite, int64_t * timeouts);
int  stack_stamps(FfnStack * st, unsigned long long * out);
int  stack_layers(const FfnStack * st);
int64_t stack_bytes(const FfnStack * st);

int launch_w0_tile(const DeviceWeight & w, int64_t o0, int64_t o1, int64_t i0, int64_t i1,
                   float * out_dev, hipStream_t s);

// f32 -> f16 bits, round to nearest even, NaN -> 0x7E00 (the result of ggml's GGML_FP32_TO_FP16)
inline uint16_t float_to_half(float fl) {
    uint32_t f;
    memcpy(&f, &fl, 4);
    const uint32_t f32infty = 255u << 23, f16max = (127u + 16u) << 23, denorm_magic_u = ((127u - 15u) + (23u - 10u) + 1u) << 23;
    const uint32_t sign = f & 0x80000000u;
    uint16_t o;
    f ^= sign;
    if (f >= f16max) o = (f > f32infty) ? 0x7E00 : 0x7C00;
    else if (f < (113u << 23)) {                       // f16 subnormal: let the float adder do the rounding
        float t, dm;
        memcpy(&t, &f, 4); memcpy(&dm, &denorm_magic_u, 4);
        t += dm;
        uint32_t tb;
        memcpy(&tb, &t, 4);
        o = (uint16_t) (tb - denorm_magic_u);
    } else {
        const uint32_t mant_odd = (f >> 13) & 1u;
        f += ((uint32_t) (15 - 127) << 23) + 0xFFFu;
        f += mant_odd;
        o = (uint16_t) (f >> 13);
    }
    return (uint16_t) (o | (sign >> 16));
}


} // namespace sdh

// C ABI boundary: turn anything thrown by host code (std::bad_alloc from a vector, ...) into a status
#define SDH_NO_THROW(call)                                                                              \
    try { return (call); }                                                                              \
    catch (const std::bad_alloc &) { ::sdh::set_error("out of host memory"); return SDH_ERR_HIP; }      \
    catch (const std::exception & e) { ::sdh::set_error("internal error: %s", e.what()); return SDH_ERR_HIP; } \
    catch (...) { ::sdh::set_error("internal error"); return SDH_ERR_HIP; }

