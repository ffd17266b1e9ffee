// sdh_delta.hip — the sparse residual Δ·x, fused with the base term, row_scale and the
// non-finite accounting, on gfx950.
//
//   y[o,t] = rs[o] * ( base[o,t] + sum_r val[r,o] * x[idx[r,o], t] )                    (COO)
//   y[o,t] = rs[o] * ( base[o,t] + sum_bi sum_j bval[j,bi,o] * x[bidx[bi,o]*block+j,t]) (block)
//
// Reference semantics (src/llama-seeddelta.cpp): COO entries with idx<0 or idx>=n_in are skipped
// (:693); blocks with blk<0 or blk*block>=n_in are skipped and the last block is clipped to n_in
// (:929-936); the base term is v_b[o-bL] for tall weights (:1106) and sum_b v_b[o], accumulated
// from 0.0f in block order, for wide ones (:1132,:1148-1150,:1154); the scale is applied only if
// it differs from 1 (:645); NaN/Inf outputs are counted (:382).
//
// Two kernels:
//   delta_rows_kernel   — the fast path. Each output row is owned by a group of G lanes of one
//                          wave64 (G = 1..64, power of two); every lane streams 8 consecutive payload
//                          elements per step with 16-byte loads (the payload is row-contiguous in
//                          the GGUF layout), gathers x through L1/L2 and the group reduces with
//                          cross-lane shuffles. fp32 accumulation, order differs from the CPU op
//                          (tolerance-checked).
//   delta_exact_kernel  — verification mode: one lane per output, the reference's sequential
//                          order with separately rounded multiply and add; bit-identical to the CPU.
#include "sdh_internal.h"

#include <hip/hip_fp16.h>

namespace sdh {

struct DeltaParams {
    const void *  idx;
    const void *  val;
    const float * rs;
    const float * x;
    float *       y;
    const float * v;        // base scratch [T][B][L] or null
    int *         nf;       // non-finite counter
    int64_t ldx, ldy;
    int n_in, n_out;
    int Ktot;               // COO: K ; block: nb*block
    int nb, block;          // block scheme only
    int L, B, tall;
    int G;                  // lanes per row
    int zero_nf;
    int x_vec_ok;           // x columns are 16-byte aligned
    int block_shift;        // log2(block) if a power of two, else -1
    int n_blk_in;           // ceil(n_in / block): block ids below this start inside x
};

__device__ __forceinline__ float h2f(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}

template <typename T> struct Vec8;     // 8 consecutive payload elements
template <> struct Vec8<uint16_t> {    // f16 values or i16 indices: one 16-byte load
    uint4 q;
    __device__ __forceinline__ void load(const uint16_t * p) { q = *reinterpret_cast<const uint4 *>(p); }
    __device__ __forceinline__ uint16_t get(int j) const {
        const uint32_t w = (j < 2) ? q.x : (j < 4) ? q.y : (j < 6) ? q.z : q.w;
        return (uint16_t) ((j & 1) ? (w >> 16) : (w & 0xFFFFu));
    }
};
template <> struct Vec8<uint32_t> {    // f32 values or i32 indices: two 16-byte loads
    uint4 a, b;
    __device__ __forceinline__ void load(const uint32_t * p) {
        a = *reinterpret_cast<const uint4 *>(p);
        b = *reinterpret_cast<const uint4 *>(p + 4);
    }
    __device__ __forceinline__ uint32_t get(int j) const {
        switch (j) {
            case 0: return a.x; case 1: return a.y; case 2: return a.z; case 3: return a.w;
            case 4: return b.x; case 5: return b.y; case 6: return b.z; default: return b.w;
        }
    }
};

template <typename VAL_T> __device__ __forceinline__ float val_to_f32(VAL_T raw);
template <> __device__ __forceinline__ float val_to_f32<uint16_t>(uint16_t raw) { return h2f(raw); }
template <> __device__ __forceinline__ float val_to_f32<uint32_t>(uint32_t raw) { return __uint_as_float(raw); }

template <typename IDX_T> __device__ __forceinline__ int idx_to_i32(IDX_T raw);
template <> __device__ __forceinline__ int idx_to_i32<uint16_t>(uint16_t raw) { return (int) (int16_t) raw; }
template <> __device__ __forceinline__ int idx_to_i32<uint32_t>(uint32_t raw) { return (int) raw; }

// base term of output (o, t): see header comment
__device__ __forceinline__ float base_term(const DeltaParams & p, int o, int64_t t) {
    if (!p.v) return 0.0f;
    const float * vt = p.v + (size_t) t * p.B * p.L;
    if (p.tall) return vt[o];                       // b*L + (o - b*L) == o
    float yh = 0.0f;
    for (int b = 0; b < p.B; ++b) yh += vt[(size_t) b * p.L + o];
    return yh;
}

__device__ __forceinline__ void finish(const DeltaParams & p, int o, int64_t t, float y) {
    const float scale = p.rs ? p.rs[o] : 1.0f;
    float out = (scale != 1.0f) ? (y * scale) : y;
    if (!isfinite(out)) {
        atomicAdd(p.nf, 1);
        if (p.zero_nf) out = 0.0f;
    }
    p.y[o + t * p.ldy] = out;
}

// Fast path. STEPS = number of 8-element steps each lane owns (compile time, so that every
// payload / index / activation load of the row is issued before the first use: one HBM round
// trip for the payload and the indices, one L2 round trip for the gathered activations).
// STEPS == 0 selects a runtime loop over chunks of 4 steps for very long rows.
// Everything is branch-free: out-of-range steps and skipped blocks load from a clamped (valid)
// address and their partial sum is discarded by a select, so hipcc keeps all loads in flight.
// Requires Ktot % 8 == 0, (block % 8 == 0 and n_in % 8 == 0 for the block scheme) and 16-byte
// aligned x columns; other shapes take delta_rows_scalar_kernel.
template <typename IDX_T, typename VAL_T, bool BLOCK, int STEPS>
__global__ __launch_bounds__(256)
void delta_rows_kernel(const DeltaParams p) {
    constexpr int NS = STEPS > 0 ? STEPS : 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = p.G;
    const int l = lane & (G - 1);
    const int o = ((int) blockIdx.y * 4 + wave) * (64 / G) + (lane / G);
    const int64_t t = blockIdx.x;
    const bool row_ok = o < p.n_out;
    const int orow = row_ok ? o : 0;

    const float * xt = p.x + t * p.ldx;
    const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) orow * p.Ktot;
    const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) orow * (BLOCK ? p.nb : p.Ktot);

    // epilogue operands, requested first so they are home long before the reduction ends
    float scale = 1.0f, vb[8];
    const bool leader = row_ok && l == 0;
    const int nbase = p.v ? (p.tall ? 1 : p.B) : 0;
    if (leader) {
        if (p.rs) scale = p.rs[o];
        const float * vt = p.v + (size_t) t * p.B * p.L;
#pragma unroll
        for (int b = 0; b < 8; ++b) vb[b] = (b < nbase) ? vt[(size_t) b * p.L + o] : 0.0f;
    }

    float acc = 0.0f;
    const int chunk = G * 8 * NS;
    for (int base = 0; base < (STEPS > 0 ? 1 : p.Ktot); base += chunk) {
        Vec8<VAL_T> vv[NS];
        bool ok[NS];
        int e0[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int e = base + (u * G + l) * 8;
            ok[u] = row_ok & (e < p.Ktot);
            e0[u] = ok[u] ? e : 0;
        }
        if constexpr (BLOCK) {
            int blk[NS], off[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {       // block indices first: they head the dependent chain
                const int bi = e0[u] >> p.block_shift;            // block is a power of two on this path
                off[u] = e0[u] & (p.block - 1);
                blk[u] = idx_to_i32<IDX_T>(irow[bi]);
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) vv[u].load(vrow + e0[u]);
            float4 xa[NS], xb[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                // valid iff 0 <= blk and blk*block < n_in  <=>  (unsigned) blk < ceil(n_in / block)
                // and the last block is clipped to n_in: with n_in % 8 == 0 an 8-run is all in or all out
                const int raw = (blk[u] << p.block_shift) + off[u];
                ok[u] = ok[u] & ((unsigned) blk[u] < (unsigned) p.n_blk_in) & (raw < p.n_in);
                const int in0 = ok[u] ? raw : 0;
                xa[u] = *reinterpret_cast<const float4 *>(xt + in0);
                xb[u] = *reinterpret_cast<const float4 *>(xt + in0 + 4);
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                float ps = val_to_f32<VAL_T>(vv[u].get(0)) * xa[u].x;
                ps += val_to_f32<VAL_T>(vv[u].get(1)) * xa[u].y;
                ps += val_to_f32<VAL_T>(vv[u].get(2)) * xa[u].z;
                ps += val_to_f32<VAL_T>(vv[u].get(3)) * xa[u].w;
                ps += val_to_f32<VAL_T>(vv[u].get(4)) * xb[u].x;
                ps += val_to_f32<VAL_T>(vv[u].get(5)) * xb[u].y;
                ps += val_to_f32<VAL_T>(vv[u].get(6)) * xb[u].z;
                ps += val_to_f32<VAL_T>(vv[u].get(7)) * xb[u].w;
                acc += ok[u] ? ps : 0.0f;
            }
        } else {
            Vec8<IDX_T> ii[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) ii[u].load(irow + e0[u]);
#pragma unroll
            for (int u = 0; u < NS; ++u) vv[u].load(vrow + e0[u]);
            float xs[NS][8];
            bool okj[NS][8];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = idx_to_i32<IDX_T>(ii[u].get(j));
                    okj[u][j] = ok[u] & ((unsigned) i < (unsigned) p.n_in);
                    xs[u][j] = xt[okj[u][j] ? i : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pr = val_to_f32<VAL_T>(vv[u].get(j)) * xs[u][j];
                    acc += okj[u][j] ? pr : 0.0f;
                }
            }
        }
    }
    for (int m = G >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (leader) {
        float yh = 0.0f;
        if (p.v) {
            if (p.tall) yh = vb[0];
            else {
#pragma unroll
                for (int b = 0; b < 8; ++b) if (b < nbase) yh += vb[b];
                for (int b = 8; b < nbase; ++b) yh += p.v[((size_t) t * p.B + b) * p.L + o];
            }
        }
        const float y = yh + acc;
        float out = (scale != 1.0f) ? (y * scale) : y;
        if (!isfinite(out)) {
            atomicAdd(p.nf, 1);
            if (p.zero_nf) out = 0.0f;
        }
        p.y[o + t * p.ldy] = out;
    }
}

// Scalar-load variant for payloads whose rows are not 16-byte multiples (K % 8 != 0, odd block
// lengths): lanes of the group stride over the row one element at a time.
template <typename IDX_T, typename VAL_T, bool BLOCK>
__global__ __launch_bounds__(256)
void delta_rows_scalar_kernel(const DeltaParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = p.G;
    const int l = lane & (G - 1);
    const int o = ((int) blockIdx.y * 4 + wave) * (64 / G) + (lane / G);
    const int64_t t = blockIdx.x;
    const bool row_ok = o < p.n_out;
    const int orow = row_ok ? o : 0;
    const float * xt = p.x + t * p.ldx;
    const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) orow * p.Ktot;
    const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) orow * (BLOCK ? p.nb : p.Ktot);
    float acc = 0.0f;
    if (row_ok) {
        for (int e = l; e < p.Ktot; e += G) {
            int i;
            if constexpr (BLOCK) {
                const int bi  = e / p.block;
                const int blk = idx_to_i32<IDX_T>(irow[bi]);
                const int64_t s = (int64_t) blk * p.block;
                i = (blk >= 0 && s < p.n_in) ? (int) s + (e - bi * p.block) : -1;
            } else {
                i = idx_to_i32<IDX_T>(irow[e]);
            }
            if (i >= 0 && i < p.n_in) acc += val_to_f32<VAL_T>(vrow[e]) * xt[i];
        }
    }
    for (int m = G >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (row_ok && l == 0) finish(p, o, t, base_term(p, o, t) + acc);
}

// Verification mode: the reference's exact accumulation order, one lane per output.
template <typename IDX_T, typename VAL_T, bool BLOCK>
__global__ __launch_bounds__(256)
void delta_exact_kernel(const DeltaParams p, int64_t T) {
#pragma clang fp contract(off)
    const int64_t gid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t) p.n_out * T) return;
    const int o = (int) (gid % p.n_out);
    const int64_t t = gid / p.n_out;
    const float * xt = p.x + t * p.ldx;
    const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) o * p.Ktot;
    const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) o * (BLOCK ? p.nb : p.Ktot);
    float y = base_term(p, o, t);
    if constexpr (BLOCK) {
        for (int bi = 0; bi < p.nb; ++bi) {
            const int blk = idx_to_i32<IDX_T>(irow[bi]);
            if (blk < 0) continue;
            const int64_t in0 = (int64_t) blk * p.block;
            if (in0 >= p.n_in) continue;
            const int len = (int) ((in0 + p.block < p.n_in) ? p.block : (p.n_in - in0));
            const VAL_T * vb = vrow + (size_t) bi * p.block;
            for (int j = 0; j < len; ++j) {
                const float prod = val_to_f32<VAL_T>(vb[j]) * xt[in0 + j];
                y = y + prod;
            }
        }
    } else {
        for (int r = 0; r < p.Ktot; ++r) {
            const int i = idx_to_i32<IDX_T>(irow[r]);
            if (i < 0 || i >= p.n_in) continue;
            const float prod = val_to_f32<VAL_T>(vrow[r]) * xt[i];
            y = y + prod;
        }
    }
    finish(p, o, t, y);
}

template <typename IDX_T, typename VAL_T, bool BLOCK>
static int launch_typed(const DeviceWeight & w, const DeltaLaunch & a, DeltaParams & p, hipStream_t s) {
    if (a.exact_order) {
        const int64_t n = (int64_t) p.n_out * a.T;
        const int64_t blocks = (n + 255) / 256;
        hipLaunchKernelGGL((delta_exact_kernel<IDX_T, VAL_T, BLOCK>), dim3((unsigned) blocks), dim3(256), 0, s, p, a.T);
        SDH_HIP_CHECK(hipGetLastError());
        return SDH_OK;
    }
    // vector path: 8 payload elements (16 B of f16) per lane per step
    const bool vec_ok = (p.Ktot % 8 == 0) && p.x_vec_ok &&
                        (!BLOCK || (p.block % 8 == 0 && p.block_shift >= 0 && p.n_in % 8 == 0));
    int G = 1;
    if (vec_ok) { while (G < 64 && G * 8 < p.Ktot) G <<= 1; }
    else        { while (G < 64 && G * 4 < p.Ktot) G <<= 1; }
    p.G = G;
    const int rows_per_block = 4 * (64 / G);
    const unsigned gy = (unsigned) ((p.n_out + rows_per_block - 1) / rows_per_block);
    if (gy > 65535u || a.T > 0x7fffffffLL) {
        set_error("grid too large (n_out=%d, T=%lld)", p.n_out, (long long) a.T);
        return SDH_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned) a.T, gy);
    if (vec_ok) {
        const int steps = (p.Ktot + G * 8 - 1) / (G * 8);
        switch (steps) {
            case 1:  hipLaunchKernelGGL((delta_rows_kernel<IDX_T, VAL_T, BLOCK, 1>), grid, dim3(256), 0, s, p); break;
            case 2:  hipLaunchKernelGGL((delta_rows_kernel<IDX_T, VAL_T, BLOCK, 2>), grid, dim3(256), 0, s, p); break;
            case 3:  hipLaunchKernelGGL((delta_rows_kernel<IDX_T, VAL_T, BLOCK, 3>), grid, dim3(256), 0, s, p); break;
            case 4:  hipLaunchKernelGGL((delta_rows_kernel<IDX_T, VAL_T, BLOCK, 4>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((delta_rows_kernel<IDX_T, VAL_T, BLOCK, 0>), grid, dim3(256), 0, s, p); break;
        }
    } else {
        hipLaunchKernelGGL((delta_rows_scalar_kernel<IDX_T, VAL_T, BLOCK>), grid, dim3(256), 0, s, p);
    }
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

int launch_delta(const DeviceWeight & w, const DeltaLaunch & a, hipStream_t s) {
    if (a.T <= 0) return SDH_OK;
    DeltaParams p{};
    p.idx = w.idx; p.val = w.val; p.rs = w.rs;
    p.x = a.x; p.y = a.y; p.v = a.v; p.nf = a.nf_counter;
    p.ldx = a.ldx; p.ldy = a.ldy;
    p.n_in = (int) w.n_in; p.n_out = (int) w.n_out;
    const bool is_block = (w.kind == SDH_OP_BLOCK || w.kind == SDH_OP_BASE_BLOCK);
    p.nb = (int) w.nb; p.block = (int) w.block;
    p.Ktot = is_block ? (int) (w.nb * w.block) : (int) w.K;
    p.L = (int) w.L; p.B = (int) w.B; p.tall = w.tall ? 1 : 0;
    p.zero_nf = a.zero_nonfinite ? 1 : 0;
    p.x_vec_ok = (((uintptr_t) a.x % 16) == 0 && (a.ldx % 4) == 0) ? 1 : 0;
    p.n_blk_in = is_block ? (int) ((w.n_in + w.block - 1) / w.block) : 0;
    p.block_shift = -1;
    if (is_block && w.block > 0 && (w.block & (w.block - 1)) == 0) { int sft = 0; while ((1LL << sft) < w.block) ++sft; p.block_shift = sft; }

    const bool i16 = w.idx_type == SDH_TYPE_I16;
    const bool f16 = w.val_type == SDH_TYPE_F16;
    if (is_block) {
        if (i16 && f16)  return launch_typed<uint16_t, uint16_t, true>(w, a, p, s);
        if (i16 && !f16) return launch_typed<uint16_t, uint32_t, true>(w, a, p, s);
        if (!i16 && f16) return launch_typed<uint32_t, uint16_t, true>(w, a, p, s);
        return launch_typed<uint32_t, uint32_t, true>(w, a, p, s);
    }
    if (i16 && f16)  return launch_typed<uint16_t, uint16_t, false>(w, a, p, s);
    if (i16 && !f16) return launch_typed<uint16_t, uint32_t, false>(w, a, p, s);
    if (!i16 && f16) return launch_typed<uint32_t, uint16_t, false>(w, a, p, s);
    return launch_typed<uint32_t, uint32_t, false>(w, a, p, s);
}

} // namespace sdh
