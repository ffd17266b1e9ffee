// sdh_delta_dev.h — device code of the sparse residual Δ·x shared by the stand-alone residual
// kernels (sdh_delta.hip) and the fused decode kernel (sdh_decode.hip).
//
//   COO  : sum_r val[r,o] * x[idx[r,o]]                      idx<0 or idx>=n_in skipped (:693)
//   block: sum_bi sum_j bval[j,bi,o] * x[bidx[bi,o]*block+j] blk<0 or blk*block>=n_in skipped,
//                                                            last block clipped to n_in (:929-936)
// (reference: src/llama-seeddelta.cpp)
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "sdh_internal.h"

namespace sdh {

struct DeltaParams {
    const void *  idx;
    const void *  val;
    const float * rs;
    const float * x;
    float *       y;
    const float * v;        // base term [T][v_ld] (token t, block b, position p at t*v_ld + b*L + p) or null
    int64_t v_ld;           // B*L for the base kernels' scratch; ldy when the base term already sits in y (in place)
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
    const float * glu_gate; // ffn_up node of a gated FFN: gate outputs [T][glu_ldg], or null
    int64_t glu_ldg;
    int glu_act;
    const uint16_t * glu_tab;
};

// host: fill the parameter block for one launch (sdh_delta.hip)
DeltaParams make_delta_params(const DeviceWeight & w, const DeltaLaunch & a);
// host: lanes per row and 8-element steps per lane for the vector path; returns false if the
// payload needs the scalar kernel
bool delta_vector_shape(const DeltaParams & p, bool is_block, int * G, int * steps);

__device__ __forceinline__ float h2f(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}

// 16 bytes of payload. STREAM: non-temporal. A decode launch reads its payload exactly once, and 13 MB
// of it allocated in the caches push out the few hundred KB that ARE reused from launch to launch and
// sit at the head of the critical path (base parameters, x). Measured on cfg3: 8.92 -> 8.24 us.
template <bool STREAM>
__device__ __forceinline__ uint4 load16(const void * p) {
    if constexpr (STREAM) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4 *>(p);
    }
}

template <typename T> struct Vec8;     // 8 consecutive payload elements
template <> struct Vec8<uint16_t> {    // f16 values or i16 indices: one 16-byte load
    uint4 q;
    template <bool STREAM = false>
    __device__ __forceinline__ void load(const uint16_t * p) { q = load16<STREAM>(p); }
    __device__ __forceinline__ uint16_t get(int j) const {
        const uint32_t w = (j < 2) ? q.x : (j < 4) ? q.y : (j < 6) ? q.z : q.w;
        return (uint16_t) ((j & 1) ? (w >> 16) : (w & 0xFFFFu));
    }
};
template <> struct Vec8<uint32_t> {    // f32 values or i32 indices: two 16-byte loads
    uint4 a, b;
    template <bool STREAM = false>
    __device__ __forceinline__ void load(const uint32_t * p) {
        a = load16<STREAM>(p);
        b = load16<STREAM>(p + 4);
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

// The fast path's arithmetic, spelled out so that it does not depend on the compiler's contraction
// choices: a run of 8 is one rounded multiply and seven fused multiply-adds, a row's partial sum is a
// chain of rounded adds. Every kernel of the fast path (one token, several tokens, one or two launches)
// goes through these two, which is what makes their results bit-identical to one another.
template <typename VAL_T>
__device__ __forceinline__ float dot8(const Vec8<VAL_T> & v, const float4 & a, const float4 & b) {
    float ps = __fmul_rn(val_to_f32<VAL_T>(v.get(0)), a.x);
    ps = __fmaf_rn(val_to_f32<VAL_T>(v.get(1)), a.y, ps);
    ps = __fmaf_rn(val_to_f32<VAL_T>(v.get(2)), a.z, ps);
    ps = __fmaf_rn(val_to_f32<VAL_T>(v.get(3)), a.w, ps);
    ps = __fmaf_rn(val_to_f32<VAL_T>(v.get(4)), b.x, ps);
    ps = __fmaf_rn(val_to_f32<VAL_T>(v.get(5)), b.y, ps);
    ps = __fmaf_rn(val_to_f32<VAL_T>(v.get(6)), b.z, ps);
    ps = __fmaf_rn(val_to_f32<VAL_T>(v.get(7)), b.w, ps);
    return ps;
}
__device__ __forceinline__ float add_if(float acc, bool ok, float term) { return __fadd_rn(acc, ok ? term : 0.0f); }
// acc + v * x as a separately rounded product and sum. (__fadd_rn(acc, __fmul_rn(v, x)) does NOT do this: the two
// library calls inline to a plain fmul / fadd pair that the backend fuses into v_fmac_f32 — 60 % of the outputs of a
// row kernel written that way differed from the one-token kernel by an ulp or two. add_if's select between the product
// and the sum is what keeps them apart there.)
__device__ __forceinline__ float mul_then_add(float acc, float v, float x) {
#pragma clang fp contract(off)
    const float prod = v * x;
    return acc + prod;
}
// two tokens per instruction (v_pk_mul_f32, v_pk_add_f32): each component rounds as the scalar forms do
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t mul_then_add2(f32x2_t acc, float v, f32x2_t x) {
#pragma clang fp contract(off)
    const f32x2_t vv = {v, v};
    const f32x2_t prod = vv * x;
    return acc + prod;
}

// One lane's share of one row's dot product on the vector path. STEPS = number of 8-element
// steps each lane owns (compile time, so that every payload / index / activation load of the row
// is issued before the first use: one HBM round trip for the payload and the indices, one L2
// round trip for the gathered activations). STEPS == 0 selects a runtime loop over chunks of 4
// steps for very long rows. Everything is branch-free: out-of-range steps and skipped blocks load
// from a clamped (valid) address and their partial sum is discarded by a select, so hipcc keeps
// all loads in flight. Requires Ktot % 8 == 0, (block a power of two >= 8 and n_in % 8 == 0 for
// the block scheme) and 16-byte aligned x columns.
template <typename IDX_T, typename VAL_T, bool BLOCK, int STEPS, bool STREAM = false>
__device__ __forceinline__ float delta_row_partial(const DeltaParams & p, const float * __restrict__ xt,
                                                   int orow, bool row_ok, int l, int G) {
    constexpr int NS = STEPS > 0 ? STEPS : 4;
    const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) orow * p.Ktot;
    const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) orow * (BLOCK ? p.nb : p.Ktot);
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
                const int bi = e0[u] >> p.block_shift;
                off[u] = e0[u] & (p.block - 1);
                blk[u] = idx_to_i32<IDX_T>(STREAM ? __builtin_nontemporal_load(irow + bi) : irow[bi]);
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) vv[u].template load<STREAM>(vrow + e0[u]);
            float4 xa[NS], xb[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                // valid iff 0 <= blk and blk*block < n_in  <=>  (unsigned) blk < ceil(n_in / block);
                // the last block is clipped to n_in: with n_in % 8 == 0 an 8-run is all in or all out
                const int raw = (blk[u] << p.block_shift) + off[u];
                ok[u] = ok[u] & ((unsigned) blk[u] < (unsigned) p.n_blk_in) & (raw < p.n_in);
                const int in0 = ok[u] ? raw : 0;
                xa[u] = *reinterpret_cast<const float4 *>(xt + in0);
                xb[u] = *reinterpret_cast<const float4 *>(xt + in0 + 4);
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                acc = add_if(acc, ok[u], dot8<VAL_T>(vv[u], xa[u], xb[u]));
            }
        } else {
            Vec8<IDX_T> ii[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) ii[u].template load<STREAM>(irow + e0[u]);
#pragma unroll
            for (int u = 0; u < NS; ++u) vv[u].template load<STREAM>(vrow + e0[u]);
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
                    acc = add_if(acc, okj[u][j], __fmul_rn(val_to_f32<VAL_T>(vv[u].get(j)), xs[u][j]));
                }
            }
        }
    }
    return acc;
}

// The same partial sums for NT tokens at once (small decode batches): the payload of the row is loaded
// once, the activations of every token are gathered with it, and each token's sum is formed by exactly
// the operations delta_row_partial performs for that token alone. xoff[t]: element offset of token t's
// column in x. Compile-time STEPS only.
template <typename IDX_T, typename VAL_T, bool BLOCK, int STEPS, int NT, bool STREAM = false>
__device__ __forceinline__ void delta_row_partial_nt(const DeltaParams & p, const float * __restrict__ x, const int (&xoff)[NT],
                                                     int orow, bool row_ok, int l, int G, float (&acc)[NT]) {
    static_assert(STEPS > 0, "compile-time steps only");
    constexpr int NS = STEPS;
    const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) orow * p.Ktot;
    const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) orow * (BLOCK ? p.nb : p.Ktot);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.0f;
    Vec8<VAL_T> vv[NS];
    bool ok[NS];
    int e0[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int e = (u * G + l) * 8;
        ok[u] = row_ok & (e < p.Ktot);
        e0[u] = ok[u] ? e : 0;
    }
    if constexpr (BLOCK) {
        int blk[NS], off[NS], in0[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int bi = e0[u] >> p.block_shift;
            off[u] = e0[u] & (p.block - 1);
            blk[u] = idx_to_i32<IDX_T>(STREAM ? __builtin_nontemporal_load(irow + bi) : irow[bi]);
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) vv[u].template load<STREAM>(vrow + e0[u]);
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int raw = (blk[u] << p.block_shift) + off[u];
            ok[u] = ok[u] & ((unsigned) blk[u] < (unsigned) p.n_blk_in) & (raw < p.n_in);
            in0[u] = ok[u] ? raw : 0;
        }
        // tokens two at a time: the activations of a pair are all in flight together, and a pair is what
        // fits the register budget that keeps three 512-thread workgroups on a CU (residency of the launch)
        constexpr int TP = NT < 2 ? NT : 2;
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TP) {
            float4 xa[TP][NS], xb[TP][NS];
#pragma unroll
            for (int t = 0; t < TP; ++t) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    xa[t][u] = *reinterpret_cast<const float4 *>(x + xoff[t0 + t] + in0[u]);
                    xb[t][u] = *reinterpret_cast<const float4 *>(x + xoff[t0 + t] + in0[u] + 4);
                }
            }
#pragma unroll
            for (int t = 0; t < TP; ++t) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    acc[t0 + t] = add_if(acc[t0 + t], ok[u], dot8<VAL_T>(vv[u], xa[t][u], xb[t][u]));
                }
            }
            if (t0 + TP < NT) __builtin_amdgcn_sched_barrier(0);     // keep the next pair's loads behind this pair's sums
        }
    } else {
        Vec8<IDX_T> ii[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) ii[u].template load<STREAM>(irow + e0[u]);
#pragma unroll
        for (int u = 0; u < NS; ++u) vv[u].template load<STREAM>(vrow + e0[u]);
        int ic[NS][8];
        bool okj[NS][8];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = idx_to_i32<IDX_T>(ii[u].get(j));
                okj[u][j] = ok[u] & ((unsigned) i < (unsigned) p.n_in);
                ic[u][j] = okj[u][j] ? i : 0;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float xs[NS][8];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xs[u][j] = x[xoff[t] + ic[u][j]];
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[t] = add_if(acc[t], okj[u][j], __fmul_rn(val_to_f32<VAL_T>(vv[u].get(j)), xs[u][j]));
                }
            }
        }
    }
}

// COO rows against activations staged token-interleaved in LDS: xs4[i] = (x_0[i], x_1[i], x_2[i], x_3[i]).
// One 16-byte LDS read per entry replaces one scattered 4-byte global gather per entry per token, which is
// what bounds the multi-token COO launch (L1 request rate). Per token the products and the order of the
// adds are those of delta_row_partial, so results stay bit-identical to the one-token kernel.
template <typename IDX_T, typename VAL_T, int STEPS, bool STREAM = false>
__device__ __forceinline__ void delta_row_partial_coo_lds4(const DeltaParams & p, const float4 * __restrict__ xs4,
                                                           int orow, bool row_ok, int l, int G, float (&acc)[4]) {
    constexpr int NS = STEPS;
    const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) orow * p.Ktot;
    const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) orow * p.Ktot;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = 0.0f;
    Vec8<VAL_T> vv[NS];
    Vec8<IDX_T> ii[NS];
    bool ok[NS];
    int e0[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int e = (u * G + l) * 8;
        ok[u] = row_ok & (e < p.Ktot);
        e0[u] = ok[u] ? e : 0;
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) ii[u].template load<STREAM>(irow + e0[u]);
#pragma unroll
    for (int u = 0; u < NS; ++u) vv[u].template load<STREAM>(vrow + e0[u]);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        float4 xv[8];
        bool okj[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = idx_to_i32<IDX_T>(ii[u].get(j));
            okj[j] = ok[u] & ((unsigned) i < (unsigned) p.n_in);
            xv[j] = xs4[okj[j] ? i : 0];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = val_to_f32<VAL_T>(vv[u].get(j));
            acc[0] = add_if(acc[0], okj[j], __fmul_rn(v, xv[j].x));
            acc[1] = add_if(acc[1], okj[j], __fmul_rn(v, xv[j].y));
            acc[2] = add_if(acc[2], okj[j], __fmul_rn(v, xv[j].z));
            acc[3] = add_if(acc[3], okj[j], __fmul_rn(v, xv[j].w));
        }
    }
}

template <int CTRL> __device__ __forceinline__ float dpp_partner(float a) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_reduce_to_lane0(float a) {      // valid in lane 0 of every 16-lane row
    a = __fadd_rn(a, dpp_partner<0x108>(a));            // row_shl:8   lane i <- i + 8
    a = __fadd_rn(a, dpp_partner<0x104>(a));            // row_shl:4
    a = __fadd_rn(a, dpp_partner<0x4E>(a));             // quad_perm [2,3,0,1]
    a = __fadd_rn(a, dpp_partner<0xB1>(a));             // quad_perm [1,0,3,2]
    return a;
}
__device__ __forceinline__ float swap32_add(float a, float b) {         // lanes 0-31: a[i] + a[i+32]; lanes 32-63: b[i-32] + b[i]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __fadd_rn(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap16_add(float a, float b) {         // even rows: a[row] + a[row+1]; odd rows: b[row-1] + b[row]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __fadd_rn(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Sum over the G lanes of a group (G a power of two <= 64, groups aligned), valid in the group's lane 0 — the only
// lane any caller reads. The pairing is the butterfly's (lane i with lane i ^ m for m = G/2 .. 1, a rounded add per
// step), so lane 0 holds the bits `for m: acc += shfl_xor(acc, m)` gives; but the partner arrives by a half-register
// swap (m = 32, 16) or a DPP move inside the 16-lane row (m <= 8) instead of a ds_bpermute with its address
// arithmetic and LDS-pipe round trip per step, which stood on every wave's critical path between its last
// multiply-add and its store. The branches are uniform (G is a kernel argument).
__device__ __forceinline__ float group_reduce(float acc, int G) {
    if (G > 32) acc = swap32_add(acc, acc);
    if (G > 16) acc = swap16_add(acc, acc);
    if (G > 8)  acc = __fadd_rn(acc, dpp_partner<0x108>(acc));          // row_shl:8   lane i <- i + 8
    if (G > 4)  acc = __fadd_rn(acc, dpp_partner<0x104>(acc));          // row_shl:4
    if (G > 2)  acc = __fadd_rn(acc, dpp_partner<0x4E>(acc));           // quad_perm [2,3,0,1]
    if (G > 1)  acc = __fadd_rn(acc, dpp_partner<0xB1>(acc));           // quad_perm [1,0,3,2]
    return acc;
}

// scale (only if != 1, :645), count / optionally zero non-finite outputs (:363-392)
__device__ __forceinline__ float finish_output(const DeltaParams & p, float y, float scale) {
    float out = (scale != 1.0f) ? (y * scale) : y;
    if (!isfinite(out)) {
        atomicAdd(p.nf, 1);
        if (p.zero_nf) out = 0.0f;
    }
    return out;
}
// act(g) * u: the gate activation of a parallel-gated FFN as ggml's CPU ops compute it on F32
// (ggml_vec_swiglu_f32, ggml/src/ggml-cpu/vec.cpp:416 with ggml_silu_f32, vec.h:1061;
// ggml_vec_geglu_f32, vec.h:1430-1443: thresholds at +-10, f16 table in between).
// act: 0 = SiLU, 1 = GELU; tab: the 65536-entry f16 GELU table (GELU only).
__device__ __forceinline__ float glu_value(int act, float g, float u, const uint16_t * __restrict__ tab) {
    if (act == 0) return (g / (1.0f + expf(-g))) * u;
    if (g <= -10.0f) return 0.0f;
    if (g >= 10.0f) return g * u;
    return h2f(tab[__half_as_ushort(__float2half_rn(g))]) * u;
}

// One output of the op. As ffn_up inside a gated FFN (p.glu_gate set) the stored value is act(gate) * up: the node's
// guard has already run on the up value, as the reference's per-node guard does before ggml_swiglu_split sees it.
__device__ __forceinline__ void store_output(const DeltaParams & p, int o, int64_t t, float y, float scale) {
    float out = finish_output(p, y, scale);
    if (p.glu_gate) out = glu_value(p.glu_act, p.glu_gate[o + t * p.glu_ldg], out, p.glu_tab);
    p.y[o + t * p.ldy] = out;
}

// ---- token tiles of x in LDS (sdh_spmm.hip, delta_rows_*_lds kernels); layout notes in sdh_spmm.hip ----
template <bool BLOCK>
__device__ __forceinline__ int tile_row(const DeltaParams & p, int i, int n4) {
    if constexpr (!BLOCK) return (i & 3) * n4 + (i >> 2);
    return i + (p.block_shift >= 0 ? (i >> p.block_shift) : (i / p.block));
}
// Tile fill, GRP tokens at a time (one 4*GRP-byte LDS store per input and token group; tokens past the end of the batch
// are zero). xs: [rows][TT].
template <bool BLOCK, int TT, int GRP, int NTH>
__device__ __forceinline__ void spmm_fill_tile(const DeltaParams & p, float * xs, int64_t t0, int nt, int n4, int tid) {
    typedef float gvec __attribute__((ext_vector_type(GRP)));
    static_assert(TT % GRP == 0, "token groups tile the tile");
    if constexpr (!BLOCK && TT <= 8) {
        if (p.x_vec_ok) {                                // all TT tokens of a quad of inputs per trip: TT 16-byte loads in flight
            const int nq = p.n_in >> 2;
            const float * xt[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) xt[t] = p.x + (t0 + min(t, nt - 1)) * p.ldx;
            for (int g = tid; g < nq; g += NTH) {
                float4 v[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) v[t] = *reinterpret_cast<const float4 *>(xt[t] + 4 * g);
#pragma unroll
                for (int q = 0; q < TT; q += GRP) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gvec w;
#pragma unroll
                        for (int u = 0; u < GRP; ++u) {
                            const float4 & c4 = v[q + u];
                            const float c = j == 0 ? c4.x : j == 1 ? c4.y : j == 2 ? c4.z : c4.w;
                            w[u] = (q + u < nt) ? c : 0.0f;
                        }
                        *reinterpret_cast<gvec *>(xs + ((size_t) j * n4 + g) * TT + q) = w;
                    }
                }
            }
            for (int i = 4 * nq + tid; i < p.n_in; i += NTH) {       // the inputs of a last partial quad
#pragma unroll
                for (int t = 0; t < TT; ++t) xs[(size_t) tile_row<false>(p, i, n4) * TT + t] = (t < nt) ? xt[t][i] : 0.0f;
            }
            return;
        }
    }
    for (int q = 0; q < TT; q += GRP) {
        if (q >= nt) {                                   // a whole group past the end of the batch
            gvec z;
#pragma unroll
            for (int u = 0; u < GRP; ++u) z[u] = 0.0f;
            for (int i = tid; i < p.n_in; i += NTH) *reinterpret_cast<gvec *>(xs + (size_t) tile_row<BLOCK>(p, i, n4) * TT + q) = z;
            continue;
        }
        const float * xt[GRP];
#pragma unroll
        for (int u = 0; u < GRP; ++u) xt[u] = p.x + (t0 + min(q + u, nt - 1)) * p.ldx;
        int i = tid;
        if constexpr (!BLOCK) {
            if (p.x_vec_ok) {                            // 16-byte loads: two quads of inputs per trip
                const int nq = p.n_in >> 2;
                auto put = [&](int g, const float4 (&v)[GRP]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gvec w;
#pragma unroll
                        for (int u = 0; u < GRP; ++u) {
                            const float c = j == 0 ? v[u].x : j == 1 ? v[u].y : j == 2 ? v[u].z : v[u].w;
                            w[u] = (q + u < nt) ? c : 0.0f;
                        }
                        *reinterpret_cast<gvec *>(xs + ((size_t) j * n4 + g) * TT + q) = w;
                    }
                };
                int g = tid;
                for (; g + NTH < nq; g += 2 * NTH) {
                    float4 va[GRP], vb[GRP];
#pragma unroll
                    for (int u = 0; u < GRP; ++u) va[u] = *reinterpret_cast<const float4 *>(xt[u] + 4 * g);
#pragma unroll
                    for (int u = 0; u < GRP; ++u) vb[u] = *reinterpret_cast<const float4 *>(xt[u] + 4 * (g + NTH));
                    put(g, va);
                    put(g + NTH, vb);
                }
                for (; g < nq; g += NTH) {
                    float4 va[GRP];
#pragma unroll
                    for (int u = 0; u < GRP; ++u) va[u] = *reinterpret_cast<const float4 *>(xt[u] + 4 * g);
                    put(g, va);
                }
                i = 4 * nq + tid;                        // the inputs of a last partial quad, one by one below
            }
        }
        // one input per lane and trip, four trips' loads in flight
        for (; i + 3 * NTH < p.n_in; i += 4 * NTH) {
            gvec v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int u = 0; u < GRP; ++u) v[k][u] = xt[u][i + k * NTH];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int u = 0; u < GRP; ++u) if (q + u >= nt) v[k][u] = 0.0f;
                *reinterpret_cast<gvec *>(xs + (size_t) tile_row<BLOCK>(p, i + k * NTH, n4) * TT + q) = v[k];
            }
        }
        for (; i < p.n_in; i += NTH) {
            gvec v;
#pragma unroll
            for (int u = 0; u < GRP; ++u) v[u] = (q + u < nt) ? xt[u][i] : 0.0f;
            *reinterpret_cast<gvec *>(xs + (size_t) tile_row<BLOCK>(p, i, n4) * TT + q) = v;
        }
    }
}


// Lane's partner value under lane ^ m (m a power of two, uniform): half-register swaps for m = 32 / 16, DPP inside a
// 16-lane row below. The data movement of one butterfly step for lanes on BOTH sides of the exchange.
__device__ __forceinline__ float lane_xor_partner(float v, int m, int lane) {
    switch (m) {
        case 32: { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                   return __uint_as_float((lane & 32) ? r[0] : r[1]); }
        case 16: { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                   return __uint_as_float((lane & 16) ? r[0] : r[1]); }
        case 8:  return dpp_partner<0x128>(v);                                  // row_ror:8
        case 4:  { const float a = dpp_partner<0x104>(v), b = dpp_partner<0x114>(v);     // row_shl:4 (i <- i+4), row_shr:4 (i <- i-4)
                   return (lane & 4) ? b : a; }
        case 2:  return dpp_partner<0x4E>(v);
        default: return dpp_partner<0xB1>(v);
    }
}

// NT sums per lane, G lanes per row (G >= NT, both powers of two): the butterfly of group_reduce for every token, with
// the first log2(NT) steps transposing — at distance m the lanes with bit m clear keep the lower half of the tokens
// they still hold and hand the upper half to their partner (and vice versa), so each step halves the values per lane
// and every add is the butterfly's own pair (lane i with lane i ^ m). After those steps a lane holds ONE token's
// partial sum; the remaining steps are plain. Token t's total ends in the lane of its row group whose bits
// G/2, G/4, ... spell t (most significant first) and whose lower bits are 0 — reduce_token_of_lane / _is_leader —
// with the bits lane 0 of the group would hold under group_reduce. One epilogue per token then runs on NT
// different lanes at once instead of NT times on one.
template <int NT, int GC = 0 /* G when known at compile time, else 0 */>
__device__ __forceinline__ float transpose_reduce(float (&acc)[NT], int G_rt, int lane) {
    const int G = GC > 0 ? GC : G_rt;
    int m = G >> 1;
#pragma unroll
    for (int n = NT; n > 1; n >>= 1, m >>= 1) {
        if (m == 32) {                                   // the swap itself transposes: one swap and one add per token pair
#pragma unroll
            for (int t = 0; t < n / 2; ++t) acc[t] = swap32_add(acc[t], acc[t + n / 2]);
        } else if (m == 16) {
#pragma unroll
            for (int t = 0; t < n / 2; ++t) acc[t] = swap16_add(acc[t], acc[t + n / 2]);
        } else {
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int t = 0; t < n / 2; ++t) {
                const float keep = upper ? acc[t + n / 2] : acc[t];
                const float send = upper ? acc[t] : acc[t + n / 2];
                acc[t] = __fadd_rn(keep, lane_xor_partner(send, m, lane));
            }
        }
    }
    return group_reduce(acc[0], G / NT);                 // the remaining steps: only the subgroup's lowest lane is read
}
template <int NT>
__device__ __forceinline__ int reduce_token_of_lane(int G, int l) {            // l: lane within the row group
    int t = 0, m = G >> 1;
#pragma unroll
    for (int n = NT; n > 1; n >>= 1, m >>= 1) t = (t << 1) | ((l & m) ? 1 : 0);
    return t;
}
template <int NT>
__device__ __forceinline__ bool reduce_is_leader(int G, int l) { return (l & (G / NT - 1)) == 0; }

} // namespace sdh
