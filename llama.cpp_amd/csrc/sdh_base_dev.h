// sdh_base_dev.h — device code of the SeedDelta base transform, shared by the stand-alone base
// kernel (sdh_base.hip) and the fused decode kernel (sdh_decode.hip).
//
//   v_b = D3 * H * [D2 * [P2] * H] * P1 * D1 * x_b      (reference: src/llama-seeddelta.cpp:497-574)
//
// One workgroup of 2^TB threads (256 by default) transforms one block of L = 2^TB * E elements
// (E = 2^LOG_E per thread). The FWHT runs entirely in registers: the index bits are processed LOG_E at a time, and
// between groups the block is transposed through LDS into the layout that makes the next LOG_E
// bits register-resident. The STAGE ORDER is the reference's (bit 0 first), every butterfly is
// (u+v, u-v) with u the element whose bit is clear, and multiplies / adds are never contracted,
// so the result is bit-identical to the CPU op.
//
// Layout j: the register index r supplies index bits [s_j, s_j + LOG_E), s_j = min(j*LOG_E, N-LOG_E);
// the thread id supplies the other N - LOG_E = TB bits (low part below s_j, high part above).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "sdh_internal.h"      // BasePackDesc

// Diagnostic stamps (tools/base_diag.hip builds with SDH_DIAG_STAMPS; the product build compiles
// them away). Thread 0 of each workgroup records the 100 MHz real-time counter.
#ifdef SDH_DIAG_STAMPS
__device__ unsigned long long * sdh_diag_buf;
#define SDH_STAMP(k)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (threadIdx.x == 0)                                                                     \
            sdh_diag_buf[(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#define SDH_DIAG_VAL(k, v)                                                                        \
    do { if (threadIdx.x == 0) sdh_diag_buf[(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (k)] = (v); } while (0)
#else
#define SDH_STAMP(k) do { } while (0)
#define SDH_DIAG_VAL(k, v) do { } while (0)
#endif

#ifndef SDH_CORE_STAMP
#define SDH_CORE_STAMP(k) do { } while (0)
#endif

namespace sdh {

template <int LOG_E_, int TB_ = 8>
struct BaseLay {
    static constexpr int LOG_E = LOG_E_;
    static constexpr int E  = 1 << LOG_E;
    static constexpr int TB = TB_;                   // thread-id bits (256 threads by default)
    static constexpr int N  = LOG_E + TB;
    static constexpr int L  = 1 << N;
    static constexpr int J  = (N + LOG_E - 1) / LOG_E;
    static constexpr int THREADS = 1 << TB;
    static __host__ __device__ constexpr int shift(int j) { return (j * LOG_E <= N - LOG_E) ? j * LOG_E : N - LOG_E; }
    static __device__ __forceinline__ int idx(int j, int tid, int r) {
        const int s = shift(j);
        return (tid & ((1 << s) - 1)) | (r << s) | ((tid >> s) << (s + LOG_E));
    }
    // LDS position of block element i. The 4-byte LDS accesses bank on (word % 32) per 32-lane half
    // (cdna_hip_programming.md §2); in the middle layouts the lanes of a half differ in a few low
    // index bits and a few bits above bit 5, which alias (E=16: 2-way, E=8 and E=4: 4-way). XOR-ing
    // those upper bits into free bank bits makes every middle layout's accesses conflict-free. Only
    // bits >= LOG_E change, so a thread's E consecutive elements stay consecutive (vector accesses).
    // Build variant only (-DSDH_BASE_SWIZZLE): measured within +-1.5 % of the plain layout on every
    // decode shape (the conflicted reads are ~10 % of the transform's LDS cycles, and the permutation
    // gathers pay three more VALU ops per element), so the plain layout stays the default.
#ifdef SDH_BASE_SWIZZLE
    static __device__ __forceinline__ int phys(int i) {
        if constexpr (TB == 8 && LOG_E == 4) return i ^ (((i >> 8) & 3) << 4);
        else if constexpr (TB == 8 && LOG_E == 3) return i ^ (((i >> 6) & 3) << 3);
        else if constexpr (TB == 8 && LOG_E == 2) return i ^ (((i >> 5) & 3) << 2) ^ (((i >> 6) & 1) << 4);
        else return i;
    }
#else
    static __device__ __forceinline__ int phys(int i) { return i; }
#endif
};

template <class Ly, int JJ>
__device__ __forceinline__ void base_stages(float (&y)[Ly::E]) {
#pragma clang fp contract(off)
    constexpr int E = Ly::E, LOG_E = Ly::LOG_E;
    constexpr int b0 = JJ * LOG_E;
    constexpr int b1 = (b0 + LOG_E < Ly::N) ? b0 + LOG_E : Ly::N;
    // Two butterflies per v_pk_add_f32: the same IEEE additions (u - w == u + (-w) exactly), half the VALU
    // instructions of the stages. The transform of one block runs on one CU and is VALU-issue bound there
    // (8 waves on 4 SIMDs); hipcc packs only about half of the scalar form by itself. cfg3: 8.06 -> 7.81 us.
    static_assert(E % 2 == 0, "E is a power of two >= 2");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int bit = b0; bit < b1; ++bit) {
        const int rb = 1 << (bit - Ly::shift(JJ));
        if (rb == 1) {
#pragma unroll
            for (int r = 0; r < E; r += 2) {
                const f32x2 uu = {y[r], y[r]};
                const f32x2 ww = {y[r + 1], -y[r + 1]};
                const f32x2 o = uu + ww;                 // (u + w, u + (-w)) == (u + w, u - w) exactly
                y[r] = o.x; y[r + 1] = o.y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < E; r += 2) {
                if ((r & rb) == 0) {
                    const f32x2 a = {y[r], y[r + 1]};
                    const f32x2 b = {y[r | rb], y[(r | rb) + 1]};
                    const f32x2 p = a + b, m = a - b;
                    y[r] = p.x; y[r + 1] = p.y;
                    y[r | rb] = m.x; y[(r | rb) + 1] = m.y;
                }
            }
        }
    }
}

template <class Ly, int JJ>
__device__ __forceinline__ void base_lds_write(float * lds, const float (&y)[Ly::E], int tid) {
    constexpr int E = Ly::E;
    if constexpr (Ly::shift(JJ) == 0 && E >= 4) {
#pragma unroll
        for (int k = 0; k < E; k += 4)
            *reinterpret_cast<float4 *>(lds + Ly::phys(E * tid + k)) = make_float4(y[k], y[k + 1], y[k + 2], y[k + 3]);
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) lds[Ly::phys(Ly::idx(JJ, tid, r))] = y[r];
    }
}

template <class Ly, int JJ>
__device__ __forceinline__ void base_lds_read(float (&y)[Ly::E], const float * lds, int tid) {
    constexpr int E = Ly::E;
    if constexpr (Ly::shift(JJ) == 0 && E >= 4) {
#pragma unroll
        for (int k = 0; k < E; k += 4) {
            const float4 q = *reinterpret_cast<const float4 *>(lds + Ly::phys(E * tid + k));
            y[k] = q.x; y[k + 1] = q.y; y[k + 2] = q.z; y[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = lds[Ly::phys(Ly::idx(JJ, tid, r))];
    }
}

// Full FWHT of the block held in layout 0; leaves it in layout J-1. The transposes alternate
// between the two LDS buffers, transition JJ using lds0 when (JJ + PAR) is odd, so that a buffer
// is never rewritten before a barrier has separated it from its last readers.
// With ONE_BUF (lds0 == lds1, half the LDS: the fused decode kernel needs the occupancy) every
// rewrite is preceded by a barrier of its own instead.
template <class Ly, int PAR, bool ONE_BUF, int JJ = 0>
__device__ __forceinline__ void base_fwht(float (&y)[Ly::E], float * lds0, float * lds1, int tid) {
    base_stages<Ly, JJ>(y);
    if constexpr (JJ + 1 < Ly::J) {
        float * buf = ((JJ + PAR) & 1) ? lds0 : lds1;
        if constexpr (ONE_BUF) __syncthreads();
        base_lds_write<Ly, JJ>(buf, y, tid);
        __syncthreads();
        base_lds_read<Ly, JJ + 1>(y, buf, tid);
        base_fwht<Ly, PAR, ONE_BUF, JJ + 1>(y, lds0, lds1, tid);
    }
}

// ---- FWHT with the lane bits done across lanes (no LDS transposes for them) ------------------------------------
// In layout 0 the index bits are [0, LOG_E) = register, [LOG_E, LOG_E + 6) = lane, the rest = wave. A butterfly
// stage on a lane bit k pairs lane with lane ^ (1 << k): the lane whose bit is clear takes own + partner, the one
// whose bit is set takes partner - own = partner + (-own): ONE add per element after a cross-lane move and a sign
// flip, the same IEEE additions as the butterfly (u + v, u - v), so the result stays bit-identical and the STAGE
// ORDER (bit 0 first) is kept. Moves: DPP quad_perm for xor 1 / 2, two DPP mirrors for xor 4 / 8 (row_half_mirror is
// xor 7 within 8 lanes, row_mirror xor 15 within 16: composed with the xor 3 / xor 7 permutation they give xor 4 /
// xor 8), ds_swizzle for xor 16, ds_bpermute for xor 32. Only the wave bits need LDS: one transpose from layout 0
// straight to the last layout, whose register bits [N - LOG_E, N) cover them. One barrier per transform instead of
// six, 2 E LDS accesses per thread instead of 6 E, none of them bank-conflicted. Measured: DESIGN.md §3.
template <int K>
__device__ __forceinline__ float base_lane_xor(float v) {
    const int i = __float_as_int(v);
    if constexpr (K == 0) return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false));          // quad_perm [1,0,3,2]
    else if constexpr (K == 1) return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
#ifdef SDH_BASE_XOR48_DPP
    else if constexpr (K == 2) {
        const int t = __builtin_amdgcn_update_dpp(i, i, 0x141, 0xF, 0xF, false);                                    // row_half_mirror: xor 7
        return __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0x1B, 0xF, 0xF, false));                            // quad_perm [3,2,1,0]: xor 3
    } else if constexpr (K == 3) {
        const int t = __builtin_amdgcn_update_dpp(i, i, 0x140, 0xF, 0xF, false);                                    // row_mirror: xor 15
        return __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, 0xF, false));                           // row_half_mirror: xor 7
    }
#else
    // xor 4 / 8 / 16 through the LDS crossbar (ds_swizzle, bit mode: and 0x1F, xor mask << 10): the transform is
    // VALU-issue bound (two waves per SIMD), two DPP moves per element cost more than one DS instruction
    else if constexpr (K == 2) return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x101F));
    else if constexpr (K == 3) return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x201F));
#endif
    else if constexpr (K == 4) return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x401F));
    else return __shfl_xor(v, 32, 64);
}

// One stage on lane bit K: y = partner + sgn * y with sgn = +1 (bit clear) / -1 (bit set): partner - y is the fused
// multiply-add partner + (-1 * y), exact product, ONE rounding — the butterfly's own addition. Two elements per
// v_pk_fma_f32.
template <class Ly, int K>
__device__ __forceinline__ void base_lane_stage(float (&y)[Ly::E], int lane) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float sg = ((lane >> K) & 1) ? -1.0f : 1.0f;
    const f32x2 sgn = {sg, sg};
    float partner[Ly::E];
#pragma unroll
    for (int r = 0; r < Ly::E; ++r) partner[r] = base_lane_xor<K>(y[r]);
#pragma unroll
    for (int r = 0; r < Ly::E; r += 2) {
        const f32x2 a = {y[r], y[r + 1]};
        const f32x2 p = {partner[r], partner[r + 1]};
        const f32x2 o = __builtin_elementwise_fma(a, sgn, p);
        y[r] = o.x; y[r + 1] = o.y;
    }
}

// Block in layout 0 -> transformed block in layout J-1. `buf`: L floats that no thread is still reading (or
// PRE_BARRIER: a barrier separates their last readers from the write).
template <class Ly, bool PRE_BARRIER>
__device__ __forceinline__ void base_fwht_x(float (&y)[Ly::E], float * buf, int tid) {
#pragma clang fp contract(off)
    constexpr int E = Ly::E, LOG_E = Ly::LOG_E, N = Ly::N, JL = Ly::J - 1;
    static_assert(Ly::TB >= 6 && Ly::TB - 6 <= LOG_E && N > LOG_E, "wave bits must fit the last layout's register bits");
    base_stages<Ly, 0>(y);                                   // bits [0, LOG_E)
    const int lane = tid & 63;
    base_lane_stage<Ly, 0>(y, lane);                         // bits LOG_E .. LOG_E + 5
    base_lane_stage<Ly, 1>(y, lane);
    base_lane_stage<Ly, 2>(y, lane);
    base_lane_stage<Ly, 3>(y, lane);
    base_lane_stage<Ly, 4>(y, lane);
    base_lane_stage<Ly, 5>(y, lane);
    if constexpr (PRE_BARRIER) __syncthreads();
    base_lds_write<Ly, 0>(buf, y, tid);
    __syncthreads();
    base_lds_read<Ly, JL>(y, buf, tid);
    if constexpr (Ly::TB > 6) {                              // wave bits [LOG_E + 6, N) = register bits of the last layout
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int bit = LOG_E + 6; bit < N; ++bit) {
            const int rb = 1 << (bit - (N - LOG_E));
            if (rb == 1) {
#pragma unroll
                for (int r = 0; r < E; r += 2) {
                    const f32x2 uu = {y[r], y[r]};
                    const f32x2 ww = {y[r + 1], -y[r + 1]};
                    const f32x2 o = uu + ww;
                    y[r] = o.x; y[r + 1] = o.y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; r += 2) {
                    if ((r & rb) == 0) {
                        const f32x2 a = {y[r], y[r + 1]};
                        const f32x2 b = {y[r | rb], y[(r | rb) + 1]};
                        const f32x2 pp = a + b, mm = a - b;
                        y[r] = pp.x; y[r + 1] = pp.y;
                        y[r | rb] = mm.x; y[(r | rb) + 1] = mm.y;
                    }
                }
            }
        }
    }
}

template <bool P32>
__device__ __forceinline__ float base_param_f(const uint8_t * p, int i) {
    if constexpr (P32) return reinterpret_cast<const float *>(p)[i];
    else return __half2float(__ushort_as_half(reinterpret_cast<const uint16_t *>(p)[i]));
}
template <bool P32>
__device__ __forceinline__ int base_param_i(const uint8_t * p, int i) {
    if constexpr (P32) return reinterpret_cast<const int *>(p)[i];
    else return (int) reinterpret_cast<const uint16_t *>(p)[i];      // perms are clamped to [0,L) at upload
}

// Transform one block. xt: the token's activation column; in0: first input element of the block
// (0 for tall weights); the block reads x[in0 .. in0+L) zero-padded past n_in. lds: 2*L floats
// (ONE_BUF: L floats, at the price of one more barrier per transpose).
// On return y[r] holds output element BaseLay::idx(J-1, tid, r) (the last layout: consecutive
// threads hold consecutive elements, so a plain store per register is coalesced).
template <class Ly, bool P32, bool ONE_BUF>
__device__ __forceinline__ void base_transform(const float * __restrict__ xt, int n_in, int in0, bool x_vec_ok,
                                               const BasePackDesc & pd, int b, float * lds, int tid,
                                               float (&y)[Ly::E]) {
#pragma clang fp contract(off)
    constexpr int E = Ly::E, L = Ly::L, JL = Ly::J - 1;
    float * lds0 = lds;
    float * lds1 = ONE_BUF ? lds : lds + L;
    const uint8_t * pk = pd.pack + (size_t) b * pd.blk_stride;
    const bool has_p1 = pd.off_p1 >= 0, has_d2 = pd.off_d2 >= 0, has_d1 = pd.off_d1 >= 0,
               has_d3 = pd.off_d3 >= 0, has_p2 = pd.off_p2 >= 0;

    // ---- every global load is issued before the first use -------------------------------------
    float s2[E], s3[E];
    int g1[E];
    const int src0 = in0 + E * tid;
    if (x_vec_ok && src0 + E <= n_in) {
#pragma unroll
        for (int k = 0; k < E; k += 4) {
            const float4 q = *reinterpret_cast<const float4 *>(xt + src0 + k);
            y[k] = q.x; y[k + 1] = q.y; y[k + 2] = q.z; y[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = (src0 + r < n_in) ? xt[src0 + r] : 0.0f;
    }
    if (has_p1) {
#pragma unroll
        for (int r = 0; r < E; ++r) g1[r] = base_param_i<P32>(pk + pd.off_p1, E * tid + r);
    }
    if (has_d2) {
#pragma unroll
        for (int r = 0; r < E; ++r) s2[r] = base_param_f<P32>(pk + pd.off_d2, Ly::idx(JL, tid, r));
    }
    if (has_d3) {
#pragma unroll
        for (int r = 0; r < E; ++r) s3[r] = base_param_f<P32>(pk + pd.off_d3, Ly::idx(JL, tid, r));
    }
    if (has_d1) {                                            // elided when all ones (x * 1.0f == x exactly)
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = y[r] * base_param_f<P32>(pk + pd.off_d1, E * tid + r);
    }
    SDH_STAMP(1);
    // ---- P1 (gather) ----------------------------------------------------------------------------
    if (has_p1) {
        base_lds_write<Ly, 0>(lds0, y, tid);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = lds0[Ly::phys(g1[r])];
    }
    SDH_STAMP(2);
    // ---- H ----------------------------------------------------------------------------------------
#ifdef SDH_BASE_FWHT_LDS
    base_fwht<Ly, 0, ONE_BUF>(y, lds0, lds1, tid);        // y now in layout JL
#else
    base_fwht_x<Ly, ONE_BUF>(y, lds1, tid);               // lds1: not the buffer the P1 gather reads (ONE_BUF: barrier first)
#endif
    SDH_STAMP(3);
    if (!has_d2 && !has_p2) {                                // depth 1 (:537-544)
        if (has_d3) {
#pragma unroll
            for (int r = 0; r < E; ++r) y[r] = y[r] * s3[r];
        }
        return;
    }
    if (has_d2) {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = y[r] * s2[r];
    }
    // ---- back to layout 0, through the optional second permutation ------------------------------
#ifdef SDH_BASE_FWHT_LDS
    // The first FWHT's last transpose used lds0 when J is odd, lds1 when J is even; the other
    // buffer's readers are already behind a barrier, so it takes the way back. The second FWHT
    // then starts on the buffer that is not being read here (PAR = J & 1).
    float * back = (Ly::J & 1) ? lds1 : lds0;
#else
    float * back = lds0;                                     // its last readers (P1 gather) are behind the FWHT's barrier
#endif
    if constexpr (ONE_BUF) __syncthreads();
    base_lds_write<Ly, JL>(back, y, tid);
    __syncthreads();
    if (has_p2) {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = back[Ly::phys(base_param_i<P32>(pk + pd.off_p2, E * tid + r))];
    } else {
        base_lds_read<Ly, 0>(y, back, tid);
    }
    SDH_STAMP(4);
#ifdef SDH_BASE_FWHT_LDS
    base_fwht<Ly, (Ly::J & 1), ONE_BUF>(y, lds0, lds1, tid);
#else
    base_fwht_x<Ly, ONE_BUF>(y, lds1, tid);
#endif
    SDH_STAMP(5);
    if (has_d3) {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = y[r] * s3[r];
    }
}

// ---- the same transform with its inputs already in registers (persistent FFN stack, sdh_stack.hip) -------
// There a base workgroup runs one transform per layer: the block's parameters are requested BEFORE the
// activations exist (they are handed over by other workgroups), then the activations arrive in registers.
// BasePrm holds what base_transform loads up front; base_transform_core is the rest of base_transform,
// operation for operation (so the result is bit-identical to it).
template <class Ly>
struct BasePrm {
    float s1[Ly::E], s2[Ly::E], s3[Ly::E];
    int   g1[Ly::E], g2[Ly::E];
};

template <class Ly, bool P32>
__device__ __forceinline__ void base_prm_load(BasePrm<Ly> & q, const BasePackDesc & pd, int b, int tid) {
    constexpr int E = Ly::E, JL = Ly::J - 1;
    const uint8_t * pk = pd.pack + (size_t) b * pd.blk_stride;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        q.g1[r] = pd.off_p1 >= 0 ? base_param_i<P32>(pk + pd.off_p1, E * tid + r) : 0;
        q.g2[r] = pd.off_p2 >= 0 ? base_param_i<P32>(pk + pd.off_p2, E * tid + r) : 0;
        q.s1[r] = pd.off_d1 >= 0 ? base_param_f<P32>(pk + pd.off_d1, E * tid + r) : 1.0f;
        q.s2[r] = pd.off_d2 >= 0 ? base_param_f<P32>(pk + pd.off_d2, Ly::idx(JL, tid, r)) : 1.0f;
        q.s3[r] = pd.off_d3 >= 0 ? base_param_f<P32>(pk + pd.off_d3, Ly::idx(JL, tid, r)) : 1.0f;
    }
}

// y: the block's input in layout 0 (thread tid holds elements [E*tid, E*tid + E), zero past the end of x).
// On return y[r] holds output element idx(J-1, tid, r). lds: L floats (ONE_BUF) or 2 L floats: with two buffers
// the transposes alternate between them and need one barrier each instead of two.
template <class Ly, bool ONE_BUF>
__device__ __forceinline__ void base_transform_core(float (&y)[Ly::E], const BasePrm<Ly> & q, const BasePackDesc & pd,
                                                    float * lds, int tid) {
#pragma clang fp contract(off)
    constexpr int E = Ly::E, L = Ly::L, JL = Ly::J - 1;
    float * lds0 = lds;
    float * lds1 = ONE_BUF ? lds : lds + L;
    const bool has_p1 = pd.off_p1 >= 0, has_d2 = pd.off_d2 >= 0, has_d1 = pd.off_d1 >= 0,
               has_d3 = pd.off_d3 >= 0, has_p2 = pd.off_p2 >= 0;
    if (has_d1) {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = y[r] * q.s1[r];
    }
    if (has_p1) {
        base_lds_write<Ly, 0>(lds0, y, tid);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = lds0[Ly::phys(q.g1[r])];
    }
    SDH_CORE_STAMP(4);
#ifdef SDH_BASE_FWHT_LDS
    base_fwht<Ly, 0, ONE_BUF>(y, lds0, lds1, tid);
#else
    base_fwht_x<Ly, ONE_BUF>(y, lds1, tid);
#endif
    SDH_CORE_STAMP(5);
    if (!has_d2 && !has_p2) {                                // depth 1 (:537-544)
        if (has_d3) {
#pragma unroll
            for (int r = 0; r < E; ++r) y[r] = y[r] * q.s3[r];
        }
        return;
    }
    if (has_d2) {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = y[r] * q.s2[r];
    }
#ifdef SDH_BASE_FWHT_LDS
    float * back = (Ly::J & 1) ? lds1 : lds0;                // as base_transform: the buffer whose readers are behind a barrier
#else
    float * back = lds0;
#endif
    if constexpr (ONE_BUF) __syncthreads();
    base_lds_write<Ly, JL>(back, y, tid);
    __syncthreads();
    if (has_p2) {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = back[Ly::phys(q.g2[r])];
    } else {
        base_lds_read<Ly, 0>(y, back, tid);
    }
    SDH_CORE_STAMP(6);
#ifdef SDH_BASE_FWHT_LDS
    base_fwht<Ly, (Ly::J & 1), ONE_BUF>(y, lds0, lds1, tid);
#else
    base_fwht_x<Ly, ONE_BUF>(y, lds1, tid);
#endif
    if (has_d3) {
#pragma unroll
        for (int r = 0; r < E; ++r) y[r] = y[r] * q.s3[r];
    }
}

// Re-order a finished block from the last layout to layout 0 (E consecutive elements per thread,
// thread t holding [E*t, E*t+E)) through a single LDS buffer that may still be in use: a barrier
// first, then write, barrier, read.
template <class Ly>
__device__ __forceinline__ void base_to_layout0(float (&y)[Ly::E], float * lds, int tid) {
    __syncthreads();
    base_lds_write<Ly, Ly::J - 1>(lds, y, tid);
    __syncthreads();
    base_lds_read<Ly, 0>(y, lds, tid);
}

} // namespace sdh
