// sdh_direct.hip — batch-1 decode of a SeedDelta weight with a base: every workgroup forms the base term of ITS OWN rows.
//
// The fused launch of sdh_decode.hip gives the base transform to B workgroups (one CU per block: two FWHTs of L behind
// the block's loads) and hands every element over to the residual waves through memory; that chain (≈ 6 us) is what
// the launch takes, not the 13 MB payload stream (profiles/r03_d_decode_ablation.txt). Here nothing is handed over.
//
//   v = H_L D2 H_L z,   z[j] = x_b[p1[j]]                                  (src/llama-seeddelta.cpp:497-574, depth 2, no P2)
//   H_L = (H_{L/16} (x) I_16) (I (x) H_16), index i = 16 ih + il, so for the 16 rows p = 16 ph + r of one workgroup
//   v[16 ph + r] = sum_k (-1)^<r,k>  sum_jh D2h[ph ^ jh][k] * Zt[jh][k]
//   Zt[jh][.]  = H_16 (z[16 jh .. 16 jh + 15])                             one 16-point transform per 16 inputs, in registers
//   D2h[s][k]  = sum_t (-1)^<s,t> d2[16 t + k]                             a table built when the weight is created (the H_{L/16} part)
//
// i.e. L multiply-adds and L/16 small transforms per base block and workgroup instead of two transforms of L on one CU,
// and every workgroup reads x (57 KB on cfg3), the permutation (2 bytes per input) and D2h (4 bytes per input) from L2
// instead of waiting for another CU. The workgroup stages x in LDS once (zero-padded): the permutation gather and the
// residual's activation gathers both read it there. 16 rows per workgroup, 16 G threads (G lanes per row); the first half
// of the waves forms the base sums (two segments per thread on cfg3), the second half the rows' residual sums.
//
// What it costs and what it is bound by (MI355X, launch period in a stream of launches over 32 rotating payloads;
// tools/gpu/r03_direct_wl.sh, tools/direct_time.py, in-kernel timeline tools/direct_stamps.py,
// profiles/r03_q_direct_timeline_ablation.txt), direct / fused launch, us:
//   cfg3 (Mistral down, block 32, K = 1536)  7.04 / 7.84      Mistral gate COO K = 128   6.05 / 6.44      gate block16  5.51 / 6.12
//   Mistral down COO K = 128                  5.85 / 6.16      Mistral down block16       5.51 / 6.22      Gemma-3 4B gate / down  5.32 / 6.10, 5.33 / 6.06
// A CU takes in about 25 bytes a clock from L2 / HBM whatever the mix, and a cfg3 workgroup asks for 204 KB (x 57, p1b 32,
// D2h 64, payload 49 + indices): the last wave's x is stored ~2.5 us after the workgroup starts, the base sums are done
// 1-2 us later, the store follows 0.6 us after that; without the base (x + payload only) the launch is 5.3-5.6 us.
// Arithmetic is not the limit (the 16-point transforms, the gathers and the cross-lane sums: 0.6 us of the base role).
//
// Short rows and a long x (16 lanes per row would mean 256 threads for 16,384 staged inputs: 9.1 against 6.5 us, four waves
// fetching the workgroup's 200 KB) run with more lanes per row than the row needs — 16 threads per 256 staged inputs, the
// lanes past the row's end idle.
//
// Measured and left out: each workgroup walking x and the tables from its own offset (same time: the L2 is not the
// narrow place); tables requested together with x (+0.3 us: they stand ahead of the row waves' x in the CU's queue);
// payload requested ahead of x, or a fixed short time after it (+0.1-0.3 us); the base term picked up before the rows'
// sums (+0.15 us); one role for all waves (every wait then counts the other role's loads: 8.0 us); the gate | up pair of a
// gated FFN block in this form (16 rows of both tensors per workgroup: 16.95 against 16.22 us per layer with the fused pair
// launch — 80 KB of x and tables per 16 KB of payload and workgroup); 1024-thread workgroups of 64 rows for tall payloads
// (x, permutation and transforms shared by four row groups, 88 instead of 4 x 48 KB per 64 rows: Mistral gate 6.24
// against 6.05 us — four small workgroups per CU at different points of their chains overlap better than one large one);
// the same launch for 2-4 tokens in passes of two (every token bit-identical to its one-token result, but 128 VGPRs do
// not hold a role's tables or payload plus two tokens' staging: with as few as 12 spilled registers 17-44 us against
// 7-15 us for the fused multi-token launches).
//
// Arithmetic: fp32 throughout, but not the reference's butterfly order — the result is within the fast path's bound
// (measured <= 0.05 of it), not bit-identical to the FWHT forms; a token's result through this launch therefore differs in
// the low bits from the same token inside a 2-4 token batch. SDH_FWD_NO_DIRECT keeps the fused launch (whose base term has
// the FWHT form's bits); SDH_FWD_EXACT_ORDER never comes here.
#include "sdh_internal.h"
#include "sdh_base_dev.h"
#include "sdh_delta_dev.h"

#include <algorithm>

namespace sdh {

struct DirectArgs {
    const unsigned short * p1b;   // [B][L] byte offset (4 * p1[j]) inside the block of the input that permuted position j reads
    const float * d2h;            // [B][L / 16][16]
    const float * d3;             // [B][L] or null (all ones)
    int nseg_blk;                 // L / 16
    int nblk;                     // base blocks a row sums over: wide B, tall 1
    int xlen;                     // floats of x staged in LDS, zero past n_in: wide B * L, tall L
};

// D2h[b][s][k] = sum_t (-1)^popcount(s & t) d2[b][16 t + k], accumulated in double
__global__ __launch_bounds__(256) void direct_d2h_kernel(const float * __restrict__ d2, float * __restrict__ d2h, int L) {
    const int b = blockIdx.y, nseg = L >> 4;
    const int i = blockIdx.x * 256 + threadIdx.x;            // s * 16 + k
    if (i >= L) return;
    const int s = i >> 4, k = i & 15;
    const float * d = d2 + (size_t) b * L;
    double acc = 0.0;
    for (int t = 0; t < nseg; ++t) {
        const double v = (double) d[16 * t + k];
        acc += (__popc(s & t) & 1) ? -v : v;
    }
    d2h[(size_t) b * L + i] = (float) acc;
}
__global__ __launch_bounds__(256) void direct_p1b_kernel(const int * __restrict__ p1, unsigned short * __restrict__ p1b, int n, int L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = p1 ? p1[i] : (i & (L - 1));
    p1b[i] = (unsigned short) (4 * j);
}

// ---- the row's payload, loaded before x is staged, used after ------------------------------------------------------------
template <bool BLOCK, int NS>
struct RowRegs {
    Vec8<uint16_t> vv[NS];
    Vec8<uint16_t> ii[BLOCK ? 1 : NS];
    int  blk[NS];
    int  off[NS];
    bool ok[NS];
};

template <bool BLOCK, int NS>
__device__ __forceinline__ void direct_row_load(const DeltaParams & p, int orow, bool row_ok, int l, int G, RowRegs<BLOCK, NS> & r) {
    const uint16_t * vrow = reinterpret_cast<const uint16_t *>(p.val) + (size_t) orow * p.Ktot;
    const uint16_t * irow = reinterpret_cast<const uint16_t *>(p.idx) + (size_t) orow * (BLOCK ? p.nb : p.Ktot);
    int e0[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int e = (u * G + l) * 8;
        r.ok[u] = row_ok & (e < p.Ktot);
        e0[u] = r.ok[u] ? e : 0;
    }
    if constexpr (BLOCK) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            r.off[u] = e0[u] & (p.block - 1);
            r.blk[u] = idx_to_i32<uint16_t>(__builtin_nontemporal_load(irow + (e0[u] >> p.block_shift)));
        }
    } else {
#pragma unroll
        for (int u = 0; u < NS; ++u) r.ii[u].template load<true>(irow + e0[u]);
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) r.vv[u].template load<true>(vrow + e0[u]);
}

// the operations of delta_row_partial on the loaded row, activations from LDS
template <bool BLOCK, int NS>
__device__ __forceinline__ float direct_row_dot(const DeltaParams & p, const float * xs, RowRegs<BLOCK, NS> & r) {
    float acc = 0.0f;
    if constexpr (BLOCK) {
        float4 xa[NS], xb[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int raw = (r.blk[u] << p.block_shift) + r.off[u];
            r.ok[u] = r.ok[u] & ((unsigned) r.blk[u] < (unsigned) p.n_blk_in) & (raw < p.n_in);
            const int in0 = r.ok[u] ? raw : 0;
            xa[u] = *reinterpret_cast<const float4 *>(xs + in0);
            xb[u] = *reinterpret_cast<const float4 *>(xs + in0 + 4);
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) acc = add_if(acc, r.ok[u], dot8<uint16_t>(r.vv[u], xa[u], xb[u]));
    } else {
        float xv[NS][8];
        bool okj[NS][8];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = idx_to_i32<uint16_t>(r.ii[u].get(j));
                okj[u][j] = r.ok[u] & ((unsigned) i < (unsigned) p.n_in);
                xv[u][j] = xs[okj[u][j] ? i : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = add_if(acc, okj[u][j], __fmul_rn(val_to_f32<uint16_t>(r.vv[u].get(j)), xv[u][j]));
        }
    }
    return acc;
}

// 16-point Walsh-Hadamard transform in registers, two butterflies per packed add
__device__ __forceinline__ void wht16(float (&z)[16]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 uu = {z[r], z[r]};
        const f32x2 ww = {z[r + 1], -z[r + 1]};
        const f32x2 o = uu + ww;
        z[r] = o.x; z[r + 1] = o.y;
    }
#pragma unroll
    for (int rb = 2; rb < 16; rb <<= 1) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            if ((r & rb) == 0) {
                const f32x2 a = {z[r], z[r + 1]};
                const f32x2 b = {z[r | rb], z[(r | rb) + 1]};
                const f32x2 s = a + b, d = a - b;
                z[r] = s.x; z[r + 1] = s.y;
                z[r | rb] = d.x; z[(r | rb) + 1] = d.y;
            }
        }
    }
}

struct DirectSegTabs { uint4 pa, pb; float4 d0, d1, d2, d3; };

__device__ __forceinline__ void direct_tabs_load(DirectSegTabs & t, const DirectArgs & da, int L, int b0, int ph, int sg, int nseg) {
    const int sgc = sg < nseg ? sg : 0;                      // past the end: a valid address, the result is not used
    const int bb = sgc / da.nseg_blk, jh = sgc - bb * da.nseg_blk;
    const size_t blk = (size_t) (b0 + bb) * L;
    const uint4 * pp = reinterpret_cast<const uint4 *>(da.p1b + blk + (jh << 4));
    const float4 * dd = reinterpret_cast<const float4 *>(da.d2h + blk + ((ph ^ jh) << 4));
#ifdef SDH_DD_NOTABLOAD            // dev A/B: the segment's arithmetic without its table loads
    const unsigned a = (unsigned) (size_t) pp & 0x3FFCu, c = (unsigned) (size_t) dd;
    t.pa = make_uint4(a, a ^ 0x40, a ^ 0x80, a ^ 0xC0); t.pb = make_uint4(a ^ 0x100, a ^ 0x140, a ^ 0x180, a ^ 0x1C0);
    t.d0 = make_float4(__uint_as_float(c), 1.0f, 2.0f, 3.0f); t.d1 = t.d0; t.d2 = t.d0; t.d3 = t.d0;
#else
    t.pa = pp[0]; t.pb = pp[1];
    t.d0 = dd[0]; t.d1 = dd[1]; t.d2 = dd[2]; t.d3 = dd[3];
#endif
}

// Timeline stamps (dev build -DSDH_DD_STAMPS, tools/direct_stamps.py): lane 0 of the first base wave and of the first row
// wave keep the 100 MHz real-time counter in registers and store them when the role is over.
#ifdef SDH_DD_STAMPS
__device__ unsigned long long dd_stamps[4096 * 2 * 8];
#define DD_T(k) do { __builtin_amdgcn_sched_barrier(0); tstamp[k] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DD_DECL unsigned long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DD_FLUSH(role, cond) do { if (cond) { for (int q_ = 0; q_ < 8; ++q_) dd_stamps[(blockIdx.x * 2 + (role)) * 8 + q_] = tstamp[q_]; } } while (0)
#else
#define DD_T(k) do { } while (0)
#define DD_DECL do { } while (0)
#define DD_FLUSH(role, cond) do { } while (0)
#endif

// One segment of 16 permuted inputs: gather from the LDS copy of x (xb: the block's first byte), transform, multiply-add.
__device__ __forceinline__ void direct_segment(float (&acc)[16], const DirectSegTabs & t, const char * xb, bool valid) {
    const unsigned pw[8] = {t.pa.x, t.pa.y, t.pa.z, t.pa.w, t.pb.x, t.pb.y, t.pb.z, t.pb.w};
    float z[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#ifdef SDH_DD_NOGATHER             // dev A/B: no LDS gather (the offsets stand in for the values)
        z[2 * i]     = __uint_as_float(pw[i] & 0xFFFFu);
        z[2 * i + 1] = __uint_as_float(pw[i] >> 16);
#else
        z[2 * i]     = *reinterpret_cast<const float *>(xb + (pw[i] & 0xFFFFu));
        z[2 * i + 1] = *reinterpret_cast<const float *>(xb + (pw[i] >> 16));
#endif
    }
#ifndef SDH_DD_NOWHT               // dev A/B: no 16-point transforms
    wht16(z);
#endif
    const float dv[16] = {t.d0.x, t.d0.y, t.d0.z, t.d0.w, t.d1.x, t.d1.y, t.d1.z, t.d1.w,
                          t.d2.x, t.d2.y, t.d2.z, t.d2.w, t.d3.x, t.d3.y, t.d3.z, t.d3.w};
    if (valid) {                                             // a trip's second segment may lie past the end (uniform per wave from L = 1024 on)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(dv[k], z[k], acc[k]);
    }
}

// A base wave's 16 sums across its lanes, reduce-scatter: each step halves the values a lane holds and adds the partner's
// half. Lane bits 5 and 4 by half-register swaps (the swap is the exchange: one swap and one add per pair), bits 1 and 0 by
// DPP moves; the four lanes that differ in bits 2 and 3 keep separate partial sums (the row waves add them up).
// red: [waves][4][16].
__device__ __forceinline__ void direct_scatter_store(float (&acc)[16], float * red, int wave, int lane) {
#ifndef SDH_DD_NOSCATTER           // dev A/B: no cross-lane reduction
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = swap32_add(acc[i], acc[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = swap16_add(acc[i], acc[i + 4]);
    {
        const bool hi = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = hi ? acc[i + 2] : acc[i], send = hi ? acc[i] : acc[i + 2];
            acc[i] = keep + dpp_partner<0x4E>(send);                          // quad_perm [2,3,0,1]
        }
    }
    {
        const bool hi = (lane & 1) != 0;
        const float keep = hi ? acc[1] : acc[0], send = hi ? acc[0] : acc[1];
        acc[0] = keep + dpp_partner<0xB1>(send);                              // quad_perm [1,0,3,2]
    }
#endif
    const int k = (((lane >> 5) & 1) << 3) | (((lane >> 4) & 1) << 2) | (lane & 3);
    red[(wave * 4 + ((lane >> 2) & 3)) * 16 + k] = acc[0];
}

// A row wave: the 16 sums from the base waves' partials, then H_16 across the lanes; lane r (mod 16) returns v[16 ph + r].
// Cross-lane moves by half-register swaps and DPP only: no LDS round trip stands between the barrier and the store.
__device__ __forceinline__ float direct_final(const float * red, int nbw, int lane) {
    const int kk = lane & 15;
    float sv = 0.0f;
    float part[8];                                           // all reads in flight together (at most 8 base waves x 4 partials / 4 lane groups)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = (lane >> 4) + 4 * q;
        part[q] = red[(e < nbw * 4 ? e : 0) * 16 + kk];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) sv += ((lane >> 4) + 4 * q < nbw * 4) ? part[q] : 0.0f;
    sv = swap16_add(sv, sv);
    sv = swap32_add(sv, sv);                                 // every 16-lane row now holds the 16 sums
    float partner;
    partner = dpp_partner<0xB1>(sv);         sv = (lane & 1) ? partner - sv : partner + sv;
    partner = dpp_partner<0x4E>(sv);         sv = (lane & 2) ? partner - sv : partner + sv;
    partner = lane_xor_partner(sv, 4, lane); sv = (lane & 4) ? partner - sv : partner + sv;
    partner = lane_xor_partner(sv, 8, lane); sv = (lane & 8) ? partner - sv : partner + sv;
    return sv;
}
__device__ __forceinline__ float direct_pick(float sv, int r, int G) {      // v of row r (0..15) for the lanes of that row
    if (G == 64) return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sv), r));     // one row per slot: uniform
    return __shfl(sv, r, 64);
}

// x -> LDS, zero past n_in: NU x 16 bytes per thread (the host launches only when xlen <= 4 * NU * nthr), split in two so
// that a role can issue its other loads in between: stage_issue first, stage_store when x is needed. Branch-free (clamped
// address, select): the compiler's waits stay exact counts. n_in % 4 == 0 and x 16-byte aligned (host).
template <int NU>
__device__ __forceinline__ void stage_issue(const DeltaParams & p, int tid, int nthr, float4 (&xq)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = (tid + u * nthr) * 4;
#ifndef SDH_DD_NOX                 // dev A/B (tools/build_variants.sh; results wrong on purpose): LDS zero-filled, x not read
        xq[u] = *reinterpret_cast<const float4 *>(p.x + (i < p.n_in ? i : 0));
#else
        xq[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#endif
    }
}
template <int NU, bool FULL>                                 // FULL: xlen == 4 NU nthr — the stores are unconditional and the waits before them exact
__device__ __forceinline__ void stage_store(const DeltaParams & p, float * xs, int tid, int nthr, int xlen, const float4 (&xq)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = (tid + u * nthr) * 4;
        if (FULL || i < xlen) *reinterpret_cast<float4 *>(xs + i) = i < p.n_in ? xq[u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// Workgroup = 16 rows = 16 G threads (G lanes per row). The first half of the waves forms the base sums, the second half
// the rows' residual sums, two 64-lane row slots per wave: each role's loads are its own (a wave's loads return in
// order and its waits count them in order: a wave that had the payload stream ahead of x or the tables in its queue waited
// for HBM before it could start). Order of issue in both roles: x first.
template <bool BLOCK, int STEPS, int NU, bool FULL>
__global__ __launch_bounds__(1024)
void decode_direct_kernel(const DeltaParams p, const DirectArgs da) {
    extern __shared__ __align__(16) float xs[];              // [xlen] activations, then [8 waves][4][16] partial base sums
    float * const red = xs + da.xlen;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int nbw = nthr >> 7;                               // base waves; the others are row waves
    const int row0 = blockIdx.x * 16;
    const int L = p.L;
    const int b0 = p.tall ? row0 / L : 0;

    if (wave < nbw) {
        // ---- base role: acc[k] = sum over this thread's segments of D2h[ph ^ jh][k] * Zt[jh][k] ----------------------------
        const int ph = (row0 - b0 * L) >> 4;
        const int nseg = da.nblk * da.nseg_blk;
        const int bt = nbw * 64;
        DD_DECL;
        DD_T(0);
        float4 xq[NU];
        stage_issue<NU>(p, tid, nthr, xq);
        // The CU's vector-memory front end takes the waves' requests in order, 64 bytes a clock: tables requested here
        // (96 KB per workgroup) would stand ahead of the x requests of the row waves, which start a little later, and the
        // barrier below waits for those. So: x first, stored, and only then the tables (in flight across the barrier).
        stage_store<NU, FULL>(p, xs, tid, nthr, da.xlen, xq);
        __builtin_amdgcn_sched_barrier(0);
        DirectSegTabs tb0, tb1;
        direct_tabs_load(tb0, da, L, b0, ph, tid, nseg);
        direct_tabs_load(tb1, da, L, b0, ph, tid + bt, nseg);
        DD_T(1);
        __syncthreads();
        DD_T(2);
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
#ifndef SDH_DD_NOBASE              // dev A/B: no base sums
        for (int sg = tid; sg < nseg; sg += 2 * bt) {
            if (sg != tid) {                                 // later trips (narrow workgroups): loaded here, latency not hidden
                direct_tabs_load(tb0, da, L, b0, ph, sg, nseg);
                direct_tabs_load(tb1, da, L, b0, ph, sg + bt, nseg);
            }
            const bool two = sg + bt < nseg;
            const char * xc = reinterpret_cast<const char *>(xs);
            direct_segment(acc, tb0, xc + (p.tall ? 0 : (size_t) (sg / da.nseg_blk) * L * 4), true);
            direct_segment(acc, tb1, xc + (p.tall || !two ? 0 : (size_t) ((sg + bt) / da.nseg_blk) * L * 4), two);
        }
#endif
        DD_T(3);
        direct_scatter_store(acc, red, wave, lane);
        DD_T(4);
        __syncthreads();
        DD_T(5);
        DD_FLUSH(0, tid == 0);
        return;
    }

    // ---- row role: two 64-lane slots of rows per wave ------------------------------------------------------------------------
    const int G = p.G, l = lane & (G - 1);
    int rloc[2], o[2];
    bool row_ok[2], leader[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        rloc[u] = ((wave - nbw) * 2 + u) * (64 / G) + lane / G;
        o[u] = row0 + rloc[u];
        row_ok[u] = o[u] < p.n_out;
        leader[u] = row_ok[u] && l == 0;
    }
    DD_DECL;
    DD_T(0);
    float4 xq[NU];
    stage_issue<NU>(p, tid, nthr, xq);
    RowRegs<BLOCK, STEPS> rr[2];
    // The payload is requested once x is stored, like the base role's tables: a CU takes in ~25 bytes a clock from L2 /
    // HBM whatever the mix (x + tables + payload: 209 KB per workgroup on cfg3 — that, not arithmetic, is what the launch
    // takes), and the barrier below waits for the LAST wave's x. Requested ahead of x: 7.33 us per launch, behind: 7.08.
#ifdef SDH_DD_PAYLOAD_EARLY        // dev A/B
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) direct_row_load<BLOCK, STEPS>(p, row_ok[u] ? o[u] : 0, row_ok[u], l, G, rr[u]);
#endif
    stage_store<NU, FULL>(p, xs, tid, nthr, da.xlen, xq);          // nothing conditional has been issued so far: the waits for x are exact
#ifndef SDH_DD_PAYLOAD_EARLY
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#ifdef SDH_DD_NODELTA              // dev A/B: no payload stream
        direct_row_load<BLOCK, STEPS>(p, 0, row_ok[u], l, G, rr[u]);
#else
        direct_row_load<BLOCK, STEPS>(p, row_ok[u] ? o[u] : 0, row_ok[u], l, G, rr[u]);
#endif
    }
#endif
    DD_T(1);
    __syncthreads();
    DD_T(2);
    float scale[2] = {1.0f, 1.0f}, s3[2] = {1.0f, 1.0f};     // first used at the very end
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (leader[u] && p.rs) scale[u] = p.rs[o[u]];
        if (leader[u] && da.d3) s3[u] = da.d3[o[u]];         // tall: block b0, position o - b0 L; wide: no d3 here (host)
    }
    float dsum[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#ifdef SDH_DD_NODOT                // dev A/B: payload loaded, not multiplied
        dsum[u] = __uint_as_float(rr[u].vv[0].q.x ^ rr[u].vv[STEPS - 1].q.w);
#else
        dsum[u] = direct_row_dot<BLOCK, STEPS>(p, xs, rr[u]);
#endif
        dsum[u] = group_reduce(dsum[u], G);
    }
    DD_T(3);
    __syncthreads();
    DD_T(4);
    const float sv = direct_final(red, nbw, lane);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float vrow = direct_pick(sv, rloc[u] & 15, G);
        if (leader[u]) store_output(p, o[u], 0, vrow * s3[u] + dsum[u], scale[u]);
    }
    DD_T(5);
    DD_FLUSH(1, tid == nbw * 64);
}

// ---- host ---------------------------------------------------------------------------------------------------------------
// The static part of what launch_decode_direct asks of a payload.
static bool direct_weight_ok(const DeviceWeight & w) {
    if (w.kind != SDH_OP_BASE_BLOCK && w.kind != SDH_OP_BASE_COO) return false;
    if (w.idx_type != SDH_TYPE_I16 || w.val_type != SDH_TYPE_F16) return false;
    if (!w.w0_closed_form || !w.d2 || w.pack.off_d1 >= 0 || w.p2) return false;
    if (w.L < 256 || w.L > 16384 || (w.L & (w.L - 1)) != 0) return false;
    if ((w.n_in % 4) != 0 || (w.n_out % 16) != 0) return false;
    if (!w.tall && (w.n_out > w.L || w.pack.off_d3 >= 0)) return false;      // wide: the blocks' sums are formed before d3 could apply
    const int64_t xlen = w.tall ? w.L : w.B * w.L;
    return w.n_in <= xlen && (size_t) xlen * 4 + 8 * 64 * 4 <= 150 * 1024;
}

// Tables of the direct launch, built when the weight is created (sdh_weight_create) by two small kernels on `s`, into room
// that sdh_weight_create left behind the packed base parameters (D2h, then the 16-bit offsets): no allocation of their own,
// and none at the first forward — a forward that allocates cannot be recorded into a graph.
int direct_build_tables(DeviceWeight & w, hipStream_t s) {
    static const bool off = [] { const char * e = getenv("SDH_DECODE_DIRECT"); return e && atoi(e) == 0; }();
    if (off || w.dir_d2h || !w.dir_space || !direct_weight_ok(w)) return SDH_OK;
    const size_t n = (size_t) w.B * (size_t) w.L;
    float * d2h = reinterpret_cast<float *>(w.dir_space);
    unsigned short * p1b = reinterpret_cast<unsigned short *>(d2h + n);
    hipLaunchKernelGGL(direct_d2h_kernel, dim3((unsigned) ((w.L + 255) / 256), (unsigned) w.B), dim3(256), 0, s, w.d2, d2h, (int) w.L);
    hipLaunchKernelGGL(direct_p1b_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, w.p1, p1b, (int) n, (int) w.L);
    if (hipGetLastError() != hipSuccess) { set_error("direct decode: table kernels failed"); return SDH_ERR_HIP; }
    w.dir_d2h = d2h; w.dir_p1b = p1b;
    return SDH_OK;
}
void direct_free(DeviceWeight & w) { w.dir_d2h = nullptr; w.dir_p1b = nullptr; w.dir_space = nullptr; }   // the tables live inside pack_dev

template <bool BLOCK, int NU, bool FULL>
static int direct_launch_steps(int steps, const DeltaParams & p, const DirectArgs & da, unsigned grid, unsigned nthr, size_t lds, hipStream_t s) {
    static const bool attr_ok = [] {
        bool ok = true;
        auto set = [&](const void * k) { ok &= hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
        set((const void *) decode_direct_kernel<BLOCK, 1, NU, FULL>); set((const void *) decode_direct_kernel<BLOCK, 2, NU, FULL>);
        set((const void *) decode_direct_kernel<BLOCK, 3, NU, FULL>); set((const void *) decode_direct_kernel<BLOCK, 4, NU, FULL>);
        return ok;
    }();
    if (!attr_ok) return 1;
    switch (steps) {
        case 1: hipLaunchKernelGGL((decode_direct_kernel<BLOCK, 1, NU, FULL>), dim3(grid), dim3(nthr), lds, s, p, da); break;
        case 2: hipLaunchKernelGGL((decode_direct_kernel<BLOCK, 2, NU, FULL>), dim3(grid), dim3(nthr), lds, s, p, da); break;
        case 3: hipLaunchKernelGGL((decode_direct_kernel<BLOCK, 3, NU, FULL>), dim3(grid), dim3(nthr), lds, s, p, da); break;
        default: hipLaunchKernelGGL((decode_direct_kernel<BLOCK, 4, NU, FULL>), dim3(grid), dim3(nthr), lds, s, p, da); break;
    }
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

// Returns SDH_OK if launched, 1 if the payload / call does not qualify (caller takes the fused launch), <0 on error.
int launch_decode_direct(DeviceWeight & w, const DeltaLaunch & a, hipStream_t s) {
    static const bool off = [] { const char * e = getenv("SDH_DECODE_DIRECT"); return e && atoi(e) == 0; }();       // A/B switch
    if (off || a.T != 1 || a.exact_order) return 1;
    if (!w.dir_d2h) return 1;                                // no tables: the payload does not qualify (direct_build_tables)
    const bool is_block = (w.kind == SDH_OP_BASE_BLOCK);
    const int64_t xlen = w.tall ? w.L : w.B * w.L;
    const size_t lds = (size_t) xlen * 4 + 8 * 64 * 4;
    DeltaParams p = make_delta_params(w, a);
    if (!p.x_vec_ok) return 1;
    int G = 1, steps = 0;
    if (!delta_vector_shape(p, is_block, &G, &steps) || steps < 1 || steps > 4) return 1;
    p.G = G;
    DirectArgs da{};
    da.p1b = w.dir_p1b; da.d2h = w.dir_d2h;
    da.d3 = w.pack.off_d3 >= 0 ? w.d3 : nullptr;
    da.nseg_blk = (int) (w.L / 16);
    da.nblk = w.tall ? 1 : (int) w.B;
    da.xlen = (int) xlen;
    // short rows and a long x (wide payloads above all): more lanes per row than the row needs (lanes past its end idle), so
    // that the workgroup has 16 threads per 256 staged inputs
    static const bool widen = getenv("SDH_DIRECT_NO_WIDEN") == nullptr;                 // A/B switch
    if (widen && xlen <= 16384) {
        while (G < 64 && (256 * (int64_t) G < xlen || G < 8)) G <<= 1;
        steps = (p.Ktot + 8 * G - 1) / (8 * G);
        p.G = G;
    }
    if (G < 8) return 1;                                     // a workgroup is at least one base wave and one row wave
    const unsigned nthr = 16u * (unsigned) G;
    const unsigned grid = (unsigned) (w.n_out / 16);
    // x staged by one pass of at most 4 x 16 bytes per thread (wide payloads with short rows were widened to 64 lanes per row
    // above for this; what still does not fit — more than 16384 staged inputs — keeps the fused launch)
    if (xlen > 16 * (int64_t) nthr) return 1;
    if (xlen == 16 * (int64_t) nthr)
        return is_block ? direct_launch_steps<true, 4, true>(steps, p, da, grid, nthr, lds, s)
                        : direct_launch_steps<false, 4, true>(steps, p, da, grid, nthr, lds, s);
    return is_block ? direct_launch_steps<true, 4, false>(steps, p, da, grid, nthr, lds, s)
                    : direct_launch_steps<false, 4, false>(steps, p, da, grid, nthr, lds, s);
}

} // namespace sdh

#ifdef SDH_DD_STAMPS
extern "C" int sdh_debug_direct_stamps(unsigned long long * out, int n_wg) {
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(sdh::dd_stamps), (size_t) n_wg * 16 * sizeof(unsigned long long));
}
#endif
