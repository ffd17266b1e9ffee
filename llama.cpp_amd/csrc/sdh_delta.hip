// sdh_delta.hip — the sparse residual Δ·x, fused with the base term (read from the scratch the
// base kernel wrote), row_scale and the non-finite accounting: stand-alone kernels used for
// prefill (T > 1), for payloads without a base, and as the generic path. Batch-1 decode with a
// base goes through the fused kernel of sdh_decode.hip.
//
//   y[o,t] = rs[o] * ( base[o,t] + sum_r val[r,o] * x[idx[r,o], t] )                    (COO)
//   y[o,t] = rs[o] * ( base[o,t] + sum_bi sum_j bval[j,bi,o] * x[bidx[bi,o]*block+j,t]) (block)
//
// Reference semantics (src/llama-seeddelta.cpp): the base term is v_b[o-bL] for tall weights
// (:1106) and sum_b v_b[o], accumulated from 0.0f in block order, for wide ones
// (:1132,:1148-1150,:1154); the scale is applied only if it differs from 1 (:645); NaN/Inf outputs
// are counted (:382).
//
//   delta_rows_kernel        — fast path. Each output row is owned by a group of G lanes of one
//                              wave64 (G = 1..64, power of two); every lane streams 8 consecutive
//                              payload elements per step with 16-byte loads (the payload is
//                              row-contiguous in the GGUF layout), gathers x through L1/L2 and the
//                              group reduces with cross-lane shuffles. fp32 accumulation, order
//                              differs from the CPU op (tolerance-checked).
//   delta_rows_scalar_kernel — same ownership, element-wise loads, for rows that are not 16-byte
//                              multiples.
//   delta_rows_{coo,block}_lds_kernel
//                            — several tokens, the same arithmetic per token: x of up to 8 tokens in LDS,
//                              rows walked pass by pass with the next pass's operands requested ahead,
//                              transposing reduction (one epilogue lane per token). Block payloads with
//                              rows of more than 256 entries from 10 tokens, otherwise from 5 tokens
//                              when the tiled SpMM is not asked for (gated FFN node launches,
//                              SDH_FWD_NO_SPMM). The *_pair_kernel forms do gate and up of a gated FFN
//                              from one tile of x and write act(gate) * up.
//   delta_exact_kernel       — verification mode: one lane per output, the reference's sequential
//                              order with separately rounded multiply and add; bit-identical.
#include "sdh_internal.h"
#include <mutex>
#include <unordered_map>
#include <algorithm>
#include <type_traits>
#include "sdh_delta_dev.h"

namespace sdh {

// base term of output (o, t) read from the base kernel's scratch
__device__ __forceinline__ float base_term(const DeltaParams & p, int o, int64_t t) {
    if (!p.v) return 0.0f;
    const float * vt = p.v + (size_t) t * (size_t) p.v_ld;
    if (p.tall) return vt[o];                       // b*L + (o - b*L) == o
    float yh = 0.0f;
    for (int b = 0; b < p.B; ++b) yh += vt[(size_t) b * p.L + o];
    return yh;
}

template <typename IDX_T, typename VAL_T, bool BLOCK, int STEPS>
__global__ __launch_bounds__(256)
void delta_rows_kernel(const DeltaParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = p.G;
    const int l = lane & (G - 1);
    const int o = ((int) blockIdx.y * 4 + wave) * (64 / G) + (lane / G);
    const int64_t t = blockIdx.x;
    const bool row_ok = o < p.n_out;
    const int orow = row_ok ? o : 0;
    const float * xt = p.x + t * p.ldx;

    // epilogue operands, requested first so they are home long before the reduction ends
    float scale = 1.0f, vb[8];
    const bool leader = row_ok && l == 0;
    const int nbase = p.v ? (p.tall ? 1 : p.B) : 0;
    if (leader) {
        if (p.rs) scale = p.rs[o];
        const float * vt = p.v + (size_t) t * (size_t) p.v_ld;
#pragma unroll
        for (int b = 0; b < 8; ++b) vb[b] = (b < nbase) ? vt[(size_t) b * p.L + o] : 0.0f;
    }

    float acc = delta_row_partial<IDX_T, VAL_T, BLOCK, STEPS>(p, xt, orow, row_ok, l, G);
    acc = group_reduce(acc, G);
    if (leader) {
        float yh = 0.0f;
        if (p.v) {
            if (p.tall) yh = vb[0];
            else {
#pragma unroll
                for (int b = 0; b < 8; ++b) if (b < nbase) yh += vb[b];
                for (int b = 8; b < nbase; ++b) yh += p.v[(size_t) t * (size_t) p.v_ld + (size_t) b * p.L + o];
            }
        }
        store_output(p, o, t, yh + acc, scale);
    }
}

// Many tokens of a block payload with long rows and long inputs (cfg3: block 32, K = 1536, n_in = 14336, T = 512),
// where the SpMM form's lane-per-row walk re-fetches every payload line eight times and loses to the per-token
// kernel. The per-token kernel in turn gathers 6 KB of activations per (row, token) through the L2 — 12.9 GB per
// ubatch — and lives for one row: a wave's set-up, its L2 round trips and its reduction are all exposed.
// Here the activations of NT tokens live in the LDS and a wave walks rows with all 64 lanes on a row (Ktot > 256),
// coalesced 16-byte payload loads as in the per-token kernel: the L2 sees the payload once per NT tokens and the
// gather becomes two 16-byte LDS reads per step and token. The tile is stored as two planes per token, plane h
// holding floats 8c + 4h .. 8c + 4h + 3 of every 8-float chunk c, so that the lanes of a block read consecutive
// 16-byte words (no bank conflicts inside a block). With the tile filling the LDS a CU holds sixteen waves, too few
// to stand behind an L2 round trip per row, so everything a row needs from memory (index and value words, its scale,
// up to four base terms) is requested one row ahead.
// Per token the products and sums are delta_row_partial's (dot8, add_if in step order) and the lanes are summed in
// group_reduce's pairing (lane i with lane i ^ m, m = 32 .. 1), so results are bit-identical to the one-token kernel;
// only the data movement of the reduction differs: the m = 32 (and, for four tokens, m = 16) exchange swaps half
// registers between the tokens' sums, which leaves token t's sum in a lane segment of its own, and the steps inside
// a 16-lane row are DPP moves.
template <typename IDX_T, typename VAL_T, int STEPS, int NT, int GC /* 64: rows of 64 lanes, else 0: p.G */>
__global__ __launch_bounds__(1024)
void delta_rows_block_lds_kernel(const DeltaParams p, const int64_t T, const int rows_per_wg) {
    extern __shared__ float4 xs4[];                     // [NT][2][n_in / 8]
    static_assert(NT == 2 || NT == 4 || NT == 8, "tokens per tile");
    constexpr int NS = STEPS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = GC > 0 ? GC : p.G;                    // host: G >= 8, G >= NT
    const int l = lane & (G - 1);
    const int rows_wave = 64 / G;
    const int nch = p.n_in >> 3;
    const int64_t t0 = (int64_t) blockIdx.x * NT;
    const int nt = (int) ((T - t0 < NT) ? (T - t0) : NT);
    const int r0 = (int) blockIdx.y * rows_per_wg;
    const int r1 = min(p.n_out, r0 + rows_per_wg);
    int e0[NS], bi[NS], off[NS];
    bool eok[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int e = (u * G + l) * 8;
        eok[u] = e < p.Ktot;
        e0[u] = eok[u] ? e : 0;
        bi[u] = e0[u] >> p.block_shift;
        off[u] = e0[u] & (p.block - 1);
    }
    const VAL_T * val = reinterpret_cast<const VAL_T *>(p.val);
    const IDX_T * idx = reinterpret_cast<const IDX_T *>(p.idx);
    // the token whose sum the transposing reduction leaves in this lane (transpose_reduce): its epilogue runs here
    const int my_t = reduce_token_of_lane<NT>(G, l);
    const bool lead = reduce_is_leader<NT>(G, l) && my_t < nt;
    constexpr int NB = 4;
    const int nbase = p.v ? (p.tall ? 1 : p.B) : 0;
    // a row's operands from memory; two sets used in turn (the loop is unrolled by two passes), so that the set a
    // pass computes on is not copied out of the set the next pass's loads land in
    struct Pre { IDX_T blk[NS]; Vec8<VAL_T> vv[NS]; float vb[NB]; float scale; };
    Pre pa, pb;
    pa.scale = pb.scale = 1.0f;
    const unsigned vt_off = (unsigned) ((t0 + (my_t < nt ? my_t : 0)) * p.v_ld);       // host: the buffers here are < 4 GiB
    auto at = [](auto * base, unsigned i) {            // uniform base + 32-bit offset: no 64-bit vector address arithmetic
        using T = std::remove_pointer_t<decltype(base)>;
        return reinterpret_cast<T *>(reinterpret_cast<const char *>(base) + (size_t) (i * (unsigned) sizeof(T)));
    };
    auto request = [&](Pre & d, int on) {
        const unsigned orow = (unsigned) (on < r1 ? on : r0);
#pragma unroll
        for (int u = 0; u < NS; ++u) d.blk[u] = *at(idx, orow * (unsigned) p.nb + (unsigned) bi[u]);
#pragma unroll
        for (int u = 0; u < NS; ++u) d.vv[u].load(at(val, orow * (unsigned) p.Ktot + (unsigned) e0[u]));
        if (lead) {
            if (p.rs) d.scale = *at(p.rs, orow);
#pragma unroll
            for (int b = 0; b < NB; ++b) d.vb[b] = (b < nbase) ? *at(p.v, vt_off + (unsigned) b * (unsigned) p.L + orow) : 0.0f;
        }
    };
    const int stride = 16 * rows_wave;                  // rows per pass of the workgroup
    int o = r0 + wave * rows_wave + lane / G;
    request(pa, o);                                      // the first rows' payload travels while the tile is filled
    for (int c = tid; c < nch; c += 1024) {              // all tokens of a chunk per trip: 2 NT 16-byte loads in flight
        float4 va[NT], vb2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float * xt = p.x + (t0 + (t < nt ? t : nt - 1)) * p.ldx;
            va[t] = *reinterpret_cast<const float4 *>(xt + c * 8);
            vb2[t] = *reinterpret_cast<const float4 *>(xt + c * 8 + 4);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            xs4[(t * 2 + 0) * nch + c] = va[t];
            xs4[(t * 2 + 1) * nch + c] = vb2[t];
        }
    }
    __syncthreads();
    auto pass = [&](const Pre & cur, Pre & nxt, int o) {
        const bool row_ok = o < r1;
        int c0[NS];
        bool ok[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int blk = idx_to_i32<IDX_T>(cur.blk[u]);
            const int raw = (blk << p.block_shift) + off[u];
            ok[u] = row_ok & eok[u] & ((unsigned) blk < (unsigned) p.n_blk_in) & (raw < p.n_in);
            c0[u] = (ok[u] ? raw : 0) >> 3;
        }
        request(nxt, o + stride);
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float4 xa[NS], xb[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                xa[u] = xs4[(t * 2 + 0) * nch + c0[u]];
                xb[u] = xs4[(t * 2 + 1) * nch + c0[u]];
            }
            acc[t] = 0.0f;
#pragma unroll
            for (int u = 0; u < NS; ++u) acc[t] = add_if(acc[t], ok[u], dot8<VAL_T>(cur.vv[u], xa[u], xb[u]));
            if (NT > 2 && (t & 1) && t + 1 < NT) __builtin_amdgcn_sched_barrier(0);    // two tokens' reads in flight
        }
        const float a = transpose_reduce<NT, GC>(acc, G, lane);
        if (row_ok && lead) {
            float yh = 0.0f;
            if (p.v) {
                if (p.tall) yh = cur.vb[0];
                else {
#pragma unroll
                    for (int b = 0; b < NB; ++b) if (b < nbase) yh += cur.vb[b];
                    for (int b = NB; b < nbase; ++b) yh += *at(p.v, vt_off + (unsigned) b * (unsigned) p.L + (unsigned) o);
                }
            }
            store_output(p, o, t0 + my_t, yh + a, cur.scale);
        }
    };
    const int passes = (r1 - r0 + stride - 1) / stride;  // uniform over the workgroup
    for (int it = 0; it < passes; it += 2) {
        pass(pa, pb, o);
        o += stride;
        if (it + 1 < passes) { pass(pb, pa, o); o += stride; }
    }
}

// COO rows for several tokens with x in LDS (the fast-path arithmetic, not the tiled SpMM's exact order): the gated
// FFN's node launches from 5 tokens up. The SpMM form gives a lane a whole row: with few tokens that is one token tile,
// a few hundred waves, each walking 128 entries behind one 32-byte payload request at a time — far too little memory
// parallelism to stream the payload (Mistral gate, 8 tokens: 15 us for 5.5 MB). Here a row has G lanes as in the
// one-token kernel (coalesced 16-byte index / value loads, the whole row requested at once and the next pass's rows
// a pass ahead), x of NT tokens sits token-interleaved in the de-interleaved tile rows of sdh_spmm.hip (one 16-byte
// LDS read per entry and four tokens), and the NT sums of a row are reduced by the transposing butterfly
// (transpose_reduce): token t's total lands in its own lane, so the NT epilogues of a row run side by side.
// Per token: the products, the order of the adds and the pairing of the lane sums are delta_rows_kernel's.
template <typename IDX_T, typename VAL_T, int STEPS, int NT>
__global__ __launch_bounds__(1024)
void delta_rows_coo_lds_kernel(const DeltaParams p, const int64_t T, const int rows_per_wg) {
#pragma clang fp contract(off)                           // the products and the adds round separately, as in delta_row_partial
    extern __shared__ float xs[];                       // [4 * n4][NT]
    static_assert(NT == 2 || NT == 4 || NT == 8, "tokens per tile");
    constexpr int NS = STEPS;
    constexpr int GRP = NT >= 4 ? 4 : 2;                // tokens per LDS read
    typedef float gvec __attribute__((ext_vector_type(GRP)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = p.G;                                  // host: G >= 8, G >= NT
    const int l = lane & (G - 1);
    const int rows_wave = 64 / G;
    const int n4 = (p.n_in + 3) >> 2;
    const int64_t t0 = (int64_t) blockIdx.x * NT;
    const int nt = (int) ((T - t0 < NT) ? (T - t0) : NT);
    const int r0 = (int) blockIdx.y * rows_per_wg;
    const int r1 = min(p.n_out, r0 + rows_per_wg);
    int e0[NS];
    bool eok[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int e = (u * G + l) * 8;
        eok[u] = e < p.Ktot;
        e0[u] = eok[u] ? e : 0;
    }
    const VAL_T * val = reinterpret_cast<const VAL_T *>(p.val);
    const IDX_T * idx = reinterpret_cast<const IDX_T *>(p.idx);
    const int my_t = reduce_token_of_lane<NT>(G, l);
    const bool lead = reduce_is_leader<NT>(G, l) && my_t < nt;
    constexpr int NB = 4;
    const int nbase = p.v ? (p.tall ? 1 : p.B) : 0;
    const float * vt = p.v + (size_t) (t0 + (my_t < nt ? my_t : 0)) * (size_t) p.v_ld;
    Vec8<IDX_T> ii_n[NS];
    Vec8<VAL_T> vv_n[NS];
    float vb_n[NB], scale_n = 1.0f;
    auto request = [&](int on) {
        const int orow = on < r1 ? on : r0;
#pragma unroll
        for (int u = 0; u < NS; ++u) ii_n[u].load(idx + (size_t) orow * p.Ktot + e0[u]);
#pragma unroll
        for (int u = 0; u < NS; ++u) vv_n[u].load(val + (size_t) orow * p.Ktot + e0[u]);
        if (lead) {
            if (p.rs) scale_n = p.rs[orow];
#pragma unroll
            for (int b = 0; b < NB; ++b) vb_n[b] = (b < nbase) ? vt[(size_t) b * p.L + orow] : 0.0f;
        }
    };
    // eight tokens are two 16-byte reads per entry: odd lanes take the halves in the other order, so that one read
    // instruction touches both halves of the 32-byte rows — all sixteen 16-byte bank slots instead of the eight of one
    // parity (the sums of the odd lanes sit in swapped slots until just before the reduction)
    const int half_sw = NT == 8 ? (lane & 1) : 0;
    const int stride = 16 * rows_wave;                  // rows per pass of the workgroup
    int o = r0 + wave * rows_wave + lane / G;
    request(o);                                          // the first rows' payload travels while the tile is filled
    spmm_fill_tile<false, NT, GRP, 1024>(p, xs, t0, nt, n4, tid);
    const int zero_row = 4 * n4;                         // one row past the tile: all zeros, what skipped entries read
    if (tid < NT) xs[(size_t) zero_row * NT + tid] = 0.0f;
    __syncthreads();
    const int passes = (r1 - r0 + stride - 1) / stride;  // uniform over the workgroup
    for (int it = 0; it < passes; ++it, o += stride) {
        const bool row_ok = o < r1;
        Vec8<IDX_T> ii[NS];
        Vec8<VAL_T> vv[NS];
        float vb[NB];
        const float scale = scale_n;
#pragma unroll
        for (int u = 0; u < NS; ++u) { ii[u] = ii_n[u]; vv[u] = vv_n[u]; }
#pragma unroll
        for (int b = 0; b < NB; ++b) vb[b] = vb_n[b];
        request(o + stride);
        f32x2_t acc2[NT / 2];
#pragma unroll
        for (int t = 0; t < NT / 2; ++t) acc2[t] = f32x2_t{0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const bool oku = row_ok & eok[u];
#pragma unroll
            for (int h = 0; h < 8; h += 4) {             // four entries' reads in flight
                gvec xv[4][NT / GRP];
                bool okj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = idx_to_i32<IDX_T>(ii[u].get(h + j));
                    okj[j] = oku & ((unsigned) i < (unsigned) p.n_in);
                    const gvec * row = reinterpret_cast<const gvec *>(xs + (size_t) (okj[j] ? tile_row<false>(p, i, n4) : zero_row) * NT);
#pragma unroll
                    for (int q = 0; q < NT / GRP; ++q) xv[j][q] = row[q ^ half_sw];
                }
                // a skipped entry contributes +0.0f * +0.0f = +0.0f (its value masked, its activations the tile's zero
                // row): acc + (+0) is add_if's acc + (ok ? term : 0) — one select per entry instead of one per entry and token
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = okj[j] ? val_to_f32<VAL_T>(vv[u].get(h + j)) : 0.0f;
#pragma unroll
                    for (int t = 0; t < NT; t += 2)
                        acc2[t / 2] = mul_then_add2(acc2[t / 2], v, f32x2_t{xv[j][t / GRP][t % GRP], xv[j][t / GRP][t % GRP + 1]});
                }
            }
        }
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t += 2) { acc[t] = acc2[t / 2][0]; acc[t + 1] = acc2[t / 2][1]; }
        if constexpr (NT == 8) {                         // odd lanes accumulated tokens 4..7 in slots 0..3: put them back
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float lo = half_sw ? acc[t + 4] : acc[t], hi = half_sw ? acc[t] : acc[t + 4];
                acc[t] = lo; acc[t + 4] = hi;
            }
        }
        const float a = transpose_reduce<NT>(acc, G, lane);
        if (row_ok && lead) {
            float yh = 0.0f;
            if (p.v) {
                if (p.tall) yh = vb[0];
                else {
#pragma unroll
                    for (int b = 0; b < NB; ++b) if (b < nbase) yh += vb[b];
                    for (int b = NB; b < nbase; ++b) yh += vt[(size_t) b * p.L + o];
                }
            }
            store_output(p, o, t0 + my_t, yh + a, scale);
        }
    }
}

// Gate and up of a gated FFN (COO, tall, one shape, I16 / F16) in ONE rows launch: the tile of x is filled once and
// serves both tensors, a wave forms row o of the gate, then row o of up, and the leader lanes write
// h[t][o] = act(gate) * up — delta_rows_coo_lds_kernel twice plus store_output's activation epilogue, minus one tile
// fill, one launch boundary and the gate buffer. Same arithmetic per (tensor, row, token), each node's own non-finite
// guard before the activation.
template <int STEPS, int NT>
__global__ __launch_bounds__(1024)
void delta_rows_coo_lds_pair_kernel(const DeltaParams pg, const DeltaParams pu, const int64_t T, const int rows_per_wg,
                                    const int act, const uint16_t * __restrict__ tab, float * __restrict__ h, const int64_t ldh) {
#pragma clang fp contract(off)
    extern __shared__ float xs[];                       // [4 * n4][NT]
    static_assert(NT == 2 || NT == 4 || NT == 8, "tokens per tile");
    constexpr int NS = STEPS;
    constexpr int GRP = NT >= 4 ? 4 : 2;
    typedef float gvec __attribute__((ext_vector_type(GRP)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = pg.G;                                 // host: one row shape for both tensors, G >= 8, G >= NT
    const int l = lane & (G - 1);
    const int rows_wave = 64 / G;
    const int n4 = (pg.n_in + 3) >> 2;
    const int64_t t0 = (int64_t) blockIdx.x * NT;
    const int nt = (int) ((T - t0 < NT) ? (T - t0) : NT);
    const int r0 = (int) blockIdx.y * rows_per_wg;
    const int r1 = min(pg.n_out, r0 + rows_per_wg);
    int e0[NS];
    bool eok[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int e = (u * G + l) * 8;
        eok[u] = e < pg.Ktot;
        e0[u] = eok[u] ? e : 0;
    }
    const int my_t = reduce_token_of_lane<NT>(G, l);
    const bool lead = reduce_is_leader<NT>(G, l) && my_t < nt;
    const unsigned tok = (unsigned) (t0 + (my_t < nt ? my_t : 0));
    struct Pre { Vec8<uint16_t> ig[NS], vg[NS], iu[NS], vu[NS]; float bg, bu, sg, su; };
    Pre pa, pb;
    pa.sg = pa.su = pb.sg = pb.su = 1.0f;
    pa.bg = pa.bu = pb.bg = pb.bu = 0.0f;
    auto at = [](auto * base, unsigned i) {
        using E = std::remove_pointer_t<decltype(base)>;
        return reinterpret_cast<E *>(reinterpret_cast<const char *>(base) + (size_t) (i * (unsigned) sizeof(E)));
    };
    const uint16_t * gi = reinterpret_cast<const uint16_t *>(pg.idx), * gv = reinterpret_cast<const uint16_t *>(pg.val);
    const uint16_t * ui = reinterpret_cast<const uint16_t *>(pu.idx), * uv = reinterpret_cast<const uint16_t *>(pu.val);
    auto request = [&](Pre & d, int on) {
        const unsigned orow = (unsigned) (on < r1 ? on : r0);
        const unsigned ro = orow * (unsigned) pg.Ktot;
#pragma unroll
        for (int u = 0; u < NS; ++u) d.ig[u].load(at(gi, ro + (unsigned) e0[u]));
#pragma unroll
        for (int u = 0; u < NS; ++u) d.vg[u].load(at(gv, ro + (unsigned) e0[u]));
#pragma unroll
        for (int u = 0; u < NS; ++u) d.iu[u].load(at(ui, ro + (unsigned) e0[u]));
#pragma unroll
        for (int u = 0; u < NS; ++u) d.vu[u].load(at(uv, ro + (unsigned) e0[u]));
        if (lead) {                                      // tall: the base term of (token, row o) is element o of the token's block array
            if (pg.rs) d.sg = *at(pg.rs, orow);
            if (pu.rs) d.su = *at(pu.rs, orow);
            d.bg = *at(pg.v, tok * (unsigned) pg.v_ld + orow);
            d.bu = *at(pu.v, tok * (unsigned) pu.v_ld + orow);
        }
    };
    const int half_sw = NT == 8 ? (lane & 1) : 0;       // see delta_rows_coo_lds_kernel
    const int stride = 16 * rows_wave;
    int o = r0 + wave * rows_wave + lane / G;
    request(pa, o);
    spmm_fill_tile<false, NT, GRP, 1024>(pg, xs, t0, nt, n4, tid);
    const int zero_row = 4 * n4;
    if (tid < NT) xs[(size_t) zero_row * NT + tid] = 0.0f;
    __syncthreads();
    auto row_sum = [&](const Vec8<uint16_t> (&ii)[NS], const Vec8<uint16_t> (&vv)[NS], bool row_ok) -> float {
#pragma clang fp contract(off)
        f32x2_t acc2[NT / 2];
#pragma unroll
        for (int t = 0; t < NT / 2; ++t) acc2[t] = f32x2_t{0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const bool oku = row_ok & eok[u];
#pragma unroll
            for (int hh = 0; hh < 8; hh += 4) {
                gvec xv[4][NT / GRP];
                bool okj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = idx_to_i32<uint16_t>(ii[u].get(hh + j));
                    okj[j] = oku & ((unsigned) i < (unsigned) pg.n_in);
                    const gvec * row = reinterpret_cast<const gvec *>(xs + (size_t) (okj[j] ? tile_row<false>(pg, i, n4) : zero_row) * NT);
#pragma unroll
                    for (int q = 0; q < NT / GRP; ++q) xv[j][q] = row[q ^ half_sw];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {                // skipped entries: +0 * +0, see delta_rows_coo_lds_kernel
                    const float v = okj[j] ? val_to_f32<uint16_t>(vv[u].get(hh + j)) : 0.0f;
#pragma unroll
                    for (int t = 0; t < NT; t += 2)
                        acc2[t / 2] = mul_then_add2(acc2[t / 2], v, f32x2_t{xv[j][t / GRP][t % GRP], xv[j][t / GRP][t % GRP + 1]});
                }
            }
        }
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t += 2) { acc[t] = acc2[t / 2][0]; acc[t + 1] = acc2[t / 2][1]; }
        if constexpr (NT == 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float lo = half_sw ? acc[t + 4] : acc[t], hi = half_sw ? acc[t] : acc[t + 4];
                acc[t] = lo; acc[t + 4] = hi;
            }
        }
        return transpose_reduce<NT>(acc, G, lane);
    };
    auto pass = [&](const Pre & cur, Pre & nxt, int o) {
        const bool row_ok = o < r1;
        request(nxt, o + stride);
        const float ag = row_sum(cur.ig, cur.vg, row_ok);
        const float au = row_sum(cur.iu, cur.vu, row_ok);
        if (row_ok && lead) {
            const float yg = finish_output(pg, cur.bg + ag, cur.sg);
            const float yu = finish_output(pu, cur.bu + au, cur.su);
            h[(size_t) (t0 + my_t) * (size_t) ldh + o] = glu_value(act, yg, yu, tab);
        }
    };
    const int passes = (r1 - r0 + stride - 1) / stride;
    for (int it = 0; it < passes; it += 2) {
        pass(pa, pb, o);
        o += stride;
        if (it + 1 < passes) { pass(pb, pa, o); o += stride; }
    }
}

// The block-payload form of the gate | up rows launch (delta_rows_block_lds_kernel's tile and row walk, rows of one
// step per lane): one fill of the two planes per token serves both tensors.
template <int NT>
__global__ __launch_bounds__(1024)
void delta_rows_block_lds_pair_kernel(const DeltaParams pg, const DeltaParams pu, const int64_t T, const int rows_per_wg,
                                      const int act, const uint16_t * __restrict__ tab, float * __restrict__ h, const int64_t ldh) {
    extern __shared__ float4 xs4[];                     // [NT][2][n_in / 8]
    static_assert(NT == 2 || NT == 4 || NT == 8, "tokens per tile");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = pg.G;                                 // host: one row shape for both tensors, 8 <= G, NT <= G, one step
    const int l = lane & (G - 1);
    const int rows_wave = 64 / G;
    const int nch = pg.n_in >> 3;
    const int64_t t0 = (int64_t) blockIdx.x * NT;
    const int nt = (int) ((T - t0 < NT) ? (T - t0) : NT);
    const int r0 = (int) blockIdx.y * rows_per_wg;
    const int r1 = min(pg.n_out, r0 + rows_per_wg);
    const int e = l * 8;
    const bool eok = e < pg.Ktot;
    const int e0 = eok ? e : 0;
    const int bi = e0 >> pg.block_shift, off = e0 & (pg.block - 1);
    const int my_t = reduce_token_of_lane<NT>(G, l);
    const bool lead = reduce_is_leader<NT>(G, l) && my_t < nt;
    const unsigned tok = (unsigned) (t0 + (my_t < nt ? my_t : 0));
    struct Pre { uint16_t kg, ku; Vec8<uint16_t> vg, vu; float bg, bu, sg, su; };
    Pre pa, pb;
    pa.sg = pa.su = pb.sg = pb.su = 1.0f;
    pa.bg = pa.bu = pb.bg = pb.bu = 0.0f;
    auto at = [](auto * base, unsigned i) {
        using E = std::remove_pointer_t<decltype(base)>;
        return reinterpret_cast<E *>(reinterpret_cast<const char *>(base) + (size_t) (i * (unsigned) sizeof(E)));
    };
    const uint16_t * gi = reinterpret_cast<const uint16_t *>(pg.idx), * gv = reinterpret_cast<const uint16_t *>(pg.val);
    const uint16_t * ui = reinterpret_cast<const uint16_t *>(pu.idx), * uv = reinterpret_cast<const uint16_t *>(pu.val);
    auto request = [&](Pre & d, int on) {
        const unsigned orow = (unsigned) (on < r1 ? on : r0);
        d.kg = *at(gi, orow * (unsigned) pg.nb + (unsigned) bi);
        d.ku = *at(ui, orow * (unsigned) pu.nb + (unsigned) bi);
        d.vg.load(at(gv, orow * (unsigned) pg.Ktot + (unsigned) e0));
        d.vu.load(at(uv, orow * (unsigned) pu.Ktot + (unsigned) e0));
        if (lead) {
            if (pg.rs) d.sg = *at(pg.rs, orow);
            if (pu.rs) d.su = *at(pu.rs, orow);
            d.bg = *at(pg.v, tok * (unsigned) pg.v_ld + orow);
            d.bu = *at(pu.v, tok * (unsigned) pu.v_ld + orow);
        }
    };
    const int stride = 16 * rows_wave;
    int o = r0 + wave * rows_wave + lane / G;
    request(pa, o);
    for (int c = tid; c < nch; c += 1024) {
        float4 va[NT], vb2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float * xt = pg.x + (t0 + (t < nt ? t : nt - 1)) * pg.ldx;
            va[t] = *reinterpret_cast<const float4 *>(xt + c * 8);
            vb2[t] = *reinterpret_cast<const float4 *>(xt + c * 8 + 4);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            xs4[(t * 2 + 0) * nch + c] = va[t];
            xs4[(t * 2 + 1) * nch + c] = vb2[t];
        }
    }
    __syncthreads();
    auto row_sum = [&](uint16_t k, const Vec8<uint16_t> & vv, bool row_ok) -> float {
        const int blk = idx_to_i32<uint16_t>(k);
        const int raw = (blk << pg.block_shift) + off;
        const bool ok = row_ok & eok & ((unsigned) blk < (unsigned) pg.n_blk_in) & (raw < pg.n_in);
        const int c0 = (ok ? raw : 0) >> 3;
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 xa = xs4[(t * 2 + 0) * nch + c0];
            const float4 xb = xs4[(t * 2 + 1) * nch + c0];
            acc[t] = add_if(0.0f, ok, dot8<uint16_t>(vv, xa, xb));
        }
        return transpose_reduce<NT>(acc, G, lane);
    };
    auto pass = [&](const Pre & cur, Pre & nxt, int o) {
        const bool row_ok = o < r1;
        request(nxt, o + stride);
        const float ag = row_sum(cur.kg, cur.vg, row_ok);
        const float au = row_sum(cur.ku, cur.vu, row_ok);
        if (row_ok && lead) {
            const float yg = finish_output(pg, cur.bg + ag, cur.sg);
            const float yu = finish_output(pu, cur.bu + au, cur.su);
            h[(size_t) (t0 + my_t) * (size_t) ldh + o] = glu_value(act, yg, yu, tab);
        }
    };
    const int passes = (r1 - r0 + stride - 1) / stride;
    for (int it = 0; it < passes; it += 2) {
        pass(pa, pb, o);
        o += stride;
        if (it + 1 < passes) { pass(pb, pa, o); o += stride; }
    }
}

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
    acc = group_reduce(acc, G);
    if (row_ok && l == 0) store_output(p, o, t, base_term(p, o, t) + acc, p.rs ? p.rs[o] : 1.0f);
}

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
    store_output(p, o, t, y, p.rs ? p.rs[o] : 1.0f);
}

bool delta_vector_shape(const DeltaParams & p, bool is_block, int * G_out, int * steps_out) {
    const bool vec_ok = (p.Ktot % 8 == 0) && p.x_vec_ok &&
                        (!is_block || (p.block % 8 == 0 && p.block_shift >= 0 && p.n_in % 8 == 0));
    int G = 1;
    if (vec_ok) { while (G < 64 && G * 8 < p.Ktot) G <<= 1; }
    else        { while (G < 64 && G * 4 < p.Ktot) G <<= 1; }
    *G_out = G;
    *steps_out = vec_ok ? (p.Ktot + G * 8 - 1) / (G * 8) : 0;
    return vec_ok;
}

// Raise a kernel's dynamic-LDS limit to the 160 KiB of a gfx950 CU, once per kernel. false = this device does not
// grant it (a 64 KiB LDS part): the LDS rows forms "do not fit" and the caller takes the per-token kernels.
static bool lds_tile_ok(const void * kern) {
    static std::mutex mu;
    static std::unordered_map<const void *, bool> seen;
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find(kern);
    if (it != seen.end()) return it->second;
    const bool ok = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (!ok) (void) hipGetLastError();
    seen[kern] = ok;
    return ok;
}

template <typename IDX_T, typename VAL_T, bool BLOCK>
static int launch_typed(const DeltaLaunch & a, DeltaParams & p, hipStream_t s) {
    if (a.exact_order) {
        const int64_t n = (int64_t) p.n_out * a.T;
        const int64_t blocks = (n + 255) / 256;
        hipLaunchKernelGGL((delta_exact_kernel<IDX_T, VAL_T, BLOCK>), dim3((unsigned) blocks), dim3(256), 0, s, p, a.T);
        SDH_HIP_CHECK(hipGetLastError());
        return SDH_OK;
    }
    int G = 1, steps = 0;
    const bool vec_ok = delta_vector_shape(p, BLOCK, &G, &steps);
    p.G = G;
    const int rows_per_block = 4 * (64 / G);
    const unsigned gy = (unsigned) ((p.n_out + rows_per_block - 1) / rows_per_block);
    if (gy > 65535u || a.T > 0x7fffffffLL) {
        set_error("grid too large (n_out=%d, T=%lld)", p.n_out, (long long) a.T);
        return SDH_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned) a.T, gy);
    // several tokens of a COO payload with x in LDS (delta_rows_coo_lds_kernel)
    if constexpr (!BLOCK) {
        static const int coo_lds_min_t = getenv("SDH_ROWS_COO_LDS_MIN_T") ? atoi(getenv("SDH_ROWS_COO_LDS_MIN_T")) : 5;   // dev: A/B
        const size_t col = (size_t) ((p.n_in + 3) / 4) * 4 * sizeof(float);
        const size_t budget = 152 * 1024;
        const int max_steps = sizeof(VAL_T) == 4 || sizeof(IDX_T) == 4 ? 1 : 2;
        if (vec_ok && a.T >= coo_lds_min_t && G >= 8 && steps >= 1 && steps <= max_steps && col * 2 <= budget) {
            int nt = col * 8 <= budget ? 8 : col * 4 <= budget ? 4 : 2;
            while (nt > 2 && (nt > G || nt / 2 >= a.T)) nt >>= 1;   // no wider than the batch needs
            const int64_t tiles = (a.T + nt - 1) / nt;
            const int rows_pass = 16 * (64 / G);
            // a workgroup fills its tile (up to 128 KB) before its first row: one round of workgroups, one per CU
            static const int wgs_per_cu = getenv("SDH_ROWS_COO_WGS_PER_CU") ? atoi(getenv("SDH_ROWS_COO_WGS_PER_CU")) : 1;   // dev: A/B
            int64_t chunks = (wgs_per_cu * (int64_t) lib().num_cu) / tiles;
            const int64_t max_chunks = (p.n_out + rows_pass - 1) / rows_pass;
            if (chunks > max_chunks) chunks = max_chunks;
            if (chunks < 1) chunks = 1;
            int rows_per_wg = (int) ((p.n_out + chunks - 1) / chunks);
            rows_per_wg = (rows_per_wg + rows_pass - 1) / rows_pass * rows_pass;
            const unsigned gl = (unsigned) ((p.n_out + rows_per_wg - 1) / rows_per_wg);
            void (*kern)(const DeltaParams, int64_t, int) = nullptr;
            if (steps == 1) kern = nt == 8 ? delta_rows_coo_lds_kernel<IDX_T, VAL_T, 1, 8> : nt == 4 ? delta_rows_coo_lds_kernel<IDX_T, VAL_T, 1, 4> : delta_rows_coo_lds_kernel<IDX_T, VAL_T, 1, 2>;
            else if constexpr (sizeof(VAL_T) == 2 && sizeof(IDX_T) == 2)
                kern = nt == 8 ? delta_rows_coo_lds_kernel<IDX_T, VAL_T, 2, 8> : nt == 4 ? delta_rows_coo_lds_kernel<IDX_T, VAL_T, 2, 4> : delta_rows_coo_lds_kernel<IDX_T, VAL_T, 2, 2>;
            // a device whose LDS cannot be raised to the tile size ("does not fit"): the per-token kernels below
            if (kern && lds_tile_ok((const void *) kern)) {
                hipLaunchKernelGGL(kern, dim3((unsigned) tiles, gl), dim3(1024), col * nt + (size_t) nt * 4, s, p, a.T, rows_per_wg);   // + the zero row
                SDH_HIP_CHECK(hipGetLastError());
                return SDH_OK;
            }
        }
    }
    // several tokens of a block payload: activations of 2 / 4 / 8 tokens in the LDS, rows walked pass by pass. A tile fill
    // per workgroup has to pay for itself. Long rows (G = 64; cfg3, us per call, LDS form / per-token kernels): T=4 17.2 /
    // 17.1, T=8 25.8 / 24.9, T=12 27.5 / 29.8, T=16 32.6 / 35.5, T=32 42.4 / 60.1, T=512 360 / 800: from 10 tokens on.
    // Short rows come here from the gated FFN's node launches (SDH_FWD_INTERNAL_ROWS) and with 5 tokens: from 5 on.
    if constexpr (BLOCK) {
        static const bool no_lds = getenv("SDH_ROWS_NO_LDS") != nullptr;           // dev: A/B
        static const int lds_min_t = getenv("SDH_ROWS_LDS_MIN_T") ? atoi(getenv("SDH_ROWS_LDS_MIN_T")) : 10;  // dev: A/B
        static const int wgs_per_cu = getenv("SDH_ROWS_BLOCK_WGS_PER_CU") ? atoi(getenv("SDH_ROWS_BLOCK_WGS_PER_CU")) : 0;
        const size_t col = (size_t) p.n_in * sizeof(float);
        const size_t budget = 152 * 1024;
        const int max_steps = sizeof(VAL_T) == 4 ? 3 : 4;          // fp32 values at four steps would spill (128 registers per lane)
        const int64_t min_t = G == 64 ? lds_min_t : 5;
        const bool off32 = (size_t) p.n_out * (size_t) p.Ktot * sizeof(VAL_T) < ((size_t) 1 << 32) && (size_t) p.n_out * (size_t) p.nb * sizeof(IDX_T) < ((size_t) 1 << 32) &&
                           (!p.v || ((size_t) a.T * (size_t) p.v_ld + (size_t) p.B * (size_t) p.L) * 4 < ((size_t) 1 << 32));
        if (vec_ok && a.T >= min_t && G >= 8 && steps >= 1 && steps <= max_steps && !no_lds && col * 2 <= budget && off32) {
            int nt = (col * 8 <= budget && steps == 1) ? 8 : (col * 4 <= budget && steps <= 2) ? 4 : 2;     // register budget
            while (nt > 2 && (nt > G || nt / 2 >= a.T)) nt >>= 1;  // no wider than the batch needs
            const int64_t tiles = (a.T + nt - 1) / nt;
            const int rows_pass = 16 * (64 / G);
            // long rows: ~4 workgroups per CU over the launch; short rows: one round (the fill is a larger share)
            const int wpc = wgs_per_cu > 0 ? wgs_per_cu : (G == 64 ? 4 : 1);
            int64_t chunks = (wpc * (int64_t) lib().num_cu + tiles - 1) / tiles;
            const int64_t max_chunks = (p.n_out + rows_pass - 1) / rows_pass;
            if (chunks > max_chunks) chunks = max_chunks;
            if (chunks < 1) chunks = 1;
            int rows_per_wg = (int) ((p.n_out + chunks - 1) / chunks);
            rows_per_wg = (rows_per_wg + rows_pass - 1) / rows_pass * rows_pass;
            const unsigned gl = (unsigned) ((p.n_out + rows_per_wg - 1) / rows_per_wg);
            void (*kern)(const DeltaParams, int64_t, int) = nullptr;
            if (G == 64) {                                  // rows of 64 lanes: the reduction's steps are compile-time
                switch (steps) {
                    case 1:  kern = nt == 8 ? delta_rows_block_lds_kernel<IDX_T, VAL_T, 1, 8, 64> : nt == 4 ? delta_rows_block_lds_kernel<IDX_T, VAL_T, 1, 4, 64> : delta_rows_block_lds_kernel<IDX_T, VAL_T, 1, 2, 64>; break;
                    case 2:  kern = nt == 4 ? delta_rows_block_lds_kernel<IDX_T, VAL_T, 2, 4, 64> : delta_rows_block_lds_kernel<IDX_T, VAL_T, 2, 2, 64>; break;
                    case 3:  kern = delta_rows_block_lds_kernel<IDX_T, VAL_T, 3, 2, 64>; break;
                    default: if constexpr (sizeof(VAL_T) == 2) kern = delta_rows_block_lds_kernel<IDX_T, VAL_T, 4, 2, 64>; break;
                }
            } else {                                        // G < 64: one step per lane
                kern = nt == 8 ? delta_rows_block_lds_kernel<IDX_T, VAL_T, 1, 8, 0> : nt == 4 ? delta_rows_block_lds_kernel<IDX_T, VAL_T, 1, 4, 0> : delta_rows_block_lds_kernel<IDX_T, VAL_T, 1, 2, 0>;
            }
            if (kern && lds_tile_ok((const void *) kern)) {
                hipLaunchKernelGGL(kern, dim3((unsigned) tiles, gl), dim3(1024), col * nt, s, p, a.T, rows_per_wg);
                SDH_HIP_CHECK(hipGetLastError());
                return SDH_OK;
            }
        }
    }
    if (vec_ok) {
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

DeltaParams make_delta_params(const DeviceWeight & w, const DeltaLaunch & a) {
    DeltaParams p{};
    p.idx = w.idx; p.val = w.val; p.rs = w.rs;
    p.x = a.x; p.y = a.y; p.v = a.v; p.nf = a.nf_counter;
    p.ldx = a.ldx; p.ldy = a.ldy;
    p.n_in = (int) w.n_in; p.n_out = (int) w.n_out;
    const bool is_block = (w.kind == SDH_OP_BLOCK || w.kind == SDH_OP_BASE_BLOCK);
    p.nb = (int) w.nb; p.block = (int) w.block;
    p.Ktot = is_block ? (int) (w.nb * w.block) : (int) w.K;
    p.L = (int) w.L; p.B = a.v_blocks > 0 ? a.v_blocks : (int) w.B; p.tall = w.tall ? 1 : 0;
    p.v_ld = a.v_ld > 0 ? a.v_ld : (int64_t) p.B * p.L;
    p.glu_gate = a.glu_gate; p.glu_ldg = a.glu_ldg; p.glu_act = a.glu_act; p.glu_tab = a.glu_tab;
    p.G = 1;
    p.zero_nf = a.zero_nonfinite ? 1 : 0;
    p.x_vec_ok = (((uintptr_t) a.x % 16) == 0 && (a.ldx % 4) == 0) ? 1 : 0;
    p.n_blk_in = is_block ? (int) ((w.n_in + w.block - 1) / w.block) : 0;
    p.block_shift = -1;
    if (is_block && w.block > 0 && (w.block & (w.block - 1)) == 0) {
        int sft = 0;
        while ((1LL << sft) < w.block) ++sft;
        p.block_shift = sft;
    }
    return p;
}

int launch_delta(const DeviceWeight & w, const DeltaLaunch & a, hipStream_t s) {
    if (a.T <= 0) return SDH_OK;
    DeltaParams p = make_delta_params(w, a);
    const bool is_block = (w.kind == SDH_OP_BLOCK || w.kind == SDH_OP_BASE_BLOCK);
    const bool i16 = w.idx_type == SDH_TYPE_I16;
    const bool f16 = w.val_type == SDH_TYPE_F16;
    if (is_block) {
        if (i16 && f16)  return launch_typed<uint16_t, uint16_t, true>(a, p, s);
        if (i16 && !f16) return launch_typed<uint16_t, uint32_t, true>(a, p, s);
        if (!i16 && f16) return launch_typed<uint32_t, uint16_t, true>(a, p, s);
        return launch_typed<uint32_t, uint32_t, true>(a, p, s);
    }
    if (i16 && f16)  return launch_typed<uint16_t, uint16_t, false>(a, p, s);
    if (i16 && !f16) return launch_typed<uint16_t, uint32_t, false>(a, p, s);
    if (!i16 && f16) return launch_typed<uint32_t, uint16_t, false>(a, p, s);
    return launch_typed<uint32_t, uint32_t, false>(a, p, s);
}

// Gate and up rows of a gated FFN in one launch (delta_rows_coo_lds_pair_kernel). ag / au: the two nodes' launches with
// their base terms in v (tall). Returns SDH_OK if launched, 1 if the pair does not fit this form.
int launch_delta_rows_pair(const DeviceWeight & g, const DeviceWeight & u, const DeltaLaunch & ag, const DeltaLaunch & au,
                           int act, const uint16_t * gelu_tab, float * h, int64_t ldh, hipStream_t s) {
    static const bool off = [] { const char * e = getenv("SDH_ROWS_PAIR"); return e && atoi(e) == 0; }();          // dev: A/B
    if (off || ag.T < 5 || ag.exact_order) return 1;
    const bool is_block = g.kind == SDH_OP_BASE_BLOCK;
    if (g.kind != u.kind || (g.kind != SDH_OP_BASE_COO && !is_block) || !g.tall || !u.tall) return 1;
    if (g.idx_type != SDH_TYPE_I16 || g.val_type != SDH_TYPE_F16 || u.idx_type != SDH_TYPE_I16 || u.val_type != SDH_TYPE_F16) return 1;
    if (g.n_in != u.n_in || g.n_out != u.n_out || !ag.v || !au.v || ag.x != au.x || ag.ldx != au.ldx) return 1;
    DeltaParams pg = make_delta_params(g, ag), pu = make_delta_params(u, au);
    if (pg.Ktot != pu.Ktot || pg.nb != pu.nb || pg.block != pu.block) return 1;
    int G = 1, steps = 0;
    if (!delta_vector_shape(pg, is_block, &G, &steps) || !delta_vector_shape(pu, is_block, &G, &steps) || steps != 1 || G < 8) return 1;
    pg.G = pu.G = G;
    // block payloads with eight-token tiles leave the rows kernels for the tiled path above 32 tokens (sdh_forward_impl)
    if (is_block && ag.T > 32 && (size_t) pg.n_in * 16 <= (size_t) 152 * 1024) return 1;
    const size_t col = is_block ? (size_t) pg.n_in * sizeof(float) : (size_t) ((pg.n_in + 3) / 4) * 4 * sizeof(float);
    const size_t budget = 152 * 1024;
    if (col * 2 > budget) return 1;
    if (is_block && (pg.n_in % 8) != 0) return 1;
    if ((size_t) pg.n_out * (size_t) pg.Ktot * 2 >= ((size_t) 1 << 32) ||
        ((size_t) ag.T * (size_t) std::max(pg.v_ld, pu.v_ld) + (size_t) pg.n_out) * 4 >= ((size_t) 1 << 32)) return 1;    // 32-bit offsets
    int nt = col * 8 <= budget ? 8 : col * 4 <= budget ? 4 : 2;
    while (nt > 2 && (nt > G || nt / 2 >= ag.T)) nt >>= 1;
    const int64_t tiles = (ag.T + nt - 1) / nt;
    const int rows_pass = 16 * (64 / G);
    int64_t chunks = (int64_t) lib().num_cu / tiles;
    const int64_t max_chunks = (pg.n_out + rows_pass - 1) / rows_pass;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    int rows_per_wg = (int) ((pg.n_out + chunks - 1) / chunks);
    rows_per_wg = (rows_per_wg + rows_pass - 1) / rows_pass * rows_pass;
    const unsigned gl = (unsigned) ((pg.n_out + rows_per_wg - 1) / rows_per_wg);
    void (*kern)(const DeltaParams, const DeltaParams, int64_t, int, int, const uint16_t *, float *, int64_t) =
        is_block ? (nt == 8 ? delta_rows_block_lds_pair_kernel<8> : nt == 4 ? delta_rows_block_lds_pair_kernel<4> : delta_rows_block_lds_pair_kernel<2>)
                 : (nt == 8 ? delta_rows_coo_lds_pair_kernel<1, 8> : nt == 4 ? delta_rows_coo_lds_pair_kernel<1, 4> : delta_rows_coo_lds_pair_kernel<1, 2>);
    if (!lds_tile_ok((const void *) kern)) return 1;           // does not fit this device's LDS: the caller goes node by node
    hipLaunchKernelGGL(kern, dim3((unsigned) tiles, gl), dim3(1024), col * nt + (is_block ? 0 : (size_t) nt * 4), s, pg, pu, ag.T, rows_per_wg, act, gelu_tab, h, ldh);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

} // namespace sdh
