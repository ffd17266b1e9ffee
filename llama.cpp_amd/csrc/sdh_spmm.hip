// sdh_spmm.hip — the residual Δ·X for many tokens (prefill) as a token-tiled sparse × dense product.
//
// One workgroup (1024 threads) stages the activations of TT tokens transposed in LDS,
// xs[i][t] (token fastest, up to 128 KB), then walks a chunk of output rows. A lane owns one
// (row, token) pair and runs over the row's entries IN ORDER, one separately rounded multiply and
// one add per entry, starting from the base term — the very sequence the CPU op performs
// (src/llama-seeddelta.cpp:676-700, :905-945, :1104-1125, :1150-1174), so this path is bit-identical
// to the reference, not merely within tolerance. The TT lanes of a row read the same idx / val
// addresses (one broadcast transaction, 8 entries per 16-byte load) and TT consecutive floats of LDS.
//
// Costs per launch: the payload is streamed once per token tile (T/TT times, from L2 after the first
// tile of a row chunk — consecutive workgroups share the chunk), every multiply-add costs one LDS
// read of 4 bytes. No atomics, no cross-lane reduction.
#include "sdh_internal.h"
#include "sdh_delta_dev.h"

#include <cstdlib>

namespace sdh {

// Row of input i in the LDS tile. COO: i. Block scheme: one spare row after every block, i + i / block — the rows a
// wave works on start their blocks at in0 * TT * 4 bytes, and with block * TT * 4 a multiple of the 256-byte bank
// cycle (block 16 or 32, TT 8) every lane of a pass hit the same banks (rocprof: 87 % of the LDS cycles of a
// Mistral gate block16 ubatch were bank conflicts, 8-way). With the spare row the start bank follows blk mod 8,
// as scattered as the COO gathers are.
//
// COO: the rows are de-interleaved by fours, input i in row (i & 3) * n4 + (i >> 2) with n4 = ceil(n_in / 4). The
// gathers do not care (any bijection serves), and the FILL can then read x sixteen bytes at a time: a lane loads
// inputs 4g .. 4g+3 of each token of a group and stores input 4g+j's tokens to row j * n4 + g — consecutive lanes,
// consecutive rows, as conflict-free as the one-input-per-lane fill was, at a quarter of the global loads. With few
// tokens the fill IS the kernel (a 256-thread workgroup fills 128 KB through a chain of L2 round trips: 15 of the
// 59 us of a Mistral FFN layer at 8 tokens were tile fills).
template <bool BLOCK>
static size_t spmm_tile_rows(const DeltaParams & p) {
    return BLOCK ? (size_t) p.n_in + (size_t) ((p.n_in + p.block - 1) / p.block) : (size_t) ((p.n_in + 3) / 4) * 4;
}

template <typename IDX_T, typename VAL_T, bool BLOCK, bool VEC, int TT, int NTH>
__global__ __launch_bounds__(NTH)
void delta_spmm_kernel(const DeltaParams p, int64_t T, int rows_per_wg) {
#pragma clang fp contract(off)
    extern __shared__ float xs[];                       // [n_in][TT]
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t) blockIdx.x * TT;
    const int nt = (int) ((T - t0 < TT) ? (T - t0) : TT);
    const int n4 = (p.n_in + 3) >> 2;
    spmm_fill_tile<BLOCK, TT, (TT % 4 == 0 ? 4 : TT % 2 == 0 ? 2 : 1), NTH>(p, xs, t0, nt, n4, tid);   // ragged last tile: zeros
    __syncthreads();

    const int t = tid % TT;
    const bool t_ok = t < nt;
    const int64_t tok = t0 + (t_ok ? t : 0);
    const int r1 = min(p.n_out, ((int) blockIdx.y + 1) * rows_per_wg);
    for (int o = (int) blockIdx.y * rows_per_wg + tid / TT; o < r1; o += NTH / TT) {
        const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) o * p.Ktot;
        const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) o * (BLOCK ? p.nb : p.Ktot);
        float y = 0.0f;
        if (p.v) {                                      // base term: tall picks, wide sums blocks in order
            const float * vt = p.v + (size_t) tok * (size_t) p.v_ld;
            if (p.tall) y = vt[o];
            else for (int b = 0; b < p.B; ++b) y += vt[(size_t) b * p.L + o];
        }
        if constexpr (BLOCK) {
            for (int bi = 0; bi < p.nb; ++bi) {
                const int blk = idx_to_i32<IDX_T>(irow[bi]);
                const int64_t in0 = (int64_t) blk * p.block;
                if (blk < 0 || in0 >= p.n_in) continue;             // uniform over the TT lanes of the row
                const int len = (int) ((in0 + p.block < p.n_in) ? p.block : (p.n_in - in0));
                const VAL_T * vb = vrow + (size_t) bi * p.block;
                const float * xb = xs + (size_t) (in0 + blk) * TT + t;      // tile_row(in0 + j) = in0 + blk + j inside the block
                if constexpr (VEC) {
                    for (int j = 0; j < p.block; j += 8) {
                        Vec8<VAL_T> vv;
                        vv.load(vb + j);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool ok = j + u < len;
                            const float prod = val_to_f32<VAL_T>(vv.get(u)) * xb[(ok ? j + u : 0) * TT];
                            y = ok ? y + prod : y;
                        }
                    }
                } else {
                    for (int j = 0; j < len; ++j) {
                        const float prod = val_to_f32<VAL_T>(vb[j]) * xb[j * TT];
                        y = y + prod;
                    }
                }
            }
        } else {
            if constexpr (VEC) {
                for (int e = 0; e < p.Ktot; e += 8) {
                    Vec8<IDX_T> ii;
                    Vec8<VAL_T> vv;
                    ii.load(irow + e);
                    vv.load(vrow + e);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = idx_to_i32<IDX_T>(ii.get(u));
                        const bool ok = (unsigned) i < (unsigned) p.n_in;
                        const float prod = val_to_f32<VAL_T>(vv.get(u)) * xs[tile_row<false>(p, ok ? i : 0, n4) * TT + t];
                        y = ok ? y + prod : y;
                    }
                }
            } else {
                for (int e = 0; e < p.Ktot; ++e) {
                    const int i = idx_to_i32<IDX_T>(irow[e]);
                    if ((unsigned) i >= (unsigned) p.n_in) continue;
                    const float prod = val_to_f32<VAL_T>(vrow[e]) * xs[tile_row<false>(p, i, n4) * TT + t];
                    y = y + prod;
                }
            }
        }
        if (t_ok) store_output(p, o, tok, y, p.rs ? p.rs[o] : 1.0f);
    }
}

// Four tokens per lane: one 16-byte LDS read feeds four multiply-adds, and the index / value unpacking
// is paid once per entry instead of once per (entry, token). The one-token kernel above spends ~10
// VALU instructions per multiply-add and is issue-bound (rocprof: 141.5 M VALU wave-instructions for
// 14.7 M wave-level multiply-adds, profiles/r01_d_prefill_pmc.json); this form needs ~3. Same
// accumulation order per (row, token), so still bit-identical to the CPU op. Vector payload shapes only.
template <typename IDX_T, typename VAL_T, bool BLOCK, int TT, int NTH, int TPL /* tokens per lane: 4, or 2 for two-token tiles */>
__global__ __launch_bounds__(NTH)
void delta_spmm4_kernel(const DeltaParams p, int64_t T, int rows_per_wg) {
#pragma clang fp contract(off)
    extern __shared__ float xs[];                       // [n_in][TT]
    static_assert(TT % TPL == 0 && (TPL == 4 || TPL == 2), "four or two tokens per lane");
    constexpr int LPR = TT / TPL;                       // lanes per row
    typedef float float2v __attribute__((ext_vector_type(2)));
    typedef float xvec __attribute__((ext_vector_type(TPL)));           // one LDS read: the lane's tokens of one input
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t) blockIdx.x * TT;
    const int nt = (int) ((T - t0 < TT) ? (T - t0) : TT);
    const int n4 = (p.n_in + 3) >> 2;
    spmm_fill_tile<BLOCK, TT, TPL, NTH>(p, xs, t0, nt, n4, tid);
    __syncthreads();

    const int tq = (tid % LPR) * TPL;                   // first of this lane's tokens within the tile
    const xvec * xq = reinterpret_cast<const xvec *>(xs + tq);         // xq[i * LPR] = x[i][tq .. tq+TPL-1]
    const int r1 = min(p.n_out, ((int) blockIdx.y + 1) * rows_per_wg);
    for (int o = (int) blockIdx.y * rows_per_wg + tid / LPR; o < r1; o += NTH / LPR) {
        const VAL_T * vrow = reinterpret_cast<const VAL_T *>(p.val) + (size_t) o * p.Ktot;
        const IDX_T * irow = reinterpret_cast<const IDX_T *>(p.idx) + (size_t) o * (BLOCK ? p.nb : p.Ktot);
        float y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (p.v) {
#pragma unroll
            for (int u = 0; u < TPL; ++u) {
                const int64_t tok = t0 + ((tq + u < nt) ? tq + u : 0);
                const float * vt = p.v + (size_t) tok * (size_t) p.v_ld;
                if (p.tall) y[u] = vt[o];
                else for (int b = 0; b < p.B; ++b) y[u] += vt[(size_t) b * p.L + o];
            }
        }
        // two tokens per instruction: v_pk_mul_f32 / v_pk_add_f32 round each component as the scalar forms do
        // (contraction is off), so the sums keep the CPU op's bits at half the multiply and add instructions
        float2v y01 = {y[0], y[1]}, y23 = {y[2], y[3]};
        auto mac = [&](float v, const xvec x) {
            const float2v vv = {v, v}, x01 = {x[0], x[1]};
            const float2v p01 = vv * x01;
            y01 = y01 + p01;
            if constexpr (TPL == 4) {
                const float2v x23 = {x[2], x[3]};
                const float2v p23 = vv * x23;
                y23 = y23 + p23;
            }
        };
        if constexpr (BLOCK) {
            for (int bi = 0; bi < p.nb; ++bi) {
                const int blk = idx_to_i32<IDX_T>(irow[bi]);
                const int64_t in0 = (int64_t) blk * p.block;
                if (blk < 0 || in0 >= p.n_in) continue;
                const int len = (int) ((in0 + p.block < p.n_in) ? p.block : (p.n_in - in0));
                const VAL_T * vb = vrow + (size_t) bi * p.block;
                const xvec * xb = xq + (size_t) (in0 + blk) * LPR;        // tile_row(in0 + j) = in0 + blk + j inside the block
                for (int j = 0; j < p.block; j += 8) {
                    Vec8<VAL_T> vv;
                    vv.load(vb + j);
                    if (j + 8 <= len) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) mac(val_to_f32<VAL_T>(vv.get(u)), xb[(j + u) * LPR]);
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (j + u < len) mac(val_to_f32<VAL_T>(vv.get(u)), xb[(j + u) * LPR]);
                    }
                }
            }
        } else {
            // the next 8 entries are requested before the current 8 are used: with one workgroup per CU (the tile fills
            // the LDS) only four waves per SIMD stand behind a step's L2 round trip
            Vec8<IDX_T> ii_n;
            Vec8<VAL_T> vv_n;
            ii_n.load(irow);
            vv_n.load(vrow);
            for (int e = 0; e < p.Ktot; e += 8) {
                const Vec8<IDX_T> ii = ii_n;
                const Vec8<VAL_T> vv = vv_n;
                if (e + 8 < p.Ktot) { ii_n.load(irow + e + 8); vv_n.load(vrow + e + 8); }
                int ix[8];
                bool all_ok = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ix[u] = idx_to_i32<IDX_T>(ii.get(u));
                    all_ok = all_ok & ((unsigned) ix[u] < (unsigned) p.n_in);
                }
                if (all_ok) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) mac(val_to_f32<VAL_T>(vv.get(u)), xq[tile_row<false>(p, ix[u], n4) * LPR]);
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) if ((unsigned) ix[u] < (unsigned) p.n_in) mac(val_to_f32<VAL_T>(vv.get(u)), xq[tile_row<false>(p, ix[u], n4) * LPR]);
                }
            }
        }
        const float scale = p.rs ? p.rs[o] : 1.0f;
        y[0] = y01[0]; y[1] = y01[1]; y[2] = y23[0]; y[3] = y23[1];
#pragma unroll
        for (int u = 0; u < TPL; ++u) if (tq + u < nt) store_output(p, o, t0 + tq + u, y[u], scale);
    }
}

// Two-token tiles: a lane takes both tokens of a row (half the LDS reads and payload unpacking per multiply-add) only
// when the launch has tiles to spare; with few tokens that halves the lanes at work and the one-token form wins
// (Mistral down tensor, T=8: 12.7 us against 27.8 us; FFN layer, COO / block16, us: T=64 98.6 / 79.7 against 106 / 83.4,
// T=128 170 / 136 against 169 / 131, T=512 542 / 438 against 543 / 429).
static inline bool spmm_two_per_lane(int64_t T) { return T >= 96; }

// NTH threads per workgroup. A tile fills the LDS of a CU, so a launch has at most tiles x ceil(n_out / rows of one
// pass) workgroups: with 1024 threads and few tokens that is a handful (T=8, Mistral gate: 28 workgroups on 256 CUs —
// a gated FFN layer took 75 us at 8 tokens against 36 us at 4). When the 1024-thread form would leave most CUs
// without a workgroup the 256-thread form is used: a quarter of the rows per pass, four times the workgroups.
template <typename IDX_T, typename VAL_T, bool BLOCK, bool VEC, int TT, int NTH>
static int launch_tt_n(const DeltaParams & p, int64_t T, hipStream_t s) {
    const size_t lds = spmm_tile_rows<BLOCK>(p) * TT * sizeof(float);
    static const bool one_token = getenv("SDH_SPMM_ONE_TOKEN") != nullptr;      // dev: A/B against the four-token form
    void (*kern)(const DeltaParams, int64_t, int) = delta_spmm_kernel<IDX_T, VAL_T, BLOCK, VEC, TT, NTH>;
    int lanes_per_row = TT;
    if constexpr (VEC && TT % 4 == 0) {
        if (!one_token) { kern = delta_spmm4_kernel<IDX_T, VAL_T, BLOCK, TT, NTH, 4>; lanes_per_row = TT / 4; }
    } else if constexpr (VEC && TT == 2) {               // long inputs: both tokens of the tile in one lane, one 8-byte LDS read per entry
        if (!one_token && spmm_two_per_lane(T)) { kern = delta_spmm4_kernel<IDX_T, VAL_T, BLOCK, TT, NTH, 2>; lanes_per_row = 1; }
    }
    SDH_HIP_CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int64_t tiles = (T + TT - 1) / TT;
    if (tiles > 0x7fffffff) { set_error("spmm: too many token tiles"); return SDH_ERR_UNSUPPORTED; }
    const int rows_pass = NTH / lanes_per_row;
    // ~4 workgroups per CU in flight over the whole launch; a chunk is a whole number of row passes
    int64_t chunks = (4 * (int64_t) lib().num_cu + tiles - 1) / tiles;
    const int64_t max_chunks = (p.n_out + rows_pass - 1) / rows_pass;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    if (chunks > 65535) chunks = 65535;
    int rows_per_wg = (int) ((p.n_out + chunks - 1) / chunks);
    rows_per_wg = (rows_per_wg + rows_pass - 1) / rows_pass * rows_pass;
    const unsigned gy = (unsigned) ((p.n_out + rows_per_wg - 1) / rows_per_wg);
    hipLaunchKernelGGL(kern, dim3((unsigned) tiles, gy), dim3(NTH), lds, s, p, T, rows_per_wg);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <typename IDX_T, typename VAL_T, bool BLOCK, bool VEC, int TT>
static int launch_tt(const DeltaParams & p, int64_t T, hipStream_t s) {
    const int64_t tiles = (T + TT - 1) / TT;
    const int lanes_per_row = (VEC && TT % 4 == 0) ? TT / 4 : (VEC && TT == 2 && spmm_two_per_lane(T)) ? 1 : TT;
    const int64_t wgs_1024 = tiles * ((p.n_out + 1024 / lanes_per_row - 1) / (1024 / lanes_per_row));
    static const int force = [] { const char * e = getenv("SDH_SPMM_NTH"); return e ? atoi(e) : 0; }();        // dev: A/B
    // measured, Mistral FFN layer (gate, up, down; us): T=8 75 -> 59, T=16 77 -> 61, but T=32 87 -> 98 and T=64 98 -> 171:
    // the narrow form pays once the wide one would leave three quarters of the CUs without a workgroup
    const bool narrow = force ? force == 256 : 4 * wgs_1024 <= (int64_t) lib().num_cu;
    return narrow ? launch_tt_n<IDX_T, VAL_T, BLOCK, VEC, TT, 256>(p, T, s) : launch_tt_n<IDX_T, VAL_T, BLOCK, VEC, TT, 1024>(p, T, s);
}

template <typename IDX_T, typename VAL_T, bool BLOCK, bool VEC>
static int launch_vec(const DeltaParams & p, int64_t T, hipStream_t s) {
    const size_t budget = 152 * 1024;                   // LDS for the tile, leaving headroom
    const size_t col = spmm_tile_rows<BLOCK>(p) * sizeof(float);
    if (col * 16 <= budget / 2 && T >= 16) return launch_tt<IDX_T, VAL_T, BLOCK, VEC, 16>(p, T, s);   // two workgroups per CU
    if (col * 8 <= budget) return launch_tt<IDX_T, VAL_T, BLOCK, VEC, 8>(p, T, s);
    // narrower tiles re-stream the payload T/TT times: measured on MI355X (T=512, n_in=14336) the COO
    // form still wins 4x over the per-token kernels, the block form loses (0.25 vs 0.15 ms at K=128,
    // 2.6 vs 0.8 ms at K=1536), so block payloads with long inputs kept the per-token kernels
    // (re-measured with the spare-row layout: at K=128 the block form now wins as well, 0.161 -> 0.132 ms per Mistral
    // down block16 ubatch; at K=1536 it still loses, 0.81 -> 1.07 ms — the payload is re-streamed T/TT times)
    if (BLOCK && p.Ktot > 256) return 1;                 // -> delta_rows_block_lds_kernel (0.33 ms at K=1536)
    if (col * 4 <= budget) return launch_tt<IDX_T, VAL_T, BLOCK, VEC, 4>(p, T, s);
    if (col * 2 <= budget) return launch_tt<IDX_T, VAL_T, BLOCK, VEC, 2>(p, T, s);
    return 1;                                           // does not fit: caller uses the row kernels (sdh_delta.hip)
}

template <typename IDX_T, typename VAL_T, bool BLOCK>
static int launch_typed(const DeltaParams & p, int64_t T, hipStream_t s) {
    // 16-byte payload loads need 8-element runs that never straddle a row or a block
    const bool vec = (p.Ktot % 8 == 0) && (!BLOCK || p.block % 8 == 0);
    return vec ? launch_vec<IDX_T, VAL_T, BLOCK, true>(p, T, s) : launch_vec<IDX_T, VAL_T, BLOCK, false>(p, T, s);
}

// Returns SDH_OK if launched, 1 if the shape does not fit this path, <0 on error.
int launch_delta_spmm(const DeviceWeight & w, const DeltaLaunch & a, hipStream_t s) {
    if (a.T <= 0) return SDH_OK;
    DeltaParams p = make_delta_params(w, a);
    const bool is_block = (w.kind == SDH_OP_BLOCK || w.kind == SDH_OP_BASE_BLOCK);
    const bool i16 = w.idx_type == SDH_TYPE_I16, f16 = w.val_type == SDH_TYPE_F16;
    if (is_block) {
        if (i16 && f16)  return launch_typed<uint16_t, uint16_t, true>(p, a.T, s);
        if (i16 && !f16) return launch_typed<uint16_t, uint32_t, true>(p, a.T, s);
        if (!i16 && f16) return launch_typed<uint32_t, uint16_t, true>(p, a.T, s);
        return launch_typed<uint32_t, uint32_t, true>(p, a.T, s);
    }
    if (i16 && f16)  return launch_typed<uint16_t, uint16_t, false>(p, a.T, s);
    if (i16 && !f16) return launch_typed<uint16_t, uint32_t, false>(p, a.T, s);
    if (!i16 && f16) return launch_typed<uint32_t, uint16_t, false>(p, a.T, s);
    return launch_typed<uint32_t, uint32_t, false>(p, a.T, s);
}

} // namespace sdh
