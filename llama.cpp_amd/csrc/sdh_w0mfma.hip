// sdh_w0mfma.hip — prefill: the base term as a dense contraction on the matrix cores.
//
//   Y0[t, o] = sum_i W0[o, i] * x[t, i],   W0[o, i] = (d3 * G_b[(p ^ inv_b[q]) & (L-1)]) * d1
//
// W0 is never stored: every wave regenerates the 32 x 16 fragment of W0 it needs, in registers,
// from the per-block table G_b = FWHT(d2_b) and the inverse permutation (integer XOR indexing, the
// closed form of sdh_w0_tile — bit-exact in fp32), rounds it to f16 and feeds it to
// v_mfma_f32_32x32x16_f16 as the B operand; the A operand is a 32-token x 16 fragment of x read
// straight from global memory. Accumulation is fp32. Operands are swapped (tokens on the rows of C)
// so that a lane's accumulator column is an output row o and the stores coalesce.
//
// This path exists because the north star asks for it; SDH_FWD_W0_MFMA selects it. Measured against
// the FWHT form it is not the faster one for the shapes of this model family:
// 2*n_in*n_out*T flops here versus T*B*(2 L log2 L + 3L) there (60 GFLOP vs 0.23 GFLOP for a
// Mistral gate at T=512), and f16 operands cost ~1e-3 relative accuracy where the FWHT form is exact.
#include "sdh_internal.h"
#include <type_traits>

namespace sdh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// workgroup: 4 waves; wave w owns output rows [o_blk + 32w, +32) for 128 tokens (4 MFMA tiles)
template <bool TALL, bool VECX>
__global__ __launch_bounds__(256)
void w0_mfma_kernel(const float * __restrict__ x, int64_t ldx, int64_t T, int n_in, int n_out, int L,
                    const float * __restrict__ d1, const float * __restrict__ d3,
                    const float * __restrict__ G, const int * __restrict__ inv,
                    float * __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int o = (int) blockIdx.x * 128 + wave * 32 + r;
    const bool o_ok = o < n_out;
    const int64_t t_blk = (int64_t) blockIdx.y * 128;
    const int Lm = L - 1;
    // row constants (tall: the row fixes the base block)
    const int bt = TALL ? (o_ok ? o / L : 0) : 0;
    const int p = TALL ? (o - bt * L) : o;
    const float d3_tall = (TALL && d3 && o_ok) ? d3[(size_t) bt * L + p] : 1.0f;

    floatx16 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.0f;

    for (int k0 = 0; k0 < n_in; k0 += 16) {
        const int kk = k0 + 8 * h;
        half8 wf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = kk + j;
            float wv = 0.0f;
            if (o_ok && i < n_in) {
                const int b = TALL ? bt : i / L;
                const int q = TALL ? i : i - b * L;
                const size_t pb = (size_t) b * L;
                const float g = G[pb + ((p ^ inv[pb + q]) & Lm)];
                const float s3 = TALL ? d3_tall : (d3 ? d3[pb + p] : 1.0f);
                wv = (s3 * g) * d1[pb + q];
            }
            wf[j] = (_Float16) wv;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t t = t_blk + 32 * s + r;
            half8 xf;
            if (VECX && t < T && kk + 8 <= n_in) {
                const float4 a = *reinterpret_cast<const float4 *>(x + t * ldx + kk);
                const float4 c = *reinterpret_cast<const float4 *>(x + t * ldx + kk + 4);
                xf[0] = (_Float16) a.x; xf[1] = (_Float16) a.y; xf[2] = (_Float16) a.z; xf[3] = (_Float16) a.w;
                xf[4] = (_Float16) c.x; xf[5] = (_Float16) c.y; xf[6] = (_Float16) c.z; xf[7] = (_Float16) c.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = (_Float16) ((t < T && kk + j < n_in) ? x[t * ldx + kk + j] : 0.0f);
            }
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, wf, acc[s], 0, 0, 0);
        }
    }
    // C[row = token][col = o]: row = (reg & 3) + 8 * (reg >> 2) + 4 * h, col = r
    if (o_ok) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t t = t_blk + 32 * s + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (t < T) out[t * ldo + o] = acc[s][e];
            }
    }
}

// ---- the tiled form: W0 tiles built once per workgroup in LDS ---------------------------------------------
//
// GEMM view: C[t, o] = sum_k X16[t, k] * W0[o, k] with M = tokens, N = output rows, K = the inputs of one base
// block (tall: n_in; wide: one block b of L inputs per blockIdx.z, the B partial products land in the
// [T][B][L] layout the residual kernels already sum in block order).
//
//   workgroup  512 threads = 8 waves; macro tile 128 rows x 256 tokens x 64 k per slab, double-buffered in LDS:
//              Ws[2][128][72] f16 (W0 tile, built here), Xs[2][256][72] f16 (x tile, copied from the f16 image of x
//              the pre-pass makes). Rows are 144 bytes apart: the 16 lanes of a ds_read_b128 pass then fall on 16
//              distinct 4-bank groups (r * 36 mod 64 runs through the multiples of 4) — conflict-free.
//   wave       64 rows x 64 tokens = 2 x 2 tiles of v_mfma_f32_32x32x16_f16, 64 accumulator registers; per 16-k
//              step 4 ds_read_b128 feed 4 MFMAs (the guide's free budget is 2 reads per MFMA).
//   W0 tile    wave w builds columns 8w..8w+7 of the slab for all 128 rows. For one column the 128 rows of the tile
//              read G_b[(p0 + r) ^ inv] = an aligned 128-float segment of G_b permuted by XOR with (inv & 127):
//              coalesced 256-byte gathers with a wave-uniform base (inv, d1 are scalar loads). (d3 * g) * d1 is
//              rounded to f16 once, 8 columns go out as one ds_write_b128 per row.
//   pipeline   global loads (G gathers, x16 lines) for slab s+2 are issued at the top of slab s into one of two
//              register sets; the set loaded a slab earlier is turned into the LDS image of slab s+1, in four
//              parts placed between the four MFMA groups of slab s, so that the conversion work issues while the
//              matrix pipe is busy; one barrier per slab.
// Everything outside the tile (rows >= n_out, k >= the block's inputs, tokens >= T) is generated as zero; the
// f16 image of x is zero-padded to whole tiles, so the main loop has no bounds checks on x.
constexpr int MT_R = 128, MT_N = 256, MT_K = 64, MT_LD = 72;

__global__ __launch_bounds__(256)
void w0_x16_kernel(const float * __restrict__ x, int64_t ldx, int64_t T, int n_in, _Float16 * __restrict__ x16, int64_t ld16, int64_t T_pad) {
    // blockIdx.y = token, threads over k in runs of 8 (no 64-bit division per thread; rows of x are 16-byte aligned when ldx % 4 == 0)
    const int64_t t = blockIdx.y;
    const int k = ((int) blockIdx.x * 256 + (int) threadIdx.x) * 8;
    if (k >= ld16) return;
    half8 h;
    const float * xr = x + t * ldx + k;
    if (t < T && k + 8 <= n_in && (((uintptr_t) xr) & 15) == 0) {
        const float4 a = *reinterpret_cast<const float4 *>(xr), c = *reinterpret_cast<const float4 *>(xr + 4);
        h[0] = (_Float16) a.x; h[1] = (_Float16) a.y; h[2] = (_Float16) a.z; h[3] = (_Float16) a.w;
        h[4] = (_Float16) c.x; h[5] = (_Float16) c.y; h[6] = (_Float16) c.z; h[7] = (_Float16) c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16) ((t < T && k + j < n_in) ? xr[j] : 0.0f);
    }
    *reinterpret_cast<half8 *>(x16 + t * ld16 + k) = h;
}

// f16 image of x, zero-padded to whole tiles (also the pre-pass of the folded contraction, sdh_w0fold.hip)
int launch_w0_x16(const float * x, int64_t ldx, int64_t T, int n_in, _Float16 * x16, int64_t ld16, int64_t T_pad, hipStream_t s) {
    if (T_pad > 65535) { set_error("w0 x16: token chunk too large"); return SDH_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(w0_x16_kernel, dim3((unsigned) ((ld16 / 8 + 255) / 256), (unsigned) T_pad), dim3(256), 0, s, x, ldx, T, n_in, x16, ld16, T_pad);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <bool TALL, bool SCALED /* d1 or d3 present and not all ones */>
__global__ __launch_bounds__(512)
void w0_mfma_tile_kernel(const _Float16 * __restrict__ x16, int64_t ld16, int64_t T, int n_in, int n_out, int L,
                         const float * __restrict__ d1, const float * __restrict__ d3,
                         const float * __restrict__ G, const int * __restrict__ inv,
                         float * __restrict__ out, int64_t ldo) {
    extern __shared__ __align__(16) unsigned char mt_smem[];
    _Float16 * Ws = (_Float16 *) mt_smem;                         // [2][MT_R][MT_LD]
    _Float16 * Xs = Ws + 2 * MT_R * MT_LD;                        // [2][MT_N][MT_LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int o_tile = (int) blockIdx.x * MT_R;
    const int64_t t_tile = (int64_t) blockIdx.y * MT_N;
    const int b = TALL ? o_tile / L : (int) blockIdx.z;
    const size_t pb = (size_t) b * L;
    const int p0 = TALL ? o_tile - b * L : o_tile;                // first row of the tile inside its block
    const int k_begin = TALL ? 0 : b * L;                         // first input of this workgroup's K range
    const int q_len = TALL ? n_in : min(L, n_in - b * L);
    const int n_slab = (q_len + MT_K - 1) / MT_K;

    // rows this lane builds: lane and lane + 64 of the tile. Byte offsets into G_b: (row ^ inv) * 4 = row*4 ^ inv*4,
    // in range because both are below L; a row outside the tensor gathers from row 0 and is written as zero.
    float d3r[2];
    bool rok[2];
    unsigned rowoff[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = lane + 64 * rr;
        rok[rr] = o_tile + row < n_out && p0 + row < L;
        rowoff[rr] = rok[rr] ? (unsigned) (p0 + row) << 2 : 0u;
        d3r[rr] = (SCALED && rok[rr] && d3) ? d3[pb + p0 + row] : 1.0f;
    }
    const char * Gb = (const char *) (G + pb);
    // x16 line of this thread inside a 64-token group: 8 lanes per 128-byte line
    const unsigned xoff = (unsigned) ((tid >> 3) * ld16 + (tid & 7) * 8);
    const unsigned ws_w = (unsigned) (lane * MT_LD + wave * 8);                  // W0 tile element this lane writes (row half 0)
    const unsigned xs_w = (unsigned) ((tid >> 3) * MT_LD + (tid & 7) * 8);       // x tile element (token group 0)

    // register sets of the two slabs in flight
    float gq[2][2][8];          // [set][row half][column]
    float d1q[2][8];
    unsigned ivq[8];            // inverse permutation (x4) of the columns whose gathers are issued next
    half8 xq[2][4];
    // the gathers of a slab depend on its inv entries: those are loaded one slab earlier still (fetch_idx for slab
    // s+3 at the top of slab s, right after the gathers of slab s+2 have used the previous ones), so that no wave
    // sits out a dependent load chain before its MFMAs
    // One vector load (lane j & 7 reads entry j; unconditional: a slab's 64 columns lie inside the block's L table
    // entries even where they are past q_len — those columns are blanked through d1q — and a slab index past the end
    // is clamped). Not scalar loads: eight conditional one-word s_loads cost a round trip each (~1600 cycles per slab),
    // and any scalar load in flight turns every LDS wait of the wave into lgkmcnt(0).
    int ivl = 0;
    auto fetch_idx = [&](int s) {
        const int q0 = min(s, n_slab - 1) * MT_K + wave * 8;      // this wave's 8 columns of the slab
        ivl = inv[pb + q0 + (lane & 7)];
    };
    auto fetch = [&](auto set_c, int s) {
        constexpr int S = decltype(set_c)::value;
        const int q0 = s * MT_K + wave * 8;
        float dl = 1.0f;
        if constexpr (SCALED) dl = d1[pb + q0 + (lane & 7)];
#pragma unroll
        for (int j = 0; j < 8; ++j) ivq[j] = (unsigned) __builtin_amdgcn_readlane(ivl, j) << 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // a zero d1 blanks a column past the end of the block's inputs
            d1q[S][j] = q0 + j < q_len ? (SCALED ? __shfl(dl, j, 64) : 1.0f) : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) gq[S][rr][j] = *reinterpret_cast<const float *>(Gb + (rowoff[rr] ^ ivq[j]));
        }
        const _Float16 * xs0 = x16 + t_tile * ld16 + k_begin + s * MT_K;         // uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) xq[S][i] = *reinterpret_cast<const half8 *>(xs0 + (size_t) i * 64 * ld16 + xoff);
    };
    // part (0..3) of turning register set S into the LDS image `buf`
    auto commit_part = [&](auto set_c, int buf, auto part_c) {
        constexpr int S = decltype(set_c)::value;
        constexpr int part = decltype(part_c)::value;
        if constexpr (part < 2) {
            constexpr int rr = part;
            half8 w8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = SCALED ? (d3r[rr] * gq[S][rr][j]) * d1q[S][j] : gq[S][rr][j] * d1q[S][j];
                w8[j] = (_Float16) (rok[rr] ? v : 0.0f);
            }
            *reinterpret_cast<half8 *>(Ws + (size_t) buf * MT_R * MT_LD + ws_w + rr * 64 * MT_LD) = w8;
        } else {
#pragma unroll
            for (int i = 2 * (part - 2); i < 2 * (part - 2) + 2; ++i)
                *reinterpret_cast<half8 *>(Xs + (size_t) buf * MT_N * MT_LD + xs_w + i * 64 * MT_LD) = xq[S][i];
        }
    };

    const int wr = wave & 1, wt = wave >> 1;                      // wave tile: rows wr*64.., tokens wt*64..
    floatx16 acc[2][2];                                           // [token tile][row tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.0f;
    const unsigned wfrag = (unsigned) ((wr * 64 + r) * MT_LD + 8 * h), xfrag = (unsigned) ((wt * 64 + r) * MT_LD + 8 * h);

    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    auto slab = [&](auto cur_c, auto nxt_c, int s) {
        // cur_c: parity of slab s = the register set slab s+2 is fetched into; nxt_c holds slab s+1
        constexpr int buf = decltype(cur_c)::value;
        if (s + 2 < n_slab) fetch(cur_c, s + 2);
        fetch_idx(s + 3);
        const _Float16 * Wb = Ws + (size_t) buf * MT_R * MT_LD + wfrag;
        const _Float16 * Xb = Xs + (size_t) buf * MT_N * MT_LD + xfrag;
        // fragments of step ks+1 are requested before the MFMAs of step ks are issued
        half8 xf[2][2], wf[2][2];
        auto frags = [&](int q, int ks) {
#pragma unroll
            for (int a = 0; a < 2; ++a) xf[q][a] = *reinterpret_cast<const half8 *>(Xb + a * 32 * MT_LD + ks * 16);
#pragma unroll
            for (int c = 0; c < 2; ++c) wf[q][c] = *reinterpret_cast<const half8 *>(Wb + c * 32 * MT_LD + ks * 16);
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) frags((ks + 1) & 1, ks + 1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[ks & 1][a], wf[ks & 1][c], acc[a][c], 0, 0, 0);
            if (s + 1 < n_slab) {
                if (ks == 0) commit_part(nxt_c, buf ^ 1, std::integral_constant<int, 0>{});
                if (ks == 1) commit_part(nxt_c, buf ^ 1, std::integral_constant<int, 1>{});
                if (ks == 2) commit_part(nxt_c, buf ^ 1, std::integral_constant<int, 2>{});
                if (ks == 3) commit_part(nxt_c, buf ^ 1, std::integral_constant<int, 3>{});
            }
        }
        __syncthreads();
    };

    fetch_idx(0);
    fetch(set0{}, 0);
    commit_part(set0{}, 0, std::integral_constant<int, 0>{});
    commit_part(set0{}, 0, std::integral_constant<int, 1>{});
    commit_part(set0{}, 0, std::integral_constant<int, 2>{});
    commit_part(set0{}, 0, std::integral_constant<int, 3>{});
    fetch_idx(1);
    if (1 < n_slab) fetch(set1{}, 1);
    fetch_idx(2);
    __syncthreads();
    for (int s = 0; s < n_slab; s += 2) {
        slab(set0{}, set1{}, s);
        if (s + 1 < n_slab) slab(set1{}, set0{}, s + 1);
    }

    // C[row = token][col = o]: row = (e & 3) + 8 * (e >> 2) + 4 * h, col = r
    const size_t col_off = TALL ? 0 : pb;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int orow = o_tile + wr * 64 + c * 32 + r;
        if (orow >= n_out) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t t = t_tile + wt * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (t < T) out[t * ldo + col_off + orow] = acc[a][c][e];
            }
    }
}

size_t w0_mfma_x16_bytes(const DeviceWeight & w, int64_t T) {
    const int64_t T_pad = (T + MT_N - 1) / MT_N * MT_N;
    const int64_t ld16 = (w.n_in + MT_K - 1) / MT_K * MT_K + MT_K;
    return (size_t) T_pad * (size_t) ld16 * 2;
}

// Tiled form. tall: Y0 goes straight to `out` = y (ldo = ldy), the residual kernel then updates y in place;
// wide: partial products per base block go to `out` = the [T][B][L] workspace (ldo = B*L). x16_ws: T_pad x ld16
// f16 words of workspace (w0_mfma_x16_bytes). Returns 1 when the payload does not fit the tile scheme.
int launch_w0_mfma_tiled(const DeviceWeight & w, const float * x, int64_t ldx, int64_t T, _Float16 * x16_ws,
                         float * out, int64_t ldo, hipStream_t s) {
    if (!w.w0_closed_form || w.L < MT_R || (w.L % MT_R) != 0) return 1;
    if (T <= 0) return SDH_OK;
    const int64_t T_pad = (T + MT_N - 1) / MT_N * MT_N;
    const int64_t ld16 = (w.n_in + MT_K - 1) / MT_K * MT_K + MT_K;
    { int rc = launch_w0_x16(x, ldx, T, (int) w.n_in, x16_ws, ld16, T_pad, s); if (rc != SDH_OK) return rc; }
    const size_t lds = (size_t) 2 * (MT_R + MT_N) * MT_LD * 2;
    static const bool attr_ok =
        hipFuncSetAttribute((const void *) w0_mfma_tile_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
        hipFuncSetAttribute((const void *) w0_mfma_tile_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
        hipFuncSetAttribute((const void *) w0_mfma_tile_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
        hipFuncSetAttribute((const void *) w0_mfma_tile_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (!attr_ok) { set_error("w0 mfma: raising the dynamic LDS limit failed"); return SDH_ERR_HIP; }
    const dim3 grid((unsigned) ((w.n_out + MT_R - 1) / MT_R), (unsigned) (T_pad / MT_N), (unsigned) (w.tall ? 1 : w.B));
    if (grid.y > 65535 || grid.z > 65535) { set_error("w0 mfma: token chunk too large"); return SDH_ERR_UNSUPPORTED; }
    const bool scaled = w.pack.off_d1 >= 0 || w.pack.off_d3 >= 0;          // else d1 = d3 = 1 (what the builder writes)
#define SDH_MT_LAUNCH(TALL_, SC_) hipLaunchKernelGGL((w0_mfma_tile_kernel<TALL_, SC_>), grid, dim3(512), lds, s, x16_ws, ld16, T, \
        (int) w.n_in, (int) w.n_out, (int) w.L, w.d1, w.d3, w.G, w.inv, out, ldo)
    if (w.tall) { if (scaled) SDH_MT_LAUNCH(true, true); else SDH_MT_LAUNCH(true, false); }
    else        { if (scaled) SDH_MT_LAUNCH(false, true); else SDH_MT_LAUNCH(false, false); }
#undef SDH_MT_LAUNCH
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

// v_out: tall -> [T][B*L] with Y0 at column o; wide -> [T][L] (all base blocks already summed)
int launch_w0_mfma(const DeviceWeight & w, const float * x, int64_t ldx, int64_t T, float * v_out, hipStream_t s) {
    if (!w.w0_closed_form) return 1;
    if (T <= 0) return SDH_OK;
    const dim3 grid((unsigned) ((w.n_out + 127) / 128), (unsigned) ((T + 127) / 128));
    if (grid.y > 65535) { set_error("w0 mfma: token chunk too large"); return SDH_ERR_UNSUPPORTED; }
    const bool vecx = ((uintptr_t) x % 16) == 0 && (ldx % 4) == 0;
    const int64_t ldo = w.tall ? w.B * w.L : w.L;
    if (w.tall) {
        if (vecx) hipLaunchKernelGGL((w0_mfma_kernel<true, true>), grid, dim3(256), 0, s, x, ldx, T, (int) w.n_in, (int) w.n_out, (int) w.L, w.d1, w.d3, w.G, w.inv, v_out, ldo);
        else      hipLaunchKernelGGL((w0_mfma_kernel<true, false>), grid, dim3(256), 0, s, x, ldx, T, (int) w.n_in, (int) w.n_out, (int) w.L, w.d1, w.d3, w.G, w.inv, v_out, ldo);
    } else {
        if (vecx) hipLaunchKernelGGL((w0_mfma_kernel<false, true>), grid, dim3(256), 0, s, x, ldx, T, (int) w.n_in, (int) w.n_out, (int) w.L, w.d1, w.d3, w.G, w.inv, v_out, ldo);
        else      hipLaunchKernelGGL((w0_mfma_kernel<false, false>), grid, dim3(256), 0, s, x, ldx, T, (int) w.n_in, (int) w.n_out, (int) w.L, w.d1, w.d3, w.G, w.inv, v_out, ldo);
    }
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

} // namespace sdh
