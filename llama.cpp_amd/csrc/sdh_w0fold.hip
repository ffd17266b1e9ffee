// sdh_w0fold.hip — prefill as ONE dense contraction per tensor: the residual folded into the W0 tile.
//
//   y[t, o] = rs[o] * sum_i (W0[o, i] + D[o, i]) * x[t, i]
//
// The reference evaluates the base term and the residual side by side for every output (src/llama-seeddelta.cpp
// :1088-1127 tall, :1128-1175 wide). For many tokens the residual's per-row gathers are what costs (0.94 G scattered
// LDS reads per ubatch of a Mistral gate); the dense product W0.x is a real GEMM. So this kernel builds the W0 tile of
// a k-slab in LDS exactly as w0_mfma_tile_kernel does (closed form G_b[(p ^ inv_b[q])], bit-exact in fp32, rounded to
// f16 once) and then ADDS the residual entries that fall into that (row tile, slab) to it, in LDS, before the matrix
// cores read it: the residual becomes part of the B operand and costs one LDS atomic per entry and workgroup instead
// of one gather-MAC per entry and token. The payload is re-laid out once per handle, on the device, for that: entries
// bucketed by (128-row tile, 64-wide k-slab), each a 32-bit word {f16 value, row in tile, column in slab}; block
// payloads are expanded to entries, invalid indices dropped (reference skip rules :693, :929-936).
//
// Numerics: f16 operands (x, W0 + D), fp32 accumulation, row_scale in fp32 in the epilogue — the tolerance of the
// MFMA base path (tests state it), not the bit-exact order of the SpMM / FWHT forms, which stay available.
//
//   workgroup  512 threads = 8 waves; tile 128 rows x 256 tokens x 64 k per slab.
//   LDS        Ws[3][128][72] f16 (three W tiles in rotation: slab s is multiplied while the residual of slab s+1 is
//              added to the tile written one slab earlier and the W0 tile of slab s+2 is written — one barrier per
//              slab), Xs[2][256][72] f16. 129 KB.
//   adds       ds_pk_add_f16 (the pair that holds the element, the other half +0): no read-modify-write sequence,
//              and duplicate (row, column) entries of a hand-made payload sum as the reference sums them.
//   tried      sixteen waves of 32 rows x 64 tokens per workgroup (four waves per SIMD, 128 registers) instead of eight of
//              64 x 64: the same time within 1 % (136.7 / 135.4 us per call on a Mistral gate) — the kernel is not short
//              of independent chains; SQ counters (profiles/r03_e_pmc_fold_and_scan.json): MFMA busy 0.22, VALU 0.29,
//              LDS 0.22, waves waiting 0.41 of their cycles, no pipe saturated. Per slab and CU the texture path
//              moves 64 KB, the LDS 176 KB, the matrix pipe runs 128 MFMAs: ~0.5 + 0.7 + 0.5 us if nothing overlapped,
//              1.9 us measured.
//   wide       K is split by base block over blockIdx.z (as the tiled W0 kernel); partial products go to [T][B][L]
//              and fold_reduce_kernel sums them in block order, applies row_scale and the non-finite census.
#include "sdh_internal.h"
#include <type_traits>

namespace sdh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int FD_R = 128, FD_N = 256, FD_K = 64, FD_LD = 72;

// ---- payload re-layout: entries bucketed by (row tile, k-slab) ------------------------------------------------
struct FoldArgs {
    const void * idx; const void * val; int idx32, val32;
    int n_in, n_out, K, nb, block, is_block;
    int L, tall, n_g, spb;             // n_g: slabs per offsets row; spb: slabs per base block (wide)
};

__device__ inline int fold_slab_of(const FoldArgs & a, int k) {
    if (a.tall) return k / FD_K;
    const int b = k / a.L;
    return b * a.spb + (k - b * a.L) / FD_K;
}

// pass 0: count, pass 1: place (cursor starts at the bucket's offset)
template <int PASS>
__global__ __launch_bounds__(256)
void fold_bucket_kernel(const FoldArgs a, int * __restrict__ counts, int * __restrict__ cursor, unsigned * __restrict__ ent) {
    const int64_t per_row = a.is_block ? a.nb : a.K;
    const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t) a.n_out * per_row) return;
    const int o = (int) (e / per_row), r = (int) (e - (int64_t) o * per_row);
    const int tile = o / FD_R, row = o - tile * FD_R;
    const int ii = a.idx32 ? ((const int *) a.idx)[e] : (int) ((const short *) a.idx)[e];
    if (!a.is_block) {
        if (ii < 0 || ii >= a.n_in) return;                                        // :693
        const int g = fold_slab_of(a, ii);
        int * slot = (PASS == 0 ? counts : cursor) + (size_t) tile * (a.n_g + 1) + g;
        if (PASS == 0) { atomicAdd(slot, 1); return; }
        const int pos = atomicAdd(slot, 1);
        const _Float16 v = a.val32 ? (_Float16) ((const float *) a.val)[e] : ((const _Float16 *) a.val)[e];
        unsigned short vb; __builtin_memcpy(&vb, &v, 2);
        const int col = a.tall ? (ii & (FD_K - 1)) : ((ii % a.L) & (FD_K - 1));
        ent[pos] = ((unsigned) vb << 16) | (unsigned) (row << 6 | col);
    } else {
        if (ii < 0) return;                                                        // :929-936
        const int64_t in0 = (int64_t) ii * a.block;
        if (in0 >= a.n_in) return;
        const int in1 = (int) min((int64_t) a.n_in, in0 + a.block);
        for (int k = (int) in0; k < in1; ++k) {
            const int g = fold_slab_of(a, k);
            int * slot = (PASS == 0 ? counts : cursor) + (size_t) tile * (a.n_g + 1) + g;
            if (PASS == 0) { atomicAdd(slot, 1); continue; }
            const int pos = atomicAdd(slot, 1);
            const int64_t vi = ((int64_t) o * a.nb + r) * a.block + (k - in0);
            const _Float16 v = a.val32 ? (_Float16) ((const float *) a.val)[vi] : ((const _Float16 *) a.val)[vi];
            unsigned short vb; __builtin_memcpy(&vb, &v, 2);
            const int col = a.tall ? (k & (FD_K - 1)) : ((k % a.L) & (FD_K - 1));
            ent[pos] = ((unsigned) vb << 16) | (unsigned) (row << 6 | col);
        }
    }
}

// counts [tiles][n_g + 1] -> bucket bounds: every bucket starts at a multiple of 4 entries (16-byte loads) and is padded
// to one with null entries (value +0: skipped by the kernel); cursor = the bucket starts, bounds[tile * n_g + g] = {begin, end}
__global__ __launch_bounds__(1024)
void fold_scan_kernel(const int * __restrict__ counts, int * __restrict__ cursor, int2 * __restrict__ bounds, int tiles, int n_g) {
    __shared__ int tile_base[1];
    __shared__ int totals[1024];
    int carry = 0;
    for (int t0 = 0; t0 < tiles; t0 += 1024) {
        const int t = t0 + (int) threadIdx.x;
        int tot = 0;
        if (t < tiles) for (int g = 0; g < n_g; ++g) tot += (counts[(size_t) t * (n_g + 1) + g] + 3) & ~3;
        totals[threadIdx.x] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = carry;
            for (int i = 0; i < 1024; ++i) { const int v = totals[i]; totals[i] = run; run += v; }
            tile_base[0] = run;
        }
        __syncthreads();
        if (t < tiles) {
            int run = totals[threadIdx.x];
            for (int g = 0; g < n_g; ++g) {
                const size_t at = (size_t) t * (n_g + 1) + g;
                const int v = (counts[at] + 3) & ~3;
                cursor[at] = run;
                bounds[(size_t) t * n_g + g] = make_int2(run, run + v);
                run += v;
            }
        }
        carry = tile_base[0];
        __syncthreads();
    }
}

// does any bucket hold the same (row, column) twice? one workgroup per bucket, a bitmap of the 8192 positions of a tile
__global__ __launch_bounds__(256)
void fold_dup_kernel(const unsigned * __restrict__ ent, const int2 * __restrict__ bounds, int * __restrict__ dup) {
    __shared__ unsigned bm[256];
    bm[threadIdx.x] = 0u;
    __syncthreads();
    const int2 o = bounds[blockIdx.x];
    bool hit = false;
    for (int e = o.x + (int) threadIdx.x; e < o.y; e += 256) {
        const unsigned wd = ent[e];
        if ((wd & 0x7FFF0000u) == 0u) continue;                                  // padding, or a value of zero
        const unsigned key = wd & 0x1FFFu, bit = 1u << (key & 31u);
        hit |= (atomicOr(&bm[key >> 5], bit) & bit) != 0u;
    }
    if (hit) atomicOr(dup, 1);
}

int fold_ensure(DeviceWeight & w, hipStream_t s) {
    if (w.fold_ent) return SDH_OK;
    if (!w.w0_closed_form || w.L < FD_R || (w.L % FD_R) != 0) return 1;
    const bool is_block = w.kind == SDH_OP_BASE_BLOCK;
    FoldArgs a{};
    a.idx = w.idx; a.val = w.val; a.idx32 = w.idx_type == SDH_TYPE_I32; a.val32 = w.val_type == SDH_TYPE_F32;
    a.n_in = (int) w.n_in; a.n_out = (int) w.n_out; a.K = (int) w.K; a.nb = (int) w.nb; a.block = (int) w.block; a.is_block = is_block;
    a.L = (int) w.L; a.tall = w.tall ? 1 : 0;
    a.spb = (int) (w.L / FD_K);
    a.n_g = w.tall ? (int) ((w.n_in + FD_K - 1) / FD_K) : (int) (w.B * a.spb);          // wide: a row of buckets per base block the grid has
    const int tiles = (int) ((w.n_out + FD_R - 1) / FD_R);
    const int64_t per_row = is_block ? w.nb : w.K;
    const int64_t cap = (int64_t) w.n_out * (is_block ? w.nb * w.block : w.K);
    if (cap <= 0 || cap > 0x7fffffff) return 1;
    const size_t off_bytes = (size_t) tiles * (a.n_g + 1) * sizeof(int);
    const size_t ent_words = (size_t) cap + (size_t) 4 * tiles * a.n_g + 4;       // every bucket padded to a multiple of 4
    int * counts = nullptr, * cursor = nullptr; unsigned * ent = nullptr; int2 * bounds = nullptr;
    SDH_HIP_CHECK(hipMalloc((void **) &counts, off_bytes));
    SDH_HIP_CHECK(hipMalloc((void **) &cursor, off_bytes));
    SDH_HIP_CHECK(hipMalloc((void **) &bounds, (size_t) tiles * a.n_g * sizeof(int2)));
    SDH_HIP_CHECK(hipMalloc((void **) &ent, ent_words * sizeof(unsigned)));
    SDH_HIP_CHECK(hipMemsetAsync(counts, 0, off_bytes, s));
    SDH_HIP_CHECK(hipMemsetAsync(ent, 0, ent_words * sizeof(unsigned), s));       // padding = null entries
    const unsigned grid = (unsigned) ((w.n_out * per_row + 255) / 256);
    hipLaunchKernelGGL(fold_bucket_kernel<0>, dim3(grid), dim3(256), 0, s, a, counts, cursor, ent);
    hipLaunchKernelGGL(fold_scan_kernel, dim3(1), dim3(1024), 0, s, counts, cursor, bounds, tiles, a.n_g);
    hipLaunchKernelGGL(fold_bucket_kernel<1>, dim3(grid), dim3(256), 0, s, a, counts, cursor, ent);
    SDH_HIP_CHECK(hipGetLastError());
    // counts is free again: its first word receives the "a (row, input) pair repeats" flag
    SDH_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int), s));
    hipLaunchKernelGGL(fold_dup_kernel, dim3((unsigned) (tiles * a.n_g)), dim3(256), 0, s, ent, bounds, counts);
    SDH_HIP_CHECK(hipGetLastError());
    int dup = 0;
    SDH_HIP_CHECK(hipMemcpyAsync(&dup, counts, sizeof(int), hipMemcpyDeviceToHost, s));
    SDH_HIP_CHECK(hipStreamSynchronize(s));
    SDH_HIP_CHECK(hipFree(cursor));
    SDH_HIP_CHECK(hipFree(counts));
    int2 * off = bounds;
    w.fold_dup = dup != 0;
    w.fold_ent = ent; w.fold_off = off; w.fold_ng = a.n_g; w.fold_spb = a.spb;
    return SDH_OK;
}

void fold_free(DeviceWeight & w) {
    (void) hipFree(w.fold_ent); (void) hipFree(w.fold_off);
    w.fold_ent = nullptr; w.fold_off = nullptr;
}

// ---- the contraction ---------------------------------------------------------------------------------------------
template <bool TALL, bool SCALED /* d1 or d3 present and not all ones */, bool ATOMIC /* the payload repeats a (row, input) pair */>
__global__ __launch_bounds__(512)
void w0d_mfma_kernel(const _Float16 * __restrict__ x16, int64_t ld16, int64_t T, int n_in, int n_out, int L,
                     const float * __restrict__ d1, const float * __restrict__ d3,
                     const float * __restrict__ G, const int * __restrict__ inv,
                     const unsigned * __restrict__ ent, const int2 * __restrict__ off, int n_g, int spb,
                     const float * __restrict__ rs, float * __restrict__ out, int64_t ldo,
                     int * __restrict__ nf_counter, int zero_nf) {
    extern __shared__ __align__(16) unsigned char fd_smem[];
    _Float16 * Ws = (_Float16 *) fd_smem;                         // [3][FD_R][FD_LD]
    _Float16 * Xs = Ws + 3 * FD_R * FD_LD;                        // [2][FD_N][FD_LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int o_tile = (int) blockIdx.x * FD_R;
    const int64_t t_tile = (int64_t) blockIdx.y * FD_N;
    const int b = TALL ? o_tile / L : (int) blockIdx.z;
    const size_t pb = (size_t) b * L;
    const int p0 = TALL ? o_tile - b * L : o_tile;                // first row of the tile inside its block
    const int k_begin = TALL ? 0 : b * L;                         // first input of this workgroup's K range
    const int q_len = TALL ? n_in : min(L, n_in - b * L);
    const int n_slab = (q_len + FD_K - 1) / FD_K;
    const int2 * offr = off + (size_t) blockIdx.x * n_g + (TALL ? 0 : b * spb);             // bounds of bucket s of this workgroup: offr[s]

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
    const unsigned xoff = (unsigned) ((tid >> 3) * ld16 + (tid & 7) * 8);
    const unsigned ws_w = (unsigned) (lane * FD_LD + wave * 8);                  // W0 tile element this lane writes (row half 0)
    const unsigned xs_w = (unsigned) ((tid >> 3) * FD_LD + (tid & 7) * 8);       // x tile element (token group 0)

    float gq[2][2][8];          // [set][row half][column]
    float d1q[2][8];
    unsigned ivq[8];
    half8 xq[2][4];
    uint4 eq[2];                // residual entries (four per thread) of the slab after next / the next slab
    int ebeg[2], eend[2];       // their bucket bounds
    int ivl = 0;
    int2 obe = make_int2(0, 0); // bucket bounds loaded a slab ahead of the entry loads that use them
    auto fetch_idx = [&](int s) {
        const int q0 = max(min(s, n_slab - 1), 0) * FD_K + wave * 8;
        ivl = inv[pb + q0 + (lane & 7)];
    };
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));                 // a zero the compiler cannot see: keeps the loads below vector loads
    auto fetch_off = [&](int s) {                                  // (a scalar load in flight turns every LDS wait into lgkmcnt(0))
        const int sc = max(min(s, n_slab - 1), 0);
        obe = offr[sc + vzero];
    };
    auto fetchW = [&](auto set_c, int s) {
        constexpr int S = decltype(set_c)::value;
        const int q0 = s * FD_K + wave * 8;
        float dl = 1.0f;
        if constexpr (SCALED) dl = d1[pb + q0 + (lane & 7)];
#pragma unroll
        for (int j = 0; j < 8; ++j) ivq[j] = (unsigned) __builtin_amdgcn_readlane(ivl, j) << 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            d1q[S][j] = q0 + j < q_len ? (SCALED ? __shfl(dl, j, 64) : 1.0f) : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) gq[S][rr][j] = *reinterpret_cast<const float *>(Gb + (rowoff[rr] ^ ivq[j]));
        }
    };
    auto fetchX = [&](auto set_c, int s) {
        constexpr int S = decltype(set_c)::value;
        const _Float16 * xs0 = x16 + t_tile * ld16 + k_begin + s * FD_K;         // uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) xq[S][i] = *reinterpret_cast<const half8 *>(xs0 + (size_t) i * 64 * ld16 + xoff);
    };
    auto fetchE = [&](auto set_c) {                                // entries of the bucket [ob, oe)
        constexpr int S = decltype(set_c)::value;
        ebeg[S] = __builtin_amdgcn_readfirstlane(obe.x); eend[S] = __builtin_amdgcn_readfirstlane(obe.y);
        const int e = ebeg[S] + 4 * tid;                           // buckets start and end at multiples of 4: one 16-byte load
        eq[S] = *reinterpret_cast<const uint4 *>(ent + (e < eend[S] ? e : max(eend[S] - 4, 0)));      // clamped: read unconditionally
    };
    auto commitW = [&](auto set_c, unsigned wbuf, int rr) {
        _Float16 * Wb = Ws + wbuf;
        constexpr int S = decltype(set_c)::value;
        half8 w8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = SCALED ? (d3r[rr] * gq[S][rr][j]) * d1q[S][j] : gq[S][rr][j] * d1q[S][j];
            w8[j] = (_Float16) (rok[rr] ? v : 0.0f);
        }
        *reinterpret_cast<half8 *>(Wb + ws_w + rr * 64 * FD_LD) = w8;
    };
    auto commitX = [&](auto set_c, int buf, int half) {
        constexpr int S = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<half8 *>(Xs + (size_t) buf * FD_N * FD_LD + xs_w + (2 * half + i) * 64 * FD_LD) = xq[S][2 * half + i];
    };
    // Adding the entries of a bucket to a W tile. Entries of a payload without repeated (row, input) pairs touch distinct
    // elements, so each is a plain 16-bit read, add, 16-bit write — the reads of a slab's entries are issued at the first
    // multiply step and the writes two steps later, so no wave waits out an LDS round trip between its MFMAs (a read-add-
    // write per entry in one place cost 0.3 us per slab: three dependent round trips with the matrix pipe idle). Payloads
    // that do repeat a pair (hand-made ones; the builder's top-K never does) take ds_pk_add_f16, which sums them as the
    // reference does.
    auto entry_pos = [&](unsigned wd) { return ((wd >> 6) & 127u) * FD_LD + (wd & 63u); };
    auto entry_val = [&](unsigned wd) { const unsigned short vb = (unsigned short) (wd >> 16); _Float16 v; __builtin_memcpy(&v, &vb, 2); return v; };
    auto add_atomic = [&](unsigned wbuf, unsigned wd) {
        const unsigned pos = entry_pos(wd);
        const _Float16 v = entry_val(wd);
        const half2v add = (pos & 1u) ? half2v{(_Float16) 0.0f, v} : half2v{v, (_Float16) 0.0f};
        __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2v *) (Ws + wbuf + (pos & ~1u)), add);
    };
    _Float16 eold[4];
    auto live = [&](unsigned wd) { return (wd & 0x7FFF0000u) != 0u; };          // not padding (and not a value of zero)
    auto scatter_read = [&](auto set_c, unsigned Wb) {
        constexpr int S = decltype(set_c)::value;
        if constexpr (!ATOMIC) {
            const unsigned wd[4] = {eq[S].x, eq[S].y, eq[S].z, eq[S].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) eold[i] = Ws[Wb + entry_pos(wd[i])];      // unconditional: a clamped or null entry reads a valid element
        }
    };
    auto scatter = [&](auto set_c, unsigned Wb) {
        constexpr int S = decltype(set_c)::value;
        if (ebeg[S] + 4 * tid < eend[S]) {
            const unsigned wd[4] = {eq[S].x, eq[S].y, eq[S].z, eq[S].w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (live(wd[i])) {
                    if constexpr (ATOMIC) add_atomic(Wb, wd[i]);
                    else Ws[Wb + entry_pos(wd[i])] = eold[i] + entry_val(wd[i]);
                }
        }
        for (int e = ebeg[S] + 2048 + tid; e < eend[S]; e += 512) {                       // a bucket beyond the prefetch
            const unsigned wd = ent[e];
            if (!live(wd)) continue;
            if constexpr (ATOMIC) add_atomic(Wb, wd);
            else { const unsigned pos = Wb + entry_pos(wd); Ws[pos] = Ws[pos] + entry_val(wd); }
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
    const unsigned wfrag = (unsigned) ((wr * 64 + r) * FD_LD + 8 * h), xfrag = (unsigned) ((wt * 64 + r) * FD_LD + 8 * h);

    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    unsigned Wcur = 0, Wnxt = FD_R * FD_LD, Wnx2 = 2 * FD_R * FD_LD;      // element offsets of the three W tiles in rotation
    // slab s: p_c = parity of s. Register sets on entry: gq[p] = W0 of slab s+2, xq[!p] / eq[!p] = x tile / entries of slab
    // s+1; LDS: Wcur = W0+D of slab s, Wnxt = W0 of slab s+1 (its entries not yet added), Xs[p] = x of slab s.
    auto slab = [&](auto p_c, auto q_c, int s) {
        constexpr int P = decltype(p_c)::value;
        const _Float16 * Wb = Ws + Wcur + wfrag;
        const _Float16 * Xb = Xs + (size_t) P * FD_N * FD_LD + xfrag;
        half8 xf[2][2], wf[2][2];
        auto frags = [&](int q, int ks) {
#pragma unroll
            for (int a = 0; a < 2; ++a) xf[q][a] = *reinterpret_cast<const half8 *>(Xb + a * 32 * FD_LD + ks * 16);
#pragma unroll
            for (int c = 0; c < 2; ++c) wf[q][c] = *reinterpret_cast<const half8 *>(Wb + c * 32 * FD_LD + ks * 16);
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
            // the global loads of the slabs ahead, spread over the multiply steps (one burst at the top of the slab
            // measured 135 against 127 us per call on a Mistral gate)
            if (ks == 0 && s + 3 < n_slab) fetchW(q_c, s + 3);
            if (ks == 1 && s + 2 < n_slab) fetchX(p_c, s + 2);
            if (ks == 2) { if (s + 2 < n_slab) fetchE(p_c); fetch_idx(s + 4); fetch_off(s + 3); }
            if (ks == 0 && s + 1 < n_slab) scatter_read(q_c, Wnxt);
            if (ks == 0 && s + 2 < n_slab) commitW(p_c, Wnx2, 0);
            if (ks == 1 && s + 2 < n_slab) commitW(p_c, Wnx2, 1);
            if (ks == 2 && s + 1 < n_slab) { commitX(q_c, P ^ 1, 0); scatter(q_c, Wnxt); }
            if (ks == 3 && s + 1 < n_slab) commitX(q_c, P ^ 1, 1);
        }
        __syncthreads();
        const unsigned t = Wcur; Wcur = Wnxt; Wnxt = Wnx2; Wnx2 = t;
    };

    // prologue: Wcur = W0(0), Wnxt = W0(1), Xs[0] = x(0); then the entries of slab 0 once W0(0) is complete
    fetch_idx(0);
    fetch_off(0);
    fetchW(set0{}, 0);
    fetchX(set0{}, 0);
    fetchE(set0{});
    fetch_idx(1);
    commitW(set0{}, Wcur, 0); commitW(set0{}, Wcur, 1);
    commitX(set0{}, 0, 0); commitX(set0{}, 0, 1);
    if (1 < n_slab) {
        fetchW(set1{}, 1);
        commitW(set1{}, Wnxt, 0); commitW(set1{}, Wnxt, 1);
    }
    __syncthreads();
    scatter_read(set0{}, Wcur);
    scatter(set0{}, Wcur);
    fetch_idx(2);
    fetch_off(1);
    if (2 < n_slab) fetchW(set0{}, 2);          // -> gq[0] = W0 of slab 2
    if (1 < n_slab) { fetchX(set1{}, 1); fetchE(set1{}); }      // -> xq[1], eq[1] = slab 1
    fetch_idx(3);
    fetch_off(2);
    __syncthreads();
    for (int s = 0; s < n_slab; s += 2) {
        slab(set0{}, set1{}, s);
        if (s + 1 < n_slab) slab(set1{}, set0{}, s + 1);
    }

    // C[row = token][col = o]: row = (e & 3) + 8 * (e >> 2) + 4 * h, col = r
    const size_t col_off = TALL ? 0 : pb;
    int bad = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int orow = o_tile + wr * 64 + c * 32 + r;
        if (orow >= n_out) continue;
        const float sc = (TALL && rs) ? rs[orow] : 1.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t t = t_tile + wt * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (t >= T) continue;
                float v = acc[a][c][e];
                if (TALL) {
                    if (sc != 1.0f) v *= sc;                                           // :645
                    if (!__builtin_isfinite(v)) { ++bad; if (zero_nf) v = 0.0f; }
                }
                out[t * ldo + col_off + orow] = v;
            }
    }
    if (TALL && bad) atomicAdd(nf_counter, bad);
}

// wide: y[t][o] = rs[o] * (part[t][0][o] + part[t][1][o] + ...) in block order, non-finite census
__global__ __launch_bounds__(256)
void fold_reduce_kernel(const float * __restrict__ part, int B, int L, int n_out, int64_t T, const float * __restrict__ rs,
                        float * __restrict__ y, int64_t ldy, int * __restrict__ nf_counter, int zero_nf) {
    const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= T * n_out) return;
    const int64_t t = e / n_out;
    const int o = (int) (e - t * n_out);
    const float * p = part + (size_t) t * B * L + o;
    float v = p[0];
    for (int b = 1; b < B; ++b) v += p[(size_t) b * L];
    const float sc = rs ? rs[o] : 1.0f;
    if (sc != 1.0f) v *= sc;
    if (!__builtin_isfinite(v)) { atomicAdd(nf_counter, 1); if (zero_nf) v = 0.0f; }
    y[t * ldy + o] = v;
}

size_t w0_fold_ws_bytes(const DeviceWeight & w, int64_t T) {
    const int64_t T_pad = (T + FD_N - 1) / FD_N * FD_N;
    const int64_t ld16 = (w.n_in + FD_K - 1) / FD_K * FD_K + FD_K;
    const size_t part = w.tall ? 0 : (size_t) T * (size_t) w.B * (size_t) w.L * 4;
    return ((part + 255) & ~(size_t) 255) + (size_t) T_pad * (size_t) ld16 * 2;
}

// The whole op for T tokens: x16 pre-pass, the folded contraction, (wide) the block-order reduction. ws: w0_fold_ws_bytes.
// Returns 1 when the payload does not fit the scheme (no closed-form W0, L < 128).
int launch_w0_fold(DeviceWeight & w, const float * x, int64_t ldx, int64_t T, void * ws, float * y, int64_t ldy,
                   int * nf_counter, bool zero_nf, hipStream_t s) {
    if (!w.w0_closed_form || w.L < FD_R || (w.L % FD_R) != 0) return 1;
    if (T <= 0) return SDH_OK;
    int rc = fold_ensure(w, s);
    if (rc != SDH_OK) return rc;
    const int64_t T_pad = (T + FD_N - 1) / FD_N * FD_N;
    const int64_t ld16 = (w.n_in + FD_K - 1) / FD_K * FD_K + FD_K;
    const size_t part_bytes = w.tall ? 0 : (((size_t) T * (size_t) w.B * (size_t) w.L * 4 + 255) & ~(size_t) 255);
    float * part = (float *) ws;
    _Float16 * x16 = (_Float16 *) ((unsigned char *) ws + part_bytes);
    if ((rc = launch_w0_x16(x, ldx, T, (int) w.n_in, x16, ld16, T_pad, s)) != SDH_OK) return rc;
    const size_t lds = (size_t) (3 * FD_R + 2 * FD_N) * FD_LD * 2;
    static const bool attr_ok = [] {
        bool ok = true;
        const void * ks[8] = {(const void *) w0d_mfma_kernel<true, true, true>, (const void *) w0d_mfma_kernel<true, true, false>,
                              (const void *) w0d_mfma_kernel<true, false, true>, (const void *) w0d_mfma_kernel<true, false, false>,
                              (const void *) w0d_mfma_kernel<false, true, true>, (const void *) w0d_mfma_kernel<false, true, false>,
                              (const void *) w0d_mfma_kernel<false, false, true>, (const void *) w0d_mfma_kernel<false, false, false>};
        for (const void * k : ks) ok &= hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        return ok;
    }();
    if (!attr_ok) return 1;                                     // a device without the LDS for it: the caller's other forms
    const dim3 grid((unsigned) ((w.n_out + FD_R - 1) / FD_R), (unsigned) (T_pad / FD_N), (unsigned) (w.tall ? 1 : w.B));
    if (grid.y > 65535 || grid.z > 65535) { set_error("w0 fold: token chunk too large"); return SDH_ERR_UNSUPPORTED; }
    const bool scaled = w.pack.off_d1 >= 0 || w.pack.off_d3 >= 0;          // else d1 = d3 = 1 (what the builder writes)
    float * out = w.tall ? y : part;
    const int64_t ldo = w.tall ? ldy : (int64_t) w.B * w.L;
#define SDH_FD_ARGS x16, ld16, T, (int) w.n_in, (int) w.n_out, (int) w.L, w.d1, w.d3, w.G, w.inv, w.fold_ent, w.fold_off, w.fold_ng, w.fold_spb, \
        w.rs, out, ldo, nf_counter, zero_nf ? 1 : 0
#define SDH_FD_LAUNCH(TALL_, SC_, AT_) hipLaunchKernelGGL((w0d_mfma_kernel<TALL_, SC_, AT_>), grid, dim3(512), lds, s, SDH_FD_ARGS)
#define SDH_FD_LAUNCH2(TALL_, SC_) do { if (w.fold_dup) SDH_FD_LAUNCH(TALL_, SC_, true); else SDH_FD_LAUNCH(TALL_, SC_, false); } while (0)
    if (w.tall) { if (scaled) SDH_FD_LAUNCH2(true, true); else SDH_FD_LAUNCH2(true, false); }
    else        { if (scaled) SDH_FD_LAUNCH2(false, true); else SDH_FD_LAUNCH2(false, false); }
#undef SDH_FD_ARGS
#undef SDH_FD_LAUNCH2
#undef SDH_FD_LAUNCH
    SDH_HIP_CHECK(hipGetLastError());
    if (!w.tall) {
        const int64_t n = T * w.n_out;
        hipLaunchKernelGGL(fold_reduce_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, part, (int) w.B, (int) w.L, (int) w.n_out, T,
                           w.rs, y, ldy, nf_counter, zero_nf ? 1 : 0);
        SDH_HIP_CHECK(hipGetLastError());
    }
    return SDH_OK;
}

} // namespace sdh
