// sdh_w0fold.hip — prefill as ONE dense contraction per tensor: the residual folded into the W0 tile.
//
//   y[t, o] = rs[o] * sum_i (W0[o, i] + D[o, i]) * x[t, i]
//
// The reference evaluates the base term and the residual side by side for every output (src/llama-seeddelta.cpp
// :1088-1127 tall, :1128-1175 wide). For many tokens the residual's per-row gathers are what costs (0.94 G scattered
// LDS reads per ubatch of a Mistral gate); the dense product W0.x is a real GEMM. So the kernels here build the W0 tile
// of a k-slab in LDS (closed form G_b[(p ^ inv_b[q])], bit-exact in fp32, rounded to f16 once) and then ADD the residual
// entries that fall into that (row tile, slab) to it, in LDS, before the matrix cores read it: the residual becomes part
// of the B operand and costs one LDS add per entry and workgroup instead of one gather-MAC per entry and token. The
// payload is re-laid out once per handle, on the device, for that: entries bucketed by (128-row tile, 64-wide k-slab),
// each a 32-bit word {f16 value, byte offset in the tile}; buckets padded to multiples of four with null entries (value
// +0 at a pad column); block payloads are expanded to entries, invalid indices dropped (reference skip rules :693,
// :929-936).
//
// Numerics: f16 operands (x, W0 + D), fp32 accumulation, row_scale in fp32 in the epilogue — the stated f16 bound of the
// tests, not the bit-exact order of the SpMM / FWHT forms, which stay available (SDH_FWD_NO_MFMA).
//
// Two kernels:
//   w0p_mfma_kernel  the contraction in the PERMUTED order j = inv_b[q], for payloads whose d1 and d3 are all ones (what
//                    the builder writes) and whose base blocks are at most a third longer than their inputs. W0[p][j] =
//                    G_b[p ^ j] is then a plain XOR-circulant, the x image is permuted and fragment-tiled by a pre-pass, and
//                    the x fragments never touch LDS. Its notes stand above the kernel. 0.074-0.086 ms per Mistral-7B
//                    tensor and ubatch of 512.
//   w0d_mfma_kernel  the contraction in q order, for everything else that has a closed-form base: W0 tile gathered from G_b
//                    (16 dword gathers, 8 readlanes, 16 multiplies and 16 conversions per thread and slab), x tile staged
//                    in LDS, 8 waves of 64 rows x 64 tokens of v_mfma_f32_32x32x16_f16. 0.11-0.13 ms. Its layout:
//   workgroup  512 threads = 8 waves; tile 128 rows x 256 tokens x 64 k per slab.
//   LDS        Ws[3][128][72] f16 (three W tiles in rotation: slab s is multiplied while the residual of slab s+1 is
//              added to the tile written one slab earlier and the W0 tile of slab s+2 is written — one barrier per
//              slab), Xs[2][256][72] f16, the bucket bounds. 133 KB.
//   adds       a 16-bit LDS read, v_add_f16, a 16-bit LDS write per entry (the reads issued at the first multiply step, the
//              writes two steps later); ds_pk_add_f16 when a payload repeats a (row, input) pair, which sums them as the
//              reference sums them.
//   tried      (on the q-order kernel, before the permuted one existed) sixteen waves of 32 rows x 64 tokens per workgroup
//              (four waves per SIMD, 128 registers) instead of eight of 64 x 64: the same time within 1 % (136.7 / 135.4 us
//              per call on a Mistral gate); SQ counters (profiles/r03_e_pmc_fold_and_scan.json): MFMA busy 0.22, VALU
//              0.29, LDS 0.22, waves waiting 0.41 of their cycles. What those waits were: the vmcnt note above
//              w0p_mfma_kernel.
//   wide       K is split by base block over blockIdx.z (as the tiled W0 kernel); partial products go to [T][B][L]
//              and fold_reduce_kernel sums them in block order, applies row_scale and the non-finite census.
#include "sdh_internal.h"
#include "sdh_delta_dev.h"
#include <type_traits>

namespace sdh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int FD_R = 128, FD_N = 256, FD_K = 64, FD_LD = 72;
constexpr int FD_GS = 8 * 128;                 // elements of one staged segment image
// the second tensor (up) of the two-pass launch for a gated FFN: same shape and layout constants as the first
struct FoldSecond { const _Float16 * z16; const _Float16 * gp; const unsigned * ent; const int2 * off; const float * rs; int * nf; };
constexpr int FP_LD = 80;                      // row pitch of the permuted kernel's W tiles: 160 bytes, conflict-free for the 16x16x32 B fragments
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- payload re-layout: entries bucketed by (row tile, k-slab) ------------------------------------------------
struct FoldArgs {
    const void * idx; const void * val; int idx32, val32;
    int n_in, n_out, K, nb, block, is_block;
    int L, tall, n_g, spb;             // n_g: slabs per offsets row; spb: slabs per base block
    const int * inv; int perm;         // perm: bucket by j = inv_b[q]
};

constexpr unsigned FD_NULL = 2u * FD_K;        // a null entry: value +0 at byte offset of (row 0, column 64) — a pad column nothing reads

// bucket and column of input k in the row o
__device__ inline void fold_place(const FoldArgs & a, int o, int k, int & g, int & col) {
    if (a.tall) {
        const int j = a.perm ? a.inv[(size_t) (o / a.L) * a.L + k] : k;
        g = j / FD_K; col = j & (FD_K - 1);
    } else {
        const int b = k / a.L;
        int j = k - b * a.L;
        if (a.perm) j = a.inv[(size_t) b * a.L + j];
        g = b * a.spb + j / FD_K; col = j & (FD_K - 1);
    }
}

__device__ inline unsigned fold_entry(const FoldArgs & a, _Float16 v, int row, int col) {
    unsigned short vb; __builtin_memcpy(&vb, &v, 2);
    return ((unsigned) vb << 16) | (unsigned) ((row * (a.perm ? FP_LD : FD_LD) + col) * 2);
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
        int g, col;
        fold_place(a, o, ii, g, col);
        int * slot = (PASS == 0 ? counts : cursor) + (size_t) tile * (a.n_g + 1) + g;
        if (PASS == 0) { atomicAdd(slot, 1); return; }
        const int pos = atomicAdd(slot, 1);
        const _Float16 v = a.val32 ? (_Float16) ((const float *) a.val)[e] : ((const _Float16 *) a.val)[e];
        ent[pos] = fold_entry(a, v, row, col);
    } else {
        if (ii < 0) return;                                                        // :929-936
        const int64_t in0 = (int64_t) ii * a.block;
        if (in0 >= a.n_in) return;
        const int in1 = (int) min((int64_t) a.n_in, in0 + a.block);
        for (int k = (int) in0; k < in1; ++k) {
            int g, col;
            fold_place(a, o, k, g, col);
            int * slot = (PASS == 0 ? counts : cursor) + (size_t) tile * (a.n_g + 1) + g;
            if (PASS == 0) { atomicAdd(slot, 1); continue; }
            const int pos = atomicAdd(slot, 1);
            const int64_t vi = ((int64_t) o * a.nb + r) * a.block + (k - in0);
            const _Float16 v = a.val32 ? (_Float16) ((const float *) a.val)[vi] : ((const _Float16 *) a.val)[vi];
            ent[pos] = fold_entry(a, v, row, col);
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

// does any bucket hold the same (row, column) twice? one workgroup per bucket, a bitmap of the positions of a tile
__global__ __launch_bounds__(256)
void fold_dup_kernel(const unsigned * __restrict__ ent, const int2 * __restrict__ bounds, int * __restrict__ dup /* [0] flag, [1] largest bucket */) {
    __shared__ unsigned bm[FD_R * FP_LD / 32];
    for (int i = threadIdx.x; i < FD_R * FP_LD / 32; i += 256) bm[i] = 0u;
    __syncthreads();
    const int2 o = bounds[blockIdx.x];
    bool hit = false;
    for (int e = o.x + (int) threadIdx.x; e < o.y; e += 256) {
        const unsigned wd = ent[e];
        if ((wd & 0xFFFFu) == FD_NULL) continue;                                 // padding
        const unsigned key = (wd & 0xFFFFu) >> 1, bit = 1u << (key & 31u);
        hit |= (atomicOr(&bm[key >> 5], bit) & bit) != 0u;
    }
    if (hit) atomicOr(dup, 1);
    if (threadIdx.x == 0) atomicMax(dup + 1, o.y - o.x);
}

// fold_gp[b][c][i] = f16(G_b[i ^ c]), c = 0..7
__global__ __launch_bounds__(256)
void fold_gp_kernel(const float * __restrict__ G, _Float16 * __restrict__ gp, int L, int B) {
    const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t) B * 8 * L) return;
    const int i = (int) (e % L), c = (int) ((e / L) & 7), b = (int) (e / (8 * (int64_t) L));
    gp[e] = (_Float16) G[(size_t) b * L + (i ^ c)];
}

// The pre-pass of the permuted form: z[t][j] = f16(x[t][perm1_b[j]]) (0 where block b has no such input, and for the padding
// tokens), written in the order the contraction's waves read it: tiled [token / 32][column / 32][32 tokens][32 columns], so
// that an A fragment of v_mfma_f32_16x16x32_f16 (16 tokens x 32 columns) is 1 KB of consecutive bytes and the four a wave
// needs for a slab are 4 KB. One workgroup of 16 waves per four tokens and half of the columns: the rows of x go to LDS once (f16), every
// base block gathers from there; the permutation is read as 16-bit indices (fold_p16, -1 = no input), once per four tokens.
//   tall: one image per base block (z + b * zcopy) of kz = L columns;   wide: one image of kz = B * L columns
#ifndef SDH_FZ_T
#define SDH_FZ_T 4
#endif
#ifndef SDH_FZ_NTH
#define SDH_FZ_NTH 1024
#endif
constexpr int FZ_T = SDH_FZ_T, FZ_NTH = SDH_FZ_NTH;
template <bool TALL>
__global__ __launch_bounds__(FZ_NTH)
void fold_z16_kernel(const float * __restrict__ x, int64_t ldx, int64_t T, int n_in, int L, int B, const short * __restrict__ p16a,
                     _Float16 * __restrict__ za, int kz, int64_t zcopy,
                     const short * __restrict__ p16b = nullptr, _Float16 * __restrict__ zb = nullptr) {
    // blockIdx.z == 1: the second tensor of a pair that shares x and the shape (gate | up of a gated FFN: one launch for both images)
    const short * __restrict__ p16 = blockIdx.z ? p16b : p16a;
    _Float16 * __restrict__ z = blockIdx.z ? zb : za;
    extern __shared__ __align__(16) unsigned char fz_smem[];
    _Float16 * xr = (_Float16 *) fz_smem;                          // [FZ_T][n_in]
    const int64_t t0 = FZ_T * (int64_t) blockIdx.x;
    typedef short short8 __attribute__((ext_vector_type(8)));
    const int gpb = L / 8, total = B * gpb, half = (total + 1) / 2;
    const int g_end = min(total, ((int) blockIdx.y + 1) * half);
    const int g_first = (int) blockIdx.y * half + (int) threadIdx.x;
    // the first group's indices are asked for before the rows of x: the two round trips overlap
    short8 q_next = *reinterpret_cast<const short8 *>(p16 + (size_t) min(g_first, total - 1) * 8);
    const bool vec = (ldx & 3) == 0 && (n_in & 3) == 0 && ((uintptr_t) x & 15) == 0;
#pragma unroll
    for (int u = 0; u < FZ_T; ++u) {
        if (t0 + u >= T) continue;
        const float * xt = x + (t0 + u) * ldx;
        _Float16 * xu = xr + (size_t) u * n_in;
        if (vec) {
            for (int k = 4 * threadIdx.x; k < n_in; k += 4 * FZ_NTH) {
                const float4 v = *reinterpret_cast<const float4 *>(xt + k);
                typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<half4 *>(xu + k) = half4{(_Float16) v.x, (_Float16) v.y, (_Float16) v.z, (_Float16) v.w};
            }
        } else {
            for (int k = threadIdx.x; k < n_in; k += FZ_NTH) xu[k] = (_Float16) xt[k];
        }
    }
    __syncthreads();
    for (int g = g_first; g < g_end; g += FZ_NTH) {
        const int b = g / gpb, j0 = (g - b * gpb) * 8;
        const short8 q = q_next;
        q_next = *reinterpret_cast<const short8 *>(p16 + (size_t) min(g + FZ_NTH, total - 1) * 8);
        const int jj = TALL ? j0 : b * L + j0;                     // column in the image
        _Float16 * zo = z + (TALL ? (int64_t) b * zcopy : (int64_t) 0) + ((t0 >> 5) * (int64_t) (kz >> 5) + (jj >> 5)) * 1024 + (t0 & 31) * 32 + (jj & 31);
#pragma unroll
        for (int u = 0; u < FZ_T; ++u) {
            half8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (t0 + u < T && q[e] >= 0) ? xr[(size_t) u * n_in + q[e]] : (_Float16) 0.0f;
            *reinterpret_cast<half8 *>(zo + u * 32) = v;
        }
    }
}

// fold_p16[b][j] = the input (index into the row of x) that column j of block b multiplies, or -1
__global__ __launch_bounds__(256)
void fold_p16_kernel(const int * __restrict__ p1, short * __restrict__ p16, int L, int B, int n_in, int tall) {
    const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t) B * L) return;
    const int b = (int) (e / L), j = (int) (e - (int64_t) b * L);
    const int q = p1 ? p1[e] : j;
    const int q_len = tall ? n_in : max(0, min(L, n_in - b * L));
    p16[e] = (short) ((q >= 0 && q < q_len) ? (tall ? q : b * L + q) : -1);
}

static bool fold_scaled(const DeviceWeight & w) { return w.pack.off_d1 >= 0 || w.pack.off_d3 >= 0; }   // else d1 = d3 = 1 (what the builder writes)

// The permuted form multiplies over whole base blocks (tall: L columns for n_in inputs, wide: B * L): taken when that is at
// most a third more than the inputs, and the row of x fits the pre-pass's LDS.
static bool fold_perm_ok(const DeviceWeight & w) {
    const int64_t kz = w.tall ? w.L : w.B * w.L;
    static const bool off = getenv("SDH_FOLD_NO_PERM") != nullptr;               // A/B switch
    return !off && !fold_scaled(w) && kz * 3 <= w.n_in * 4 && w.n_in <= 20480;     // 16-bit input indices; four f16 rows of x in the pre-pass's LDS
}

bool w0_fold_permuted(const DeviceWeight & w) { return w.fold_ent ? w.fold_perm : fold_perm_ok(w); }

int fold_ensure(DeviceWeight & w, hipStream_t s) {
    if (w.fold_ent) return SDH_OK;
    if (!w.w0_closed_form || w.L < FD_R || (w.L % FD_R) != 0) return 1;
    const bool is_block = w.kind == SDH_OP_BASE_BLOCK;
    const bool perm = fold_perm_ok(w);
    FoldArgs a{};
    a.idx = w.idx; a.val = w.val; a.idx32 = w.idx_type == SDH_TYPE_I32; a.val32 = w.val_type == SDH_TYPE_F32;
    a.n_in = (int) w.n_in; a.n_out = (int) w.n_out; a.K = (int) w.K; a.nb = (int) w.nb; a.block = (int) w.block; a.is_block = is_block;
    a.L = (int) w.L; a.tall = w.tall ? 1 : 0;
    a.inv = w.inv; a.perm = perm ? 1 : 0;
    a.spb = (int) (w.L / FD_K);
    a.n_g = w.tall ? (perm ? a.spb : (int) ((w.n_in + FD_K - 1) / FD_K)) : (int) (w.B * a.spb);      // wide: a row of buckets per base block the grid has
    const int tiles = (int) ((w.n_out + FD_R - 1) / FD_R);
    const int64_t per_row = is_block ? w.nb : w.K;
    const int64_t cap = (int64_t) w.n_out * (is_block ? w.nb * w.block : w.K);
    if (cap <= 0 || cap > 0x7fffffff) return 1;
    const size_t off_bytes = (size_t) tiles * (a.n_g + 1) * sizeof(int);
    const size_t ent_words = (size_t) cap + (size_t) 4 * tiles * a.n_g + 4;       // every bucket padded to a multiple of 4
    int * counts = nullptr, * cursor = nullptr; unsigned * ent = nullptr; int2 * bounds = nullptr; _Float16 * gp = nullptr; short * p16 = nullptr;
    bool done = false;
    struct Cleanup {                                               // whatever way this function is left early, nothing it allocated stays behind
        int *& counts; int *& cursor; unsigned *& ent; int2 *& bounds; _Float16 *& gp; short *& p16; bool & done;
        ~Cleanup() {
            (void) hipFree(counts); (void) hipFree(cursor);
            if (!done) { (void) hipFree(ent); (void) hipFree(bounds); (void) hipFree(gp); (void) hipFree(p16); }
        }
    } cleanup{counts, cursor, ent, bounds, gp, p16, done};
    SDH_HIP_CHECK(hipMalloc((void **) &counts, off_bytes));
    SDH_HIP_CHECK(hipMalloc((void **) &cursor, off_bytes));
    SDH_HIP_CHECK(hipMalloc((void **) &bounds, (size_t) tiles * a.n_g * sizeof(int2)));
    SDH_HIP_CHECK(hipMalloc((void **) &ent, ent_words * sizeof(unsigned)));
    SDH_HIP_CHECK(hipMemsetAsync(counts, 0, off_bytes, s));
    SDH_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t) ent, (int) FD_NULL, ent_words, s));     // padding = null entries
    const unsigned grid = (unsigned) ((w.n_out * per_row + 255) / 256);
    hipLaunchKernelGGL(fold_bucket_kernel<0>, dim3(grid), dim3(256), 0, s, a, counts, cursor, ent);
    hipLaunchKernelGGL(fold_scan_kernel, dim3(1), dim3(1024), 0, s, counts, cursor, bounds, tiles, a.n_g);
    hipLaunchKernelGGL(fold_bucket_kernel<1>, dim3(grid), dim3(256), 0, s, a, counts, cursor, ent);
    SDH_HIP_CHECK(hipGetLastError());
    // counts is free again: its first word receives the "a (row, input) pair repeats" flag
    SDH_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * sizeof(int), s));
    hipLaunchKernelGGL(fold_dup_kernel, dim3((unsigned) (tiles * a.n_g)), dim3(256), 0, s, ent, bounds, counts);
    SDH_HIP_CHECK(hipGetLastError());
    if (perm) {
        const int64_t n = w.B * 8 * w.L;
        SDH_HIP_CHECK(hipMalloc((void **) &gp, (size_t) n * sizeof(_Float16)));
        hipLaunchKernelGGL(fold_gp_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, w.G, gp, (int) w.L, (int) w.B);
        SDH_HIP_CHECK(hipGetLastError());
        SDH_HIP_CHECK(hipMalloc((void **) &p16, (size_t) (w.B * w.L) * sizeof(short)));
        hipLaunchKernelGGL(fold_p16_kernel, dim3((unsigned) ((w.B * w.L + 255) / 256)), dim3(256), 0, s, w.p1, p16, (int) w.L, (int) w.B, (int) w.n_in, w.tall ? 1 : 0);
        SDH_HIP_CHECK(hipGetLastError());
    }
    int dup[2] = {0, 0};
    SDH_HIP_CHECK(hipMemcpyAsync(dup, counts, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SDH_HIP_CHECK(hipStreamSynchronize(s));
    done = true;
    w.fold_dup = dup[0] != 0;
    w.fold_over = dup[1] > 4 * 512;                               // some bucket holds more than the four entries per thread of the prefetch
    w.fold_perm = perm; w.fold_gp = gp; w.fold_p16 = p16;
    w.fold_ent = ent; w.fold_off = bounds; w.fold_ng = a.n_g; w.fold_spb = a.spb;
    return SDH_OK;
}

void fold_free(DeviceWeight & w) {
    (void) hipFree(w.fold_ent); (void) hipFree(w.fold_off); (void) hipFree(w.fold_gp); (void) hipFree(w.fold_p16);
    w.fold_ent = nullptr; w.fold_off = nullptr; w.fold_gp = nullptr; w.fold_p16 = nullptr;
}

// ---- the contraction ---------------------------------------------------------------------------------------------
template <bool TALL, bool SCALED /* d1 or d3 present and not all ones */, bool ATOMIC /* the payload repeats a (row, input) pair */,
          bool OVER /* some bucket holds more than 2048 entries: a loop over the rest in the scatter step (its loads cost the
                       main loop its exact vmcnt bookkeeping, so it is a variant of its own) */>
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
    int2 * Bn = (int2 *) (Xs + 2 * FD_N * FD_LD + 2 * FD_GS);     // [n_slab] bucket bounds of this workgroup (behind the permuted kernel's staging area)
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
    auto clamp_slab = [&](int s) { return max(min(s, n_slab - 1), 0); };
    auto fetch_idx = [&](int s) {
        const int q0 = clamp_slab(s) * FD_K + wave * 8;
        ivl = inv[pb + q0 + (lane & 7)];
    };
    // Every global load of the main loop is issued unconditionally (slab indices clamped to the last slab): a load behind a
    // branch makes the compiler's vmcnt bookkeeping assume the path that skipped it, and every later wait then drains
    // the queue almost completely (vmcnt(1), vmcnt(2) where ten loads could stay in flight; the bounds -> entries chain
    // alone was a vmcnt(0) per slab that exposed the x loads issued just before it: 127 -> 9x us per call on a Mistral gate).
    // The bucket bounds come from LDS for the same reason.
    auto fetch_off = [&](int s) { obe = Bn[clamp_slab(s)]; };
    auto fetchW = [&](auto set_c, int s) {
        constexpr int S = decltype(set_c)::value;
        const int q0 = clamp_slab(s) * FD_K + wave * 8;
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
        const _Float16 * xs0 = x16 + t_tile * ld16 + k_begin + clamp_slab(s) * FD_K;         // uniform
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
#ifdef SDH_FD_NOCOMMIT
            { const half8 keep = xq[S][2 * half + i]; asm volatile("" :: "v"(keep)); }
#else
            *reinterpret_cast<half8 *>(Xs + (size_t) buf * FD_N * FD_LD + xs_w + (2 * half + i) * 64 * FD_LD) = xq[S][2 * half + i];
#endif
    };
    // Adding the entries of a bucket to a W tile. Entries of a payload without repeated (row, input) pairs touch distinct
    // elements, so each is a plain 16-bit read, add, 16-bit write — the reads of a slab's entries are issued at the first
    // multiply step and the writes two steps later, so no wave waits out an LDS round trip between its MFMAs (a read-add-
    // write per entry in one place cost 0.3 us per slab: three dependent round trips with the matrix pipe idle). Payloads
    // that do repeat a pair (hand-made ones; the builder's top-K never does) take ds_pk_add_f16, which sums them as the
    // reference does.
    auto entry_pos = [&](unsigned wd) { return (wd & 0xFFFFu) >> 1; };             // element offset in the tile (null entries: a pad column)
    auto entry_val = [&](unsigned wd) { const unsigned short vb = (unsigned short) (wd >> 16); _Float16 v; __builtin_memcpy(&v, &vb, 2); return v; };
    auto add_atomic = [&](unsigned wbuf, unsigned wd) {
        const unsigned pos = entry_pos(wd);
        const _Float16 v = entry_val(wd);
        const half2v add = (pos & 1u) ? half2v{(_Float16) 0.0f, v} : half2v{v, (_Float16) 0.0f};
        __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2v *) (Ws + wbuf + (pos & ~1u)), add);
    };
    _Float16 eold[4];
    auto scatter_read = [&](auto set_c, unsigned Wb) {
        constexpr int S = decltype(set_c)::value;
        if constexpr (!ATOMIC) {
            const unsigned wd[4] = {eq[S].x, eq[S].y, eq[S].z, eq[S].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) eold[i] = Ws[Wb + entry_pos(wd[i])];      // unconditional: a clamped or null entry reads a valid element
        }
    };
    // no test per entry: a null entry adds +0 to a pad element (whatever races there is never read); the threads beyond the
    // bucket's end hold clamped copies of its last entries and must not write
    auto scatter = [&](auto set_c, unsigned Wb) {
        constexpr int S = decltype(set_c)::value;
        if (ebeg[S] + 4 * tid < eend[S]) {
            const unsigned wd[4] = {eq[S].x, eq[S].y, eq[S].z, eq[S].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (ATOMIC) add_atomic(Wb, wd[i]);
                else Ws[Wb + entry_pos(wd[i])] = eold[i] + entry_val(wd[i]);
            }
        }
        if constexpr (OVER)
            for (int e = ebeg[S] + 2048 + tid; e < eend[S]; e += 512) {                   // a bucket beyond the prefetch
                const unsigned wd = ent[e];
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
    // The fragments of a slab's first multiply step are read right after the barrier that completes its tiles, and the last
    // multiply step of the slab before is issued behind those reads: the matrix pipe works through the LDS latency that
    // every wave would otherwise wait out together once per slab.
    half8 xf[2][2], wf[2][2];
#ifdef SDH_FD_NOFRAGS
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int a = 0; a < 2; ++a) { asm volatile("" : "=v"(xf[q][a])); asm volatile("" : "=v"(wf[q][a])); }
#endif
    auto frags = [&](int q, int xbuf, unsigned wbuf, int ks) {
#ifndef SDH_FD_NOFRAGS
        const _Float16 * Wb = Ws + wbuf + wfrag;
        const _Float16 * Xb = Xs + (size_t) xbuf * FD_N * FD_LD + xfrag;
#pragma unroll
        for (int a = 0; a < 2; ++a) xf[q][a] = *reinterpret_cast<const half8 *>(Xb + a * 32 * FD_LD + ks * 16);
#pragma unroll
        for (int c = 0; c < 2; ++c) wf[q][c] = *reinterpret_cast<const half8 *>(Wb + c * 32 * FD_LD + ks * 16);
#endif
    };
    auto multiply = [&](int q) {
#ifndef SDH_FD_NOMFMA
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[q][a], wf[q][c], acc[a][c], 0, 0, 0);
#endif
    };
    auto slab = [&](auto p_c, auto q_c, int s) {
        constexpr int P = decltype(p_c)::value;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            frags((ks + 1) & 1, P, Wcur, ks + 1);
            multiply(ks & 1);
            // the global loads of the slabs ahead, spread over the multiply steps (one burst at the top of the slab
            // measured 135 against 127 us per call on a Mistral gate)
            // (the work for slabs past the end lands in tiles nothing reads any more)
#ifndef SDH_FD_NOSCATTER
            if (ks == 0) fetchE(p_c);                               // bucket s + 2 (its bounds: fetch_off of the slab before)
#endif
#ifndef SDH_FD_NOW
            if (ks == 0) fetchW(q_c, s + 3);
#endif
#ifndef SDH_FD_NOX
            if (ks == 1) fetchX(p_c, s + 2);
#endif
#ifndef SDH_FD_NOSCATTER
            if (ks == 2) { fetch_idx(s + 4); fetch_off(s + 3); }
            if (ks == 0) scatter_read(q_c, Wnxt);
#endif
#ifndef SDH_FD_NOW
            if (ks == 0) commitW(p_c, Wnx2, 0);
            if (ks == 1) commitW(p_c, Wnx2, 1);
#endif
#ifndef SDH_FD_NOX
            if (ks == 1) commitX(q_c, P ^ 1, 0);
            if (ks == 2) commitX(q_c, P ^ 1, 1);
#endif
#ifndef SDH_FD_NOSCATTER
            if (ks == 2) scatter(q_c, Wnxt);
#endif
        }
#ifndef SDH_FD_NOBARRIER
        __syncthreads();
#endif
        const unsigned t = Wcur; Wcur = Wnxt; Wnxt = Wnx2; Wnx2 = t;
        frags(0, P ^ 1, Wcur, 0);                                   // the next slab's first step
        multiply(1);                                                // this slab's last
    };

    // prologue: Wcur = W0(0), Wnxt = W0(1), Xs[0] = x(0); then the entries of slab 0 once W0(0) is complete. The last loads
    // are issued in the order a slab of the main loop issues them (entries, W0 source, x): the loop's first waits count the
    // loads behind the one they need, and the count they may assume is the smaller of this path's and the loop's.
    for (int i = tid; i < n_slab; i += 512) Bn[i] = offr[i];
    fetch_idx(0);
    fetchW(set0{}, 0);
    fetchX(set0{}, 0);
    fetch_idx(1);
    __syncthreads();
    fetch_off(0);
    fetchE(set0{});
    commitW(set0{}, Wcur, 0); commitW(set0{}, Wcur, 1);
    commitX(set0{}, 0, 0); commitX(set0{}, 0, 1);
    fetchW(set1{}, 1);
    commitW(set1{}, Wnxt, 0); commitW(set1{}, Wnxt, 1);
    __syncthreads();
    scatter_read(set0{}, Wcur);
    scatter(set0{}, Wcur);
    fetch_idx(2);
    fetch_off(1);
    fetchE(set1{});                             // -> eq[1] = entries of slab 1
    fetchW(set0{}, 2);                          // -> gq[0] = W0 of slab 2
    fetchX(set1{}, 1);                          // -> xq[1] = x of slab 1
    fetch_idx(3);
    fetch_off(2);
    __syncthreads();
    frags(0, 0, Wcur, 0);
    for (int s = 0; s + 1 < n_slab; s += 2) {           // no branch inside the pair (see the note on vmcnt above)
        slab(set0{}, set1{}, s);
        slab(set1{}, set0{}, s + 1);
    }
    if (n_slab & 1) slab(set0{}, set1{}, n_slab - 1);

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

// ---- the contraction in the permuted order --------------------------------------------------------------------------
// Same W tiles in rotation and the same scatter step as w0d_mfma_kernel; what differs:
//   W0 tile   The 128 x 64 tile of slab s is seg[pl ^ jl] for ONE aligned 128-element segment of G_b (at (p0 ^ 64 s) & ~127): 256
//             bytes, 2 KB under the 8 permutations of fold_gp. Each thread brings one dword of that image per slab (global ->
//             register -> Gs, laid out [group of 8][copy][8] so that the 16 lanes of a read pass fall on 16 distinct bank quads),
//             and the tile is built LDS -> LDS: thread (pl, gsel) copies the groups (pl >> 3) ^ g ^ 8 hi of copy pl & 7 to row pl,
//             columns 8 g .. 8 g + 7, g = gsel and gsel + 4. No arithmetic, 2 KB of L2 traffic per slab instead of 32 KB.
//   x         never in LDS: the pre-pass (fold_z16_kernel) writes z[t][j] = f16(x[t][perm1_b[j]]) tiled [token / 32][column / 32]
//             [32][32], an A fragment of v_mfma_f32_16x16x32_f16 is 1 KB of consecutive bytes, and every wave owns all 128 rows
//             of the tile for its own 32 (or 16) tokens, so no two waves want the same fragment. W tiles at a 160-byte pitch
//             make the B fragments of that MFMA shape conflict-free; under it the chip holds a higher clock than under
//             32x32x16 at the same cycles per flop (79.7 -> 74.0 us).
//   depth     x fragments and entries are loaded THREE slabs ahead into three register sets in rotation (the loop is unrolled
//             by three; which W tile a slab reads is a runtime offset): a CU gets 60-70 GB/s from L2 with ~70 KB in flight, and
//             one slab ahead left the waves waiting on vmcnt with the matrix pipe idle. The last multiply step of a slab is
//             issued behind the barrier, after the next slab's first W fragments have been asked for.
//   clocks    the chip holds ~1.6 GHz under this kernel (126 k shader cycles in ~78 us) and needs 20+ ms of load to settle:
//             time it behind a few hundred calls (tools/fold_time.py does), and read "0.35 of the f16 peak" as 0.52 of the
//             matrix-pipe cycles the chip actually delivers.
//   tried     the workgroups that read one x panel (base block, token tile) on one XCD instead of dealt round-robin: the same
//             time within 1 % (85.4 / 86.4 us per call on a Mistral gate). In the pre-pass, the four tokens' values of an input
//             side by side in LDS (8 gathers of 8 bytes per thread instead of 32 of 2, staged by dword loads): 12.3 / 21.4 us
//             against 10.0 / 12.1 us (tall / wide Mistral tensors). Neither kept.
//             Token tiles of 128 at 512 tokens as well (448 workgroups, two per CU when compiled for 128 registers): 122 against
//             89 us per call; the 128-token tiles compiled for two workgroups per CU at up to 256 tokens: no difference.
//   vmcnt     every global load of the loop is issued unconditionally (indices clamped; work for slabs past the end lands in
//             tiles nothing reads) and the prologue issues its last loads in the loop's order: a load behind a branch, or a
//             different order on the way into the loop, makes the compiler's counted waits assume the shortest queue and
//             drain it (vmcnt(0..2) where ten loads may stay in flight). The bucket bounds sit in LDS for the same reason.
template <bool TALL, bool ATOMIC /* the payload repeats a (row, input) pair */, bool OVER /* some bucket holds more than 2048 entries */,
          int NT /* tokens per wave: 32 (token tile 256) or 16 (token tile 128: up to 256 tokens, twice the workgroups) */,
          bool DUAL = false /* gate and up of a gated FFN in one launch: two passes over the same tile (first the tensor of the
                               main arguments = gate, then `second` = up), out = act(gate) * up; tall tensors of one shape */>
__global__ __launch_bounds__(512)
void w0p_mfma_kernel(const _Float16 * __restrict__ z16, int64_t zcopy, int64_t T, int n_out, int L, int kz /* columns of one z image */,
                     const _Float16 * __restrict__ gp, const unsigned * __restrict__ ent, const int2 * __restrict__ off, int n_g, int spb,
                     const float * __restrict__ rs, float * __restrict__ out, int64_t ldo, int * __restrict__ nf_counter, int zero_nf,
                     const float * __restrict__ glu_gate /* tall only: out = act(gate) * value (ffn_up inside a gated FFN) */, int64_t glu_ldg,
                     int glu_act, const uint16_t * __restrict__ glu_tab, const FoldSecond second) {
    extern __shared__ __align__(16) unsigned char fd_smem[];
    _Float16 * Ws = (_Float16 *) fd_smem;                         // [3][FD_R][FP_LD]
    _Float16 * Gs = Ws + 3 * FD_R * FP_LD;                        // [2][16 groups][8 copies][8]
    int2 * Bn = (int2 *) (Gs + 2 * FD_GS);                        // [n_slab] bucket bounds of this workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, h = lane >> 4;                       // 16x16x32 operands: row (token or output row) r, columns 8 h .. 8 h + 7 of the 32
    const int o_tile = (int) blockIdx.x * FD_R;
    constexpr int NA = NT / 16;                                   // token halves (A fragments per 32-column step) of a wave
    const int64_t t_tile = (int64_t) blockIdx.y * (8 * NT);
    const int b = TALL ? o_tile / L : (int) blockIdx.z;
    const int p0 = TALL ? o_tile - b * L : o_tile;                // first row of the tile inside its block
    const int n_slab = spb;
    const int2 * offr = off + (size_t) blockIdx.x * n_g + (TALL ? 0 : b * spb);
    const unsigned * entp = ent;                                  // (offr, entp, zw, Gb: rebound for the second pass of DUAL)
    // x fragments of this wave: tokens t_tile + 32 wave .. + 32; the image is tiled [token / 32][column / 32][token half][16][32]
    const int64_t tok0 = t_tile + wave * NT;                      // this wave's first token (a multiple of 16)
    const int64_t z_off = (TALL ? (int64_t) b * zcopy : (int64_t) 0) + (tok0 >> 5) * ((int64_t) kz * 32) + ((tok0 >> 4) & 1) * 512
                        + (TALL ? (int64_t) 0 : (int64_t) b * L * 32) + r * 32 + h * 8;
    const _Float16 * zw = z16 + z_off;
    const char * Gb = (const char *) (gp + (size_t) 8 * b * L);

    const int pl = tid & 127, gsel = tid >> 7;
    const unsigned gs_r = (unsigned) (((((pl >> 3) ^ gsel) << 3) | (pl & 7)) << 3);          // element offset in a Gs image at hi = 0, g = gsel
    const unsigned ws_p = (unsigned) (pl * FP_LD + 8 * gsel);
    // staging: dword tid of the image = group tid >> 5, copy (tid >> 2) & 7, halves 2 (tid & 3) ..
    const unsigned sg_src = (((unsigned) ((tid >> 2) & 7) * (unsigned) L + (unsigned) ((tid >> 5) << 3) + (unsigned) ((tid & 3) << 1)) << 1);

    unsigned sg = 0;            // the staged dword of the segment three slabs on
    half8 gh[2];                // the two column groups of the tile two slabs on
    half8 xq[3][2 * NA];        // x fragments of three slabs, [set][NA * (32-column step) + token half]
    uint4 eq[3];                // residual entries (four per thread) in flight
    int ebeg[3], eend[3];       // their bucket bounds
    int2 obe = make_int2(0, 0); // bucket bounds read a slab ahead of the entry loads that use them
    auto clamp_slab = [&](int s) { return max(min(s, n_slab - 1), 0); };
    auto fetch_off = [&](int s) { obe = Bn[clamp_slab(s)]; };
    auto fetchX = [&](auto set_c, int s) {
        constexpr int S = decltype(set_c)::value;
        const _Float16 * xs0 = zw + (size_t) clamp_slab(s) * (4 * 512);                   // NT = 32: 4 KB per slab and wave, contiguous
#pragma unroll
        for (int i = 0; i < 2 * NA; ++i) xq[S][i] = *reinterpret_cast<const half8 *>(xs0 + (i / NA) * 1024 + (i % NA) * 512);
    };
    auto fetchE = [&](auto set_c) {
        constexpr int S = decltype(set_c)::value;
        ebeg[S] = __builtin_amdgcn_readfirstlane(obe.x); eend[S] = __builtin_amdgcn_readfirstlane(obe.y);
        const int e = ebeg[S] + 4 * tid;                           // buckets start and end at multiples of 4: one 16-byte load
        eq[S] = *reinterpret_cast<const uint4 *>(entp + (e < eend[S] ? e : max(eend[S] - 4, 0)));      // clamped: read unconditionally
    };
    auto seg_load = [&](int s) {                                   // global -> sg: the image dword of slab s
        const unsigned base = ((unsigned) (p0 ^ (clamp_slab(s) << 6)) & ~127u) << 1;
        sg = *reinterpret_cast<const unsigned *>(Gb + base + sg_src);
    };
    auto seg_commit = [&](int buf) { reinterpret_cast<unsigned *>(Gs + buf * FD_GS)[tid] = sg; };
    auto tile_read = [&](int buf, int hi) {                        // hi: slab parity (the upper 64 of the segment's 128 positions)
        const _Float16 * g = Gs + buf * FD_GS;
        gh[0] = *reinterpret_cast<const half8 *>(g + (gs_r ^ (unsigned) (hi << 9)));
        gh[1] = *reinterpret_cast<const half8 *>(g + (gs_r ^ (unsigned) (hi << 9) ^ (4u << 6)));
    };
    auto tile_write = [&](unsigned wbuf, int rr) { *reinterpret_cast<half8 *>(Ws + wbuf + ws_p + rr * 32) = gh[rr]; };

    // the scatter step (see w0d_mfma_kernel)
    auto entry_pos = [&](unsigned wd) { return (wd & 0xFFFFu) >> 1; };
    auto entry_val = [&](unsigned wd) { const unsigned short vb = (unsigned short) (wd >> 16); _Float16 v; __builtin_memcpy(&v, &vb, 2); return v; };
    auto add_atomic = [&](unsigned wbuf, unsigned wd) {
        const unsigned pos = entry_pos(wd);
        const _Float16 v = entry_val(wd);
        const half2v add = (pos & 1u) ? half2v{(_Float16) 0.0f, v} : half2v{v, (_Float16) 0.0f};
        __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2v *) (Ws + wbuf + (pos & ~1u)), add);
    };
    _Float16 eold[4];
    auto scatter_read = [&](auto set_c, unsigned Wb) {
        constexpr int S = decltype(set_c)::value;
        if constexpr (!ATOMIC) {
            if (ebeg[S] + 256 * wave < eend[S]) {                  // (wave-uniform) a wave beyond the bucket's end has nothing to add
                const unsigned wd[4] = {eq[S].x, eq[S].y, eq[S].z, eq[S].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) eold[i] = Ws[Wb + entry_pos(wd[i])];
            }
        }
    };
    auto scatter = [&](auto set_c, unsigned Wb) {
        constexpr int S = decltype(set_c)::value;
        if (ebeg[S] + 4 * tid < eend[S]) {
            const unsigned wd[4] = {eq[S].x, eq[S].y, eq[S].z, eq[S].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (ATOMIC) add_atomic(Wb, wd[i]);
                else Ws[Wb + entry_pos(wd[i])] = eold[i] + entry_val(wd[i]);
            }
        }
        if constexpr (OVER)
            for (int e = ebeg[S] + 2048 + tid; e < eend[S]; e += 512) {                   // a bucket beyond the prefetch
                const unsigned wd = entp[e];
                if constexpr (ATOMIC) add_atomic(Wb, wd);
                else { const unsigned pos = Wb + entry_pos(wd); Ws[pos] = Ws[pos] + entry_val(wd); }
            }
    };

    // wave tile: all 128 rows x 32 tokens (no two waves want the same x: its fragments come straight from global memory), as
    // 2 x 8 tiles of v_mfma_f32_16x16x32_f16 (the chip holds a higher clock under this shape than under 32x32x16 at the same
    // cycles per flop). A multiply step = one 32-column half of the slab x one half of the rows: 4 W fragments, 8 MFMAs.
    floatx4 acc[NA][8];                                           // [token half][row tile of 16]
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][c][e] = 0.0f;
    const unsigned wfrag = (unsigned) (r * FP_LD + 8 * h);
    half8 wf[2][4];
    auto frags = [&](int q, unsigned wbuf, int ks) {                // step ks: columns 32 (ks >> 1) .., rows 64 (ks & 1) ..
        const _Float16 * Wb = Ws + wbuf + wfrag + (ks & 1) * 64 * FP_LD + (ks >> 1) * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) wf[q][c] = *reinterpret_cast<const half8 *>(Wb + c * 16 * FP_LD);
    };
    auto multiply = [&](int q, int ks, const half8 * xk) {          // xk: the NA fragments of this step's 32 columns
#ifndef SDH_FD_NOMFMA
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int a = 0; a < NA; ++a)
                acc[a][4 * (ks & 1) + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xk[a], wf[q][c], acc[a][4 * (ks & 1) + c], 0, 0, 0);
#endif
    };

    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    using set2 = std::integral_constant<int, 2>;
    unsigned Wcur = 0, Wnxt = FD_R * FP_LD, Wnx2 = 2 * FD_R * FP_LD;      // element offsets of the three W tiles in rotation
    int par = 0;                                                          // parity of the slab
    // slab s, on entry: LDS: Wcur = W0+D of slab s, Wnxt = W0 of slab s+1 (entries not yet added), Gs[par] = segment of slab
    // s+2; registers: sg = segment dword of slab s+3, x fragments of slabs s (set a_c), s+1, s+2, entries of slabs s+1 (set
    // c_c, added here) and s+2. The set a_c receives the entries of slab s+3 at once and its x fragments after the last
    // multiply step. The W fragments of the first multiply step are in flight.
    auto slab = [&](auto a_c, auto c_c, int s) {
        constexpr int A = decltype(a_c)::value;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            frags((ks + 1) & 1, Wcur, ks + 1);
            multiply(ks & 1, ks, &xq[A][NA * (ks >> 1)]);
            // (dev ablations: -DSDH_FP_NOSC / NOTILE / NOX drop the entries, the tile build, the x loads: results wrong, times
            // 82.6 / 75.6 / 72.3 us against 81.0 per call on a Mistral gate; all three: 65.1; no multiplies: 39.8)
#ifndef SDH_FP_NOSC
            if (ks == 0) fetchE(a_c);                                // bucket s + 3 (bounds: fetch_off of the slab before)
#endif
#ifndef SDH_FP_NOTILE
            if (ks == 0) { seg_commit(par ^ 1); seg_load(s + 4); tile_read(par, par); }
            if (ks == 1) tile_write(Wnx2, 0);
            if (ks == 2) tile_write(Wnx2, 1);
#endif
#ifndef SDH_FP_NOSC
            if (ks == 0) scatter_read(c_c, Wnxt);
            if (ks == 2) { fetch_off(s + 4); scatter(c_c, Wnxt); }
#endif
        }
        __syncthreads();
        { const unsigned t = Wcur; Wcur = Wnxt; Wnxt = Wnx2; Wnx2 = t; }
        par ^= 1;
        frags(0, Wcur, 0);                                           // the next slab's first step, behind the barrier
        multiply(1, 3, &xq[A][NA]);                                  // this slab's last: the matrix pipe works through that LDS latency
#ifndef SDH_FP_NOX
        fetchX(a_c, s + 3);
#endif
    };

    auto run_pass = [&]() {
        par = 0;
        for (int i = tid; i < n_slab; i += 512) Bn[i] = offr[i];
        seg_load(0); seg_commit(0);
        seg_load(1); seg_commit(1);
        __syncthreads();
        fetch_off(0);
        fetchE(set0{});
        tile_read(0, 0); tile_write(Wcur, 0); tile_write(Wcur, 1);
        tile_read(1, 1); tile_write(Wnxt, 0); tile_write(Wnxt, 1);
        __syncthreads();
        seg_load(2); seg_commit(0);
        scatter_read(set0{}, Wcur);
        scatter(set0{}, Wcur);
        // the loads in flight on the way into the loop, in the order the loop's slabs -3, -2, -1 would have issued them
        fetchX(set0{}, 0);
        fetch_off(1); fetchE(set1{}); fetchX(set1{}, 1);
        fetch_off(2); fetchE(set2{}); seg_load(3); fetchX(set2{}, 2);
        fetch_off(3);
        __syncthreads();
        frags(0, Wcur, 0);
        int s = 0;
        for (; s + 2 < n_slab; s += 3) {                    // no branch inside the triple
            slab(set0{}, set1{}, s);
            slab(set1{}, set2{}, s + 1);
            slab(set2{}, set0{}, s + 2);
        }
        if (s < n_slab) slab(set0{}, set1{}, s);
        if (s + 1 < n_slab) slab(set1{}, set2{}, s + 1);
    };
    run_pass();
    floatx4 accg[DUAL ? NA : 1][8];
    if constexpr (DUAL) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                accg[a][c] = acc[a][c];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][c][e] = 0.0f;
            }
        __syncthreads();                                            // the first pass's last LDS reads are behind us
        offr = second.off + (size_t) blockIdx.x * n_g;
        entp = second.ent;
        zw = second.z16 + z_off;
        Gb = (const char *) (second.gp + (size_t) 8 * b * L);
        run_pass();
    }

    // C[row = token][col = o] of a 16 x 16 tile: row = 4 h + e, col = r
    const size_t col_off = TALL ? 0 : (size_t) b * L;
    int bad = 0, bad2 = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int orow = o_tile + c * 16 + r;
        if (orow >= n_out) continue;
        const float sc = (TALL && rs) ? rs[orow] : 1.0f;
        const float sc2 = (DUAL && second.rs) ? second.rs[orow] : 1.0f;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t t = tok0 + a * 16 + 4 * h + e;
                if (t >= T) continue;
                float v = acc[a][c][e];
                if constexpr (DUAL) {                                      // each node's guard, then the activation (store_output)
                    float g = accg[a][c][e];
                    if (sc != 1.0f) g *= sc;
                    if (!__builtin_isfinite(g)) { ++bad; if (zero_nf) g = 0.0f; }
                    if (sc2 != 1.0f) v *= sc2;
                    if (!__builtin_isfinite(v)) { ++bad2; if (zero_nf) v = 0.0f; }
                    v = glu_value(glu_act, g, v, glu_tab);
                } else if (TALL) {
                    if (sc != 1.0f) v *= sc;                                           // :645
                    if (!__builtin_isfinite(v)) { ++bad; if (zero_nf) v = 0.0f; }
                    if (glu_gate) v = glu_value(glu_act, glu_gate[t * glu_ldg + orow], v, glu_tab);   // after the node's guard, as store_output
                }
                out[t * ldo + col_off + orow] = v;
            }
    }
    if (DUAL && bad2) atomicAdd(second.nf, bad2);
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

// x image: q order: [T_pad][round64(n_in) + 64]; j order (tiled): tall B images of T_pad x L, wide one of T_pad x B * L
static void fold_x_layout(const DeviceWeight & w, int64_t T, bool perm, int64_t & T_pad, int64_t & ld16, int64_t & zcopy, size_t & bytes) {
    T_pad = (T + FD_N - 1) / FD_N * FD_N;
    if (!perm) { ld16 = (w.n_in + FD_K - 1) / FD_K * FD_K + FD_K; zcopy = 0; bytes = (size_t) T_pad * (size_t) ld16 * 2; return; }
    ld16 = w.tall ? w.L : w.B * w.L;                             // columns of one image (tiled, not a row stride: see fold_z16_kernel)
    zcopy = w.tall ? T_pad * ld16 : 0;
    bytes = (size_t) T_pad * (size_t) ld16 * 2 * (size_t) (w.tall ? w.B : 1);
}

size_t w0_fold_ws_bytes(const DeviceWeight & w, int64_t T) {
    int64_t T_pad, ld16, zcopy; size_t xb;
    fold_x_layout(w, T, w.fold_ent ? w.fold_perm : fold_perm_ok(w), T_pad, ld16, zcopy, xb);
    const size_t part = w.tall ? 0 : (size_t) T * (size_t) w.B * (size_t) w.L * 4;
    return ((part + 255) & ~(size_t) 255) + xb;
}

// The whole op for T tokens: x16 pre-pass, the folded contraction, (wide) the block-order reduction. ws: w0_fold_ws_bytes.
// Returns 1 when the payload does not fit the scheme (no closed-form W0, L < 128).
int launch_w0_fold(DeviceWeight & w, const float * x, int64_t ldx, int64_t T, void * ws, float * y, int64_t ldy,
                   int * nf_counter, bool zero_nf, hipStream_t s,
                   const float * glu_gate, int64_t glu_ldg, int glu_act, const uint16_t * glu_tab) {
    if (!w.w0_closed_form || w.L < FD_R || (w.L % FD_R) != 0) return 1;
    if (glu_gate && !(w.tall && (w.fold_ent ? w.fold_perm : fold_perm_ok(w)))) return 1;      // the activation epilogue: permuted kernel, tall tensors
    if (T <= 0) return SDH_OK;
    int rc = fold_ensure(w, s);
    if (rc != SDH_OK) return rc;
    const bool perm = w.fold_perm;
    int64_t T_pad, ld16, zcopy; size_t xb;
    fold_x_layout(w, T, perm, T_pad, ld16, zcopy, xb);
    const size_t part_bytes = w.tall ? 0 : (((size_t) T * (size_t) w.B * (size_t) w.L * 4 + 255) & ~(size_t) 255);
    float * part = (float *) ws;
    _Float16 * x16 = (_Float16 *) ((unsigned char *) ws + part_bytes);
    const size_t lds = perm ? (size_t) 3 * FD_R * FP_LD * 2 + (size_t) 2 * FD_GS * 2 + (size_t) (w.L / FD_K) * sizeof(int2)
                            : (size_t) (3 * FD_R + 2 * FD_N) * FD_LD * 2 + (size_t) 2 * FD_GS * 2 + (size_t) (w.L / FD_K) * sizeof(int2);
    if (lds > 160 * 1024) return 1;
    static const bool attr_ok = [] {
        bool ok = true;
        auto set = [&](const void * k) { ok &= hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
#define SDH_FD_SET4(TALL_, SC_) set((const void *) w0d_mfma_kernel<TALL_, SC_, false, false>); set((const void *) w0d_mfma_kernel<TALL_, SC_, false, true>); \
        set((const void *) w0d_mfma_kernel<TALL_, SC_, true, false>); set((const void *) w0d_mfma_kernel<TALL_, SC_, true, true>)
        SDH_FD_SET4(true, true); SDH_FD_SET4(true, false); SDH_FD_SET4(false, true); SDH_FD_SET4(false, false);
#undef SDH_FD_SET4
#define SDH_FP_SET4(TALL_, NT_) set((const void *) w0p_mfma_kernel<TALL_, false, false, NT_>); set((const void *) w0p_mfma_kernel<TALL_, false, true, NT_>); \
        set((const void *) w0p_mfma_kernel<TALL_, true, false, NT_>); set((const void *) w0p_mfma_kernel<TALL_, true, true, NT_>)
        SDH_FP_SET4(true, 32); SDH_FP_SET4(false, 32); SDH_FP_SET4(true, 16); SDH_FP_SET4(false, 16);
#undef SDH_FP_SET4
        set((const void *) fold_z16_kernel<true>); set((const void *) fold_z16_kernel<false>);
        return ok;
    }();
    if (!attr_ok) return 1;                                     // a device without the LDS for it: the caller's other forms
    if (perm) {
        const size_t zl = ((size_t) w.n_in * 2 * FZ_T + 15) & ~(size_t) 15;
        const dim3 zg((unsigned) (T_pad / FZ_T), 2);
        if (w.tall) hipLaunchKernelGGL(fold_z16_kernel<true>, zg, dim3(FZ_NTH), zl, s, x, ldx, T, (int) w.n_in, (int) w.L, (int) w.B, w.fold_p16, x16, (int) ld16, zcopy);
        else        hipLaunchKernelGGL(fold_z16_kernel<false>, zg, dim3(FZ_NTH), zl, s, x, ldx, T, (int) w.n_in, (int) w.L, (int) w.B, w.fold_p16, x16, (int) ld16, zcopy);
        SDH_HIP_CHECK(hipGetLastError());
    } else if ((rc = launch_w0_x16(x, ldx, T, (int) w.n_in, x16, ld16, T_pad, s)) != SDH_OK) return rc;
    const dim3 grid((unsigned) ((w.n_out + FD_R - 1) / FD_R), (unsigned) (T_pad / FD_N), (unsigned) (w.tall ? 1 : w.B));
    if (grid.y > 65535 || grid.z > 65535) { set_error("w0 fold: token chunk too large"); return SDH_ERR_UNSUPPORTED; }
    const bool scaled = fold_scaled(w);
    // up to 256 tokens the permuted kernel takes token tiles of 128 (16 tokens per wave): twice the workgroups, each with
    // half the multiplies, where tiles of 256 would leave half the CUs (or more) without a workgroup
    static const bool no_half = getenv("SDH_FOLD_NO_HALF_TILES") != nullptr;      // A/B switch
    const bool half_tiles = perm && T <= 256 && !no_half;
    const dim3 pgrid(grid.x, half_tiles ? (unsigned) ((T + 127) / 128) : grid.y, grid.z);
    float * out = w.tall ? y : part;
    const int64_t ldo = w.tall ? ldy : (int64_t) w.B * w.L;
#define SDH_FD_ARGS x16, ld16, T, (int) w.n_in, (int) w.n_out, (int) w.L, w.d1, w.d3, w.G, w.inv, w.fold_ent, w.fold_off, w.fold_ng, \
        w.fold_spb, w.rs, out, ldo, nf_counter, zero_nf ? 1 : 0
#define SDH_FP_ARGS x16, zcopy, T, (int) w.n_out, (int) w.L, (int) ld16, w.fold_gp, w.fold_ent, w.fold_off, w.fold_ng, \
        w.fold_spb, w.rs, out, ldo, nf_counter, zero_nf ? 1 : 0, glu_gate, glu_ldg, glu_act, glu_tab, FoldSecond{}
#define SDH_FD_LAUNCH(TALL_, SC_, AT_) do { if (w.fold_over) hipLaunchKernelGGL((w0d_mfma_kernel<TALL_, SC_, AT_, true>), grid, dim3(512), lds, s, SDH_FD_ARGS); \
        else hipLaunchKernelGGL((w0d_mfma_kernel<TALL_, SC_, AT_, false>), grid, dim3(512), lds, s, SDH_FD_ARGS); } while (0)
#define SDH_FD_LAUNCH2(TALL_, SC_) do { if (w.fold_dup) SDH_FD_LAUNCH(TALL_, SC_, true); else SDH_FD_LAUNCH(TALL_, SC_, false); } while (0)
#define SDH_FP_LAUNCH0(TALL_, AT_, OV_) do { if (half_tiles) hipLaunchKernelGGL((w0p_mfma_kernel<TALL_, AT_, OV_, 16>), pgrid, dim3(512), lds, s, SDH_FP_ARGS); \
        else hipLaunchKernelGGL((w0p_mfma_kernel<TALL_, AT_, OV_, 32>), pgrid, dim3(512), lds, s, SDH_FP_ARGS); } while (0)
#define SDH_FP_LAUNCH(TALL_, AT_) do { if (w.fold_over) SDH_FP_LAUNCH0(TALL_, AT_, true); else SDH_FP_LAUNCH0(TALL_, AT_, false); } while (0)
#define SDH_FP_LAUNCH2(TALL_) do { if (w.fold_dup) SDH_FP_LAUNCH(TALL_, true); else SDH_FP_LAUNCH(TALL_, false); } while (0)
    if (perm)        { if (w.tall) SDH_FP_LAUNCH2(true); else SDH_FP_LAUNCH2(false); }
    else if (w.tall) { if (scaled) SDH_FD_LAUNCH2(true, true); else SDH_FD_LAUNCH2(true, false); }
    else             { if (scaled) SDH_FD_LAUNCH2(false, true); else SDH_FD_LAUNCH2(false, false); }
#undef SDH_FD_ARGS
#undef SDH_FP_ARGS
#undef SDH_FD_LAUNCH2
#undef SDH_FD_LAUNCH
#undef SDH_FP_LAUNCH2
#undef SDH_FP_LAUNCH0
#undef SDH_FP_LAUNCH
    SDH_HIP_CHECK(hipGetLastError());
    if (!w.tall) {
        const int64_t n = T * w.n_out;
        hipLaunchKernelGGL(fold_reduce_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, part, (int) w.B, (int) w.L, (int) w.n_out, T,
                           w.rs, y, ldy, nf_counter, zero_nf ? 1 : 0);
        SDH_HIP_CHECK(hipGetLastError());
    }
    return SDH_OK;
}

// Gate and up of a gated FFN as ONE two-pass launch: every workgroup contracts its (row tile, token tile) for gate, keeps the
// accumulators, contracts the same tile for up and writes act(gate) * up — no gate output in memory, no second launch. Tall
// tensors of one shape, both in the permuted order; the caller sends more than 256 tokens per call (fewer: the two-launch form
// with token tiles of 128). Each tensor's permuted x image goes to its own workspace (w0_fold_ws_bytes). Arithmetic per tensor
// and the activation are those of the two-launch form: the same bits. Returns 1 when the pair does not fit (before any launch).
int launch_w0_fold_pair(DeviceWeight & g, DeviceWeight & u, const float * x, int64_t ldx, int64_t T, float * h, int64_t ldh,
                        int * nf_g, int * nf_u, bool zero_nf, int act, const uint16_t * tab, hipStream_t s) {
    auto fits = [](const DeviceWeight & w) { return w.w0_closed_form && w.L >= FD_R && (w.L % FD_R) == 0 && w.tall && w0_fold_permuted(w); };
    if (!fits(g) || !fits(u) || g.n_in != u.n_in || g.n_out != u.n_out || g.L != u.L || g.B != u.B) return 1;
    if (T <= 0) return SDH_OK;
    int rc = fold_ensure(g, s);
    if (rc == SDH_OK) rc = fold_ensure(u, s);
    if (rc != SDH_OK) return rc;
    if (!g.fold_perm || !u.fold_perm || g.fold_ng != u.fold_ng) return 1;
    int64_t T_pad, kz, zcopy; size_t xb;
    fold_x_layout(g, T, true, T_pad, kz, zcopy, xb);
    const size_t lds = (size_t) 3 * FD_R * FP_LD * 2 + (size_t) 2 * FD_GS * 2 + (size_t) (g.L / FD_K) * sizeof(int2);
    if (lds > 160 * 1024) return 1;
    static const bool attr_ok = [] {
        bool ok = true;
        auto set = [&](const void * k) { ok &= hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
        set((const void *) w0p_mfma_kernel<true, false, false, 32, true>); set((const void *) w0p_mfma_kernel<true, false, true, 32, true>);
        set((const void *) w0p_mfma_kernel<true, true, false, 32, true>); set((const void *) w0p_mfma_kernel<true, true, true, 32, true>);
        set((const void *) fold_z16_kernel<true>);
        return ok;
    }();
    if (!attr_ok) return 1;
    _Float16 * zg = (_Float16 *) g.ws, * zu = (_Float16 *) u.ws;
    const size_t zl = ((size_t) g.n_in * 2 * FZ_T + 15) & ~(size_t) 15;
    const dim3 zgrid((unsigned) (T_pad / FZ_T), 2);
    static const bool one_prepass = getenv("SDH_FOLD_TWO_PREPASSES") == nullptr;       // A/B switch
    if (one_prepass) {
        const dim3 zgrid2(zgrid.x, zgrid.y, 2);
        hipLaunchKernelGGL(fold_z16_kernel<true>, zgrid2, dim3(FZ_NTH), zl, s, x, ldx, T, (int) g.n_in, (int) g.L, (int) g.B, g.fold_p16, zg, (int) kz, zcopy,
                           u.fold_p16, zu);
    } else {
        hipLaunchKernelGGL(fold_z16_kernel<true>, zgrid, dim3(FZ_NTH), zl, s, x, ldx, T, (int) g.n_in, (int) g.L, (int) g.B, g.fold_p16, zg, (int) kz, zcopy);
        hipLaunchKernelGGL(fold_z16_kernel<true>, zgrid, dim3(FZ_NTH), zl, s, x, ldx, T, (int) u.n_in, (int) u.L, (int) u.B, u.fold_p16, zu, (int) kz, zcopy);
    }
    SDH_HIP_CHECK(hipGetLastError());
    const dim3 grid((unsigned) ((g.n_out + FD_R - 1) / FD_R), (unsigned) (T_pad / FD_N), 1);
    if (grid.y > 65535) { set_error("w0 fold: token chunk too large"); return SDH_ERR_UNSUPPORTED; }
    const FoldSecond sec{zu, u.fold_gp, u.fold_ent, u.fold_off, u.rs, nf_u};
    const bool atomic = g.fold_dup || u.fold_dup, over = g.fold_over || u.fold_over;     // the general variants are correct for either tensor
#define SDH_FP2_ARGS zg, zcopy, T, (int) g.n_out, (int) g.L, (int) kz, g.fold_gp, g.fold_ent, g.fold_off, g.fold_ng, g.fold_spb, g.rs, h, ldh, \
        nf_g, zero_nf ? 1 : 0, (const float *) nullptr, (int64_t) 0, act, tab, sec
    if (atomic) { if (over) hipLaunchKernelGGL((w0p_mfma_kernel<true, true, true, 32, true>), grid, dim3(512), lds, s, SDH_FP2_ARGS);
                  else      hipLaunchKernelGGL((w0p_mfma_kernel<true, true, false, 32, true>), grid, dim3(512), lds, s, SDH_FP2_ARGS); }
    else        { if (over) hipLaunchKernelGGL((w0p_mfma_kernel<true, false, true, 32, true>), grid, dim3(512), lds, s, SDH_FP2_ARGS);
                  else      hipLaunchKernelGGL((w0p_mfma_kernel<true, false, false, 32, true>), grid, dim3(512), lds, s, SDH_FP2_ARGS); }
#undef SDH_FP2_ARGS
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

} // namespace sdh
