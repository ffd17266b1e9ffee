// sdh_scan.hip — the builder's candidate evaluation as a GPU pipeline (gfx950).
//
// One candidate = one run_trial of the reference builder (tools/seeddelta-build/sd_build.cpp:463-800)
// on one dense FFN weight: [fit the XOR-circulant base] -> per output row: residual, top-K / top-block
// selection, least-squares row scale -> weight-space metrics on sampled rows -> activation-space
// metrics on Gaussian x -> [FFN proxy]. The dense weight stays in HBM in its stored type (Q8_0 split
// into an int8 plane and an fp16 scale plane at upload) and is dequantised on the fly with the
// reference's rule (q * fp32(d)), so every value a kernel sees is the value the CPU builder sees.
//
// What is exact and what is not:
//   * base fit: each thread owns one XOR bin u and adds its samples in sample order — the same
//     order the CPU loop produces for that bin — so sum / sum2 / wgt, hence h2, d2 and the chosen
//     perm, are bit-identical (sd_eval.cpp:486-555);
//   * selection: keys are the fp32 scores' bit patterns (COO) or sequentially accumulated double
//     block energies (block), so the selected sets are identical; equal scores are ordered by
//     ascending index (the reference leaves tie order to std::sort);
//   * sums of doubles over a whole row (row-scale LSQ, metric norms and dots) are tree-reduced, so
//     they agree with the CPU's sequential sums to ~1e-15 relative, not bit-for-bit.
//
// Host side: libstdc++'s std::mt19937 / std::shuffle / std::normal_distribution<float> — the same
// library routines the reference calls, consumed in the same order.
#include "sdh_internal.h"
#include "sdh_rng.h"
#include <atomic>
#include <array>
#include <chrono>
#include <map>
#include <memory>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <vector>

#include <hip/hip_fp16.h>

#pragma clang fp contract(off)

namespace sdh {

// ---- dense weight on device ----------------------------------------------------------------------
struct DenseView {
    const void *    data;   // F32 / F16 rows, contiguous [n_out][n_in]
    const int8_t *  q;      // Q8_0: [n_out][n_in]
    const __half *  d;      // Q8_0: [n_out][n_in/32]
    int     type;
    int64_t n_in, n_out;
};

struct DeviceDense {
    DenseView v{};
    void * alloc0 = nullptr, * alloc1 = nullptr;
    int64_t stored_bytes = 0;
    uint64_t serial = next_serial();    // identity for caches (an address can be reused by a later tensor)
    static uint64_t next_serial() { static std::atomic<uint64_t> n{0}; return ++n; }
};

// The stored type is a template parameter of every kernel that reads the weight: with a run-time
// branch on W.type inside the accessor each load sits in a basic block of its own and hipcc waits for
// it before issuing the next (one memory round trip per element instead of one per batch).
template <int DT>
__device__ __forceinline__ float dense_at(const DenseView & W, int64_t row, int64_t i) {
    if constexpr (DT == SDH_TYPE_Q8_0) {
        const float d = __half2float(W.d[row * (W.n_in >> 5) + (i >> 5)]);
        return (float) W.q[row * W.n_in + i] * d;
    } else if constexpr (DT == SDH_TYPE_F16) {
        return __half2float(((const __half *) W.data)[row * W.n_in + i]);
    } else {
        return ((const float *) W.data)[row * W.n_in + i];
    }
}

#define SDH_LAUNCH_DT(VIEW, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                              \
    do {                                                                                                        \
        switch ((VIEW).type) {                                                                                  \
            case SDH_TYPE_Q8_0: hipLaunchKernelGGL((KERNEL<SDH_TYPE_Q8_0>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break; \
            case SDH_TYPE_F16:  hipLaunchKernelGGL((KERNEL<SDH_TYPE_F16>),  GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break; \
            default:            hipLaunchKernelGGL((KERNEL<SDH_TYPE_F32>),  GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break; \
        }                                                                                                       \
    } while (0)

struct BaseView {            // fitted base on device, [B][L]; L is a power of two (fit_dims)
    const float * h2;
    const int *   inv;
    int L, B, tall, log2L;
};

// W0[row, i] of the fitted base: h2_b[(p ^ inv_b[q]) & (L-1)]  (sd_eval.cpp:506,528,714-716)
__device__ __forceinline__ int base_inv_index(const BaseView & a, int row, int i) {
    return a.tall ? ((row >> a.log2L) << a.log2L) + i : i;      // b*L + q
}
__device__ __forceinline__ int base_h2_index(const BaseView & a, int row, int i, int j) {
    const int b = a.tall ? row >> a.log2L : i >> a.log2L;
    return (b << a.log2L) + ((row ^ j) & (a.L - 1));             // (p ^ j) & (L-1): the bits of b drop out of p
}
__device__ __forceinline__ float base_weight(const BaseView & a, int64_t row, int64_t i) {
    const int j = a.inv[base_inv_index(a, (int) row, (int) i)];
    return a.h2[base_h2_index(a, (int) row, (int) i, j)];
}


// ---- block-wide helpers (256 threads = 4 waves) -----------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double * scratch /* >= 4 doubles */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// exclusive prefix over the block of one int per thread; total returned through *total
__device__ __forceinline__ int block_excl_scan(int v, int * scratch /* >= 5 ints */, int * total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += scratch[w];
    *total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    return base + inc - v;
}

// ---- residual row into LDS (both selection kernels) -------------------------------------------------
// r[i] = W[row,i] - W0[row,i] (or W[row,i]); dot / nn: this thread's partial sums of w*bw*s^2 and bw^2*s^2
// over its elements i = tid, tid+256, ... in ascending order. Eight elements per thread per round, every
// load of a round issued before the first use and no branch around a load (clamped addresses): three
// memory round trips per round (weights + inv, then the h2 gather) instead of three per element.
template <int DT>
__device__ __forceinline__ void load_residual_row(const DenseView & W, const BaseView & base, bool has_base,
                                                  const float * __restrict__ ws, int64_t row, float * r,
                                                  double & dot, double & nn) {
    constexpr int U = 8;
    const int n_in = (int) W.n_in, tid = threadIdx.x;
    for (int i0 = 0; i0 < n_in; i0 += 256 * U) {
        int ic[U];
        float w[U], bw[U], s1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256 + tid;
            ic[u] = i < n_in ? i : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = dense_at<DT>(W, row, ic[u]);
        if (has_base) {
            int j[U];
#pragma unroll
            for (int u = 0; u < U; ++u) j[u] = base.inv[base_inv_index(base, (int) row, ic[u])];
#pragma unroll
            for (int u = 0; u < U; ++u) bw[u] = base.h2[base_h2_index(base, (int) row, ic[u], j[u])];
        }
        if (ws) {
#pragma unroll
            for (int u = 0; u < U; ++u) s1[u] = ws[ic[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256 + tid;
            if (i < n_in) {
                if (has_base) {
                    r[i] = w[u] - bw[u];
                    const double sd = ws ? (double) s1[u] : 1.0;
                    dot += (double) w[u] * (double) bw[u] * sd * sd;
                    nn  += (double) bw[u] * (double) bw[u] * sd * sd;
                } else {
                    r[i] = w[u];
                }
            }
        }
    }
}

// ---- kernel: base fit accumulation ---------------------------------------------------------------------
// grid (L/256, n_jobs); job = (base block, perm trial). Thread u adds, for every sampled row in sample
// order, the one element of that row that falls into XOR bin u: j = u ^ p, src = in0 + perm[j].
struct FitJob { int b; int rows_off; int n_rows; int perm_off; };

template <int DT>
__global__ __launch_bounds__(256)
void scan_fit_kernel(DenseView W, int L, int tall, const FitJob * __restrict__ jobs, const int * __restrict__ rows,
                     const int * __restrict__ perms, const float * __restrict__ ws, double * __restrict__ out /* [n_jobs][3][L] */) {
    extern __shared__ int perm_s[];
    const FitJob job = jobs[blockIdx.y];
    for (int i = threadIdx.x; i < L; i += 256) perm_s[i] = perms[job.perm_off + i];
    __syncthreads();
    const int u = blockIdx.x * 256 + threadIdx.x;
    const int64_t in0 = tall ? 0 : (int64_t) job.b * L;
    const int64_t in1 = tall ? W.n_in : min((int64_t) W.n_in, in0 + L);
    // Branch-free so that the compiler keeps several rows' gathers in flight: an out-of-range source
    // contributes +0.0, which leaves every partial sum bit-for-bit unchanged (x + 0.0 == x, and the sums
    // start at +0.0), so the bin still accumulates its valid samples in sample order.
    double s = 0.0, s2 = 0.0, n = 0.0;
#pragma unroll 8
    for (int si = 0; si < job.n_rows; ++si) {
        const int row = rows[job.rows_off + si];
        const int p = tall ? row - job.b * L : row;
        const int j = (u ^ p) & (L - 1);
        const int64_t src = in0 + perm_s[j];
        const bool ok = src >= in0 && src < in1;
        const int64_t srcc = ok ? src : in0;
        const double v = ok ? (double) dense_at<DT>(W, row, srcc) : 0.0;
        if (ws) {
            const double w1 = ok ? (double) ws[srcc] : 0.0;
            const double ww = w1 * w1;
            s  += ww * v;
            s2 += ww * v * v;
            n  += ww;
        } else {            // ww == 1: ww*v == v, ww*v*v == v*v exactly
            s  += v;
            s2 += v * v;
            n  += ok ? 1.0 : 0.0;
        }
    }
    double * o = out + (size_t) blockIdx.y * 3 * L;
    o[u] = s; o[L + u] = s2; o[2 * L + u] = n;
}

// ---- kernel: base fit accumulation, rows staged through LDS (Q8_0) ---------------------------------------
// Same arithmetic and the same per-bin sample order as scan_fit_kernel, but instead of one random
// one-byte gather per (bin, sample) — 64 cache lines per wave instruction, ~134 M L2 requests per fit at
// Mistral dims — a workgroup copies the sampled row's block [in0, in0+L) into LDS with 16-byte loads
// (8 rows per stage, register-prefetched) and its threads pick their elements there. The workgroups
// of a job re-read the same 4 KB row from L2, coalesced.
constexpr int FIT_ROWS_PER_STAGE = 8;

// FT threads per workgroup, one bin each: the staging of a row is shared by FT bins, so FT = 1024 reads
// each sampled row from L2 L/1024 times instead of L/256 times and keeps 16 waves per CU behind the
// LDS round trips of a stage.
// FAST: no imatrix weighting; a job whose block lies wholly inside the inputs (span == L) then takes the short body:
// every sample of the job is valid for every bin (the permutation maps into [0, L)), the per-sample validity selects
// drop out and the weight sum is the sample count — written once at the end, exactly what n += 1.0 per sample gives.
// Samples past the end of the job read the zero-filled tail of the stage: +0.0 to both sums (exact).
template <int FT, bool FAST>
__global__ __launch_bounds__(FT)
void scan_fit_lds_q8_kernel(DenseView W, int L, int tall, const FitJob * __restrict__ jobs, const int * __restrict__ rows,
                            const int * __restrict__ perms, const float * __restrict__ ws, double * __restrict__ out /* [n_jobs][3][L] */,
                            int xcd_per, int jobs_per_block) {
    extern __shared__ __align__(16) unsigned char fit_smem[];
    constexpr int R = FIT_ROWS_PER_STAGE;
    // Which (job, bin chunk) this workgroup takes. The jobs of one base block (its permutation trials) and their bin
    // chunks all stage the SAME sampled row segments; workgroups are dealt to the 8 XCDs round-robin in launch order,
    // so with the plain (chunk, job) = blockIdx mapping the 16 workgroups of a block sat on all 8 XCDs and every row
    // segment was fetched into 8 L2s (measured: 8.5x the algorithmic bytes). xcd_per > 0: the workgroups of a block
    // go to xcd_per neighbouring XCDs (Mistral: 4 blocks x 2 XCDs), so a segment crosses the fabric xcd_per times.
    int job_id = (int) blockIdx.y, chunk_id = (int) blockIdx.x;
    if (xcd_per > 0) {
        const int lin = (int) (blockIdx.x + gridDim.x * blockIdx.y), chunks = (int) gridDim.x;
        const int xcd = lin & 7, slot = lin >> 3, M = jobs_per_block * chunks;
        const int item = (xcd % xcd_per) * (M / xcd_per) + slot;
        job_id = (xcd / xcd_per) * jobs_per_block + item / chunks;
        chunk_id = item % chunks;
    }
    const FitJob job = jobs[job_id];
    const int n_rows_pad = (job.n_rows + 3) & ~3;
    int *     perm_s = (int *) fit_smem;                               // [L]
    int *     rows_s = perm_s + L;                                     // [n_rows]: the job's sampled rows
    int8_t *  qs     = (int8_t *) (rows_s + n_rows_pad);               // [2][R][L]
    __half *  ds     = (__half *) (qs + 2 * R * L);                    // [2][R][L/32]
    float *   wss    = (float *) (ds + 2 * R * (L >> 5));              // [L] when ws
    const int tid = threadIdx.x;
    const int lg = 31 - __clz(L);                                      // L is a power of two
    const int64_t in0 = tall ? 0 : (int64_t) job.b * L;
    const int64_t in1 = tall ? W.n_in : min((int64_t) W.n_in, in0 + L);
    const int span = (int) (in1 - in0);                                // valid elements of the block, a multiple of 32
    for (int i = tid; i < L; i += FT) {
        perm_s[i] = perms[job.perm_off + i];
        if (ws) wss[i] = i < span ? ws[in0 + i] : 0.0f;
    }
    for (int i = tid; i < job.n_rows; i += FT) rows_s[i] = rows[job.rows_off + i];
    __syncthreads();
    const int u = chunk_id * FT + tid;
    const int nd = L >> 5;

    // a stage = R rows of the job: R*L/16 16-byte pieces and R*L/32 scales, fetched into registers
    // (fetch), consumed from LDS one stage later (commit): the fetch of stage k+1 is in flight while
    // stage k is being accumulated. L <= 4096 (host check): at most R*256/FT pieces and R*128/FT scales per thread.
    constexpr int MAXP = R * 256 / FT, MAXD = (R * 128 + FT - 1) / FT;
    const int n_piece = R * (L >> 4), n_scale = R * nd;
    // two register sets: the fetch of stage k+2 is issued at the start of stage k and committed at the end of
    // stage k+1, so a stage no longer waits out one full memory round trip (one workgroup per CU at this LDS
    // footprint: nothing else hides it)
    uint4 pq[2][MAXP];
    __half pd[2][MAXD];
    auto fetch = [&](auto set_c, int s0) {
        constexpr int S = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int e = tid + FT * i;
            pq[S][i] = make_uint4(0, 0, 0, 0);
            if (e < n_piece) {
                const int k = e >> (lg - 4), c = e & ((L >> 4) - 1);
                if (s0 + k < job.n_rows && c * 16 < span) {
                    const int row = rows_s[s0 + k];
                    pq[S][i] = *reinterpret_cast<const uint4 *>(W.q + (int64_t) row * W.n_in + in0 + c * 16);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXD; ++i) {
            const int e = tid + FT * i;
            pd[S][i] = __float2half(0.0f);
            if (e < n_scale) {
                const int k = e >> (lg - 5), c = e & (nd - 1);
                if (s0 + k < job.n_rows && c * 32 < span) {
                    const int row = rows_s[s0 + k];
                    pd[S][i] = W.d[(int64_t) row * (W.n_in >> 5) + (in0 >> 5) + c];
                }
            }
        }
    };
    auto commit = [&](auto set_c) {                 // register set S holds a stage of parity S: LDS buffer S
        constexpr int S = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int e = tid + FT * i;
            if (e < n_piece) *reinterpret_cast<uint4 *>(qs + (size_t) S * R * L + (size_t) e * 16) = pq[S][i];
        }
#pragma unroll
        for (int i = 0; i < MAXD; ++i) {
            const int e = tid + FT * i;
            if (e < n_scale) ds[(size_t) S * R * nd + e] = pd[S][i];
        }
    };

    double s = 0.0, s2 = 0.0, n = 0.0;
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    // one stage: parity P = (s0 / R) & 1 is its LDS buffer and the register set the stage after next is fetched into
    const bool full = FAST && span == L;                               // uniform over the workgroup
    auto stage = [&](auto par_c, auto other_c, auto fast_c, int s0) {
        constexpr int buf = decltype(par_c)::value;
        if (s0 + 2 * R < job.n_rows) fetch(par_c, s0 + 2 * R);
        // the R samples of the stage: all LDS reads first (branch-free, clamped), then the sums in sample order;
        // a sample past the end of the job, like one outside the block, adds +0.0 (exact)
        if constexpr (decltype(fast_c)::value) {
            int srcv[R];
            float qv[R], dv[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int row = rows_s[min(s0 + k, job.n_rows - 1)];
                const int p = tall ? row - job.b * L : row;
                srcv[k] = perm_s[(u ^ p) & (L - 1)];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                dv[k] = __half2float(ds[((size_t) buf * R + k) * nd + (srcv[k] >> 5)]);
                qv[k] = (float) qs[((size_t) buf * R + k) * L + srcv[k]];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double v = (double) (qv[k] * dv[k]);                 // dense_at<Q8_0>: (float) q * d
                s  += v;
                s2 += v * v;
            }
            if (s0 + R < job.n_rows) commit(other_c);
            __syncthreads();
            return;
        }
        int srcv[R];
        bool okv[R];
        float qv[R], dv[R], wv1[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool valid = s0 + k < job.n_rows;
            const int row = rows_s[valid ? s0 + k : 0];
            const int p = tall ? row - job.b * L : row;
            srcv[k] = perm_s[(u ^ p) & (L - 1)];                       // offset inside the block
            okv[k] = valid;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            okv[k] = okv[k] && srcv[k] >= 0 && srcv[k] < span;
            const int sc = okv[k] ? srcv[k] : 0;
            dv[k] = __half2float(ds[((size_t) buf * R + k) * nd + (sc >> 5)]);
            qv[k] = (float) qs[((size_t) buf * R + k) * L + sc];
            wv1[k] = ws ? wss[sc] : 1.0f;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const float wv = qv[k] * dv[k];                            // dense_at<Q8_0>: (float) q * d
            const double v = okv[k] ? (double) wv : 0.0;
            if (ws) {
                const double w1 = okv[k] ? (double) wv1[k] : 0.0;
                const double ww = w1 * w1;
                s  += ww * v;
                s2 += ww * v * v;
                n  += ww;
            } else {
                s  += v;
                s2 += v * v;
                n  += okv[k] ? 1.0 : 0.0;
            }
        }
        if (s0 + R < job.n_rows) commit(other_c);                      // the next stage, fetched one stage ago
        __syncthreads();
    };
    fetch(set0{}, 0);
    commit(set0{});
    if (R < job.n_rows) fetch(set1{}, R);
    __syncthreads();
    if (full) {
        for (int s0 = 0; s0 < job.n_rows; s0 += 2 * R) {
            stage(set0{}, set1{}, std::true_type{}, s0);
            if (s0 + R < job.n_rows) stage(set1{}, set0{}, std::true_type{}, s0 + R);
        }
        n = (double) job.n_rows;
    } else {
        for (int s0 = 0; s0 < job.n_rows; s0 += 2 * R) {
            stage(set0{}, set1{}, std::false_type{}, s0);
            if (s0 + R < job.n_rows) stage(set1{}, set0{}, std::false_type{}, s0 + R);
        }
    }
    double * o = out + (size_t) job_id * 3 * L;
    o[u] = s; o[L + u] = s2; o[2 * L + u] = n;
}

// ---- kernels: the fit's tail on the device (sd_eval.cpp:600-651) -----------------------------------------
// The host used to fetch the [n_jobs][3][L] sums, pick the best permutation trial per block and upload h2,
// d2, perm1 and its inverse: one synchronisation in the middle of every candidate. The same arithmetic here:
//   scan_fit_sse_kernel   grid n_jobs: sse_j = sum_u max(sm2 - sm^2/wg, 0) over bins with wg > 0, the terms
//                         computed by all threads, added by ONE thread in bin order (the host's order);
//   scan_fit_pick_kernel  grid B: first trial with the strictly smallest sse (trial order), h2 = (float)(sm/wg),
//                         d2 = FWHT(h2)/L with the host's butterfly order (independent pairs per stage: the
//                         values do not depend on which thread forms them), perm1, its inverse, d1 = 1.
__global__ __launch_bounds__(256)
void scan_fit_sse_kernel(const double * __restrict__ sums /* [n_jobs][3][L] */, int L, double * __restrict__ sse /* [n_jobs] */) {
    extern __shared__ __align__(16) unsigned char fit_tail_smem[];
    double * term = (double *) fit_tail_smem;                           // [L]
    const double * sm = sums + (size_t) blockIdx.x * 3 * L, * sm2 = sm + L, * wg = sm2 + L;
    for (int u = threadIdx.x; u < L; u += 256) {
        double t = 0.0;                                                 // a bin without samples adds +0.0: exact
        if (wg[u] > 0.0) t = fmax(sm2[u] - (sm[u] * sm[u]) / wg[u], 0.0);
        term[u] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int u = 0; u < L; u += 8) {                                // L is a power of two >= 256
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = term[u + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += t[k];
        }
        sse[blockIdx.x] = acc;
    }
}

__global__ __launch_bounds__(256)
void scan_fit_pick_kernel(const double * __restrict__ sums, const double * __restrict__ sse, const FitJob * __restrict__ jobs, int n_jobs,
                          const int * __restrict__ perms, int L,
                          float * __restrict__ h2_out, float * __restrict__ d2_out, int * __restrict__ perm1_out,
                          int * __restrict__ inv_out, float * __restrict__ d1_out /* each [B][L] */) {
    extern __shared__ __align__(16) unsigned char fit_tail_smem[];
    float * a = (float *) fit_tail_smem;                                // [L]
    __shared__ int best_s;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        double best = INFINITY;
        int bj = -1;
        for (int j = 0; j < n_jobs; ++j)                                // jobs are in (block, trial) order
            if (jobs[j].b == b && sse[j] < best) { best = sse[j]; bj = j; }
        best_s = bj;
    }
    __syncthreads();
    const int bj = best_s;
    const size_t o = (size_t) b * L;
    for (int u = threadIdx.x; u < L; u += 256) { inv_out[o + u] = 0; d1_out[o + u] = 1.0f; }     // D1 = I in the fit (:450)
    if (bj < 0) {                                                       // block without rows / inputs: stays zero
        for (int u = threadIdx.x; u < L; u += 256) { h2_out[o + u] = 0.0f; d2_out[o + u] = 0.0f; perm1_out[o + u] = 0; }
        return;
    }
    const double * sm = sums + (size_t) bj * 3 * L, * wg = sm + 2 * (size_t) L;
    const int * perm = perms + jobs[bj].perm_off;
    for (int u = threadIdx.x; u < L; u += 256) {
        const float h = wg[u] > 0.0 ? (float) (sm[u] / wg[u]) : 0.0f;
        a[u] = h;
        h2_out[o + u] = h;
        perm1_out[o + u] = perm[u];
    }
    __syncthreads();                                                    // also orders inv_out = 0 before the scatter below
    for (int i = threadIdx.x; i < L; i += 256) {
        const int src = perm[i];
        if (src >= 0 && src < L) inv_out[o + src] = i;
    }
    for (int len = 1; len < L; len <<= 1) {
        for (int t = threadIdx.x; t < (L >> 1); t += 256) {
            const int i = ((t & ~(len - 1)) << 1) | (t & (len - 1));
            const float u = a[i], v = a[i + len];
            a[i] = u + v; a[i + len] = u - v;
        }
        __syncthreads();
    }
    const float inv_L = 1.0f / (float) L;
    for (int u = threadIdx.x; u < L; u += 256) d2_out[o + u] = a[u] * inv_L;
}

// ---- kernel: COO selection ---------------------------------------------------------------------------
// One workgroup per output row. LDS: r[n_in] f32 | sort[Kpad] u64 | hist[256] | misc.
__device__ __forceinline__ uint32_t score_key(float r) { return __float_as_uint(fabsf(r)); }

// hist[bin] += 1 for every active lane, one LDS atomic per distinct bin of the wave. The top byte of an
// fp32 score (sign + 7 exponent bits) takes a handful of values over a weight row: plain per-lane
// atomics would put all 64 lanes of a wave on one or two LDS words, which the LDS serialises.
__device__ __forceinline__ void hist_add_aggregated(uint32_t * hist, uint32_t bin, bool active) {
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const uint32_t b = (uint32_t) __shfl((int) bin, leader, 64);
        const unsigned long long same = __ballot(active && bin == b);
        if ((int) (threadIdx.x & 63) == leader) atomicAdd(&hist[b], (uint32_t) __popcll(same));
        todo &= ~same;
    }
}

template <typename CMP>
__device__ __forceinline__ void bitonic_desc(int n, CMP swap_if_needed) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n >> 1); t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                swap_if_needed(i, i | j, (i & k) == 0);
            }
            __syncthreads();
        }
    }
}

// lane ^ m partner of a 32-bit value (m a power of two, uniform) without the LDS crossbar: half-register swaps for
// m = 32 / 16, DPP moves inside a 16-lane row below (a ds_bpermute per value made the register sort LDS-bound)
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t a) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) a, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int m, int lane) {
    switch (m) {
        case 32: { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
        case 16: { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
        case 8:  return dpp_u32<0x128>(v);                                       // row_ror:8
        case 4:  { const uint32_t a = dpp_u32<0x104>(v), b = dpp_u32<0x114>(v);  // row_shl:4 (i <- i+4), row_shr:4 (i <- i-4)
                   return (lane & 4) ? b : a; }
        case 2:  return dpp_u32<0x4E>(v);                                        // quad_perm [2,3,0,1]
        default: return dpp_u32<0xB1>(v);                                        // quad_perm [1,0,3,2]
    }
}

// all lanes of the wave receive the sum / the maximum (strides written out: each exchange folds to one DPP move or swap)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t c, int lane) {
    c += lane_xor_u32(c, 32, lane); c += lane_xor_u32(c, 16, lane); c += lane_xor_u32(c, 8, lane);
    c += lane_xor_u32(c, 4, lane);  c += lane_xor_u32(c, 2, lane);  c += lane_xor_u32(c, 1, lane);
    return c;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t c, int lane) {
    c = max(c, lane_xor_u32(c, 32, lane)); c = max(c, lane_xor_u32(c, 16, lane)); c = max(c, lane_xor_u32(c, 8, lane));
    c = max(c, lane_xor_u32(c, 4, lane));  c = max(c, lane_xor_u32(c, 2, lane));  c = max(c, lane_xor_u32(c, 1, lane));
    return c;
}

// Bitonic sort (descending) of n = 256 * E 64-bit keys held E per thread, element i = tid * E + e. The network and
// its result are those of the LDS form below (the keys are distinct, so any correct sorting network gives the same
// order); what changes is where a compare-exchange finds its partner: strides below E in the thread's own registers,
// strides below 64 * E in another lane of the wave (two DPP moves or half-register swaps per key), and only the strides that cross waves —
// three stages for 1024 or 2048 keys instead of 55 or 66 — go through LDS with a barrier.
template <int E, int K2, int J>
__device__ __forceinline__ void bitonic_stage(uint64_t (&v)[E], uint64_t * sbuf, int tid) {
    if constexpr (J >= 64 * E) {                              // partner in another wave
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) sbuf[tid * E + e] = v[e];
        __syncthreads();
        const int pt = tid ^ (J / E);
        const bool keep_max = (((tid * E) & K2) == 0) == (((tid * E) & J) == 0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint64_t y = sbuf[pt * E + e];
            v[e] = ((v[e] > y) == keep_max) ? v[e] : y;            // one 64-bit compare, two selects (equal keys are padding: either is right)
        }
    } else if constexpr (J >= E) {                            // partner in this wave
        const bool keep_max = (((tid * E) & K2) == 0) == (((tid * E) & J) == 0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t lo = lane_xor_u32((uint32_t) v[e], J / E, tid & 63);
            const uint32_t hi = lane_xor_u32((uint32_t) (v[e] >> 32), J / E, tid & 63);
            const uint64_t y = ((uint64_t) hi << 32) | lo;
            v[e] = ((v[e] > y) == keep_max) ? v[e] : y;            // one 64-bit compare, two selects (equal keys are padding: either is right)
        }
    } else {                                                  // partner in this thread
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if ((e & J) == 0) {
                const bool desc = (((tid * E + e) & K2) == 0);
                const uint64_t a = v[e], b = v[e | J];
                const bool sw = (a < b) == desc;
                v[e] = sw ? b : a; v[e | J] = sw ? a : b;
            }
        }
    }
    if constexpr (J > 1) bitonic_stage<E, K2, J / 2>(v, sbuf, tid);
}
template <int E, int K2>
__device__ __forceinline__ void bitonic_merge_levels(uint64_t (&v)[E], uint64_t * sbuf, int tid) {
    bitonic_stage<E, K2, K2 / 2>(v, sbuf, tid);
    if constexpr (K2 < 256 * E) bitonic_merge_levels<E, K2 * 2>(v, sbuf, tid);
}
template <int E>
__device__ __forceinline__ void bitonic_desc_regs(uint64_t * sbuf) {
    const int tid = threadIdx.x;
    uint64_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = sbuf[tid * E + e];
    bitonic_merge_levels<E, 2>(v, sbuf, tid);                 // every stride a compile-time constant: the lane exchanges fold to one DPP move or swap
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) sbuf[tid * E + e] = v[e];
    __syncthreads();
}

// A K schedule evaluated from ONE selection (the builder fits once per tensor and runs a trial per K,
// sd_build.cpp:444-451, :843-913): rows are selected and sorted at the largest K; descending order with
// ties by ascending index makes the top-K of every smaller K a prefix of that row. k[j] counts idx entries
// (COO: residuals; block: kept blocks). Row scales and metric results are laid out [j][slot].
struct KGrid { int n; int k[SDH_SCAN_MAX_K]; int ord[SDH_SCAN_MAX_K]; };      // ord: the members by ascending k (ties by position)

template <int DT>
__global__ __launch_bounds__(256)
void scan_select_coo_kernel(DenseView W, BaseView base, int has_base, int K /* the largest of kg */, int Kpad, int want_rs, KGrid kg,
                            const int * __restrict__ rows_sel /* null: every row; else outputs are compact */,
                            const float * __restrict__ ws,
                            int * __restrict__ idx_out, float * __restrict__ val_out, float * __restrict__ rs_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_in = (int) W.n_in;
    float *    r    = (float *) smem;
    uint64_t * sbuf = (uint64_t *) (smem + (((size_t) n_in * 4 + 15) & ~(size_t) 15));
    uint32_t * hist = (uint32_t *) (sbuf + Kpad);
    double *   dsc  = (double *) (hist + 256);       // 4 doubles
    int *      isc  = (int *) (dsc + 4);             // 8 ints
    const int64_t row = rows_sel ? rows_sel[blockIdx.x] : blockIdx.x;      // weight row
    const int64_t orow = blockIdx.x;                                       // output slot
    const int tid = threadIdx.x;

    double dot = 0.0, nn = 0.0;
    load_residual_row<DT>(W, base, has_base != 0, ws, row, r, dot, nn);
    __syncthreads();
    // the score is |r| (times the imatrix scale, one fp32 product, sd_eval.cpp:53-56); keys are its bits
    auto key_of = [&](int i) -> uint32_t {
        const float sc = fabsf(r[i]);
        return __float_as_uint(ws ? sc * ws[i] : sc);
    };

    // radix select: the K-th largest key, 8 bits per pass from the top
    uint32_t prefix = 0;
    int need = K;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        if (pass == 0) {
            // The top byte (sign + 7 exponent bits) of a weight row's scores takes a handful of values just below the
            // row's largest. Each thread counts its elements into eight 8-bit fields of one register pair — the eight
            // bins from the row maximum's down — the waves add their fields up with lane exchanges and four lanes
            // per bin touch the LDS histogram: no ballot loop and no LDS atomic per element (0.10 of 0.43 ms per
            // selection of a Mistral gate went there, 0.27 of 0.70 for down). If the K-th key lies below the window,
            // or a thread owns more elements than a field counts, the general form runs instead.
            uint32_t tmax = 0;
            for (int i = tid; i < n_in; i += 256) tmax = max(tmax, key_of(i) >> 24);
            tmax = wave_max_u32(tmax, tid & 63);
            if ((tid & 63) == 0) isc[2 + (tid >> 6)] = (int) tmax;
            __syncthreads();
            const uint32_t bmax = (uint32_t) max(max(isc[2], isc[3]), max(isc[4], isc[5]));
            __syncthreads();
            bool windowed = n_in <= 255 * 256;
            if (windowed) {
                uint64_t packed = 0;
                for (int i = tid; i < n_in; i += 256) {
                    const uint32_t d = bmax - (key_of(i) >> 24);
                    packed += d < 8u ? (1ull << (8u * d)) : 0ull;
                }
                int in_window = 0;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const uint32_t c = wave_sum_u32((uint32_t) (packed >> (8 * d)) & 255u, tid & 63);
                    if ((tid & 63) == 0 && c && bmax >= (uint32_t) d) atomicAdd(&hist[bmax - d], c);
                    in_window += (int) c;
                }
                if ((tid & 63) == 0) isc[2 + (tid >> 6)] = in_window;
                __syncthreads();
                windowed = isc[2] + isc[3] + isc[4] + isc[5] >= need;      // uniform: every thread reads the same four words
                __syncthreads();
                if (!windowed) { hist[tid] = 0; __syncthreads(); }
            }
            if (!windowed) {
                for (int i0 = 0; i0 < n_in; i0 += 256) {      // whole waves stay in the loop: the ballots need them
                    const int i = i0 + tid;
                    const bool act = i < n_in;
                    hist_add_aggregated(hist, act ? (key_of(i) >> 24) : 0u, act);
                }
            }
        } else {
            for (int i = tid; i < n_in; i += 256) {
                const uint32_t key = key_of(i);
                if ((key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        {   // digit = the largest d with (count of keys in bins > d) < need <= (count in bins >= d):
            // suffix sums over the 256 bins, one bin per thread
            const int mine = (int) hist[tid];
            int suf = mine;                               // inclusive suffix sum within the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(suf, o, 64);
                if ((tid & 63) + o < 64) suf += t;
            }
            if ((tid & 63) == 0) isc[2 + (tid >> 6)] = suf;          // wave totals
            __syncthreads();
            int above = 0;                                // bins of higher waves
            for (int w = (tid >> 6) + 1; w < 4; ++w) above += isc[2 + w];
            const int ge = suf + above, gt = ge - mine;   // keys in bins >= tid, > tid
            if (gt < need && need <= ge) { isc[0] = tid; isc[1] = need - gt; }
        }
        __syncthreads();
        prefix = (prefix << 8) | (uint32_t) isc[0];
        need = isc[1];
        __syncthreads();
    }
    const uint32_t T = prefix;          // key of the K-th largest; `need` entries equal to T are kept
    const int n_greater = K - need;

    // ordered compaction: thread t owns the contiguous range [lo, hi)
    const int per = (n_in + 255) / 256;
    const int lo = min(n_in, tid * per), hi = min(n_in, lo + per);
    int cg = 0, ce = 0;
    for (int i = lo; i < hi; ++i) {
        const uint32_t key = key_of(i);
        cg += key > T;
        ce += key == T;
    }
    int tot;
    int og = block_excl_scan(cg, isc, &tot);
    int oe = block_excl_scan(ce, isc, &tot);
    for (int i = lo; i < hi; ++i) {
        const uint32_t key = key_of(i);
        const uint64_t comp = ((uint64_t) key << 32) | (uint64_t) (0xffffffffu - (uint32_t) i);
        if (key > T) sbuf[og++] = comp;
        else if (key == T) { if (oe < need) sbuf[n_greater + oe] = comp; ++oe; }
    }
    for (int i = K + tid; i < Kpad; i += 256) sbuf[i] = 0;
    __syncthreads();

    // bitonic sort, descending; a thread's pairs of one stage are disjoint, so their loads are issued
    // together (four LDS round trips per stage become one)
    // 512 keys and more: E per thread in registers (bitonic_desc_regs); fewer: the LDS form
    int k_first = 2;
    switch (Kpad) {
        case 512:  bitonic_desc_regs<2>(sbuf);  k_first = Kpad + 1; break;
        case 1024: bitonic_desc_regs<4>(sbuf);  k_first = Kpad + 1; break;
        case 2048: bitonic_desc_regs<8>(sbuf);  k_first = Kpad + 1; break;
        case 4096: bitonic_desc_regs<16>(sbuf); k_first = Kpad + 1; break;
        default: break;
    }
    for (int k = k_first; k <= Kpad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t0 = 0; t0 < (Kpad >> 1); t0 += 256 * 4) {
                uint64_t x[4], y[4];
                int ia[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * 256 + tid;
                    const int tc = t < (Kpad >> 1) ? t : 0;
                    ia[u] = ((tc & ~(j - 1)) << 1) | (tc & (j - 1));
                    x[u] = sbuf[ia[u]];
                    y[u] = sbuf[ia[u] | j];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * 256 + tid;
                    const bool desc = (ia[u] & k) == 0;
                    if (t < (Kpad >> 1) && (desc ? (x[u] < y[u]) : (x[u] > y[u]))) { sbuf[ia[u]] = y[u]; sbuf[ia[u] | j] = x[u]; }
                }
            }
            __syncthreads();
        }
    }

#ifdef SEL_ABL_NOWRITE
    if (K < 0)
#endif
    for (int rr = tid; rr < K; rr += 256) {
        const int i = (int) (0xffffffffu - (uint32_t) sbuf[rr]);
        idx_out[orow * K + rr] = i;
        val_out[orow * K + rr] = r[i];
    }
    if (want_rs) {
        // the least-squares scale of every K of the schedule: each thread adds its entries rr = tid, tid + 256, ...
        // below K_j, the order a selection at K_j alone would use
        // ONE walk over the stored row in ascending k: a thread's running sums at the boundary k_j are exactly the sums
        // a selection at k_j alone forms (its own entries rr = tid, tid + 256, .. below k_j, in that order), so every
        // member's scale has the bits of a lone call and the row is read once instead of once per member
        const double d_row = block_sum(dot, dsc), n_row = block_sum(nn, dsc);
        double dot2 = 0.0, nn2 = 0.0;
        int rr = tid;
        for (int jj = 0; jj < kg.n; ++jj) {
            const int j = kg.ord[jj];
            for (; rr < kg.k[j]; rr += 256) {
                const int i = (int) (0xffffffffu - (uint32_t) sbuf[rr]);
                const double wi = (double) dense_at<DT>(W, row, i), di = (double) r[i];
                const double s1 = ws ? (double) ws[i] : 1.0;
                if (has_base) { dot2 += wi * di * s1 * s1; nn2 += (2.0 * wi * di - di * di) * s1 * s1; }
                else          { dot2 += wi * wi * s1 * s1; nn2 += wi * wi * s1 * s1; }
            }
            const double d_all = d_row + block_sum(dot2, dsc);
            const double n_all = n_row + block_sum(nn2, dsc);
            if (tid == 0) rs_out[(size_t) j * gridDim.x + orow] = n_all > 1e-30 ? (float) (d_all / n_all) : 1.0f;
        }
    }
}

// ---- kernel: block selection -------------------------------------------------------------------------
// LDS: r[n_in] f32 | energy[NBpad] f64 | bid[NBpad] i32 | misc
template <int DT>
__global__ __launch_bounds__(256)
void scan_select_block_kernel(DenseView W, BaseView base, int has_base, int block, int keep /* the largest of kg */, int n_blocks, int NBpad,
                              int want_rs, KGrid kg, const int * __restrict__ rows_sel, const float * __restrict__ ws,
                              int * __restrict__ idx_out, float * __restrict__ val_out, float * __restrict__ rs_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_in = (int) W.n_in;
    float *  r   = (float *) smem;
    double * en  = (double *) (smem + (((size_t) n_in * 4 + 15) & ~(size_t) 15));
    int *    bid = (int *) (en + NBpad);
    double * dsc = (double *) (bid + NBpad + (NBpad & 1));
    const int64_t row = rows_sel ? rows_sel[blockIdx.x] : blockIdx.x;
    const int64_t orow = blockIdx.x;
    const int tid = threadIdx.x;

    double dot = 0.0, nn = 0.0;
    load_residual_row<DT>(W, base, has_base != 0, ws, row, r, dot, nn);
    __syncthreads();
    for (int b = tid; b < NBpad; b += 256) {
        double e = -1.0;                                   // padding sorts last
        if (b < n_blocks) {
            e = 0.0;
            const int i1 = min(n_in, (b + 1) * block);
            for (int i = b * block; i < i1; ++i) {                                                 // sequential: exact
                const double x = (double) r[i] * (ws ? (double) ws[i] : 1.0);
                e += x * x;
            }
        }
        en[b] = e;
        bid[b] = b;
    }
    __syncthreads();
    bitonic_desc(NBpad, [&](int a, int b, bool desc) {
        const double ea = en[a], eb = en[b];
        const int ia = bid[a], ib = bid[b];
        const bool a_first = ea > eb || (ea == eb && ia < ib);      // a belongs before b in descending order
        if (desc ? !a_first : a_first) { en[a] = eb; en[b] = ea; bid[a] = ib; bid[b] = ia; }
    });

    for (int e = tid; e < keep * block; e += 256) {
        const int bi = e / block, t = e - bi * block;
        const int blk = bid[bi];
        const int i = blk * block + t;
        if (t == 0) idx_out[orow * keep + bi] = blk;
        val_out[((size_t) orow * keep + bi) * block + t] = i < n_in ? r[i] : 0.0f;
    }
    if (want_rs) {
        const double d_row = block_sum(dot, dsc), n_row = block_sum(nn, dsc);
        for (int j = 0; j < kg.n; ++j) {                      // per K of the schedule, in a lone selection's order
            double dot2 = 0.0, nn2 = 0.0;
            for (int e = tid; e < kg.k[j] * block; e += 256) {
                const int bi = e / block, t = e - bi * block;
                const int i = bid[bi] * block + t;
                if (i < n_in) {
                    const double wi = (double) dense_at<DT>(W, row, i), di = (double) r[i];
                    const double s1 = ws ? (double) ws[i] : 1.0;
                    if (has_base) { dot2 += wi * di * s1 * s1; nn2 += (2.0 * wi * di - di * di) * s1 * s1; }
                    else          { dot2 += wi * wi * s1 * s1; nn2 += wi * wi * s1 * s1; }
                }
            }
            const double d_all = d_row + block_sum(dot2, dsc);
            const double n_all = n_row + block_sum(nn2, dsc);
            if (tid == 0) rs_out[(size_t) j * gridDim.x + orow] = n_all > 1e-30 ? (float) (d_all / n_all) : 1.0f;
        }
    }
}

// ---- kernels: --delta-norm-clamp-down (sd_build.cpp:29-46, :705-720) --------------------------------------
// per-row sums of squares in double (tree-reduced; the host adds the rows in order), and the rescale
template <int DT>
__global__ __launch_bounds__(256)
void scan_row_sumsq_dense_kernel(DenseView W, double * __restrict__ out /* [n_out] */) {
    __shared__ double dsc[4];
    const int64_t row = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < (int) W.n_in; i += 256) {
        const double v = (double) dense_at<DT>(W, row, i);
        acc += v * v;
    }
    acc = block_sum(acc, dsc);
    if (threadIdx.x == 0) out[row] = acc;
}

__global__ __launch_bounds__(256)
void scan_row_sumsq_vals_kernel(const float * __restrict__ val, int K, double * __restrict__ out /* [n_out] */) {
    __shared__ double dsc[4];
    const int64_t row = blockIdx.x;
    double acc = 0.0;
    for (int e = threadIdx.x; e < K; e += 256) {
        const double v = (double) val[row * K + e];
        acc += v * v;
    }
    acc = block_sum(acc, dsc);
    if (threadIdx.x == 0) out[row] = acc;
}

__global__ void scan_scale_vals_kernel(float * __restrict__ val, int64_t n, double scale) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) val[i] = (float) ((double) val[i] * scale);           // v = (float) (v * scale), :716
}

// ---- kernel: weight-space metric triples (w_norm2, hat_norm2, dot) per sampled row -------------------
template <int DT>
__global__ __launch_bounds__(256)
void scan_evalw_kernel(DenseView W, BaseView base, int has_base, int scheme, KGrid kg, int ld /* idx entries per payload row */, int block,
                       const int * __restrict__ idx, const float * __restrict__ val, const float * __restrict__ rs /* [kg.n][rs_stride] */,
                       int64_t rs_stride, const float * __restrict__ ws, const int * __restrict__ rows,
                       int compact /* payload indexed by sample, not by row */,
                       double * __restrict__ out /* [kg.n][n_rows][3] */) {
    __shared__ double dsc[4];
    const int64_t row = rows[blockIdx.x];
    const int64_t pr = compact ? (int64_t) blockIdx.x : row;       // payload slot of this row
    const int n_in = (int) W.n_in;
    const int K = kg.k[blockIdx.y];                                // this K of the schedule: a prefix of the row
    if (rs) rs += (size_t) blockIdx.y * (size_t) rs_stride;
    out += (size_t) blockIdx.y * gridDim.x * 3;
    const double scale = rs ? (double) rs[pr] : 1.0;
    double wn = 0.0, hn = 0.0, dt = 0.0;
    for (int i = threadIdx.x; i < n_in; i += 256) {
        const double s1 = ws ? (double) ws[i] : 1.0;
        const double wi = (double) dense_at<DT>(W, row, i) * s1;
        wn += wi * wi;
        if (has_base) {
            const double bs = (double) base_weight(base, row, i) * s1;
            hn += bs * bs;
            dt += wi * bs;
        }
    }
    const int n_ent = scheme == 1 ? K * block : K;
    for (int e = threadIdx.x; e < n_ent; e += 256) {
        int64_t i;
        float dv;
        if (scheme == 1) {
            const int bi = e / block, t = e - bi * block;
            const int blk = idx[pr * ld + bi];
            if (blk < 0) continue;
            i = (int64_t) blk * block + t;
            if (i >= n_in) continue;
            dv = val[((size_t) pr * ld + bi) * block + t];
        } else {
            const int ii = idx[pr * ld + e];
            if (ii < 0 || ii >= n_in) continue;
            i = ii;
            dv = val[pr * ld + e];
        }
        const double s1 = ws ? (double) ws[i] : 1.0;
        if (has_base) {
            const double wi = (double) dense_at<DT>(W, row, i) * s1;
            const double ds = (double) dv * s1, bs = (double) base_weight(base, row, i) * s1;
            dt += wi * ds;
            hn += 2.0 * bs * ds + ds * ds;
        } else {
            const double wh = scale * (double) dv;
            const double whs = wh * s1;
            hn += whs * whs;
            dt += (double) dense_at<DT>(W, row, i) * s1 * whs;
        }
    }
    wn = block_sum(wn, dsc); hn = block_sum(hn, dsc); dt = block_sum(dt, dsc);
    if (has_base) { dt = scale * dt; hn = scale * scale * hn; }
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = wn; out[blockIdx.x * 3 + 1] = hn; out[blockIdx.x * 3 + 2] = dt; }
}

// ---- kernel: y_true[xi][ci] = sum_i W[row_ci][i] * x[xi][i] in double -----------------------------------
template <int DT>
__global__ __launch_bounds__(256)
void scan_dense_dot_kernel(DenseView W, const int * __restrict__ rows, int n_rows, const float * __restrict__ xs,
                           double * __restrict__ out /* [n_x][n_rows] */) {
    __shared__ double dsc[4];
    const int ci = blockIdx.x, xi = blockIdx.y;
    const int64_t row = rows[ci];
    const float * x = xs + (size_t) xi * W.n_in;
    double acc = 0.0;
    for (int i = threadIdx.x; i < (int) W.n_in; i += 256) acc += (double) dense_at<DT>(W, row, i) * (double) x[i];
    acc = block_sum(acc, dsc);
    if (threadIdx.x == 0) out[(size_t) xi * n_rows + ci] = acc;
}

// ---- kernel: y_hat[xi][ci] = (base(x)[row] + sum_r val * x[idx]) * row_scale -----------------------------
// v = base transforms [n_x][B][L] (fp32, exact); tall picks v[b][p], wide sums blocks in fp32 from 0.0f
// in block order (sd_eval.cpp:1119-1134)
__global__ __launch_bounds__(256)
void scan_delta_dot_kernel(int n_in, int scheme, KGrid kg, int ld /* idx entries per payload row */, int block, const int * __restrict__ idx,
                           const float * __restrict__ val, const float * __restrict__ rs /* [kg.n][rs_stride] */, int64_t rs_stride,
                           const float * __restrict__ v, int L, int B, int tall,
                           const int * __restrict__ rows, int n_rows, int compact, const float * __restrict__ xs,
                           double * __restrict__ out /* [kg.n][n_x][n_rows] */) {
    __shared__ double dsc[4];
    const int ci = blockIdx.x, xi = blockIdx.y;
    const int K = kg.k[blockIdx.z];                                // this K of the schedule: a prefix of the row
    if (rs) rs += (size_t) blockIdx.z * (size_t) rs_stride;
    out += (size_t) blockIdx.z * gridDim.y * (size_t) n_rows;
    const int64_t row = rows[ci];
    const int64_t pr = compact ? (int64_t) ci : row;
    const float * x = xs + (size_t) xi * n_in;
    double acc = 0.0;
    const int n_ent = scheme == 1 ? K * block : K;
    for (int e = threadIdx.x; e < n_ent; e += 256) {
        if (scheme == 1) {
            const int bi = e / block, t = e - bi * block;
            const int blk = idx[pr * ld + bi];
            if (blk < 0) continue;
            const int64_t i = (int64_t) blk * block + t;
            if (i >= n_in) continue;
            acc += (double) val[((size_t) pr * ld + bi) * block + t] * (double) x[i];
        } else {
            const int ii = idx[pr * ld + e];
            if (ii < 0 || ii >= n_in) continue;
            acc += (double) val[pr * ld + e] * (double) x[ii];
        }
    }
    acc = block_sum(acc, dsc);
    if (threadIdx.x == 0) {
        double y0 = 0.0;
        if (v) {
            const float * vx = v + (size_t) xi * B * L;
            if (tall) {
                const int64_t b = row / L;
                if (b < B) y0 = (double) vx[b * L + (row - b * L)];
            } else if (row < L) {
                float s = 0.0f;
                for (int b = 0; b < B; ++b) s += vx[(size_t) b * L + row];
                y0 = (double) s;
            }
        }
        out[(size_t) xi * n_rows + ci] = (y0 + acc) * (rs ? (double) rs[pr] : 1.0);
    }
}

template <int DT>
__global__ void scan_gather_kernel(DenseView W, const int * __restrict__ rows, int n_rows, const int * __restrict__ cols,
                                   int n_cols, float * __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * n_cols) return;
    const int c = cols[e % n_cols];
    out[e] = c < W.n_in ? dense_at<DT>(W, rows[e / n_cols], c) : 0.0f;
}

// ---- kernels: the builder's packing step (tools/seeddelta-build/sd_build.cpp:1104-1236) -----------------
// indices: int32 staging -> I16, anything outside [-32768, 32767] becomes -1 (:1132, :1162)
__global__ __launch_bounds__(256)
void scan_pack_i16_kernel(const int * __restrict__ in, int16_t * __restrict__ out, int64_t n, int64_t w /* entries per row */, int64_t ld /* row stride of in */) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const int v = in[(i / w) * ld + i % w]; out[i] = (v < -32768 || v > 32767) ? (int16_t) -1 : (int16_t) v; }
}
// values: fp32 -> F16 as ggml_fp32_to_fp16_row does it (ggml/src/ggml.c:457-462): round to nearest even,
// overflow to infinity, subnormals kept, every NaN -> sign | 0x7E00
__global__ __launch_bounds__(256)
void scan_pack_f16_kernel(const float * __restrict__ in, uint16_t * __restrict__ out, int64_t n, int64_t per_row /* entries per row */, int64_t ld /* row stride of in */) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f = in[(i / per_row) * ld + i % per_row];
    const uint32_t w = __float_as_uint(f);
    const uint16_t h = __half_as_ushort(__float2half_rn(f));
    out[i] = ((w & 0x7FFFFFFFu) > 0x7F800000u) ? (uint16_t) (((w >> 16) & 0x8000u) | 0x7E00u) : h;
}

// ---- host: scan state --------------------------------------------------------------------------------
// Device scratch of a scan handle. Capacities grow geometrically and never shrink: hipFree synchronises the
// whole device, and with 16 handles working through a K grid exact-size buffers were reallocated on
// almost every candidate (rocprofv3 --hip-runtime-trace: 1160 hipFree calls, 0.36 ms each, in one sweep).
// put() copies straight from pageable memory: the runtime stages it and returns when the source is
// consumed. Small uploads through a pinned mirror (truly asynchronous, one SDMA hop each in stream order)
// measured slower end to end: +0.3 ms of stream latency per candidate for the 5-8 uploads it makes.
template <typename T>
struct DevBuf {
    T * p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return SDH_OK;
        size_t want = 256;
        while (want < n) want <<= 1;
        if (p) (void) hipFree(p);
        p = nullptr; cap = 0;
        SDH_HIP_CHECK(hipMalloc((void **) &p, want * sizeof(T)));
        cap = want;
        return SDH_OK;
    }
    int put(const T * h, size_t n, hipStream_t s) {
        int rc = ensure(n ? n : 1);
        if (rc != SDH_OK) return rc;
        if (!n) return SDH_OK;
        SDH_HIP_CHECK(hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, s));
        return SDH_OK;
    }
    int put(const std::vector<T> & h, hipStream_t s) { return put(h.data(), h.size(), s); }
    ~DevBuf() { if (p) (void) hipFree(p); }
};

struct FitPlan;
struct Scan {
    std::mt19937 rng;
    // last candidate
    int64_t n_in = 0, n_out = 0, K = 0, block = 0, L = 0, B = 0;     // K: idx entries per row of the candidate sdh_scan_fetch returns
    int64_t ld = 0;                     // idx entries per row as stored (the largest K of the last schedule)
    int     kg_n = 1, pick = 0;         // K schedule of the last call and the member sdh_scan_fetch* returns
    std::mt19937 rng_after[SDH_SCAN_MAX_K];   // generator after each trial of the last call (sequential draws), else [0] only
    int     n_after = 0;
    int64_t kg_ksel[SDH_SCAN_MAX_K] = {0};
    int scheme = 0; bool has_base = false, has_rs = false;
    bool emitted = false;               // the last candidate selected every row (sdh_scan_fetch can return the payload)
    const void * fit_of = nullptr;      // the dense handle the current fit belongs to
    DevBuf<int> idx; DevBuf<float> val; DevBuf<float> rs;
    DevBuf<float> h2, d2, d1; DevBuf<int> inv, perm1;
    std::vector<float> h_h2, h_d2; std::vector<int> h_perm1, h_inv;   // host copies of the fit, fetched on demand
    bool host_fit_valid = false;
    std::shared_ptr<const FitPlan> fit_plan;   // keeps the device copy of the plan alive while the fit kernels may run
    DevBuf<double> sse;
    // scratch
    DevBuf<uint16_t> packed;            // sdh_scan_fetch_packed: converted idx / val / row_scale
    DevBuf<int> rows, perms, cols; DevBuf<FitJob> jobs; DevBuf<double> dbl, dbl2; DevBuf<float> xs, v, gat, wsd;
    std::vector<float> h_ws;
    // host side of the result copies. Pageable on purpose: with pinned buffers the (then truly asynchronous)
    // copies measured 30 % slower end to end (7.8k vs 5.4k candidates/s; some hipMemcpyAsync calls blocked for 2-7 ms)
    std::vector<double> h_dbl; std::vector<float> h_flt;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream = nullptr;       // one stream per handle: candidates prepared by different host threads overlap on the GPU
    // proxy: the true gate/up dots and the W_down samples depend on the three dense tensors and the proxy
    // seed only, not on the candidate: kept for the next K of the same tensor
    struct ProxyTruth {
        uint64_t g = 0, u = 0, d = 0; int64_t seed = -1, cap = -1; int nx = 0;
        std::vector<double> gu; std::vector<float> wd;
    } proxy_truth;
    // dev: host wall time per phase of sdh_scan_tensor, printed at destroy when SDH_SCAN_TIMING is set
    double t_phase[7] = {0, 0, 0, 0, 0, 0, 0};   // fit, select-all + clamp, metric draws, wait, summaries, tail, metric launches
    long   n_cand = 0;
    explicit Scan(uint32_t seed) : rng(seed) {}
};
static bool scan_timing() { static const bool v = getenv("SDH_SCAN_TIMING") != nullptr; return v; }
struct PhaseTimer {
    double & acc; std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(double & a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~PhaseTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

static void fit_dims(int64_t n_in, int64_t n_out, int64_t * L, int64_t * B) {
    int64_t p = 1;
    while (p < std::min(n_in, n_out)) p <<= 1;
    *L = p;
    *B = (std::max(n_in, n_out) + p - 1) / p;
}

static void host_fwht(std::vector<float> & a) {
    const size_t n = a.size();
    for (size_t len = 1; len < n; len <<= 1)
        for (size_t i = 0; i < n; i += len << 1)
            for (size_t j = 0; j < len; ++j) {
                const float u = a[i + j], v = a[i + j + len];
                a[i + j] = u + v; a[i + j + len] = u - v;
            }
}

static double percentile(std::vector<double> v, double p) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    const double x = p * double(v.size() - 1);
    const size_t i = (size_t) x, j = std::min(i + 1, v.size() - 1);
    const double a = x - double(i);
    return v[i] * (1.0 - a) + v[j] * a;
}

static void triple_metrics(double n_true, double n_hat, double dot, double * rel, double * cs, double * nr) {
    const double err2 = std::max(n_true + n_hat - 2.0 * dot, 0.0);
    const double dt = std::sqrt(std::max(n_true, 1e-20)), dh = std::sqrt(std::max(n_hat, 1e-20));
    *rel = std::sqrt(err2) / dt; *cs = dot / (dt * dh); *nr = dh / dt;
}

static void summarise(const std::vector<double> & rel, const std::vector<double> & cs, const std::vector<double> & nr, double * o) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (double x : rel) a += x;
    for (double x : cs) b += x;
    for (double x : nr) c += x;
    const double n = (double) rel.size();
    o[0] = rel.empty() ? 0.0 : a / n; o[1] = percentile(rel, 0.95);
    o[2] = cs.empty() ? 0.0 : b / n;  o[3] = percentile(cs, 0.05);
    o[4] = nr.empty() ? 0.0 : c / n;
}

// ---- host routines that consume the mt19937 stream, memoised on the generator state ------------------
// The evaluators draw eval_x * n_in Gaussians per pass (sd_eval.cpp:1084-1087, sd_ffn_proxy.cpp:77-80) at
// ~26 ns each, shuffle iota(n_out) for every row sample (~80 us at n_out = 14336) and the fit shuffles
// its rows and permutations — and the stream position each of them starts from depends only on the seed
// and the tensor shapes: not on the layer, not on K. Across a sweep the same (state, arguments) recurs
// for every layer and every K of a tensor kind, so a result — identical by construction — is computed
// once and the generator is moved to the state it would have reached. Lookups compare the full mt19937
// state (operator==) and the arguments.
using MemoKey = std::array<int64_t, 8>;
template <class V>
class RngMemo {
    struct Entry {
        std::mt19937 before, after;
        MemoKey key{};
        bool ready = false;                 // false while the first caller is still computing
        std::shared_ptr<const V> val;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Entry>> q;
    const size_t cap;
public:
    explicit RngMemo(size_t cap_) : cap(cap_) {}
    // make(rng) computes the value and advances rng
    template <class F>
    std::shared_ptr<const V> get(std::mt19937 & rng, const MemoKey & key, F && make) {
        std::shared_ptr<Entry> mine;
        {
            std::unique_lock<std::mutex> lock(mu);
            for (size_t i = 0; i < q.size(); ++i) {
                if (q[i]->key == key && q[i]->before == rng) {
                    const std::shared_ptr<Entry> hit = q[i];            // the deque may change while we wait
                    cv.wait(lock, [&] { return hit->ready; });          // someone is computing exactly this
                    if (!hit->val) break;                               // its computation failed: compute here
                    rng = hit->after;
                    return hit->val;
                }
            }
            mine = std::make_shared<Entry>();
            mine->before = rng;
            mine->key = key;
            q.push_back(mine);
            while (q.size() > cap) q.pop_front();                       // holders keep evicted values alive
        }
        std::shared_ptr<const V> v;
        try {
            v = make(rng);
        } catch (...) {                                                 // e.g. bad_alloc: release the waiters, forget the entry
            {
                std::lock_guard<std::mutex> lock(mu);
                mine->ready = true;
                for (size_t i = 0; i < q.size(); ++i) if (q[i] == mine) { q.erase(q.begin() + (long) i); break; }
            }
            cv.notify_all();
            throw;
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            mine->after = rng;
            mine->val = v;
            mine->ready = true;
        }
        cv.notify_all();
        return v;
    }
    void clear() {
        std::lock_guard<std::mutex> lock(mu);
        q.clear();
    }
};

// the row sample every eval draws: iota, shuffle, truncate (sd_eval.cpp:138-141 and siblings)
static RngMemo<std::vector<int>> & rows_memo() {
    static RngMemo<std::vector<int>> & memo = *new RngMemo<std::vector<int>>(256);      // never destroyed (no exit-time teardown)
    return memo;
}
static std::shared_ptr<const std::vector<int>> draw_rows(std::mt19937 & rng, int64_t n_out, int64_t want) {
    return rows_memo().get(rng, MemoKey{1, n_out, want}, [&](std::mt19937 & g) {
        std::vector<int64_t> all((size_t) n_out);
        std::iota(all.begin(), all.end(), 0);
        std::shuffle(all.begin(), all.end(), g);
        all.resize((size_t) std::min(want, n_out));
        return std::make_shared<const std::vector<int>>(all.begin(), all.end());
    });
}

// n draws of std::normal_distribution<float>(0,1), with a device copy made on first use and shared by
// every handle (the x matrices of eval_x and of the proxy are 0.5-1.8 MB per pass at Mistral dims)
// device buffers of evicted draws are kept for the next draw of that size: hipFree synchronises the device
struct NormalsPool {
    std::mutex mu;
    std::multimap<size_t, float *> free_list;
    float * take(size_t n) {
        {
            std::lock_guard<std::mutex> lock(mu);
            auto it = free_list.find(n);
            if (it != free_list.end()) { float * p = it->second; free_list.erase(it); return p; }
        }
        float * p = nullptr;
        return hipMalloc((void **) &p, n * sizeof(float)) == hipSuccess ? p : nullptr;
    }
    void give(size_t n, float * p) {
        std::lock_guard<std::mutex> lock(mu);
        free_list.emplace(n, p);
    }
    void release_all() {
        std::lock_guard<std::mutex> lock(mu);
        for (auto & e : free_list) (void) hipFree(e.second);
        free_list.clear();
    }
};
static NormalsPool & normals_pool() {
    static NormalsPool & pool = *new NormalsPool();      // never destroyed (no exit-time teardown)
    return pool;
}
struct Normals {
    std::vector<float> v;
    mutable std::mutex mu;
    mutable float * dev = nullptr;
    const float * device() const {
        std::lock_guard<std::mutex> lock(mu);
        if (!dev && !v.empty()) {
            float * p = normals_pool().take(v.size());
            if (!p) return nullptr;
            if (hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { normals_pool().give(v.size(), p); return nullptr; }
            dev = p;
        }
        return dev;
    }
    ~Normals() { if (dev) normals_pool().give(v.size(), dev); }
};
static RngMemo<Normals> & normals_memo() {
    static RngMemo<Normals> & memo = *new RngMemo<Normals>(96);
    return memo;
}
static std::atomic<long> g_normal_misses{0}, g_normal_miss_us{0}, g_normal_calls{0}, g_normal_call_us{0};
static std::shared_ptr<const Normals> draw_normals(std::mt19937 & rng, int64_t n) {
    const auto c0 = std::chrono::steady_clock::now();
    struct CallTimer { std::chrono::steady_clock::time_point t0; ~CallTimer() {
        g_normal_calls++; g_normal_call_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); } } ct{c0};
    return normals_memo().get(rng, MemoKey{2, n}, [&](std::mt19937 & g) {
        const auto t0 = std::chrono::steady_clock::now();
        struct MissTimer { std::chrono::steady_clock::time_point t0; ~MissTimer() {
            g_normal_misses++; g_normal_miss_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); } } mt{t0};
        auto out = std::make_shared<Normals>();
        out->v.resize((size_t) n);
        static const bool std_draws = getenv("SDH_SCAN_STD_NORMALS") != nullptr;        // A/B: the library template
        if (std_draws) {
            std::normal_distribution<float> nd(0.0f, 1.0f);
            for (int64_t i = 0; i < n; ++i) out->v[(size_t) i] = nd(g);
        } else {
            fast_normals(g, n, out->v.data());
        }
        return std::shared_ptr<const Normals>(out);
    });
}

// RNG consumption of the base fit (sd_eval.cpp:557-651), with or without the work
struct FitPlan {
    std::vector<FitJob> jobs; std::vector<int> rows; std::vector<int> perms;
    // device copy of the three arrays (jobs | rows | perms in one allocation), made on first use and shared by
    // every handle: the plan of a tensor kind is the same for every layer of a sweep. Freed with the plan.
    mutable std::mutex mu;
    mutable unsigned char * dev = nullptr;
    mutable const FitJob * jobs_dev = nullptr; mutable const int * rows_dev = nullptr, * perms_dev = nullptr;
    FitPlan() = default;
    FitPlan(FitPlan && o) noexcept : jobs(std::move(o.jobs)), rows(std::move(o.rows)), perms(std::move(o.perms)) {}
    bool device() const {
        std::lock_guard<std::mutex> lock(mu);
        if (dev) return true;
        const size_t nj = (jobs.size() * sizeof(FitJob) + 255) & ~(size_t) 255, nr = (rows.size() * sizeof(int) + 255) & ~(size_t) 255,
                     np = perms.size() * sizeof(int);
        unsigned char * p = nullptr;
        if (hipMalloc((void **) &p, nj + nr + np + 256) != hipSuccess) return false;
        if (hipMemcpy(p, jobs.data(), jobs.size() * sizeof(FitJob), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p + nj, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p + nj + nr, perms.data(), np, hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(p); return false; }
        jobs_dev = (const FitJob *) p; rows_dev = (const int *) (p + nj); perms_dev = (const int *) (p + nj + nr);
        dev = p;
        return true;
    }
    ~FitPlan() { if (dev) (void) hipFree(dev); }
};

static FitPlan plan_fit_draw(std::mt19937 & rng, int64_t n_in, int64_t n_out, int64_t L, int64_t B, int64_t max_samples, int trials) {
    FitPlan pl;
    const bool tall = n_out >= n_in;
    trials = std::max(1, trials);
    auto add_block = [&](int64_t b, int rows_off, int n_rows) {
        for (int t = 0; t < trials; ++t) {
            std::vector<int32_t> perm((size_t) L);
            std::iota(perm.begin(), perm.end(), 0);
            if (t > 0) std::shuffle(perm.begin(), perm.end(), rng);
            pl.jobs.push_back(FitJob{ (int) b, rows_off, n_rows, (int) pl.perms.size() });
            pl.perms.insert(pl.perms.end(), perm.begin(), perm.end());
        }
    };
    if (tall) {
        for (int64_t b = 0; b < B; ++b) {
            const int64_t o0 = b * L, o1 = std::min(n_out, (b + 1) * L);
            if (o1 <= o0) continue;
            std::vector<int64_t> cols((size_t) (o1 - o0));
            std::iota(cols.begin(), cols.end(), o0);
            std::shuffle(cols.begin(), cols.end(), rng);
            const int64_t S = max_samples <= 0 ? (o1 - o0) : std::min(o1 - o0, max_samples);
            const int off = (int) pl.rows.size();
            pl.rows.insert(pl.rows.end(), cols.begin(), cols.begin() + S);
            add_block(b, off, (int) S);
        }
    } else {
        const int64_t n_eff = std::min(n_out, L);
        std::vector<int64_t> cols((size_t) n_eff);
        std::iota(cols.begin(), cols.end(), 0);
        std::shuffle(cols.begin(), cols.end(), rng);
        const int64_t S = max_samples <= 0 ? n_eff : std::min(n_eff, max_samples);
        pl.rows.assign(cols.begin(), cols.begin() + S);
        for (int64_t b = 0; b < B; ++b) {
            if (std::min(n_in, b * L + L) - b * L <= 0) continue;
            add_block(b, 0, (int) S);
        }
    }
    return pl;
}
static RngMemo<FitPlan> & fit_memo() {
    static RngMemo<FitPlan> & memo = *new RngMemo<FitPlan>(32);
    return memo;
}
static std::shared_ptr<const FitPlan> plan_fit(std::mt19937 & rng, int64_t n_in, int64_t n_out, int64_t L, int64_t B, int64_t max_samples, int trials) {
    return fit_memo().get(rng, MemoKey{3, n_in, n_out, L, B, max_samples, trials}, [&](std::mt19937 & g) {
        return std::make_shared<const FitPlan>(plan_fit_draw(g, n_in, n_out, L, B, max_samples, trials));
    });
}

static DenseView view_or_null(const sdh_dense * d) {
    return d ? ((const DeviceDense *) d)->v : DenseView{};
}

static void clamp_budget(const sdh_scan_params & p, int64_t n_in, int64_t * K_eff, int64_t * keep) {
    const int64_t Kc = std::max<int64_t>(1, std::min<int64_t>(p.K, n_in));
    const int64_t blk = std::max<int64_t>(1, p.block);
    *keep = (Kc + blk - 1) / blk;
    *K_eff = p.scheme == 1 ? *keep * blk : Kc;
}

} // namespace sdh

using namespace sdh;

extern "C" {

static int sdh_dense_create_impl(const sdh_tensor * W, sdh_dense ** out) {
    if (!W || !out || !W->data) { set_error("sdh_dense_create: null argument"); return SDH_ERR_INVALID; }
    LibState & ls = lib();
    if (!ls.inited) { set_error("sdh_init() has not been called"); return SDH_ERR_NO_DEVICE; }
    const int64_t n_in = W->ne[0], n_out = W->ne[1];
    if (n_in <= 0 || n_out <= 0 || n_in > INT32_MAX / 2 || n_out > INT32_MAX / 2) {
        set_error("sdh_dense_create: bad dims %lld x %lld", (long long) n_in, (long long) n_out);
        return SDH_ERR_INVALID;
    }
    DeviceDense * d = new DeviceDense();
    d->v.type = W->type; d->v.n_in = n_in; d->v.n_out = n_out;
    const uint8_t * src = (const uint8_t *) W->data;
    auto fail = [&](int rc) { (void) hipFree(d->alloc0); (void) hipFree(d->alloc1); delete d; return rc; };
    if (W->type == SDH_TYPE_Q8_0) {
        if (n_in % 32) { set_error("Q8_0 needs n_in %% 32 == 0"); return fail(SDH_ERR_INVALID); }
        const int64_t nblk = n_in / 32;
        std::vector<int8_t> q((size_t) (n_out * n_in));
        std::vector<uint16_t> sc((size_t) (n_out * nblk));
        for (int64_t r = 0; r < n_out; ++r) {
            const uint8_t * row = src + r * W->nb[1];
            for (int64_t b = 0; b < nblk; ++b) {
                std::memcpy(&sc[(size_t) (r * nblk + b)], row + b * 34, 2);
                std::memcpy(&q[(size_t) (r * n_in + b * 32)], row + b * 34 + 2, 32);
            }
        }
        if (hipMalloc(&d->alloc0, q.size()) != hipSuccess || hipMalloc(&d->alloc1, sc.size() * 2) != hipSuccess) {
            set_error("hipMalloc failed for a dense weight"); return fail(SDH_ERR_HIP);
        }
        (void) hipMemcpy(d->alloc0, q.data(), q.size(), hipMemcpyHostToDevice);
        (void) hipMemcpy(d->alloc1, sc.data(), sc.size() * 2, hipMemcpyHostToDevice);
        d->v.q = (const int8_t *) d->alloc0; d->v.d = (const __half *) d->alloc1;
        d->stored_bytes = n_out * nblk * 34;
    } else if (W->type == SDH_TYPE_F32 || W->type == SDH_TYPE_F16) {
        const size_t es = W->type == SDH_TYPE_F32 ? 4 : 2;
        if ((size_t) W->nb[0] != es) { set_error("dense rows must be contiguous"); return fail(SDH_ERR_INVALID); }
        if (hipMalloc(&d->alloc0, (size_t) (n_out * n_in) * es) != hipSuccess) {
            set_error("hipMalloc failed for a dense weight"); return fail(SDH_ERR_HIP);
        }
        (void) hipMemcpy2D(d->alloc0, (size_t) n_in * es, src, (size_t) W->nb[1], (size_t) n_in * es, (size_t) n_out, hipMemcpyHostToDevice);
        d->v.data = d->alloc0;
        d->stored_bytes = n_out * n_in * (int64_t) es;
    } else {
        set_error("dense weight type %d not supported (F32, F16, Q8_0)", W->type);
        return fail(SDH_ERR_UNSUPPORTED);
    }
    if (hipGetLastError() != hipSuccess) { set_error("upload of a dense weight failed"); return fail(SDH_ERR_HIP); }
    *out = (sdh_dense *) d;
    return SDH_OK;
}

void sdh_dense_destroy(sdh_dense * W) {
    if (!W) return;
    DeviceDense * d = (DeviceDense *) W;
    (void) hipFree(d->alloc0); (void) hipFree(d->alloc1);
    delete d;
}

int64_t sdh_dense_bytes(const sdh_dense * W) { return W ? ((const DeviceDense *) W)->stored_bytes : -1; }

int sdh_imatrix_scale(const float * imatrix, int64_t n_in, float eps, float power, float * scale_out) {
    if (!imatrix || !scale_out || n_in <= 0) { set_error("sdh_imatrix_scale: bad argument"); return SDH_ERR_INVALID; }
    eps = std::max(eps, 0.0f);
    power = std::max(power, 0.0f);
    double sum = 0.0;
    for (int64_t i = 0; i < n_in; ++i) {
        float sc = std::pow(std::max(imatrix[i], eps), 0.5f * power);
        if (std::isnan(sc) || std::isinf(sc)) sc = 1.0f;
        scale_out[i] = sc;
        sum += (double) sc;
    }
    const double mean = sum / std::max<double>(1.0, (double) n_in);
    if (mean > 0.0) {
        const float inv_mean = (float) (1.0 / mean);
        for (int64_t i = 0; i < n_in; ++i) scale_out[i] *= inv_mean;
    }
    return SDH_OK;
}

// test hook: n normals after `skip` raw draws of mt19937(seed), by fast_normals (out_fast) and by the standard
// library's normal_distribution (out_std); returns 1 when the two generators end in the same state
int sdh_test_normals(uint32_t seed, int64_t skip, int64_t n, float * out_fast, float * out_std) {
    if (n < 0 || skip < 0 || !out_fast || !out_std) return -1;
    std::mt19937 a(seed), b(seed);
    a.discard((unsigned long long) skip); b.discard((unsigned long long) skip);
    fast_normals(a, n, out_fast);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    for (int64_t i = 0; i < n; ++i) out_std[i] = nd(b);
    return a == b ? 1 : 0;
}

int sdh_scan_create(uint32_t seed, sdh_scan ** out) {
    if (!out) { set_error("sdh_scan_create: null argument"); return SDH_ERR_INVALID; }
    if (!lib().inited) { set_error("sdh_init() has not been called"); return SDH_ERR_NO_DEVICE; }
    Scan * s = new Scan(seed);
    if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess ||
        hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipEventCreate / hipStreamCreate failed"); delete s; return SDH_ERR_HIP;
    }
    *out = (sdh_scan *) s;
    return SDH_OK;
}

void sdh_scan_memo_clear(void) {
    normals_memo().clear();
    rows_memo().clear();
    fit_memo().clear();
    normals_pool().release_all();       // buffers still held by a running candidate return to the pool later
}

int sdh_scan_reseed(sdh_scan * h, uint32_t seed) {
    if (!h) { set_error("sdh_scan_reseed: null handle"); return SDH_ERR_INVALID; }
    ((Scan *) h)->rng.seed(seed);
    return SDH_OK;
}

void sdh_scan_destroy(sdh_scan * h) {
    if (!h) return;
    Scan * s = (Scan *) h;
    if (scan_timing() && s->n_cand)
        fprintf(stderr, "[sdh] normal draws so far: %ld calls %.1f ms total (incl. waiting), %ld misses %.1f ms drawing\n",
                g_normal_calls.load(), g_normal_call_us.load() / 1e3, g_normal_misses.load(), g_normal_miss_us.load() / 1e3);
    if (scan_timing() && s->n_cand)
        fprintf(stderr, "[sdh] scan handle %p: %ld candidates; ms per candidate: fit %.3f select-all %.3f draws %.3f launches %.3f wait %.3f summaries %.3f tail %.3f\n",
                (void *) s, s->n_cand, 1e3 * s->t_phase[0] / s->n_cand, 1e3 * s->t_phase[1] / s->n_cand, 1e3 * s->t_phase[2] / s->n_cand,
                1e3 * s->t_phase[6] / s->n_cand, 1e3 * s->t_phase[3] / s->n_cand, 1e3 * s->t_phase[4] / s->n_cand, 1e3 * s->t_phase[5] / s->n_cand);
    if (s->ev0) (void) hipEventDestroy(s->ev0);
    if (s->ev1) (void) hipEventDestroy(s->ev1);
    if (s->stream) { (void) hipStreamSynchronize(s->stream); (void) hipStreamDestroy(s->stream); }
    delete s;
}

static int sdh_scan_skip_impl(sdh_scan * h, int64_t n_in, int64_t n_out, const sdh_scan_params * p) {
    if (!h || !p || n_in <= 0 || n_out <= 0) { set_error("sdh_scan_skip: bad argument"); return SDH_ERR_INVALID; }
    Scan & s = *(Scan *) h;
    if (p->write_base) {
        int64_t L, B;
        fit_dims(n_in, n_out, &L, &B);
        (void) plan_fit(s.rng, n_in, n_out, L, B, p->base_max_samples, p->base_perm_trials);
    }
    const int passes = p->w_scale ? 2 : 1;               // the _w / _x_w passes draw their own samples
    for (int k = 0; k < passes && p->eval_cols > 0; ++k) (void) draw_rows(s.rng, n_out, p->eval_cols);
    for (int k = 0; k < passes && p->eval_cols > 0 && p->eval_x > 0; ++k) {
        (void) draw_rows(s.rng, n_out, p->eval_cols);
        (void) draw_normals(s.rng, p->eval_x * n_in);
    }
    return SDH_OK;
}

// One tensor, a schedule of nK budgets (Ks; nK = 1 is sdh_scan_tensor): one fit, one selection at the largest K,
// the metric passes of every K queued behind it, one copy, one synchronisation. outs[j] is what a lone
// sdh_scan_tensor with K = Ks[j] returns from the same generator state; the generator advances once.
static int scan_grid_impl(sdh_scan * h, const sdh_dense * Wh, const sdh_dense * Wg, const sdh_dense * Wu,
                    const sdh_dense * Wd, const sdh_scan_params * pp, const int64_t * Ks, int nK, sdh_scan_result * outs) {
    if (!h || !Wh || !pp || !outs || !Ks) { set_error("sdh_scan_tensor: null argument"); return SDH_ERR_INVALID; }
    if (nK < 1 || nK > SDH_SCAN_MAX_K) { set_error("scan: a K schedule holds 1..%d budgets, got %d", SDH_SCAN_MAX_K, nK); return SDH_ERR_INVALID; }
    Scan & s = *(Scan *) h;
    const sdh_scan_params & p = *pp;
    hipStream_t st = s.stream;
    const DenseView W = ((const DeviceDense *) Wh)->v;
    const int64_t n_in = W.n_in, n_out = W.n_out;
    const bool tall = n_out >= n_in;
    std::memset(outs, 0, sizeof(*outs) * (size_t) nK);
    if (p.scheme != 0 && p.scheme != 1) { set_error("scan: scheme must be 0 (coo) or 1 (block)"); return SDH_ERR_INVALID; }
    if (p.scheme == 1 && p.block <= 0) { set_error("scan: block must be > 0"); return SDH_ERR_INVALID; }
    // rejected before any draw or state write: a caller that falls back to one call per budget starts from the
    // generator state and the handle contents it had before this call
    const bool clamp_applies = p.delta_norm_clamp_down > 0.0 && p.kind == 2 && p.scheme == 0;   // ffn_down, d_val non-empty
    if (clamp_applies && nK > 1) {
        set_error("scan: delta_norm_clamp_down rescales the residual values of each K: evaluate such budgets one by one");
        return SDH_ERR_UNSUPPORTED;
    }

    int64_t K_eff = 0, keep = 0;                             // of the largest budget: what is selected and stored
    int64_t K_eff_j[SDH_SCAN_MAX_K], keep_j[SDH_SCAN_MAX_K];
    KGrid kg{};
    kg.n = nK;
    for (int j = 0; j < nK; ++j) {
        sdh_scan_params pj = p;
        pj.K = Ks[j];
        clamp_budget(pj, n_in, &K_eff_j[j], &keep_j[j]);
        K_eff = std::max(K_eff, K_eff_j[j]); keep = std::max(keep, keep_j[j]);
        kg.k[j] = (int) (p.scheme == 1 ? keep_j[j] : K_eff_j[j]);
        kg.ord[j] = j;
        outs[j].K_eff = K_eff_j[j]; outs[j].n_blocks = keep_j[j];
    }
    std::stable_sort(kg.ord, kg.ord + nK, [&](int a, int b) { return kg.k[a] < kg.k[b]; });
    const int64_t Ksel = p.scheme == 1 ? keep : K_eff;       // entries of idx per stored row
    int rc;
    SDH_HIP_CHECK(hipEventRecord(s.ev0, st));
    const float * ws_dev = nullptr;
    if (p.w_scale) {
        s.h_ws.assign(p.w_scale, p.w_scale + n_in);
        if ((rc = s.wsd.put(s.h_ws, st))) return rc;
        ws_dev = s.wsd.p;
    }

    // ---- base fit -------------------------------------------------------------------------------------
    int64_t L = 0, B = 0;
    BaseView bv{};
    ++s.n_cand;
    std::unique_ptr<PhaseTimer> ph(new PhaseTimer(s.t_phase[0]));
    if (p.write_base && p.base_mode != 0) {
        fit_dims(n_in, n_out, &L, &B);
        if (!s.has_base || s.fit_of != (const void *) Wh || s.L != L || s.B != B) {
            set_error("scan: base_mode %d needs a previous fit of the same tensor on this handle", p.base_mode);
            return SDH_ERR_INVALID;
        }
        if (p.base_mode == 2) (void) plan_fit(s.rng, n_in, n_out, L, B, p.base_max_samples, p.base_perm_trials);
        bv.h2 = s.h2.p; bv.inv = s.inv.p; bv.L = (int) L; bv.B = (int) B; bv.tall = tall ? 1 : 0; bv.log2L = 0; while ((1 << bv.log2L) < bv.L) ++bv.log2L;
    } else if (p.write_base) {
        fit_dims(n_in, n_out, &L, &B);
        if (L > 16384) { set_error("scan: base block length %lld too large", (long long) L); return SDH_ERR_UNSUPPORTED; }
        const std::shared_ptr<const FitPlan> plp = plan_fit(s.rng, n_in, n_out, L, B, p.base_max_samples, p.base_perm_trials);
        const FitPlan & pl = *plp;
        const int n_jobs = (int) pl.jobs.size();
        if (!pl.device()) { set_error("scan: upload of the fit plan failed"); return SDH_ERR_HIP; }
        s.fit_plan = plp;
        if ((rc = s.dbl.ensure((size_t) n_jobs * 3 * L))) return rc;
        const int threads = 256;
        if (L < threads) { set_error("scan: base block length %lld < 256 not supported on the GPU path", (long long) L); return SDH_ERR_UNSUPPORTED; }
        // Q8_0 with whole 32-blocks per base block: rows staged through LDS; anything else: the gather kernel
        size_t max_rows = 0;
        for (const FitJob & j : pl.jobs) max_rows = std::max(max_rows, (size_t) j.n_rows);
        const size_t fit_lds = (size_t) L * 4 + ((max_rows + 3) & ~(size_t) 3) * 4 +
                               2 * FIT_ROWS_PER_STAGE * ((size_t) L + (size_t) (L >> 5) * 2) + (ws_dev ? (size_t) L * 4 : 0);
        static const bool fit_gather_only = getenv("SDH_SCAN_FIT_GATHER") != nullptr;      // A/B runs
        if (W.type == SDH_TYPE_Q8_0 && L >= 1024 && L <= 4096 && fit_lds <= 160 * 1024 && !fit_gather_only) {
            static const bool fit_256 = getenv("SDH_SCAN_FIT_T256") != nullptr;               // A/B: the first shape
            static const bool attr_ok =
                hipFuncSetAttribute((const void *) scan_fit_lds_q8_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                hipFuncSetAttribute((const void *) scan_fit_lds_q8_kernel<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                hipFuncSetAttribute((const void *) scan_fit_lds_q8_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            if (!attr_ok) { set_error("scan: raising the dynamic LDS limit failed"); return SDH_ERR_HIP; }
            // XCD-aware deal (see the kernel): the jobs of a base block are consecutive in the plan
            int xcd_per = 0, jobs_per_block = 1;
            {
                int G = 0, last_b = -1;
                for (const FitJob & j : pl.jobs) if (j.b != last_b) { ++G; last_b = j.b; }
                static const bool no_remap = getenv("SDH_SCAN_FIT_NO_XCD") != nullptr;       // A/B
                if (G > 0 && n_jobs % G == 0 && 8 % G == 0 && !no_remap) {
                    const int tr = n_jobs / G, per = 8 / G, M = tr * (int) (L / 1024);
                    bool grouped = true;
                    for (int j = 0; j < n_jobs; ++j) grouped &= pl.jobs[j].b == pl.jobs[j / tr * tr].b;
                    if (grouped && M % per == 0) { xcd_per = per; jobs_per_block = tr; }
                }
            }
            static const bool fit_slow = getenv("SDH_SCAN_FIT_GENERAL") != nullptr;          // A/B: the general body
            const bool fast = !ws_dev && !fit_slow;                  // per job: blocks wholly inside the inputs take the short body
            if (fit_256)
                hipLaunchKernelGGL((scan_fit_lds_q8_kernel<256, false>), dim3((unsigned) (L / 256), (unsigned) n_jobs), dim3(256), fit_lds, st,
                                   W, (int) L, tall ? 1 : 0, pl.jobs_dev, pl.rows_dev, pl.perms_dev, ws_dev, s.dbl.p, 0, 1);
            else if (fast)
                hipLaunchKernelGGL((scan_fit_lds_q8_kernel<1024, true>), dim3((unsigned) (L / 1024), (unsigned) n_jobs), dim3(1024), fit_lds, st,
                                   W, (int) L, tall ? 1 : 0, pl.jobs_dev, pl.rows_dev, pl.perms_dev, ws_dev, s.dbl.p, xcd_per, jobs_per_block);
            else
                hipLaunchKernelGGL((scan_fit_lds_q8_kernel<1024, false>), dim3((unsigned) (L / 1024), (unsigned) n_jobs), dim3(1024), fit_lds, st,
                                   W, (int) L, tall ? 1 : 0, pl.jobs_dev, pl.rows_dev, pl.perms_dev, ws_dev, s.dbl.p, xcd_per, jobs_per_block);
        } else {
            SDH_LAUNCH_DT(W, scan_fit_kernel, dim3((unsigned) (L / threads), (unsigned) n_jobs), dim3(threads),
                          (size_t) L * sizeof(int), st, W, (int) L, tall ? 1 : 0, pl.jobs_dev, pl.rows_dev, pl.perms_dev, ws_dev, s.dbl.p);
        }
        SDH_HIP_CHECK(hipGetLastError());
        // the tail — best trial per block, h2, d2 = FWHT(h2)/L, perm1 and its inverse — stays on the device: the
        // candidate runs through to its metric passes without a synchronisation (host copies: fetch_fit_host)
        const size_t n_fit = (size_t) (L * B);
        if ((rc = s.sse.ensure((size_t) n_jobs)) || (rc = s.h2.ensure(n_fit)) || (rc = s.d2.ensure(n_fit)) || (rc = s.perm1.ensure(n_fit)) ||
            (rc = s.inv.ensure(n_fit)) || (rc = s.d1.ensure(n_fit))) return rc;
        static const bool tail_attr_ok =
            hipFuncSetAttribute((const void *) scan_fit_sse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8) == hipSuccess &&
            hipFuncSetAttribute((const void *) scan_fit_pick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4) == hipSuccess;
        if (!tail_attr_ok) { set_error("scan: raising the dynamic LDS limit failed"); return SDH_ERR_HIP; }
        hipLaunchKernelGGL(scan_fit_sse_kernel, dim3((unsigned) n_jobs), dim3(256), (size_t) L * 8, st, s.dbl.p, (int) L, s.sse.p);
        hipLaunchKernelGGL(scan_fit_pick_kernel, dim3((unsigned) B), dim3(256), (size_t) L * 4, st, s.dbl.p, s.sse.p, pl.jobs_dev, n_jobs,
                           pl.perms_dev, (int) L, s.h2.p, s.d2.p, s.perm1.p, s.inv.p, s.d1.p);
        SDH_HIP_CHECK(hipGetLastError());
        s.host_fit_valid = false;
        bv.h2 = s.h2.p; bv.inv = s.inv.p; bv.L = (int) L; bv.B = (int) B; bv.tall = tall ? 1 : 0; bv.log2L = 0; while ((1 << bv.log2L) < bv.L) ++bv.log2L;
        s.fit_of = (const void *) Wh;
    }
    for (int j = 0; j < nK; ++j) { outs[j].L = L; outs[j].B = B; s.kg_ksel[j] = kg.k[j]; }
    s.kg_n = nK; s.pick = nK - 1; s.ld = Ksel;
    s.n_in = n_in; s.n_out = n_out; s.K = s.kg_ksel[s.pick]; s.block = p.scheme == 1 ? p.block : 1; s.scheme = p.scheme;
    s.L = L; s.B = B; s.has_base = p.write_base != 0; s.has_rs = p.write_row_scale != 0;
    if (!p.write_base) s.fit_of = nullptr;

    // ---- selection ------------------------------------------------------------------------------------
    // select(rows_dev, n_sel): rows_dev == null selects every output row into [n_out] slots (the payload);
    // otherwise only the listed rows, into compact slots [n_sel] in list order (metrics_only)
    const bool metrics_only = p.metrics_only != 0 && !clamp_applies;
    const int compact = metrics_only ? 1 : 0;
    const size_t r_bytes = ((size_t) n_in * 4 + 15) & ~(size_t) 15;
    static const bool lds_attr_set = [] {
        bool ok = true;
        ok &= hipFuncSetAttribute((const void *) scan_select_coo_kernel<SDH_TYPE_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *) scan_select_coo_kernel<SDH_TYPE_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *) scan_select_coo_kernel<SDH_TYPE_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *) scan_select_block_kernel<SDH_TYPE_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *) scan_select_block_kernel<SDH_TYPE_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        ok &= hipFuncSetAttribute((const void *) scan_select_block_kernel<SDH_TYPE_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        return ok;
    }();
    if (!lds_attr_set) { set_error("scan: raising the dynamic LDS limit failed"); return SDH_ERR_HIP; }
    auto select = [&](const int * rows_dev, int64_t n_sel) -> int {
        int rc2;
        if ((rc2 = s.idx.ensure((size_t) n_sel * (size_t) Ksel)) || (rc2 = s.val.ensure((size_t) n_sel * (size_t) K_eff)) ||
            (rc2 = s.rs.ensure((size_t) n_sel * (size_t) nK))) return rc2;
        if (n_sel <= 0) return SDH_OK;
        if (p.scheme == 0) {
            int Kpad = 2;
            while (Kpad < K_eff) Kpad <<= 1;
            const size_t lds = r_bytes + (size_t) Kpad * 8 + 256 * 4 + 4 * 8 + 8 * 4;
            if (lds > 160 * 1024) {
                set_error("scan: n_in=%lld with K=%lld needs %zu B of LDS (max 163840)", (long long) n_in, (long long) K_eff, lds);
                return SDH_ERR_UNSUPPORTED;
            }
            SDH_LAUNCH_DT(W, scan_select_coo_kernel, dim3((unsigned) n_sel), dim3(256), lds, st, W, bv, p.write_base ? 1 : 0,
                          (int) K_eff, Kpad, p.write_row_scale ? 1 : 0, kg, rows_dev, ws_dev, s.idx.p, s.val.p, s.rs.p);
        } else {
            const int64_t n_blocks = (n_in + p.block - 1) / p.block;
            int NBpad = 2;
            while (NBpad < n_blocks) NBpad <<= 1;
            const size_t lds = r_bytes + (size_t) NBpad * 8 + (size_t) (NBpad + (NBpad & 1)) * 4 + 4 * 8;
            if (lds > 160 * 1024) {
                set_error("scan: n_in=%lld block=%lld needs %zu B of LDS (max 163840)", (long long) n_in, (long long) p.block, lds);
                return SDH_ERR_UNSUPPORTED;
            }
            SDH_LAUNCH_DT(W, scan_select_block_kernel, dim3((unsigned) n_sel), dim3(256), lds, st, W, bv, p.write_base ? 1 : 0,
                          (int) p.block, (int) keep, (int) n_blocks, NBpad, p.write_row_scale ? 1 : 0, kg, rows_dev, ws_dev, s.idx.p, s.val.p, s.rs.p);
        }
        SDH_HIP_CHECK(hipGetLastError());
        return SDH_OK;
    };
    s.emitted = !metrics_only;
    ph.reset(); ph.reset(new PhaseTimer(s.t_phase[1]));
    if (!metrics_only && (rc = select(nullptr, n_out))) return rc;
    sdh_scan_result * out = outs;                            // the clamp acts on a lone budget only
    for (int j = 0; j < nK; ++j) outs[j].delta_clamp_scale = 1.0;
    if (clamp_applies && K_eff > 0) {
        if ((rc = s.dbl.ensure((size_t) n_out * 2))) return rc;
        SDH_LAUNCH_DT(W, scan_row_sumsq_dense_kernel, dim3((unsigned) n_out), dim3(256), 0, st, W, s.dbl.p);
        hipLaunchKernelGGL(scan_row_sumsq_vals_kernel, dim3((unsigned) n_out), dim3(256), 0, st, s.val.p, (int) K_eff, s.dbl.p + n_out);
        SDH_HIP_CHECK(hipGetLastError());
        if (s.h_dbl.size() < (size_t) n_out * 2) s.h_dbl.resize((size_t) n_out * 2);
        const double * sq = s.h_dbl.data();
        SDH_HIP_CHECK(hipMemcpyAsync(s.h_dbl.data(), s.dbl.p, (size_t) n_out * 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        SDH_HIP_CHECK(hipStreamSynchronize(st));
        double w2 = 0.0, d2sum = 0.0;
        for (int64_t o = 0; o < n_out; ++o) { w2 += sq[(size_t) o]; d2sum += sq[(size_t) (n_out + o)]; }
        const double w_l2 = w2 > 0.0 ? std::sqrt(w2) : 0.0, delta_norm = d2sum > 0.0 ? std::sqrt(d2sum) : 0.0;
        const double limit = p.delta_norm_clamp_down * w_l2;
        if (w_l2 > 0.0 && limit > 0.0 && delta_norm > limit) {
            const double scale = limit / delta_norm;
            const int64_t nv = n_out * K_eff;
            hipLaunchKernelGGL(scan_scale_vals_kernel, dim3((unsigned) ((nv + 255) / 256)), dim3(256), 0, st, s.val.p, nv, scale);
            SDH_HIP_CHECK(hipGetLastError());
            out->delta_clamp_scale = scale;
        }
    }

    // ---- cost model (sd_cost.cpp:7-40) --------------------------------------------------------------
    for (int j = 0; j < nK; ++j) {
        double * c = outs[j].cost;
        c[2] = 2.0 * (double) n_in * (double) n_out;
        c[4] = 2.0 * (double) K_eff_j[j] * (double) n_out;
        c[5] = p.write_row_scale ? (double) n_out : 0.0;
        if (p.write_base && L > 0 && B > 0) {
            c[0] = (double) L; c[1] = (double) B;
            const double l = (double) L, lg = std::log2(std::max(2.0, l));
            c[3] = (double) B * (2.0 * l * lg + 3.0 * l + (tall ? 0.0 : l));
        }
        c[6] = c[3] + c[4] + c[5];
        c[7] = c[2] > 0.0 ? c[6] / c[2] : 0.0;
    }

    // ---- metric passes ---------------------------------------------------------------------------------
    // em, [em_w], em_x, [em_x_w] draw their samples in this order in run_trial (:735-795); the FFN proxy
    // (sd_ffn_proxy.cpp:20-247: COO, no base, gate/up only) has a generator of its own. No pass consumes
    // random numbers after its draws, so every draw is made first; the sampled rows of all passes are then
    // selected by ONE launch into consecutive compact slots, the metric kernels of all passes are queued
    // behind it, and one copy + one synchronisation bring every partial result back.
    struct Pass {
        int kind = 0;                         // 0 weight space, 1 activation space, 2 proxy
        const float * ws_dev = nullptr;       // kind 0: imatrix scale on the device, or null
        const float * x_scale_host = nullptr; // kind 1: imatrix scale on the host, or null
        int which = 0;                        // result field: 0 em, 1 em_w, 2 em_x, 3 em_x_w
        std::shared_ptr<const std::vector<int>> rows;
        std::shared_ptr<const Normals> normals;
        std::vector<float> scaled;            // kind 1 with x_scale_host
        int row0 = 0, nr = 0, nx = 0;         // position in the concatenated row list = first compact slot
        size_t out0 = 0;                      // first double of its results in s.dbl
        int j = -1;                           // -1: one pass serves every K of the schedule (shared draws); else the pass of trial j
        size_t xs0 = 0;                       // kind 1 with x_scale_host: first float of its scaled inputs in s.xs
        int nkp = 1;                          // K entries it evaluates: nK or 1
    };
    // seq: one builder run whose trials follow each other (the autotune loop, sd_build.cpp:843-913): trial j draws its
    // eval samples after trial j-1's; the generator state after every trial is kept for sdh_scan_rewind
    const bool seq = p.kgrid_draws == 1 && nK > 1;
    ph.reset(); ph.reset(new PhaseTimer(s.t_phase[2]));
    std::vector<Pass> passes;
    for (int j = 0; j < nK; ++j) outs[j].has_w = ws_dev ? 1 : 0;
    auto o5_of = [](sdh_scan_result & r, int which) { return which == 0 ? r.em : which == 1 ? r.em_w : which == 2 ? r.em_x : r.em_x_w; };
    for (int jj = 0; jj < (seq ? nK : 1); ++jj) {
    if (p.eval_cols > 0) {
        for (int k = 0; k < (ws_dev ? 2 : 1); ++k) {
            Pass ps; ps.kind = 0; ps.ws_dev = k ? ws_dev : nullptr; ps.which = k ? 1 : 0;
            ps.j = seq ? jj : -1; ps.nkp = seq ? 1 : nK;
            ps.rows = draw_rows(s.rng, n_out, p.eval_cols);
            passes.push_back(std::move(ps));
        }
    }
    if (p.eval_cols > 0 && p.eval_x > 0) {
        for (int j = 0; j < nK; ++j) outs[j].has_x = 1;
        for (int k = 0; k < (ws_dev ? 2 : 1); ++k) {
            Pass ps; ps.kind = 1; ps.x_scale_host = k ? p.w_scale : nullptr; ps.which = k ? 3 : 2;
            ps.j = seq ? jj : -1; ps.nkp = seq ? 1 : nK;
            ps.rows = draw_rows(s.rng, n_out, p.eval_cols);
            ps.nx = (int) p.eval_x;
            ps.normals = draw_normals(s.rng, (int64_t) ps.nx * n_in);
            if (ps.x_scale_host) {
                const std::vector<float> & g = ps.normals->v;
                ps.scaled.resize(g.size());
                for (int xi = 0; xi < ps.nx; ++xi)
                    for (int64_t i = 0; i < n_in; ++i)
                        ps.scaled[(size_t) xi * (size_t) n_in + (size_t) i] = g[(size_t) xi * (size_t) n_in + (size_t) i] * ps.x_scale_host[i];   // sd_eval.cpp:1084-1087
            }
            passes.push_back(std::move(ps));
        }
    }
    s.rng_after[jj] = s.rng;
    }
    s.n_after = seq ? nK : 1;
    const bool want_proxy = p.want_proxy && p.scheme == 0 && !p.write_base && p.eval_x > 0 && p.kind != 2 && Wg && Wu && Wd;
    std::shared_ptr<const std::vector<int>> proxy_outs;
    int64_t proxy_cap = 0;
    if (want_proxy) {
        const DenseView G = view_or_null(Wg), D = view_or_null(Wd);
        proxy_cap = p.eval_cols > 0 ? p.eval_cols : 128;
        std::mt19937 prng((uint32_t) p.proxy_seed);
        Pass ps; ps.kind = 2; ps.nkp = nK;
        ps.rows = draw_rows(prng, G.n_out, std::min(G.n_out, proxy_cap));          // sampled hidden units
        proxy_outs = draw_rows(prng, D.n_out, std::min(D.n_out, proxy_cap));       // sampled outputs of W_down
        ps.nx = (int) std::min<int64_t>(p.eval_x, 128);
        ps.normals = draw_normals(prng, (int64_t) ps.nx * G.n_in);
        passes.push_back(std::move(ps));
    }

    ph.reset(); ph.reset(new PhaseTimer(s.t_phase[6]));
    bool ev1_recorded = false;
    if (!passes.empty()) {
        // rows of all passes, one upload; results of all passes, one buffer
        int total_rows = 0;
        size_t total_dbl = 0, n_xpass = 0, total_xs = 0;
        Scan::ProxyTruth & pt = s.proxy_truth;
        bool have_truth = false;
        for (Pass & ps : passes) {
            ps.nr = (int) ps.rows->size();
            ps.row0 = total_rows; total_rows += ps.nr;
            ps.out0 = total_dbl;
            if (ps.kind == 0) total_dbl += (size_t) ps.nr * 3 * ps.nkp;                      // [nkp][nr][3]
            else if (ps.kind == 1) {                                                          // truth, hat [nkp]
                total_dbl += (size_t) ps.nr * ps.nx * (1 + ps.nkp); ++n_xpass;
                if (ps.x_scale_host) { ps.xs0 = total_xs; total_xs += ps.scaled.size(); }
            }
            else {
                const uint64_t id_g = ((const DeviceDense *) Wg)->serial, id_u = ((const DeviceDense *) Wu)->serial, id_d = ((const DeviceDense *) Wd)->serial;
                have_truth = pt.g == id_g && pt.u == id_u && pt.d == id_d && pt.seed == (int64_t) p.proxy_seed && pt.cap == proxy_cap &&
                             pt.nx == ps.nx && pt.gu.size() == (size_t) ps.nr * ps.nx * 2 && pt.wd.size() == proxy_outs->size() * (size_t) ps.nr;
                total_dbl += (size_t) ps.nr * ps.nx * (nK + (have_truth ? 0 : 2));   // hat [nK] [, gate truth, up truth]
            }
        }
        if ((rc = s.dbl.ensure(total_dbl))) return rc;
        if (s.h_dbl.size() < total_dbl) s.h_dbl.resize(total_dbl);
        std::vector<int> all_rows((size_t) total_rows);
        for (const Pass & ps : passes) std::copy(ps.rows->begin(), ps.rows->end(), all_rows.begin() + ps.row0);
        if ((rc = s.rows.put(all_rows, st))) return rc;
        if (metrics_only && (rc = select(s.rows.p, total_rows))) return rc;
        const float * rs_all = p.write_row_scale ? s.rs.p : nullptr;
        const int64_t rs_stride = metrics_only ? total_rows : n_out;      // slots of the selection launch
        const size_t v_per = (size_t) p.eval_x * (size_t) B * (size_t) L;
        if (p.write_base && n_xpass && (rc = s.v.ensure(n_xpass * v_per))) return rc;
        if (total_xs && (rc = s.xs.ensure(total_xs))) return rc;      // every scaled-input pass its own region: the kernels run later
        size_t xi_pass = 0;
        int no = 0;
        for (Pass & ps : passes) {
            // payload slots of this pass: compact -> its own run of slots, else indexed by output row
            const int *   idx_p = s.idx.p + (compact ? (size_t) ps.row0 * (size_t) Ksel : 0);
            const float * val_p = s.val.p + (compact ? (size_t) ps.row0 * (size_t) K_eff : 0);
            // a pass of one trial evaluates that trial's K alone: a one-entry schedule, its row scales at [j][slot]
            KGrid kgp = kg;
            if (ps.j >= 0) { kgp.n = 1; kgp.k[0] = kg.k[ps.j]; }
            const unsigned nkp = (unsigned) ps.nkp;
            const float * rs_p  = rs_all ? rs_all + (compact ? ps.row0 : 0) + (ps.j >= 0 ? (size_t) ps.j * (size_t) rs_stride : 0) : nullptr;
            const int *   rows_p = s.rows.p + ps.row0;
            double * o = s.dbl.p + ps.out0;
            if (ps.kind == 0) {
                SDH_LAUNCH_DT(W, scan_evalw_kernel, dim3((unsigned) ps.nr, nkp), dim3(256), 0, st, W, bv, p.write_base ? 1 : 0, p.scheme,
                              kgp, (int) Ksel, (int) s.block, idx_p, val_p, rs_p, rs_stride, ps.ws_dev, rows_p, compact, o);
            } else if (ps.kind == 1) {
                const float * xs_dev = nullptr;            // unscaled draws: the copy shared by all handles
                if (ps.x_scale_host) {
                    SDH_HIP_CHECK(hipMemcpyAsync(s.xs.p + ps.xs0, ps.scaled.data(), ps.scaled.size() * sizeof(float), hipMemcpyHostToDevice, st));
                    xs_dev = s.xs.p + ps.xs0;
                } else if (!(xs_dev = ps.normals->device())) {
                    set_error("scan: upload of the eval_x inputs failed"); return SDH_ERR_HIP;
                }
                const float * vdev = nullptr;
                if (p.write_base) {
                    float * vb = s.v.p + xi_pass * v_per;
                    if ((rc = launch_base_apply_generic(xs_dev, n_in, (int) n_in, (int) L, (int) B, tall, s.d1.p, s.d2.p, nullptr,
                                                        s.perm1.p, nullptr, ps.nx, vb, st))) return rc;
                    vdev = vb;
                }
                ++xi_pass;
                SDH_LAUNCH_DT(W, scan_dense_dot_kernel, dim3((unsigned) ps.nr, (unsigned) ps.nx), dim3(256), 0, st, W, rows_p, ps.nr, xs_dev, o);
                hipLaunchKernelGGL(scan_delta_dot_kernel, dim3((unsigned) ps.nr, (unsigned) ps.nx, nkp), dim3(256), 0, st, (int) n_in, p.scheme,
                                   kgp, (int) Ksel, (int) s.block, idx_p, val_p, rs_p, rs_stride, vdev, (int) L, (int) B, tall ? 1 : 0,
                                   rows_p, ps.nr, compact, xs_dev, o + (size_t) ps.nr * ps.nx);
            } else {
                const DenseView G = view_or_null(Wg), U = view_or_null(Wu), D = view_or_null(Wd);
                const float * xs_dev = ps.normals->device();
                if (!xs_dev) { set_error("scan: upload of the proxy inputs failed"); return SDH_ERR_HIP; }
                const int nh = ps.nr;
                no = (int) proxy_outs->size();
                hipLaunchKernelGGL(scan_delta_dot_kernel, dim3((unsigned) nh, (unsigned) ps.nx, (unsigned) nK), dim3(256), 0, st, (int) n_in, 0, kg, (int) Ksel, 1,
                                   idx_p, val_p, (const float *) nullptr, (int64_t) 0, (const float *) nullptr, 0, 0, 1, rows_p, nh, compact, xs_dev, o);
                double * truth = o + (size_t) nh * ps.nx * nK;
                if (!have_truth) {
                    // true gate / up pre-activations of the sampled hidden units and W_down[out rows][hidden cols]:
                    // functions of the three dense tensors and the proxy seed only, kept for the next K
                    if ((rc = s.cols.put(*proxy_outs, st)) || (rc = s.gat.ensure((size_t) no * nh))) return rc;
                    if (s.h_flt.size() < (size_t) no * nh) s.h_flt.resize((size_t) no * nh);
                    SDH_LAUNCH_DT(G, scan_dense_dot_kernel, dim3((unsigned) nh, (unsigned) ps.nx), dim3(256), 0, st, G, rows_p, nh, xs_dev, truth);
                    SDH_LAUNCH_DT(U, scan_dense_dot_kernel, dim3((unsigned) nh, (unsigned) ps.nx), dim3(256), 0, st, U, rows_p, nh, xs_dev, truth + (size_t) nh * ps.nx);
                    SDH_LAUNCH_DT(D, scan_gather_kernel, dim3((unsigned) ((no * nh + 255) / 256)), dim3(256), 0, st, D, s.cols.p, no, rows_p, nh, s.gat.p);
                    pt.seed = -1;                          // invalid until the copies below have landed
                    SDH_HIP_CHECK(hipMemcpyAsync(s.h_flt.data(), s.gat.p, (size_t) no * nh * sizeof(float), hipMemcpyDeviceToHost, st));
                }
            }
        }
        SDH_HIP_CHECK(hipGetLastError());
        SDH_HIP_CHECK(hipMemcpyAsync(s.h_dbl.data(), s.dbl.p, total_dbl * sizeof(double), hipMemcpyDeviceToHost, st));
        SDH_HIP_CHECK(hipEventRecord(s.ev1, st));              // end of the candidate's device work
        ev1_recorded = true;
        ph.reset(); ph.reset(new PhaseTimer(s.t_phase[3]));
        SDH_HIP_CHECK(hipStreamSynchronize(st));
        ph.reset(); ph.reset(new PhaseTimer(s.t_phase[4]));

        for (const Pass & ps : passes) {
            const double * r = s.h_dbl.data() + ps.out0;
            const int nr = ps.nr, nx = ps.nx;
            if (ps.kind == 0) {
                std::vector<double> rel((size_t) nr), cs((size_t) nr), nrm((size_t) nr);
                for (int jp = 0; jp < ps.nkp; ++jp) {                // results [jp] belong to trial ps.j, or to K number jp
                    const int j = ps.j >= 0 ? ps.j : jp;
                    const double * rj = r + (size_t) jp * nr * 3;
                    for (int i = 0; i < nr; ++i) triple_metrics(rj[(size_t) i * 3], rj[(size_t) i * 3 + 1], rj[(size_t) i * 3 + 2], &rel[(size_t) i], &cs[(size_t) i], &nrm[(size_t) i]);
                    summarise(rel, cs, nrm, o5_of(outs[j], ps.which));
                }
            } else if (ps.kind == 1) {
                const double * yt = r;
                std::vector<double> rel((size_t) nx), cs((size_t) nx), nrm((size_t) nx);
                for (int jp = 0; jp < ps.nkp; ++jp) {
                    const int j = ps.j >= 0 ? ps.j : jp;
                    const double * yh = r + (size_t) nr * nx * (size_t) (1 + jp);
                    for (int xi = 0; xi < nx; ++xi) {
                        double n1 = 0.0, n2 = 0.0, dot = 0.0;
                        for (int ci = 0; ci < nr; ++ci) {
                            const double a = yt[(size_t) xi * nr + ci], b = yh[(size_t) xi * nr + ci];
                            n1 += a * a; n2 += b * b; dot += a * b;
                        }
                        triple_metrics(n1, n2, dot, &rel[(size_t) xi], &cs[(size_t) xi], &nrm[(size_t) xi]);
                    }
                    summarise(rel, cs, nrm, o5_of(outs[j], ps.which));
                }
            } else {
                const int nh = nr;
                if (!have_truth) {
                    pt.gu.assign(r + (size_t) nh * nx * nK, r + (size_t) nh * nx * (nK + 2));
                    pt.wd.assign(s.h_flt.data(), s.h_flt.data() + (size_t) no * nh);
                    pt.g = ((const DeviceDense *) Wg)->serial; pt.u = ((const DeviceDense *) Wu)->serial; pt.d = ((const DeviceDense *) Wd)->serial;
                    pt.seed = (int64_t) p.proxy_seed; pt.cap = proxy_cap; pt.nx = nx;
                }
                no = (int) proxy_outs->size();
                const std::vector<double> & gu = pt.gu;
                const std::vector<float> & wd = pt.wd;
                for (int j = 0; j < nK; ++j) {
                sdh_scan_result * out = &outs[j];
                const double * hat = r + (size_t) j * nh * nx;
                std::vector<double> cs_all, l2_all, lnr_all;
                auto silu = [](double v) { return v / (1.0 + std::exp(-v)); };
                std::vector<double> at((size_t) nh), ah((size_t) nh);
                for (int xi = 0; xi < nx; ++xi) {
                    const double * gt = gu.data() + (size_t) xi * nh, * ut = gu.data() + (size_t) nh * nx + (size_t) xi * nh;
                    const double * ht = hat + (size_t) xi * nh;
                    // silu(g)*u per hidden sample, once per x (the reference evaluates it inside the output loop;
                    // (silu(g)*u)*w associates the same way, so the sums are bit-identical)
                    for (int hi = 0; hi < nh; ++hi) {
                        const double g_hat = p.kind == 0 ? ht[hi] : gt[hi], u_hat = p.kind == 1 ? ht[hi] : ut[hi];
                        at[(size_t) hi] = silu(gt[hi]) * ut[hi];
                        ah[(size_t) hi] = silu(g_hat) * u_hat;
                    }
                    double dot = 0.0, n1 = 0.0, n2 = 0.0, l2 = 0.0;
                    for (int oi = 0; oi < no; ++oi) {
                        double yt = 0.0, yh = 0.0;
                        for (int hi = 0; hi < nh; ++hi) {
                            const double w = (double) wd[(size_t) oi * nh + hi];
                            yt += at[(size_t) hi] * w;
                            yh += ah[(size_t) hi] * w;
                        }
                        dot += yt * yh; n1 += yt * yt; n2 += yh * yh;
                        const double d = yt - yh;
                        l2 += d * d;
                    }
                    const double den = std::sqrt(std::max(n1, 1e-20)) * std::sqrt(std::max(n2, 1e-20));
                    cs_all.push_back(den > 0.0 ? dot / den : 0.0);
                    l2_all.push_back(std::sqrt(std::max(l2, 0.0)));
                    const double nrat = n1 > 0.0 ? std::sqrt(std::max(n2, 1e-20)) / std::sqrt(std::max(n1, 1e-20)) : 0.0;
                    if (nrat > 0.0) lnr_all.push_back(std::log(nrat));
                }
                auto mean = [](const std::vector<double> & v) { return v.empty() ? 0.0 : std::accumulate(v.begin(), v.end(), 0.0) / (double) v.size(); };
                out->proxy[0] = mean(cs_all);  out->proxy[1] = percentile(cs_all, 0.05);
                out->proxy[2] = mean(l2_all);  out->proxy[3] = percentile(l2_all, 0.95);
                out->proxy[4] = mean(lnr_all); out->proxy[5] = percentile(lnr_all, 0.95);
                out->proxy[6] = (double) nx;   out->proxy[7] = (double) no;
                out->proxy_available = 1;
                }
            }
        }
    }

    ph.reset(); ph.reset(new PhaseTimer(s.t_phase[5]));
    if (!ev1_recorded) {
        SDH_HIP_CHECK(hipEventRecord(s.ev1, st));
        SDH_HIP_CHECK(hipEventSynchronize(s.ev1));
    }
    float ms = 0.0f;
    SDH_HIP_CHECK(hipEventElapsedTime(&ms, s.ev0, s.ev1));
    for (int j = 0; j < nK; ++j) outs[j].gpu_ms = (double) ms / (double) nK;      // an equal share of the schedule's device time
    return SDH_OK;
}

static int sdh_scan_rewind_impl(sdh_scan * h, int j) {
    if (!h) { set_error("sdh_scan_rewind: null handle"); return SDH_ERR_INVALID; }
    Scan & s = *(Scan *) h;
    if (j < 0 || j >= s.n_after) { set_error("sdh_scan_rewind: trial %d of %d", j, s.n_after); return SDH_ERR_INVALID; }
    s.rng = s.rng_after[j];
    return SDH_OK;
}

static int sdh_scan_pick_impl(sdh_scan * h, int j) {
    if (!h) { set_error("sdh_scan_pick: null handle"); return SDH_ERR_INVALID; }
    Scan & s = *(Scan *) h;
    if (s.n_out <= 0 || j < 0 || j >= s.kg_n) { set_error("sdh_scan_pick: member %d of a schedule of %d", j, s.n_out > 0 ? s.kg_n : 0); return SDH_ERR_INVALID; }
    s.pick = j;
    s.K = s.kg_ksel[j];
    return SDH_OK;
}

// host copies of the last fit (the builder's staging of base_d2 / base_h2 / base_perm1), made when first asked for
static int fetch_fit_host(Scan & s) {
    if (s.host_fit_valid) return SDH_OK;
    const size_t n = (size_t) (s.L * s.B);
    s.h_h2.resize(n); s.h_d2.resize(n); s.h_perm1.resize(n); s.h_inv.resize(n);
    SDH_HIP_CHECK(hipMemcpyAsync(s.h_h2.data(), s.h2.p, n * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    SDH_HIP_CHECK(hipMemcpyAsync(s.h_d2.data(), s.d2.p, n * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    SDH_HIP_CHECK(hipMemcpyAsync(s.h_perm1.data(), s.perm1.p, n * sizeof(int), hipMemcpyDeviceToHost, s.stream));
    SDH_HIP_CHECK(hipMemcpyAsync(s.h_inv.data(), s.inv.p, n * sizeof(int), hipMemcpyDeviceToHost, s.stream));
    SDH_HIP_CHECK(hipStreamSynchronize(s.stream));
    s.host_fit_valid = true;
    return SDH_OK;
}

static int sdh_scan_fetch_impl(sdh_scan * h, int32_t * idx, float * val, float * row_scale,
                   float * d2, float * h2, int32_t * perm1, int32_t * perm1_inv) {
    if (!h) { set_error("sdh_scan_fetch: null handle"); return SDH_ERR_INVALID; }
    Scan & s = *(Scan *) h;
    if (s.n_out <= 0) { set_error("sdh_scan_fetch: no candidate evaluated yet"); return SDH_ERR_INVALID; }
    hipStream_t st = s.stream;
    const size_t n_idx = (size_t) s.n_out * (size_t) s.K;
    if ((idx || val || row_scale) && !s.emitted) {
        set_error("sdh_scan_fetch: the candidate was evaluated with metrics_only, no payload was kept");
        return SDH_ERR_INVALID;
    }
    // the picked K is a prefix of every stored row (rows are s.ld entries apart)
    const size_t w_idx = (size_t) s.K * sizeof(int32_t), w_val = (size_t) s.K * (size_t) s.block * sizeof(float);
    if (idx && n_idx) SDH_HIP_CHECK(hipMemcpy2DAsync(idx, w_idx, s.idx.p, (size_t) s.ld * sizeof(int32_t), w_idx, (size_t) s.n_out, hipMemcpyDeviceToHost, st));
    if (val && n_idx) SDH_HIP_CHECK(hipMemcpy2DAsync(val, w_val, s.val.p, (size_t) s.ld * (size_t) s.block * sizeof(float), w_val, (size_t) s.n_out, hipMemcpyDeviceToHost, st));
    if (row_scale) {
        if (!s.has_rs) { set_error("sdh_scan_fetch: candidate had no row scale"); return SDH_ERR_INVALID; }
        SDH_HIP_CHECK(hipMemcpyAsync(row_scale, s.rs.p + (size_t) s.pick * (size_t) s.n_out, (size_t) s.n_out * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    SDH_HIP_CHECK(hipStreamSynchronize(st));
    if (d2 || h2 || perm1 || perm1_inv) {
        if (!s.has_base) { set_error("sdh_scan_fetch: candidate had no base"); return SDH_ERR_INVALID; }
        int rc;
        if ((rc = fetch_fit_host(s))) return rc;
        const size_t n = (size_t) (s.L * s.B);
        if (d2) std::memcpy(d2, s.h_d2.data(), n * sizeof(float));
        if (h2) std::memcpy(h2, s.h_h2.data(), n * sizeof(float));
        if (perm1) std::memcpy(perm1, s.h_perm1.data(), n * sizeof(int32_t));
        if (perm1_inv) std::memcpy(perm1_inv, s.h_inv.data(), n * sizeof(int32_t));
    }
    return SDH_OK;
}

static int sdh_scan_fetch_packed_impl(sdh_scan * h, int idx_type, void * idx, int val_type, void * val, uint16_t * row_scale_f16,
                          uint16_t * d2_f16, void * perm1, int * perm1_type) {
    if (!h) { set_error("sdh_scan_fetch_packed: null handle"); return SDH_ERR_INVALID; }
    Scan & s = *(Scan *) h;
    if (s.n_out <= 0) { set_error("sdh_scan_fetch_packed: no candidate evaluated yet"); return SDH_ERR_INVALID; }
    if ((idx || val || row_scale_f16) && !s.emitted) {
        set_error("sdh_scan_fetch_packed: the candidate was evaluated with metrics_only, no payload was kept");
        return SDH_ERR_INVALID;
    }
    if (idx && idx_type != SDH_TYPE_I16 && idx_type != SDH_TYPE_I32) { set_error("sdh_scan_fetch_packed: idx type must be I16 or I32"); return SDH_ERR_INVALID; }
    if (val && val_type != SDH_TYPE_F16 && val_type != SDH_TYPE_F32) { set_error("sdh_scan_fetch_packed: val type must be F16 or F32"); return SDH_ERR_INVALID; }
    if (row_scale_f16 && !s.has_rs) { set_error("sdh_scan_fetch_packed: candidate had no row scale"); return SDH_ERR_INVALID; }
    hipStream_t st = s.stream;
    const size_t n_idx = (size_t) s.n_out * (size_t) s.K, n_val = n_idx * (size_t) s.block, n_rs = (size_t) s.n_out;
    // one scratch buffer of 16-bit words: [idx | val | row_scale] as far as they are converted
    const size_t o_idx = 0, o_val = o_idx + (idx && idx_type == SDH_TYPE_I16 ? n_idx : 0),
                 o_rs = o_val + (val && val_type == SDH_TYPE_F16 ? n_val : 0), total = o_rs + (row_scale_f16 ? n_rs : 0);
    int rc;
    if ((rc = s.packed.ensure(total ? total : 1))) return rc;
    auto grid = [](size_t n) { return dim3((unsigned) ((n + 255) / 256)); };
    if (idx && idx_type == SDH_TYPE_I16 && n_idx)
        hipLaunchKernelGGL(scan_pack_i16_kernel, grid(n_idx), dim3(256), 0, st, s.idx.p, (int16_t *) (s.packed.p + o_idx), (int64_t) n_idx, s.K, s.ld);
    if (val && val_type == SDH_TYPE_F16 && n_val)
        hipLaunchKernelGGL(scan_pack_f16_kernel, grid(n_val), dim3(256), 0, st, s.val.p, s.packed.p + o_val, (int64_t) n_val, s.K * s.block, s.ld * s.block);
    if (row_scale_f16)                                                   // row_scale is always F16 (:1188-1191)
        hipLaunchKernelGGL(scan_pack_f16_kernel, grid(n_rs), dim3(256), 0, st, s.rs.p + (size_t) s.pick * n_rs, s.packed.p + o_rs, (int64_t) n_rs, (int64_t) n_rs, (int64_t) n_rs);
    SDH_HIP_CHECK(hipGetLastError());
    if (idx) {
        if (idx_type == SDH_TYPE_I16) SDH_HIP_CHECK(hipMemcpyAsync(idx, s.packed.p + o_idx, n_idx * 2, hipMemcpyDeviceToHost, st));
        else if (n_idx)               SDH_HIP_CHECK(hipMemcpy2DAsync(idx, (size_t) s.K * 4, s.idx.p, (size_t) s.ld * 4, (size_t) s.K * 4, (size_t) s.n_out, hipMemcpyDeviceToHost, st));
    }
    if (val) {
        if (val_type == SDH_TYPE_F16) SDH_HIP_CHECK(hipMemcpyAsync(val, s.packed.p + o_val, n_val * 2, hipMemcpyDeviceToHost, st));
        else if (n_val)               SDH_HIP_CHECK(hipMemcpy2DAsync(val, (size_t) s.K * s.block * 4, s.val.p, (size_t) s.ld * s.block * 4, (size_t) s.K * s.block * 4, (size_t) s.n_out, hipMemcpyDeviceToHost, st));
    }
    if (row_scale_f16) SDH_HIP_CHECK(hipMemcpyAsync(row_scale_f16, s.packed.p + o_rs, n_rs * 2, hipMemcpyDeviceToHost, st));
    SDH_HIP_CHECK(hipStreamSynchronize(st));
    if (d2_f16 || perm1) {
        if (!s.has_base) { set_error("sdh_scan_fetch_packed: candidate had no base"); return SDH_ERR_INVALID; }
        if ((rc = fetch_fit_host(s))) return rc;
        const size_t n = (size_t) (s.L * s.B);
        if (d2_f16) for (size_t i = 0; i < n; ++i) d2_f16[i] = float_to_half(s.h_d2[i]);
        if (perm1) {
            const bool small = s.L <= 32768;                             // perm type rule of :1111
            if (perm1_type) *perm1_type = small ? SDH_TYPE_I16 : SDH_TYPE_I32;
            if (small) {
                int16_t * o = (int16_t *) perm1;
                for (size_t i = 0; i < n; ++i) { const int v = s.h_perm1[i]; o[i] = (v < 0 || v > 32767) ? (int16_t) 0 : (int16_t) v; }   // :1228
            } else {
                std::memcpy(perm1, s.h_perm1.data(), n * sizeof(int32_t));
            }
        }
    }
    return SDH_OK;
}

// ---- the ABI is C: nothing is thrown across it (host allocations can fail) ---------------------------
int sdh_dense_create(const sdh_tensor * W, sdh_dense ** out) {
    SDH_NO_THROW(sdh_dense_create_impl(W, out));
}

int sdh_scan_skip(sdh_scan * h, int64_t n_in, int64_t n_out, const sdh_scan_params * p) {
    SDH_NO_THROW(sdh_scan_skip_impl(h, n_in, n_out, p));
}

int sdh_scan_tensor(sdh_scan * h, const sdh_dense * Wh, const sdh_dense * Wg, const sdh_dense * Wu,
                    const sdh_dense * Wd, const sdh_scan_params * pp, sdh_scan_result * out) {
    if (!pp) { set_error("sdh_scan_tensor: null argument"); return SDH_ERR_INVALID; }
    const int64_t K = pp->K;
    SDH_NO_THROW(scan_grid_impl(h, Wh, Wg, Wu, Wd, pp, &K, 1, out));
}

int sdh_scan_tensor_kgrid(sdh_scan * h, const sdh_dense * Wh, const sdh_dense * Wg, const sdh_dense * Wu,
                          const sdh_dense * Wd, const sdh_scan_params * pp, const int64_t * Ks, int n_k, sdh_scan_result * outs) {
    SDH_NO_THROW(scan_grid_impl(h, Wh, Wg, Wu, Wd, pp, Ks, n_k, outs));
}

int sdh_scan_pick(sdh_scan * h, int j) {
    SDH_NO_THROW(sdh_scan_pick_impl(h, j));
}

int sdh_scan_rewind(sdh_scan * h, int j) {
    SDH_NO_THROW(sdh_scan_rewind_impl(h, j));
}

int sdh_scan_fetch(sdh_scan * h, int32_t * idx, float * val, float * row_scale,
                   float * d2, float * h2, int32_t * perm1, int32_t * perm1_inv) {
    SDH_NO_THROW(sdh_scan_fetch_impl(h, idx, val, row_scale, d2, h2, perm1, perm1_inv));
}

int sdh_scan_fetch_packed(sdh_scan * h, int idx_type, void * idx, int val_type, void * val, uint16_t * row_scale_f16,
                          uint16_t * d2_f16, void * perm1, int * perm1_type) {
    SDH_NO_THROW(sdh_scan_fetch_packed_impl(h, idx_type, idx, val_type, val, row_scale_f16, d2_f16, perm1, perm1_type));
}

} // extern "C"
