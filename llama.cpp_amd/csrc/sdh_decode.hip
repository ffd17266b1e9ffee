// sdh_decode.hip — batch-1 decode of a SeedDelta weight WITH a base, as ONE launch.
//
// At T = 1 the op is latency-bound: the payload (cfg3: 13 MB) streams in ~4 us, the base transform
// (2 FWHTs per block behind a cold parameter fetch) takes ~5 us on its own, and a dependent kernel
// boundary costs another ~1.5-2.5 us. Running the two as separate launches serialises them. Here
// the first B workgroups of the grid compute the base blocks while all the others stream the
// residual; a residual wave picks up the base term of its row at the very end.
//
// Hand-off (cdna_hip_programming.md Guideline 16, form R2 "the data IS the flag"): the base
// workgroups publish every output element as a naturally aligned 8-byte {value, epoch} granule
// written by ONE agent-scope (sc1, write-through) store; a consumer lane reads the granules of its
// row with agent-scope (sc1) 8-byte
// loads and accepts them only when the tag equals this launch's epoch. A granule is written by one
// store and read by one load, so no flag, fence or ordering between different addresses is
// involved, nothing depends on workgroup placement or dispatch order, and the 4096 polling lanes
// touch 4096 different addresses (a single shared flag word measured 3 us slower: every poll of the
// launch lands on one line). The epoch is a host counter bumped per launch, so granules of earlier
// launches are never accepted.
//
// Stream capture: a launch recorded into a HIP graph would replay its epoch, so a captured launch
// carries the reserved tag CAPTURE_TAG instead (never a host epoch) and sets `reset`: every residual
// workgroup stores 0 over the granules its lanes have read, once they have all finished polling
// (reset_granules). Each granule has exactly one reader, so when the kernel ends its granules are all
// cleared and the next replay (or any other captured launch of the same
// weight, which the stream or the graph's edges order after it) starts from tags that cannot match.
// The residency census needs a host round trip, so it cannot run during capture: a kernel/grid that
// has not been checked by an earlier eager launch declines, and the caller records the two-kernel
// path (warm up once before capturing, as callers of graphs do anyway).
//
// Safety: the consumers spin, so every workgroup of the grid must be resident together. The host
// only takes this path when grid <= occupancy x CUs, and verifies that once per kernel with a
// census launch (every workgroup checks in and waits, bounded, for all the others) — on gfx950 the
// occupancy API can over-report by one workgroup per CU. All spins are bounded by the real-time
// counter; a consumer that gives up writes NaN, which the non-finite counter reports.
#include "sdh_internal.h"
#include "sdh_base_dev.h"
#include "sdh_delta_dev.h"

#include <algorithm>
#include <type_traits>

namespace sdh {

constexpr unsigned long long SPIN_TICKS = 50000;      // 100 MHz ticks = 500 us
#ifndef SDH_POLL_SLEEP
#define SDH_POLL_SLEEP 4                              // s_sleep units of 64 clocks between polls
#endif
#ifndef SDH_BASE_PRIO
#define SDH_BASE_PRIO 3
#endif

struct FusedArgs {
    unsigned long long * vg;      // [B][L] granules {value, epoch}
    unsigned   epoch;
    int        reset;             // 1: consumers clear the granules they read (captured launches, see header)
    int        resid_delay;       // residual workgroups start after this many s_sleep(8) (512 clocks each)
    int        ncu;               // CUs of the device (round-robin period of the dispatcher); 0: no shadow workgroups
    int        linger;            // how long a shadow workgroup stays, in 100 MHz ticks
    int        delay_xcd_only;    // 1: only residual workgroups on a base workgroup's XCD take the head-start delay
    int        census;            // 1: census launch (no computation)
    unsigned * census_ctr;        // [0] arrivals, [1] failures
    const float4 * x4;            // several tokens, COO: token-interleaved x in global memory (DeltaLaunch::x4) or null
};

// Element i of a buffer as (uniform base + 32-bit byte offset): keeps the base in scalar registers
// and the access a global (not flat) one; the buffers here are far below 4 GiB.
template <typename T>
__device__ __forceinline__ T * at32(T * base, int i) {
    using Byte = std::conditional_t<std::is_const_v<T>, const char, char>;
    return reinterpret_cast<T *>(reinterpret_cast<Byte *>(base) + (size_t) ((unsigned) i * (unsigned) sizeof(T)));
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long * base, int i) {
    return __hip_atomic_load(at32(base, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait for the nbase <= N base granules of output o. One poll = all N loads issued back to back,
// then the tags. The loads are branch-free (slots past nbase re-read block 0): with a branch around
// each load hipcc waits for every load in turn, one memory-side round trip per base block.
constexpr unsigned CAPTURE_TAG = 0xFFFFFFFFu;        // tag of launches recorded into a graph; never a host epoch

template <int N>
__device__ __forceinline__ bool poll_granules(const FusedArgs & fa, int o, int L, int nbase, float (&vals)[8]) {
    bool ok = false;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    do {
        unsigned long long g[N];
#pragma unroll
        for (int b = 0; b < N; ++b) g[b] = granule_load(fa.vg, (b < nbase ? b * L : 0) + o);
        ok = true;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            vals[b] = 0.0f;
            if (b < N) {
                ok = ok & (((unsigned) (g[b] >> 32) == fa.epoch) | (b >= nbase));
                vals[b] = (b < nbase) ? __uint_as_float((unsigned) g[b]) : 0.0f;
            }
        }
        if (ok) break;
        __builtin_amdgcn_s_sleep(SDH_POLL_SLEEP);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < SPIN_TICKS);
    return ok;
}

// Captured launches (reset): the granules a residual workgroup has read are cleared by the LAST of its
// waves to finish polling, counted in LDS. Clearing from each lane right after its own poll measured
// +3.3 us on cfg3: those stores land on lines that the neighbouring lanes are still polling (a
// workgroup's rows are consecutive, so its granules of one base block are one contiguous run and, from
// 16 rows per workgroup up, whole lines that only this workgroup reads).
__device__ __forceinline__ void reset_begin(int * done) {
    if (threadIdx.x == 0) *done = 0;
    __syncthreads();
}
__device__ __forceinline__ bool reset_last_wave(int * done, int lane, int waves) {
    int prev = 0;
    if (lane == 0) prev = atomicAdd(done, 1);
    return __builtin_amdgcn_readfirstlane(prev) == waves - 1;
}
__device__ __forceinline__ void reset_granules(unsigned long long * vg, int row0, int nrows, int nbase, int L, int lane) {
    for (int i = lane; i < nrows * nbase; i += 64) {
        const int b = i / nrows, r = i - b * nrows;
        __hip_atomic_store(at32(vg, b * L + row0 + r), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// LOG_T: log2 of the workgroup size. The base role's chain (2 FWHTs + gathers) is the critical path of
// the launch and its time falls with the number of threads sharing a block (fewer elements and LDS
// accesses per thread), so L = 4096 runs 512-thread workgroups (E = 8); the residual role simply has
// 8 waves per workgroup instead of 4.
// Workgroup placement. The dispatcher deals the workgroups of a launch to the CUs round-robin in
// blockIdx order (measured, tools/placement.hip: grid 256k+4 puts the 4 extra workgroups on the CUs of
// workgroups 0-3). The base workgroups are blockIdx 0..nb-1, so blockIdx = w*ncu + i, i < nb, w >= 1 are
// the residual workgroups that would share a CU with a base workgroup and slow its chain down (FWHT 1
// measured 2.4 us with two residual workgroups on the CU, 1.4 us alone). Those ids are "shadows": they
// do no work (they linger ~2 us so that the dispatcher does not refill their slot, then leave) and the
// residual rows are dealt to the other ids. A performance assumption only: any placement is correct.
//   returns -1 for a shadow, else the residual index of this workgroup
__device__ __forceinline__ int residual_index(int bid, int nb, int ncu) {
    if (ncu <= 0 || bid < ncu) return bid - nb;
    const int wrap = bid / ncu;
    if (bid - wrap * ncu < nb) return -1;
    return bid - nb - nb * wrap;
}
__device__ __forceinline__ void shadow_linger(int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long) ticks) __builtin_amdgcn_s_sleep(16);      // 100 MHz ticks
}

// RW: waves per workgroup. The base role always runs on the first 2^LOG_T threads; RW = 2^LOG_T / 64 + 1
// is the "exclusive base CU" shape (see launch_decode_fused): one more residual row per workgroup, so
// that the rows fit on the CUs that carry no base workgroup.
template <bool BLOCK, int STEPS, int LOG_E, int LOG_T, int RW = (1 << LOG_T) / 64>
__global__ __launch_bounds__(RW * 64)
void decode_fused_kernel(const DeltaParams p, const BasePackDesc pd, const FusedArgs fa) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    constexpr int E = Ly::E;
    constexpr int WAVES = RW;
    __shared__ __align__(16) float lds[L];             // one transpose buffer: occupancy over barriers (two buffers: 7.80 vs 7.80 us)

    if (fa.census) {                                   // residency self-test, see header comment
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&fa.census_ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool all = false;
            while (__builtin_amdgcn_s_memrealtime() - t0 < 2 * SPIN_TICKS) {
                if (__hip_atomic_load(&fa.census_ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!all) __hip_atomic_fetch_add(&fa.census_ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    if ((int) blockIdx.x < p.B) {
        // ---- base role: one block of the base transform, published write-through + flag --------
        const int b = blockIdx.x;
        const int tid = threadIdx.x;
        if constexpr (RW * 64 > (1 << LOG_T)) { if (tid >= (1 << LOG_T)) return; }   // whole extra waves leave: barriers count live waves
        __builtin_amdgcn_s_setprio(SDH_BASE_PRIO);     // the base blocks are the critical path of the launch
#ifdef SDH_DBG_NOBASE              // dev A/B: ... without the base role's work
        return;
#endif
        float y[E];
        base_transform<Ly, false, true>(p.x, p.n_in, p.tall ? 0 : b * L, p.x_vec_ok != 0, pd, b, lds, tid, y);
        // Publish as {value, epoch} granules, one agent-scope (sc1, write-through) 8-byte store each,
        // straight from the last layout (consecutive lanes write consecutive granules). Measured
        // alternatives, all slower end to end on the cfg3 shape: two granules per 16-byte sc1 store
        // after an LDS re-order (+5 us: the stores collide with the pollers of the same lines), plain
        // f32 values + one flag per block (+2.5 us: 4096 pollers on one line).
        unsigned long long * out = fa.vg + (size_t) b * L;
        const unsigned long long tag = (unsigned long long) fa.epoch << 32;
#pragma unroll
        for (int r = 0; r < E; ++r)
            __hip_atomic_store(out + Ly::idx(Ly::J - 1, tid, r),
                               tag | (unsigned long long) __float_as_uint(y[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SDH_STAMP(6);
        return;
    }

    // ---- residual role ------------------------------------------------------------------------
    // A short head start for the base workgroups: their parameter and activation loads head the
    // critical path of the launch, and issued together with the payload flood of ~1000 residual
    // workgroups they queue behind it. A short delay costs the residual role nothing (it is done
    // long before the base). Measured on cfg3 with the streaming payload loads: 0 / 1 / 2 / 3 / 4 / 5 / 6 / 8
    // units = 9.27 / 8.76 / 8.08 / 8.04 / 8.14 / 8.24 / 8.43 / 8.90 us; default 3 (~0.7 us).
    const int ridx = residual_index((int) blockIdx.x, p.B, fa.ncu);
    if (ridx < 0) { shadow_linger(fa.linger); return; }
    int * const done = reinterpret_cast<int *>(lds);
    if (fa.reset) reset_begin(done);
    {
        // delay_xcd_only: workgroup ids are dealt to the 8 XCDs round-robin, so only ids with (id % 8) < B
        // share an L2 with a base workgroup; the others start at once (cfg3: 7.78 -> 7.67 us, other shapes
        // unchanged). A performance assumption only.
        const bool wait = !fa.delay_xcd_only || (int) (blockIdx.x & 7u) < p.B;
        if (wait) for (int i = 0; i < fa.resid_delay; ++i) __builtin_amdgcn_s_sleep(8);
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = p.G;
    const int l = lane & (G - 1);
    const int o = (ridx * WAVES + wave) * (64 / G) + (lane / G);
    const bool row_ok = o < p.n_out;
    const int orow = row_ok ? o : 0;
    const bool leader = row_ok && l == 0;
    float scale = 1.0f;
    if (leader && p.rs) scale = p.rs[o];

    SDH_STAMP(10);
#ifdef SDH_DBG_NOSTREAM            // dev A/B (tools/build_variants.sh): what the launch costs without the payload stream
    float acc = 0.0f;
#else
    float acc = delta_row_partial<uint16_t, uint16_t, BLOCK, STEPS, true>(p, p.x, orow, row_ok, l, G);
    acc = group_reduce(acc, G);
#endif
    SDH_STAMP(11);

    if (leader) {
        // base term: tall v_b[o - bL] with b = o / L (buffer index b*L + (o - b*L) == o);
        // wide sum_b v_b[o], accumulated from 0.0f in block order
        const int nbase = p.tall ? 1 : p.B;             // host guarantees nbase <= 8
        float vals[8];
        bool ok;
#ifdef SDH_DBG_NOPOLL              // dev A/B: ... without waiting for the base term
        ok = true;
        for (int b = 0; b < 8; ++b) vals[b] = 0.0f;
#else
        if (nbase == 1)      ok = poll_granules<1>(fa, o, L, nbase, vals);
        else if (nbase <= 4) ok = poll_granules<4>(fa, o, L, nbase, vals);
        else                 ok = poll_granules<8>(fa, o, L, nbase, vals);
#endif
        SDH_STAMP(12);
        float yh;
        if (p.tall) yh = vals[0];
        else {
            yh = 0.0f;
#pragma unroll
            for (int b = 0; b < 8; ++b) if (b < nbase) yh += vals[b];
        }
        float y = yh + acc;
        if (!ok) y = __uint_as_float(0x7FC00000u);       // base never arrived: surface it as a NaN
        store_output(p, o, 0, y, scale);
    }
    if (fa.reset && reset_last_wave(done, lane, WAVES)) {
        const int rows = WAVES * (64 / G), row0 = ridx * rows;
        reset_granules(fa.vg, row0, min(rows, p.n_out - row0), p.tall ? 1 : p.B, L, lane);
    }
}

// ---- small decode batches: 2..4 tokens in one launch -------------------------------------------------
// Speculative / batched decode hands the op a few tokens at a time. Per token the fused launch would
// repeat the 13 MB payload stream; the two-kernel path reads it once but pays the kernel boundary
// (cfg3: T=2 14.9 us; Mistral gate COO K=128: T=2 14.8, T=4 19.8 us). Here the grid has B base workgroups PER TOKEN and every residual wave
// forms the sums of its row for all tokens from one pass over the payload, then picks up n_tok x nbase
// granules with one batch of loads. Granules of token t live at vg + t*B*L. NT is the compile-time
// capacity (2 or 4), n_tok <= NT the tokens of this call: slots past n_tok re-read token 0 and are
// dropped. Same per-token arithmetic as the one-token kernel, so a token's result does not depend on the
// batch it arrives in. Measured: cfg3 T=2 12.2 us; Mistral gate T=2 9.9, T=4 15.4 us.
template <int NB, int NT>
__device__ __forceinline__ bool poll_granules_nt(const FusedArgs & fa, int o, int L, int B, int nbase, int n_tok, float (&vals)[NT][NB]) {
    bool ok = false;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    do {
        unsigned long long g[NT][NB];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) g[t][b] = granule_load(fa.vg, ((t < n_tok ? t : 0) * B + (b < nbase ? b : 0)) * L + o);
        ok = true;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bool used = (b < nbase) & (t < n_tok);
                ok = ok & (((unsigned) (g[t][b] >> 32) == fa.epoch) | !used);
                vals[t][b] = used ? __uint_as_float((unsigned) g[t][b]) : 0.0f;
            }
        if (ok) break;
        __builtin_amdgcn_s_sleep(SDH_POLL_SLEEP);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < SPIN_TICKS);
    return ok;
}

template <bool BLOCK, int STEPS, int LOG_E, int LOG_T, int NT>
__global__ __launch_bounds__(1 << LOG_T, (1 << LOG_T) == 512 ? 6 : 1)    // 512 threads: <= 80 VGPRs, three workgroups per CU
void decode_fused_nt_kernel(const DeltaParams p, const BasePackDesc pd, const FusedArgs fa, const int n_tok) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    constexpr int E = Ly::E;
    constexpr int WAVES = (1 << LOG_T) / 64;
    __shared__ __align__(16) float lds[L];

    if (fa.census) {                                   // residency self-test, as in decode_fused_kernel
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&fa.census_ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool all = false;
            while (__builtin_amdgcn_s_memrealtime() - t0 < 2 * SPIN_TICKS) {
                if (__hip_atomic_load(&fa.census_ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!all) __hip_atomic_fetch_add(&fa.census_ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    const int n_base_wg = p.B * n_tok;
    if ((int) blockIdx.x < n_base_wg) {
        const int t = (int) blockIdx.x / p.B, b = (int) blockIdx.x - t * p.B;
        const int tid = threadIdx.x;
        __builtin_amdgcn_s_setprio(SDH_BASE_PRIO);
        float y[E];
        base_transform<Ly, false, true>(p.x + (size_t) t * p.ldx, p.n_in, p.tall ? 0 : b * L, p.x_vec_ok != 0, pd, b, lds, tid, y);
        unsigned long long * out = fa.vg + (size_t) (t * p.B + b) * L;
        const unsigned long long tag = (unsigned long long) fa.epoch << 32;
#pragma unroll
        for (int r = 0; r < E; ++r)
            __hip_atomic_store(out + Ly::idx(Ly::J - 1, tid, r),
                               tag | (unsigned long long) __float_as_uint(y[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    const int ridx = residual_index((int) blockIdx.x, n_base_wg, fa.ncu);
    if (ridx < 0) { shadow_linger(fa.linger); return; }
    int * const done = reinterpret_cast<int *>(lds);
    if (fa.reset) reset_begin(done);
    for (int i = 0; i < fa.resid_delay; ++i) __builtin_amdgcn_s_sleep(8);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = p.G;
    const int l = lane & (G - 1);
    const int o = (ridx * WAVES + wave) * (64 / G) + (lane / G);
    const bool row_ok = o < p.n_out;
    const int orow = row_ok ? o : 0;
    const bool leader = row_ok && l == 0;
    float scale = 1.0f;
    if (leader && p.rs) scale = p.rs[o];

    int xoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xoff[t] = (t < n_tok ? t : 0) * (int) p.ldx;
    float acc[NT];
    delta_row_partial_nt<uint16_t, uint16_t, BLOCK, STEPS, NT, true>(p, p.x, xoff, orow, row_ok, l, G, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = group_reduce(acc[t], G);

    const int nbase = p.tall ? 1 : p.B;                 // host: nbase * NT <= 16
    if (leader) {
        float yh[NT];
        bool ok;
        if (nbase == 1) {
            float vals[NT][1];
            ok = poll_granules_nt<1, NT>(fa, o, L, p.B, nbase, n_tok, vals);
#pragma unroll
            for (int t = 0; t < NT; ++t) yh[t] = p.tall ? vals[t][0] : 0.0f + vals[t][0];   // wide, B = 1: summed from 0.0f
        } else {
            constexpr int NBW = (NT <= 2) ? 8 : 4;
            float vals[NT][NBW];
            ok = poll_granules_nt<NBW, NT>(fa, o, L, p.B, nbase, n_tok, vals);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                yh[t] = 0.0f;                           // wide: sum_b v_b[o] from 0.0f in block order
#pragma unroll
                for (int b = 0; b < NBW; ++b) if (b < nbase) yh[t] += vals[t][b];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < n_tok) {
                float y = yh[t] + acc[t];
                if (!ok) y = __uint_as_float(0x7FC00000u);
                store_output(p, o, t, y, scale);
            }
        }
    }
    if (fa.reset && reset_last_wave(done, lane, WAVES)) {
        const int rows = WAVES * (64 / G), row0 = ridx * rows, n = min(rows, p.n_out - row0);
        for (int t = 0; t < n_tok; ++t) reset_granules(fa.vg + (size_t) t * p.B * L, row0, n, nbase, L, lane);
    }
}

// ---- 2..4 tokens, COO, activations staged in LDS -------------------------------------------------------
// For COO payloads the multi-token launch above is bound by the L1 request rate of its activation gathers
// (one scattered 4-byte request per entry per token). When 16 * n_in bytes fit in LDS the residual
// workgroups first copy x token-interleaved into LDS (xs4[i] = x_0..3[i], tokens past n_tok zero), and a
// row entry then costs one 16-byte LDS read for all four tokens. Dynamic LDS: max(4 L, 16 n_in) + 16 bytes.
template <int STEPS, int LOG_E, int LOG_T>
__global__ __launch_bounds__(1 << LOG_T)
void decode_fused_nt_xlds_kernel(const DeltaParams p, const BasePackDesc pd, const FusedArgs fa, const int n_tok) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    constexpr int E = Ly::E;
    constexpr int WAVES = (1 << LOG_T) / 64;
    constexpr int THREADS = 1 << LOG_T;
    extern __shared__ __align__(16) float xlds[];

    if (fa.census) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&fa.census_ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool all = false;
            while (__builtin_amdgcn_s_memrealtime() - t0 < 2 * SPIN_TICKS) {
                if (__hip_atomic_load(&fa.census_ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!all) __hip_atomic_fetch_add(&fa.census_ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    const int n_base_wg = p.B * n_tok;
    if ((int) blockIdx.x < n_base_wg) {
        const int t = (int) blockIdx.x / p.B, b = (int) blockIdx.x - t * p.B;
        const int tid = threadIdx.x;
        __builtin_amdgcn_s_setprio(SDH_BASE_PRIO);
        float y[E];
        base_transform<Ly, false, true>(p.x + (size_t) t * p.ldx, p.n_in, p.tall ? 0 : b * L, p.x_vec_ok != 0, pd, b, xlds, tid, y);
        unsigned long long * out = fa.vg + (size_t) (t * p.B + b) * L;
        const unsigned long long tag = (unsigned long long) fa.epoch << 32;
#pragma unroll
        for (int r = 0; r < E; ++r)
            __hip_atomic_store(out + Ly::idx(Ly::J - 1, tid, r),
                               tag | (unsigned long long) __float_as_uint(y[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    const int ridx = residual_index((int) blockIdx.x, n_base_wg, fa.ncu);
    if (ridx < 0) { shadow_linger(fa.linger); return; }
    float4 * const xs4 = reinterpret_cast<float4 *>(xlds);
    int * const done = fa.x4 ? reinterpret_cast<int *>(xlds) : reinterpret_cast<int *>(xlds + 4 * (size_t) p.n_in);      // behind the staged activations
    for (int i = 0; i < fa.resid_delay; ++i) __builtin_amdgcn_s_sleep(8);
    if (fa.x4) {                                         // the producer left x token-interleaved in global memory: nothing to stage
        if (threadIdx.x == 0) *done = 0;
        __syncthreads();
    } else {   // stage x: 4 inputs x 4 tokens per thread and pass (host: n_in % 4 == 0, columns 16-byte aligned)
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int i4 = threadIdx.x; i4 < (p.n_in >> 2); i4 += THREADS) {
            const float4 a = *reinterpret_cast<const float4 *>(p.x + 4 * (size_t) i4);
            const float4 b = n_tok > 1 ? *reinterpret_cast<const float4 *>(p.x + (size_t) p.ldx + 4 * (size_t) i4) : z;
            const float4 c = n_tok > 2 ? *reinterpret_cast<const float4 *>(p.x + 2 * (size_t) p.ldx + 4 * (size_t) i4) : z;
            const float4 d = n_tok > 3 ? *reinterpret_cast<const float4 *>(p.x + 3 * (size_t) p.ldx + 4 * (size_t) i4) : z;
            xs4[4 * i4 + 0] = make_float4(a.x, b.x, c.x, d.x);
            xs4[4 * i4 + 1] = make_float4(a.y, b.y, c.y, d.y);
            xs4[4 * i4 + 2] = make_float4(a.z, b.z, c.z, d.z);
            xs4[4 * i4 + 3] = make_float4(a.w, b.w, c.w, d.w);
        }
        if (threadIdx.x == 0) *done = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = p.G;
    const int l = lane & (G - 1);
    const int o = (ridx * WAVES + wave) * (64 / G) + (lane / G);
    const bool row_ok = o < p.n_out;
    const int orow = row_ok ? o : 0;
    const bool leader = row_ok && l == 0;
    float scale = 1.0f;
    if (leader && p.rs) scale = p.rs[o];

    float acc[4];
    if (fa.x4) delta_row_partial_coo_lds4<uint16_t, uint16_t, STEPS, true>(p, fa.x4, orow, row_ok, l, G, acc);     // global gathers
    else       delta_row_partial_coo_lds4<uint16_t, uint16_t, STEPS, true>(p, xs4, orow, row_ok, l, G, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = group_reduce(acc[t], G);

    const int nbase = p.tall ? 1 : p.B;                 // host: nbase <= 4 for this kernel
    if (leader) {
        float yh[4];
        bool ok;
        if (nbase == 1) {
            float vals[4][1];
            ok = poll_granules_nt<1, 4>(fa, o, L, p.B, nbase, n_tok, vals);
#pragma unroll
            for (int t = 0; t < 4; ++t) yh[t] = p.tall ? vals[t][0] : 0.0f + vals[t][0];
        } else {
            float vals[4][4];
            ok = poll_granules_nt<4, 4>(fa, o, L, p.B, nbase, n_tok, vals);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                yh[t] = 0.0f;
#pragma unroll
                for (int b = 0; b < 4; ++b) if (b < nbase) yh[t] += vals[t][b];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < n_tok) {
                float y = yh[t] + acc[t];
                if (!ok) y = __uint_as_float(0x7FC00000u);
                store_output(p, o, t, y, scale);
            }
        }
    }
    if (fa.reset && reset_last_wave(done, lane, WAVES)) {
        const int rows = WAVES * (64 / G), row0 = ridx * rows, n = min(rows, p.n_out - row0);
        for (int t = 0; t < n_tok; ++t) reset_granules(fa.vg + (size_t) t * p.B * L, row0, n, nbase, L, lane);
    }
}

// ---- gate | up in one launch (gated FFN block, T = 1) ----------------------------------------------
// Both tensors are tall base payloads of one shape (same n_in, n_out, L, B, row shape). The first 2B
// workgroups run the base blocks of the gate, then of the up tensor; a residual wave computes row o
// of BOTH tensors (all loads of the two rows in flight together), its leader picks up the two base
// terms and writes h[o] = act(gate[o]) * up[o]: the two node outputs are never materialised.
struct PairArgs {
    DeltaParams  p[2];            // [0] gate, [1] up; p[t].y is unused, p[t].nf is the node's counter
    BasePackDesc pd[2];
    unsigned long long * vg[2];
    unsigned     epoch[2];
    int          reset;
    int          act;
    const uint16_t * gelu_tab;
    float *      h;
    float4 *     h4;              // several tokens: also h token-interleaved, h4[o] = (h_0[o] .. h_3[o]), or null
    int          resid_delay;
    int          ncu;
    int          linger;
    int          census;
    unsigned *   census_ctr;
};

template <bool BLOCK, int STEPS, int LOG_E, int LOG_T>
__global__ __launch_bounds__(1 << LOG_T)
void decode_pair_kernel(const PairArgs a) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    constexpr int E = Ly::E;
    constexpr int WAVES = (1 << LOG_T) / 64;
    __shared__ __align__(16) float lds[L];

    if (a.census) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&a.census_ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool all = false;
            while (__builtin_amdgcn_s_memrealtime() - t0 < 2 * SPIN_TICKS) {
                if (__hip_atomic_load(&a.census_ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!all) __hip_atomic_fetch_add(&a.census_ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    const int B = a.p[0].B;
    if ((int) blockIdx.x < 2 * B) {
        const bool second = (int) blockIdx.x >= B;
        const int b = (int) blockIdx.x - (second ? B : 0);
        const int tid = threadIdx.x;
        __builtin_amdgcn_s_setprio(SDH_BASE_PRIO);
        float y[E];
        const BasePackDesc & pd = second ? a.pd[1] : a.pd[0];
        base_transform<Ly, false, true>(a.p[0].x, a.p[0].n_in, 0, a.p[0].x_vec_ok != 0, pd, b, lds, tid, y);
        unsigned long long * out = (second ? a.vg[1] : a.vg[0]) + (size_t) b * L;
        const unsigned long long tag = (unsigned long long) (second ? a.epoch[1] : a.epoch[0]) << 32;
#pragma unroll
        for (int r = 0; r < E; ++r)
            __hip_atomic_store(out + Ly::idx(Ly::J - 1, tid, r),
                               tag | (unsigned long long) __float_as_uint(y[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    const int ridx = residual_index((int) blockIdx.x, 2 * B, a.ncu);
    if (ridx < 0) { shadow_linger(a.linger); return; }
    int * const done = reinterpret_cast<int *>(lds);
    if (a.reset) reset_begin(done);
    for (int i = 0; i < a.resid_delay; ++i) __builtin_amdgcn_s_sleep(8);     // head start for the base role (see decode_fused_kernel)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = a.p[0].G;
    const int l = lane & (G - 1);
    const int o = (ridx * WAVES + wave) * (64 / G) + (lane / G);
    const bool row_ok = o < a.p[0].n_out;
    const int orow = row_ok ? o : 0;
    const bool leader = row_ok && l == 0;
    float sg = 1.0f, su = 1.0f;
    if (leader && a.p[0].rs) sg = a.p[0].rs[o];
    if (leader && a.p[1].rs) su = a.p[1].rs[o];

    float accg = delta_row_partial<uint16_t, uint16_t, BLOCK, STEPS, true>(a.p[0], a.p[0].x, orow, row_ok, l, G);
    float accu = delta_row_partial<uint16_t, uint16_t, BLOCK, STEPS, true>(a.p[1], a.p[0].x, orow, row_ok, l, G);
    accg = group_reduce(accg, G);
    accu = group_reduce(accu, G);

    if (leader) {
        // tall: the base term of row o is granule o of its tensor (block o / L, element o % L)
        bool ok = false;
        float vg = 0.0f, vu = 0.0f;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        do {
            const unsigned long long g0 = granule_load(a.vg[0], o);
            const unsigned long long g1 = granule_load(a.vg[1], o);
            ok = ((unsigned) (g0 >> 32) == a.epoch[0]) & ((unsigned) (g1 >> 32) == a.epoch[1]);
            vg = __uint_as_float((unsigned) g0);
            vu = __uint_as_float((unsigned) g1);
            if (ok) break;
            __builtin_amdgcn_s_sleep(SDH_POLL_SLEEP);
        } while (__builtin_amdgcn_s_memrealtime() - t0 < SPIN_TICKS);
        float yg = vg + accg, yu = vu + accu;
        if (!ok) yg = yu = __uint_as_float(0x7FC00000u);
        yg = finish_output(a.p[0], yg, sg);
        yu = finish_output(a.p[1], yu, su);
        a.h[o] = glu_value(a.act, yg, yu, a.gelu_tab);
    }
    if (a.reset && reset_last_wave(done, lane, WAVES)) {  // tall: the granules of rows row0.. are vg[t][row0..]
        const int rows = WAVES * (64 / G), row0 = ridx * rows, n = min(rows, a.p[0].n_out - row0);
        reset_granules(a.vg[0], row0, n, 1, L, lane);
        reset_granules(a.vg[1], row0, n, 1, L, lane);
    }
}

// ---- gate | up in one launch, 2..4 tokens ----------------------------------------------------------
// The pair launch for small decode batches: 2 * B * n_tok base workgroups (gate blocks of every token, then up
// blocks), and a residual wave computes row o of BOTH tensors for all tokens with the multi-token partial sums of
// the single-tensor launches (delta_row_partial_nt, or — XLDS, COO with 16 * n_in bytes in LDS — x staged
// token-interleaved once per workgroup and shared by the two tensors: delta_row_partial_coo_lds4). Per token the
// arithmetic is the one-token kernel's, so h equals the node-by-node result bit for bit.
template <bool BLOCK, int STEPS, int LOG_E, int LOG_T, int NT, bool XLDS>
__global__ __launch_bounds__(1 << LOG_T, 4)             // <= 128 VGPRs: two 512-thread workgroups per CU
void decode_pair_nt_kernel(const PairArgs a, const int n_tok, const int64_t ldh) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    constexpr int E = Ly::E;
    constexpr int WAVES = (1 << LOG_T) / 64;
    constexpr int THREADS = 1 << LOG_T;
    static_assert(!XLDS || (!BLOCK && NT == 4), "the LDS-staged form is the COO one, four token slots");
    extern __shared__ __align__(16) float xlds[];       // max(L floats, XLDS: 4 * n_in floats) + 4

    if (a.census) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&a.census_ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool all = false;
            while (__builtin_amdgcn_s_memrealtime() - t0 < 2 * SPIN_TICKS) {
                if (__hip_atomic_load(&a.census_ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!all) __hip_atomic_fetch_add(&a.census_ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    const DeltaParams & pg = a.p[0];
    const int B = pg.B;
    const int per = B * n_tok, nb_tot = 2 * per;
    if ((int) blockIdx.x < nb_tot) {
        const bool second = (int) blockIdx.x >= per;
        const int rem = (int) blockIdx.x - (second ? per : 0);
        const int t = rem / B, b = rem - t * B;
        const int tid = threadIdx.x;
        __builtin_amdgcn_s_setprio(SDH_BASE_PRIO);
        float y[E];
        const BasePackDesc & pd = second ? a.pd[1] : a.pd[0];
        base_transform<Ly, false, true>(pg.x + (size_t) t * pg.ldx, pg.n_in, 0, pg.x_vec_ok != 0, pd, b, xlds, tid, y);
        unsigned long long * out = (second ? a.vg[1] : a.vg[0]) + (size_t) (t * B + b) * L;
        const unsigned long long tag = (unsigned long long) (second ? a.epoch[1] : a.epoch[0]) << 32;
#pragma unroll
        for (int r = 0; r < E; ++r)
            __hip_atomic_store(out + Ly::idx(Ly::J - 1, tid, r),
                               tag | (unsigned long long) __float_as_uint(y[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    const int ridx = residual_index((int) blockIdx.x, nb_tot, a.ncu);
    if (ridx < 0) { shadow_linger(a.linger); return; }
    int * done;
    for (int i = 0; i < a.resid_delay; ++i) __builtin_amdgcn_s_sleep(8);
    if constexpr (XLDS) {
        float4 * const xs4 = reinterpret_cast<float4 *>(xlds);
        done = reinterpret_cast<int *>(xlds + 4 * (size_t) pg.n_in);
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int i4 = threadIdx.x; i4 < (pg.n_in >> 2); i4 += THREADS) {       // host: n_in % 4 == 0, columns 16-byte aligned
            const float4 c0 = *reinterpret_cast<const float4 *>(pg.x + 4 * (size_t) i4);
            const float4 c1 = n_tok > 1 ? *reinterpret_cast<const float4 *>(pg.x + (size_t) pg.ldx + 4 * (size_t) i4) : z;
            const float4 c2 = n_tok > 2 ? *reinterpret_cast<const float4 *>(pg.x + 2 * (size_t) pg.ldx + 4 * (size_t) i4) : z;
            const float4 c3 = n_tok > 3 ? *reinterpret_cast<const float4 *>(pg.x + 3 * (size_t) pg.ldx + 4 * (size_t) i4) : z;
            xs4[4 * i4 + 0] = make_float4(c0.x, c1.x, c2.x, c3.x);
            xs4[4 * i4 + 1] = make_float4(c0.y, c1.y, c2.y, c3.y);
            xs4[4 * i4 + 2] = make_float4(c0.z, c1.z, c2.z, c3.z);
            xs4[4 * i4 + 3] = make_float4(c0.w, c1.w, c2.w, c3.w);
        }
        if (threadIdx.x == 0) *done = 0;
        __syncthreads();
    } else {
        done = reinterpret_cast<int *>(xlds);
        if (a.reset) reset_begin(done);
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = pg.G;
    const int l = lane & (G - 1);
    const int o = (ridx * WAVES + wave) * (64 / G) + (lane / G);
    const bool row_ok = o < pg.n_out;
    const int orow = row_ok ? o : 0;
    const bool leader = row_ok && l == 0;
    float sg = 1.0f, su = 1.0f;
    if (leader && a.p[0].rs) sg = a.p[0].rs[o];
    if (leader && a.p[1].rs) su = a.p[1].rs[o];

    float accg[NT], accu[NT];
    if constexpr (XLDS) {
        const float4 * xs4 = reinterpret_cast<const float4 *>(xlds);
        delta_row_partial_coo_lds4<uint16_t, uint16_t, STEPS, true>(a.p[0], xs4, orow, row_ok, l, G, accg);
        delta_row_partial_coo_lds4<uint16_t, uint16_t, STEPS, true>(a.p[1], xs4, orow, row_ok, l, G, accu);
    } else {
        int xoff[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) xoff[t] = (t < n_tok ? t : 0) * (int) pg.ldx;
        delta_row_partial_nt<uint16_t, uint16_t, BLOCK, STEPS, NT, true>(a.p[0], pg.x, xoff, orow, row_ok, l, G, accg);
        delta_row_partial_nt<uint16_t, uint16_t, BLOCK, STEPS, NT, true>(a.p[1], pg.x, xoff, orow, row_ok, l, G, accu);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) { accg[t] = group_reduce(accg[t], G); accu[t] = group_reduce(accu[t], G); }

    if (leader) {
        // tall: the base term of (token t, row o) is granule t * B * L + o of its tensor
        bool ok = false;
        float vg[NT], vu[NT];
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        do {
            unsigned long long g0[NT], g1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int gi = (t < n_tok ? t : 0) * B * L + o;
                g0[t] = granule_load(a.vg[0], gi);
                g1[t] = granule_load(a.vg[1], gi);
            }
            ok = true;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ok = ok & ((((unsigned) (g0[t] >> 32) == a.epoch[0]) & ((unsigned) (g1[t] >> 32) == a.epoch[1])) | (t >= n_tok));
                vg[t] = __uint_as_float((unsigned) g0[t]);
                vu[t] = __uint_as_float((unsigned) g1[t]);
            }
            if (ok) break;
            __builtin_amdgcn_s_sleep(SDH_POLL_SLEEP);
        } while (__builtin_amdgcn_s_memrealtime() - t0 < SPIN_TICKS);
        float hv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < n_tok) {
                float yg = vg[t] + accg[t], yu = vu[t] + accu[t];
                if (!ok) yg = yu = __uint_as_float(0x7FC00000u);
                yg = finish_output(a.p[0], yg, sg);
                yu = finish_output(a.p[1], yu, su);
                hv[t] = glu_value(a.act, yg, yu, a.gelu_tab);
                a.h[(size_t) t * (size_t) ldh + o] = hv[t];
            }
        }
        if (a.h4) a.h4[o] = make_float4(hv[0], hv[1], hv[2], hv[3]);
    }
    if (a.reset && reset_last_wave(done, lane, WAVES)) {
        const int rows = WAVES * (64 / G), row0 = ridx * rows, n = min(rows, pg.n_out - row0);
        for (int t = 0; t < n_tok; ++t) {
            reset_granules(a.vg[0] + (size_t) t * B * L, row0, n, 1, L, lane);
            reset_granules(a.vg[1] + (size_t) t * B * L, row0, n, 1, L, lane);
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------
// workgroup shape per base block length: L = 2^(LOG_E + LOG_T)
#if defined(SDH_DECODE_L4096_T256)    // build variants (A/B): 256 threads x 16 elements was the first version
#define SDH_L4096_LOG_E 4
#define SDH_L4096_LOG_T 8
#elif defined(SDH_DECODE_L4096_T1024)
#define SDH_L4096_LOG_E 2
#define SDH_L4096_LOG_T 10
#else
#define SDH_L4096_LOG_E 3
#define SDH_L4096_LOG_T 9
#endif
#if defined(SDH_DECODE_L2048_T512)
#define SDH_L2048_LOG_E 2
#define SDH_L2048_LOG_T 9
#else
#define SDH_L2048_LOG_E 3
#define SDH_L2048_LOG_T 8
#endif
static int decode_log_t(int L) { return L == 4096 ? SDH_L4096_LOG_T : L == 2048 ? SDH_L2048_LOG_T : 8; }
// head start of the base role, in s_sleep(8) units of 512 clocks; SDH_DECODE_DELAY overrides (A/B runs)
// grid size that holds nb base workgroups and n_res working residual workgroups when the ids that
// would land on a base workgroup's CU are left empty (see residual_index)
// Shadow ids are OFF by default: measured on the product build, back to back, they cost 0.5 us on cfg3
// (9.29 vs 8.77 us) although a base workgroup alone on its CU runs its chain faster; the in-kernel stamps
// that suggested a 2 us gain distort the comparison (the stamp stores sit on the base chain).
// SDH_DECODE_SHADOW=<linger ticks> turns them on for experiments.
static int shadow_linger_ticks() {
    static const int v = [] { const char * e = getenv("SDH_DECODE_SHADOW"); return e ? atoi(e) : -1; }();
    return v;
}
static int shadow_ncu() { return shadow_linger_ticks() >= 0 ? lib().num_cu : 0; }
static unsigned shadow_grid(int nb, int n_res, int ncu) {
    if (ncu <= 0 || nb >= ncu) return (unsigned) (nb + n_res);
    int first = ncu - nb;                          // residual ids in the first wrap
    if (n_res <= first) return (unsigned) (nb + n_res);
    const int per = ncu - nb;                      // working ids per later wrap (nb shadows each)
    const int rest = n_res - first;
    const int full = rest / per, tail = rest % per;
    return (unsigned) (ncu * (1 + full) + (tail ? nb + tail : 0));
}
static int resid_delay_units() {
    static const int v = [] { const char * e = getenv("SDH_DECODE_DELAY"); return e ? atoi(e) : 3; }();
    return v < 0 ? 0 : (v > 64 ? 64 : v);
}

// host epochs run 1 .. 0xFFFFFFFE: 0 is "cleared", CAPTURE_TAG belongs to captured launches
static unsigned next_epoch(unsigned & e) {
    ++e;
    if (e == 0 || e == CAPTURE_TAG) e = 1;
    return e;
}

struct FusedKernelInfo {
    int occ = -1;                 // resident workgroups per CU (occupancy API)
    unsigned census_ok_grid = 0;  // largest grid the census has confirmed
    bool census_failed = false;
};

template <bool BLOCK, int STEPS, int LOG_E, int LOG_T, int RW>
static FusedKernelInfo & kernel_info() {
    static FusedKernelInfo info;
    return info;
}

template <bool BLOCK, int STEPS, int LOG_E, int LOG_T, int RW = (1 << LOG_T) / 64>
static int launch_inst(const DeltaParams & p, const BasePackDesc & pd, FusedArgs fa, unsigned grid, hipStream_t s) {
    FusedKernelInfo & ki = kernel_info<BLOCK, STEPS, LOG_E, LOG_T, RW>();
    LibState & ls = lib();
    if (ki.occ < 0) {
        int occ = 0;
        SDH_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, decode_fused_kernel<BLOCK, STEPS, LOG_E, LOG_T, RW>, RW * 64, 0));
        ki.occ = occ;
    }
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
    if (ki.census_failed || (long long) grid > (long long) ki.occ * ls.num_cu) {   // not all resident: caller falls back
        if (dbg) fprintf(stderr, "[sdh] fused decode declined: grid=%u occ=%d cus=%d census_failed=%d\n", grid, ki.occ, ls.num_cu, (int) ki.census_failed);
        return 1;
    }
    if (grid > ki.census_ok_grid && fa.epoch == CAPTURE_TAG) {   // capturing: the census cannot run now
        if (dbg) fprintf(stderr, "[sdh] fused decode declined: capture before the first eager launch (grid=%u)\n", grid);
        return 1;
    }
    if (grid > ki.census_ok_grid) {
        // one-time residency census for this kernel at this grid size
        if (!ls.census_ctr) {
            SDH_HIP_CHECK(hipMalloc((void **) &ls.census_ctr, 16));
        }
        SDH_HIP_CHECK(hipMemsetAsync(ls.census_ctr, 0, 16, s));
        FusedArgs ca = fa;
        ca.census = 1;
        ca.census_ctr = ls.census_ctr;
        hipLaunchKernelGGL((decode_fused_kernel<BLOCK, STEPS, LOG_E, LOG_T, RW>), dim3(grid), dim3(RW * 64), 0, s, p, pd, ca);
        SDH_HIP_CHECK(hipGetLastError());
        unsigned h[2] = {0, 0};
        SDH_HIP_CHECK(hipMemcpyAsync(h, ls.census_ctr, sizeof(h), hipMemcpyDeviceToHost, s));
        SDH_HIP_CHECK(hipStreamSynchronize(s));
        if (dbg) fprintf(stderr, "[sdh] fused decode census: grid=%u arrived=%u failures=%u\n", grid, h[0], h[1]);
        if (h[0] != grid || h[1] != 0) { ki.census_failed = true; return 1; }
        ki.census_ok_grid = grid;
    }
    fa.census = 0;
    fa.census_ctr = nullptr;
    hipLaunchKernelGGL((decode_fused_kernel<BLOCK, STEPS, LOG_E, LOG_T, RW>), dim3(grid), dim3(RW * 64), 0, s, p, pd, fa);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <bool BLOCK, int LOG_E, int LOG_T, int RW = (1 << LOG_T) / 64>
static int launch_steps(int steps, const DeltaParams & p, const BasePackDesc & pd, const FusedArgs & fa, unsigned grid, hipStream_t s) {
    switch (steps) {
        case 1: return launch_inst<BLOCK, 1, LOG_E, LOG_T, RW>(p, pd, fa, grid, s);
        case 2: return launch_inst<BLOCK, 2, LOG_E, LOG_T, RW>(p, pd, fa, grid, s);
        case 3: return launch_inst<BLOCK, 3, LOG_E, LOG_T, RW>(p, pd, fa, grid, s);
        case 4: return launch_inst<BLOCK, 4, LOG_E, LOG_T, RW>(p, pd, fa, grid, s);
        default: return 1;
    }
}

// ---- 2..4 tokens ------------------------------------------------------------------------------------
template <bool BLOCK, int STEPS, int LOG_E, int LOG_T, int NT>
static int launch_nt_inst(const DeltaParams & p, const BasePackDesc & pd, FusedArgs fa, int n_tok, unsigned grid, hipStream_t s) {
    static FusedKernelInfo ki;
    LibState & ls = lib();
    if (ki.occ < 0) {
        int occ = 0;
        SDH_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, decode_fused_nt_kernel<BLOCK, STEPS, LOG_E, LOG_T, NT>, 1 << LOG_T, 0));
        ki.occ = occ;
    }
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
    if (ki.census_failed || (long long) grid > (long long) ki.occ * ls.num_cu) {
        if (dbg) fprintf(stderr, "[sdh] %d-token decode declined: grid=%u occ=%d cus=%d census_failed=%d\n", n_tok, grid, ki.occ, ls.num_cu, (int) ki.census_failed);
        return 1;
    }
    if (grid > ki.census_ok_grid && fa.epoch == CAPTURE_TAG) {
        if (dbg) fprintf(stderr, "[sdh] %d-token decode declined: capture before the first eager launch (grid=%u)\n", n_tok, grid);
        return 1;
    }
    if (grid > ki.census_ok_grid) {
        if (!ls.census_ctr) SDH_HIP_CHECK(hipMalloc((void **) &ls.census_ctr, 16));
        SDH_HIP_CHECK(hipMemsetAsync(ls.census_ctr, 0, 16, s));
        FusedArgs ca = fa;
        ca.census = 1;
        ca.census_ctr = ls.census_ctr;
        hipLaunchKernelGGL((decode_fused_nt_kernel<BLOCK, STEPS, LOG_E, LOG_T, NT>), dim3(grid), dim3(1 << LOG_T), 0, s, p, pd, ca, n_tok);
        SDH_HIP_CHECK(hipGetLastError());
        unsigned h[2] = {0, 0};
        SDH_HIP_CHECK(hipMemcpyAsync(h, ls.census_ctr, sizeof(h), hipMemcpyDeviceToHost, s));
        SDH_HIP_CHECK(hipStreamSynchronize(s));
        if (dbg) fprintf(stderr, "[sdh] %d-token decode census: grid=%u arrived=%u failures=%u\n", n_tok, grid, h[0], h[1]);
        if (h[0] != grid || h[1] != 0) { ki.census_failed = true; return 1; }
        ki.census_ok_grid = grid;
    }
    fa.census = 0;
    fa.census_ctr = nullptr;
    hipLaunchKernelGGL((decode_fused_nt_kernel<BLOCK, STEPS, LOG_E, LOG_T, NT>), dim3(grid), dim3(1 << LOG_T), 0, s, p, pd, fa, n_tok);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <bool BLOCK, int LOG_E, int LOG_T>
static int launch_nt_steps(int steps, int n_tok, const DeltaParams & p, const BasePackDesc & pd, const FusedArgs & fa, unsigned grid, hipStream_t s) {
    if (n_tok <= 2) {
        switch (steps) {
            case 1: return launch_nt_inst<BLOCK, 1, LOG_E, LOG_T, 2>(p, pd, fa, n_tok, grid, s);
            case 2: return launch_nt_inst<BLOCK, 2, LOG_E, LOG_T, 2>(p, pd, fa, n_tok, grid, s);
            case 3: return launch_nt_inst<BLOCK, 3, LOG_E, LOG_T, 2>(p, pd, fa, n_tok, grid, s);
            case 4: return launch_nt_inst<BLOCK, 4, LOG_E, LOG_T, 2>(p, pd, fa, n_tok, grid, s);
            default: return 1;
        }
    }
    // 3-4 tokens: rows of up to 2 steps per lane. With 3 steps the kernel needs > 80 VGPRs or spills, and
    // measured slower than the two-kernel path (cfg3, T=4: 22.5 vs 18.7 us), which those shapes keep. Round 3 tried
    // two passes of two tokens inside the one launch (the second pass re-reading the row from L2, the two-token form's
    // arithmetic): the compiler still spills 31 registers around the 16-granule poll and the launch measured 25.5 us
    // at 3 and 25.9 us at 4 tokens against 15.5 / 17.1 us for the two kernels. Not kept.
    switch (steps) {
        case 1: return launch_nt_inst<BLOCK, 1, LOG_E, LOG_T, 4>(p, pd, fa, n_tok, grid, s);
        case 2: return launch_nt_inst<BLOCK, 2, LOG_E, LOG_T, 4>(p, pd, fa, n_tok, grid, s);
        default: return 1;
    }
}

// COO, activations staged in LDS (decode_fused_nt_xlds_kernel)
template <int STEPS, int LOG_E, int LOG_T>
static int launch_ntx_inst(const DeltaParams & p, const BasePackDesc & pd, FusedArgs fa, int n_tok, unsigned grid, size_t lds, hipStream_t s) {
    static FusedKernelInfo ki;
    static size_t occ_lds = 0;
    LibState & ls = lib();
    static const bool attr_ok = hipFuncSetAttribute((const void *) decode_fused_nt_xlds_kernel<STEPS, LOG_E, LOG_T>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (!attr_ok) return 1;
    if (ki.occ < 0 || lds > occ_lds) {                  // residency depends on the LDS request: keep the largest seen
        int occ = 0;
        SDH_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, decode_fused_nt_xlds_kernel<STEPS, LOG_E, LOG_T>, 1 << LOG_T, lds));
        ki.occ = occ; occ_lds = lds; ki.census_ok_grid = 0;
    }
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
    if (ki.census_failed || (long long) grid > (long long) ki.occ * ls.num_cu) {
        if (dbg) fprintf(stderr, "[sdh] %d-token LDS decode declined: grid=%u occ=%d cus=%d lds=%zu census_failed=%d\n", n_tok, grid, ki.occ, ls.num_cu, lds, (int) ki.census_failed);
        return 1;
    }
    if (grid > ki.census_ok_grid && fa.epoch == CAPTURE_TAG) return 1;
    if (grid > ki.census_ok_grid) {
        if (!ls.census_ctr) SDH_HIP_CHECK(hipMalloc((void **) &ls.census_ctr, 16));
        SDH_HIP_CHECK(hipMemsetAsync(ls.census_ctr, 0, 16, s));
        FusedArgs ca = fa;
        ca.census = 1;
        ca.census_ctr = ls.census_ctr;
        hipLaunchKernelGGL((decode_fused_nt_xlds_kernel<STEPS, LOG_E, LOG_T>), dim3(grid), dim3(1 << LOG_T), occ_lds, s, p, pd, ca, n_tok);
        SDH_HIP_CHECK(hipGetLastError());
        unsigned h[2] = {0, 0};
        SDH_HIP_CHECK(hipMemcpyAsync(h, ls.census_ctr, sizeof(h), hipMemcpyDeviceToHost, s));
        SDH_HIP_CHECK(hipStreamSynchronize(s));
        if (dbg) fprintf(stderr, "[sdh] %d-token LDS decode census: grid=%u arrived=%u failures=%u\n", n_tok, grid, h[0], h[1]);
        if (h[0] != grid || h[1] != 0) { ki.census_failed = true; return 1; }
        ki.census_ok_grid = grid;
    }
    fa.census = 0;
    fa.census_ctr = nullptr;
    hipLaunchKernelGGL((decode_fused_nt_xlds_kernel<STEPS, LOG_E, LOG_T>), dim3(grid), dim3(1 << LOG_T), lds, s, p, pd, fa, n_tok);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <int LOG_E, int LOG_T>
static int launch_ntx_steps(int steps, int n_tok, const DeltaParams & p, const BasePackDesc & pd, const FusedArgs & fa, unsigned grid, size_t lds, hipStream_t s) {
    switch (steps) {
        case 1: return launch_ntx_inst<1, LOG_E, LOG_T>(p, pd, fa, n_tok, grid, lds, s);
        case 2: return launch_ntx_inst<2, LOG_E, LOG_T>(p, pd, fa, n_tok, grid, lds, s);
        default: return 1;
    }
}

// Returns SDH_OK if the fused kernel was launched, 1 if this payload / call does not qualify (the
// caller then runs the two-kernel path), <0 on error.
int launch_decode_fused(DeviceWeight & w, const DeltaLaunch & a, hipStream_t s) {
    const bool is_block = (w.kind == SDH_OP_BASE_BLOCK);
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
#define SDH_DECLINE(why) do { if (dbg) fprintf(stderr, "[sdh] fused decode declined: %s\n", why); return 1; } while (0)
    if (a.T < 1 || a.T > MAX_FUSED_TOKENS || a.exact_order) return 1;
    if (w.kind != SDH_OP_BASE_BLOCK && w.kind != SDH_OP_BASE_COO) return 1;
    if (w.idx_type != SDH_TYPE_I16 || w.val_type != SDH_TYPE_F16) SDH_DECLINE("payload is not I16/F16");
    if (w.pack.wide32 || !w.handoff) SDH_DECLINE("base parameters are 32-bit");
    if (!(w.L == 1024 || w.L == 2048 || w.L == 4096)) SDH_DECLINE("L not in {1024,2048,4096}");
    if (!w.tall && w.B > 8) SDH_DECLINE("more than 8 base blocks per output");
    const int n_tok = (int) a.T;
    if (n_tok > 1 && !w.tall && w.B * (n_tok <= 2 ? 2 : 4) > 16) SDH_DECLINE("too many base granules per row for a multi-token launch");
    if (n_tok > 1 && (a.ldx > INT32_MAX / 4 || !((((uintptr_t) a.x) % 16) == 0 && (a.ldx % 4) == 0))) SDH_DECLINE("token columns are not 16-byte aligned");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) SDH_DECLINE("capture status unknown");
    const bool capturing = (cs != hipStreamCaptureStatusNone);

    DeltaParams p = make_delta_params(w, a);
    int G = 1, steps = 0;
    if (!delta_vector_shape(p, is_block, &G, &steps) || steps < 1 || steps > 4) SDH_DECLINE("row shape needs the scalar or looped kernel");
#undef SDH_DECLINE
    p.G = G;
    const int log_t = decode_log_t((int) w.L);
    int rows_per_block = ((1 << log_t) / 64) * (64 / G);
    int ncu = shadow_ncu();
    // Exclusive base CUs (experiment, SDH_DECODE_EXCLUSIVE=1; default off). The dispatcher deals blockIdx
    // round-robin over the CUs, so ids = b mod CUs of the later wraps can be left empty (they sleep ~1 us,
    // holding their slot until the dispatch is over) and the base workgroups have their CUs to themselves;
    // with 9 waves per workgroup the rows of cfg3 (4096 one-row waves) still fit two workgroups per
    // remaining CU (504 slots), so nothing is unbalanced. In the STAMPED build this takes the base chain
    // from 6.4 to 3.9 us and the launch from 8.0 to 5.7 us in-kernel; in the product build it measures
    // 7.93 against 7.78 us per token: the slow chain beside residual workgroups is largely an artefact of
    // the stamp stores queueing behind the payload loads of the same CU. Kept for the next look at this.
    bool exclusive = false;
#if SDH_L4096_LOG_T == 9
    {
        static const bool excl_off = [] { const char * e = getenv("SDH_DECODE_EXCLUSIVE"); return !(e && atoi(e) == 1); }();
        const int slots = 2 * (lib().num_cu - (int) w.B), rpw = 64 / G;
        const int wg8 = (p.n_out + 8 * rpw - 1) / (8 * rpw), wg9 = (p.n_out + 9 * rpw - 1) / (9 * rpw);
        exclusive = !excl_off && n_tok == 1 && w.L == 4096 && ncu == 0 && (int) w.B < lib().num_cu / 4 && wg8 > slots && wg9 <= slots;
        if (exclusive) { rows_per_block = 9 * rpw; ncu = lib().num_cu; }
    }
#endif
    const unsigned grid = shadow_grid((int) w.B * n_tok, (p.n_out + rows_per_block - 1) / rows_per_block, ncu);

    FusedArgs fa{};
    fa.vg = reinterpret_cast<unsigned long long *>(w.handoff);
    fa.epoch = capturing ? CAPTURE_TAG : next_epoch(w.epoch);
    static const bool force_reset = getenv("SDH_DECODE_FORCE_RESET") != nullptr;     // A/B: cost of the clearing stores
    fa.reset = (capturing || force_reset) ? 1 : 0;
    fa.resid_delay = resid_delay_units();
    fa.ncu = ncu;
    fa.linger = shadow_linger_ticks();
    {
        static const int dx = [] { const char * e = getenv("SDH_DECODE_DELAY_XCD"); return e ? atoi(e) : 1; }();   // 0: every workgroup waits
        fa.delay_xcd_only = dx;
    }
    if (exclusive) {
        static const int excl_linger = [] { const char * e = getenv("SDH_DECODE_EXCL_LINGER"); return e ? atoi(e) : 100; }();
        fa.linger = excl_linger;
#if SDH_L4096_LOG_T == 9
        return is_block ? launch_steps<true, SDH_L4096_LOG_E, SDH_L4096_LOG_T, 9>(steps, p, w.pack, fa, grid, s)
                        : launch_steps<false, SDH_L4096_LOG_E, SDH_L4096_LOG_T, 9>(steps, p, w.pack, fa, grid, s);
#endif
    }
    if (n_tok > 1) {
        // several tokens: the residual role is the longer chain (its activation gathers scale with the tokens:
        // Mistral gate COO, T=4: 14.5 us without a head start, 14.8 / 15.4 / 16.3 with 3 / 6 / 10 units)
        static const bool delay_forced = getenv("SDH_DECODE_DELAY") != nullptr;
        if (!delay_forced) fa.resid_delay = 0;
        // COO with x small enough for LDS: stage it token-interleaved (one LDS read per entry for all tokens).
        // (One token through this form measured slower than the plain one-token kernel: Mistral gate 7.85 vs
        // 6.81 us, Gemma-4B gate 7.23 vs 6.62 us — the staging pass sits in front of the payload stream.)
        static const bool xlds_off = [] { const char * e = getenv("SDH_DECODE_XLDS"); return e && atoi(e) == 0; }();
        static const bool x4_off = [] { const char * e = getenv("SDH_DECODE_X4"); return e && atoi(e) == 0; }();             // dev: A/B
        // Mistral FFN layer, us, with / without: 2 tokens 18.6 / 17.9 (two plain gathers per entry beat one 16-byte gather half of which is padding),
        // 3 tokens 20.7 / 22.6, 4 tokens 22.3 / 24.5
        const bool use_x4 = a.x4 != nullptr && !x4_off && n_tok >= 3;
        fa.x4 = use_x4 ? reinterpret_cast<const float4 *>(a.x4) : nullptr;
        const size_t xlds_bytes = use_x4 ? (size_t) w.L * 4 + 16 : std::max((size_t) w.L * 4, (size_t) p.n_in * 16) + 16;
        if ((!xlds_off || use_x4) && !is_block && steps <= 2 && (p.n_in % 4) == 0 && xlds_bytes <= 72 * 1024 && (w.tall || w.B <= 4)) {
            int rc = 1;
            switch (w.L) {
                case 1024: rc = launch_ntx_steps<2, 8>(steps, n_tok, p, w.pack, fa, grid, xlds_bytes, s); break;
                case 2048: rc = launch_ntx_steps<SDH_L2048_LOG_E, SDH_L2048_LOG_T>(steps, n_tok, p, w.pack, fa, grid, xlds_bytes, s); break;
                default:   rc = launch_ntx_steps<SDH_L4096_LOG_E, SDH_L4096_LOG_T>(steps, n_tok, p, w.pack, fa, grid, xlds_bytes, s); break;
            }
            if (rc != 1) return rc;                    // launched or failed; 1: does not fit, try the plain multi-token launch
        }
        fa.x4 = nullptr;
        switch (w.L) {
            case 1024: return is_block ? launch_nt_steps<true, 2, 8>(steps, n_tok, p, w.pack, fa, grid, s) : launch_nt_steps<false, 2, 8>(steps, n_tok, p, w.pack, fa, grid, s);
            case 2048: return is_block ? launch_nt_steps<true, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(steps, n_tok, p, w.pack, fa, grid, s)
                                       : launch_nt_steps<false, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(steps, n_tok, p, w.pack, fa, grid, s);
            default:   return is_block ? launch_nt_steps<true, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(steps, n_tok, p, w.pack, fa, grid, s)
                                       : launch_nt_steps<false, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(steps, n_tok, p, w.pack, fa, grid, s);
        }
    }
    switch (w.L) {
        case 1024: return is_block ? launch_steps<true, 2, 8>(steps, p, w.pack, fa, grid, s) : launch_steps<false, 2, 8>(steps, p, w.pack, fa, grid, s);
        case 2048: return is_block ? launch_steps<true, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(steps, p, w.pack, fa, grid, s)
                                   : launch_steps<false, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(steps, p, w.pack, fa, grid, s);
        default:   return is_block ? launch_steps<true, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(steps, p, w.pack, fa, grid, s)
                                   : launch_steps<false, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(steps, p, w.pack, fa, grid, s);
    }
}


// ---- gate | up pair ---------------------------------------------------------------------------------
template <bool BLOCK, int STEPS, int LOG_E, int LOG_T>
static int launch_pair_inst(PairArgs & a, unsigned grid, hipStream_t s) {
    static FusedKernelInfo ki;
    LibState & ls = lib();
    if (ki.occ < 0) {
        int occ = 0;
        SDH_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, decode_pair_kernel<BLOCK, STEPS, LOG_E, LOG_T>, 1 << LOG_T, 0));
        ki.occ = occ;
    }
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
    if (ki.census_failed || (long long) grid > (long long) ki.occ * ls.num_cu) {
        if (dbg) fprintf(stderr, "[sdh] gate|up launch declined: grid=%u occ=%d cus=%d census_failed=%d\n", grid, ki.occ, ls.num_cu, (int) ki.census_failed);
        return 1;
    }
    if (grid > ki.census_ok_grid && a.epoch[0] == CAPTURE_TAG) {  // capturing: the census cannot run now
        if (dbg) fprintf(stderr, "[sdh] gate|up launch declined: capture before the first eager launch (grid=%u)\n", grid);
        return 1;
    }
    if (grid > ki.census_ok_grid) {
        if (!ls.census_ctr) SDH_HIP_CHECK(hipMalloc((void **) &ls.census_ctr, 16));
        SDH_HIP_CHECK(hipMemsetAsync(ls.census_ctr, 0, 16, s));
        PairArgs ca = a;
        ca.census = 1;
        ca.census_ctr = ls.census_ctr;
        hipLaunchKernelGGL((decode_pair_kernel<BLOCK, STEPS, LOG_E, LOG_T>), dim3(grid), dim3(1 << LOG_T), 0, s, ca);
        SDH_HIP_CHECK(hipGetLastError());
        unsigned h[2] = {0, 0};
        SDH_HIP_CHECK(hipMemcpyAsync(h, ls.census_ctr, sizeof(h), hipMemcpyDeviceToHost, s));
        SDH_HIP_CHECK(hipStreamSynchronize(s));
        if (dbg) fprintf(stderr, "[sdh] gate|up census: grid=%u arrived=%u failures=%u\n", grid, h[0], h[1]);
        if (h[0] != grid || h[1] != 0) { ki.census_failed = true; return 1; }
        ki.census_ok_grid = grid;
    }
    a.census = 0;
    a.census_ctr = nullptr;
    hipLaunchKernelGGL((decode_pair_kernel<BLOCK, STEPS, LOG_E, LOG_T>), dim3(grid), dim3(1 << LOG_T), 0, s, a);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <bool BLOCK, int LOG_E, int LOG_T>
static int launch_pair_steps(int steps, PairArgs & a, unsigned grid, hipStream_t s) {
    switch (steps) {                                   // two rows per wave: rows of up to 2 steps per lane
        case 1: return launch_pair_inst<BLOCK, 1, LOG_E, LOG_T>(a, grid, s);
        case 2: return launch_pair_inst<BLOCK, 2, LOG_E, LOG_T>(a, grid, s);
        default: return 1;
    }
}

template <bool BLOCK, int STEPS, int LOG_E, int LOG_T, int NT, bool XLDS>
static int launch_pair_nt_inst(PairArgs & a, int n_tok, int64_t ldh, unsigned grid, size_t lds, hipStream_t s) {
    static FusedKernelInfo ki;
    static size_t occ_lds = 0;
    LibState & ls = lib();
    const auto kern = decode_pair_nt_kernel<BLOCK, STEPS, LOG_E, LOG_T, NT, XLDS>;
    static const bool attr_ok = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (!attr_ok) return 1;
    if (ki.occ < 0 || lds > occ_lds) {                  // residency depends on the LDS request: keep the largest seen
        int occ = 0;
        SDH_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 1 << LOG_T, lds));
        ki.occ = occ; occ_lds = lds; ki.census_ok_grid = 0;
    }
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
    if (ki.census_failed || (long long) grid > (long long) ki.occ * ls.num_cu) {
        if (dbg) fprintf(stderr, "[sdh] %d-token gate|up launch declined: grid=%u occ=%d cus=%d lds=%zu census_failed=%d\n", n_tok, grid, ki.occ, ls.num_cu, lds, (int) ki.census_failed);
        return 1;
    }
    if (grid > ki.census_ok_grid && a.epoch[0] == CAPTURE_TAG) return 1;       // capture before the first eager launch
    if (grid > ki.census_ok_grid) {
        if (!ls.census_ctr) SDH_HIP_CHECK(hipMalloc((void **) &ls.census_ctr, 16));
        SDH_HIP_CHECK(hipMemsetAsync(ls.census_ctr, 0, 16, s));
        PairArgs ca = a;
        ca.census = 1;
        ca.census_ctr = ls.census_ctr;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1 << LOG_T), occ_lds, s, ca, n_tok, ldh);
        SDH_HIP_CHECK(hipGetLastError());
        unsigned h[2] = {0, 0};
        SDH_HIP_CHECK(hipMemcpyAsync(h, ls.census_ctr, sizeof(h), hipMemcpyDeviceToHost, s));
        SDH_HIP_CHECK(hipStreamSynchronize(s));
        if (dbg) fprintf(stderr, "[sdh] %d-token gate|up census: grid=%u arrived=%u failures=%u\n", n_tok, grid, h[0], h[1]);
        if (h[0] != grid || h[1] != 0) { ki.census_failed = true; return 1; }
        ki.census_ok_grid = grid;
    }
    a.census = 0;
    a.census_ctr = nullptr;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1 << LOG_T), lds, s, a, n_tok, ldh);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <bool BLOCK, int LOG_E, int LOG_T>
static int launch_pair_nt_steps(int steps, PairArgs & a, int n_tok, int64_t ldh, unsigned grid, size_t xlds_bytes, hipStream_t s) {
    constexpr size_t base_lds = ((size_t) 1 << (LOG_E + LOG_T)) * 4 + 16;
    if constexpr (!BLOCK) {
        if (xlds_bytes) {
            int rc = 1;                                // rows of two steps with four token slots spill (128 VGPRs)
            if (steps == 1) rc = launch_pair_nt_inst<false, 1, LOG_E, LOG_T, 4, true>(a, n_tok, ldh, grid, std::max(base_lds, xlds_bytes), s);
            if (rc != 1) return rc;
        }
    }
    if (n_tok <= 2) {
        switch (steps) {
            case 1: return launch_pair_nt_inst<BLOCK, 1, LOG_E, LOG_T, 2, false>(a, n_tok, ldh, grid, base_lds, s);
            case 2: return launch_pair_nt_inst<BLOCK, 2, LOG_E, LOG_T, 2, false>(a, n_tok, ldh, grid, base_lds, s);
            default: return 1;
        }
    }
    if (steps == 1) return launch_pair_nt_inst<BLOCK, 1, LOG_E, LOG_T, 4, false>(a, n_tok, ldh, grid, base_lds, s);
    return 1;
}

int launch_decode_pair(DeviceWeight & g, DeviceWeight & u, int act, const uint16_t * gelu_tab,
                       const float * x, int64_t ldx, int n_tok, float * h, int64_t ldh, float * h4,
                       int * nf_g, int * nf_u, bool zero_nonfinite, hipStream_t s) {
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
#define SDH_DECLINE(why) do { if (dbg) fprintf(stderr, "[sdh] gate|up launch declined: %s\n", why); return 1; } while (0)
    if (g.kind != u.kind || (g.kind != SDH_OP_BASE_BLOCK && g.kind != SDH_OP_BASE_COO)) SDH_DECLINE("not two base payloads of one kind");
    if (&g == &u) SDH_DECLINE("gate and up are one handle (they would share hand-off granules)");
    if (!g.tall || !u.tall) SDH_DECLINE("not tall");
    if (g.n_in != u.n_in || g.n_out != u.n_out || g.L != u.L || g.B != u.B) SDH_DECLINE("shapes differ");
    if (g.idx_type != SDH_TYPE_I16 || g.val_type != SDH_TYPE_F16 || u.idx_type != SDH_TYPE_I16 || u.val_type != SDH_TYPE_F16) SDH_DECLINE("payload is not I16/F16");
    if (g.pack.wide32 || u.pack.wide32 || !g.handoff || !u.handoff) SDH_DECLINE("base parameters are 32-bit");
    if (!(g.L == 1024 || g.L == 2048 || g.L == 4096)) SDH_DECLINE("L not in {1024,2048,4096}");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) SDH_DECLINE("capture status unknown");
    const bool capturing = (cs != hipStreamCaptureStatusNone);

    const bool is_block = (g.kind == SDH_OP_BASE_BLOCK);
    DeltaLaunch dl{};
    if (n_tok < 1 || n_tok > MAX_FUSED_TOKENS) SDH_DECLINE("token count");
    if (n_tok > 1 && (ldx > INT32_MAX / 4 || !((((uintptr_t) x) % 16) == 0 && (ldx % 4) == 0))) SDH_DECLINE("token columns are not 16-byte aligned");
    dl.x = x; dl.ldx = n_tok > 1 ? ldx : g.n_in; dl.T = n_tok; dl.y = h; dl.ldy = g.n_out; dl.v = nullptr;
    dl.zero_nonfinite = zero_nonfinite; dl.exact_order = false;
    PairArgs a{};
    dl.nf_counter = nf_g; a.p[0] = make_delta_params(g, dl);
    dl.nf_counter = nf_u; a.p[1] = make_delta_params(u, dl);
    int G0 = 1, G1 = 1, st0 = 0, st1 = 0;
    if (!delta_vector_shape(a.p[0], is_block, &G0, &st0) || !delta_vector_shape(a.p[1], is_block, &G1, &st1)) SDH_DECLINE("row shape needs the scalar or looped kernel");
    if (G0 != G1 || st0 != st1 || st0 < 1 || st0 > 2) SDH_DECLINE("row shapes differ or are too long for two rows per wave");
#undef SDH_DECLINE
    a.p[0].G = a.p[1].G = G0;
    a.pd[0] = g.pack; a.pd[1] = u.pack;
    const int rows_per_block = ((1 << decode_log_t((int) g.L)) / 64) * (64 / G0);
    const int ncu = shadow_ncu();
    const unsigned grid = shadow_grid(2 * (int) g.B * n_tok, (a.p[0].n_out + rows_per_block - 1) / rows_per_block, ncu);
    a.vg[0] = reinterpret_cast<unsigned long long *>(g.handoff);
    a.vg[1] = reinterpret_cast<unsigned long long *>(u.handoff);
    a.epoch[0] = capturing ? CAPTURE_TAG : next_epoch(g.epoch);
    a.epoch[1] = capturing ? CAPTURE_TAG : next_epoch(u.epoch);
    static const bool force_reset = getenv("SDH_DECODE_FORCE_RESET") != nullptr;
    a.reset = (capturing || force_reset) ? 1 : 0;
    a.act = act; a.gelu_tab = gelu_tab; a.h = h;
    a.h4 = n_tok > 1 ? reinterpret_cast<float4 *>(h4) : nullptr;
    // the pair's residual role streams two payloads and is the longer chain: no head start by default
    // (FFN block, Mistral layer: 12.10 / 12.22 / 12.24 / 12.39 us per layer with 0 / 1 / 2 / 3 units when
    // both launches shared one setting)
    static const int pair_delay = [] { const char * e = getenv("SDH_DECODE_PAIR_DELAY"); return e ? atoi(e) : 0; }();
    a.resid_delay = pair_delay < 0 ? 0 : (pair_delay > 64 ? 64 : pair_delay);
    a.ncu = ncu;
    a.linger = shadow_linger_ticks();
    if (n_tok > 1) {
        static const bool off = [] { const char * e = getenv("SDH_DECODE_PAIR_NT"); return e && atoi(e) == 0; }();       // dev: A/B
        if (off) return 1;
        static const bool xlds_off = [] { const char * e = getenv("SDH_DECODE_XLDS"); return e && atoi(e) == 0; }();
        size_t xlds_bytes = (size_t) a.p[0].n_in * 16 + 16;
        if (xlds_off || is_block || (a.p[0].n_in % 4) != 0 || xlds_bytes > 72 * 1024) xlds_bytes = 0;
        switch (g.L) {
            case 1024: return is_block ? launch_pair_nt_steps<true, 2, 8>(st0, a, n_tok, ldh, grid, 0, s) : launch_pair_nt_steps<false, 2, 8>(st0, a, n_tok, ldh, grid, xlds_bytes, s);
            case 2048: return is_block ? launch_pair_nt_steps<true, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(st0, a, n_tok, ldh, grid, 0, s)
                                       : launch_pair_nt_steps<false, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(st0, a, n_tok, ldh, grid, xlds_bytes, s);
            default:   return is_block ? launch_pair_nt_steps<true, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(st0, a, n_tok, ldh, grid, 0, s)
                                       : launch_pair_nt_steps<false, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(st0, a, n_tok, ldh, grid, xlds_bytes, s);
        }
    }
    switch (g.L) {
        case 1024: return is_block ? launch_pair_steps<true, 2, 8>(st0, a, grid, s) : launch_pair_steps<false, 2, 8>(st0, a, grid, s);
        case 2048: return is_block ? launch_pair_steps<true, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(st0, a, grid, s)
                                   : launch_pair_steps<false, SDH_L2048_LOG_E, SDH_L2048_LOG_T>(st0, a, grid, s);
        default:   return is_block ? launch_pair_steps<true, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(st0, a, grid, s)
                                   : launch_pair_steps<false, SDH_L4096_LOG_E, SDH_L4096_LOG_T>(st0, a, grid, s);
    }
}

// ---- gate activation over a buffer (many-token path of the FFN block) --------------------------------
__global__ __launch_bounds__(256)
void glu_kernel(int act, const uint16_t * __restrict__ tab, const float * __restrict__ g, const float * __restrict__ u,
                float * __restrict__ y, int64_t n) {
    const int64_t i0 = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 + 4 <= n) {
        const float4 a = *reinterpret_cast<const float4 *>(g + i0), b = *reinterpret_cast<const float4 *>(u + i0);
        *reinterpret_cast<float4 *>(y + i0) = make_float4(glu_value(act, a.x, b.x, tab), glu_value(act, a.y, b.y, tab),
                                                          glu_value(act, a.z, b.z, tab), glu_value(act, a.w, b.w, tab));
    } else {
        for (int64_t i = i0; i < n; ++i) y[i] = glu_value(act, g[i], u[i], tab);
    }
}

int launch_glu(int act, const uint16_t * gelu_tab, const float * g, const float * u, float * y, int64_t n, hipStream_t s) {
    if (n <= 0) return SDH_OK;
    const bool vec = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (!vec) { set_error("sdh_glu: buffers must be 16-byte aligned"); return SDH_ERR_INVALID; }
    const unsigned grid = (unsigned) ((n + 1023) / 1024);
    hipLaunchKernelGGL(glu_kernel, dim3(grid), dim3(256), 0, s, act, gelu_tab, g, u, y, n);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

} // namespace sdh
