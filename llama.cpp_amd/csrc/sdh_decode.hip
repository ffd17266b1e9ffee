// sdh_decode.hip — batch-1 decode of a SeedDelta weight WITH a base, as ONE launch.
//
// At T = 1 the op is latency-bound: the payload (cfg3: 13 MB) streams in ~4 us, the base transform
// (2 FWHTs per block behind a cold parameter fetch) takes ~5 us on its own, and a dependent kernel
// boundary costs another ~1.5-2.5 us. Running the two as separate launches serialises them. Here
// the first B workgroups of the grid compute the base blocks while all the others stream the
// residual rows; a residual wave publishes its row sums and LEAVES, and the base workgroups, which
// are the last to finish anyway, add them to the base term, scale and store the outputs.
//   tall: base workgroup b owns rows [bL, bL + L): y[o] = (v_b[o - bL] + r[o]) * s[o]
//   wide: y[o] = (sum_b v_b[o] + r[o]) * s[o]; workgroup b owns a 1/B slice of the outputs and the
//         base workgroups exchange the other slices of their v_b.
//
// Hand-off (cdna_hip_programming.md Guideline 16, form R2 "the data IS the flag"): every value
// that crosses workgroups is a naturally aligned 8-byte {value, epoch} granule written by ONE
// agent-scope (sc1, write-through) store; the consumer reads it with an agent-scope 8-byte load
// and accepts it only when the tag equals this launch's epoch. A granule is written by one store
// and read by one load, so no flag, fence or ordering between different addresses is involved, and
// every polling lane has addresses of its own (a single shared flag word measured 3 us slower:
// every poll of the launch lands on one line). The epoch is a host counter bumped per launch, so
// granules of earlier launches are never accepted (which is also why this path is not used under
// stream capture: a replayed launch would replay its epoch).
//
// Progress: only the B base workgroups ever wait. The residual workgroups never do, so they drain
// whatever the dispatch order or the occupancy, and the base workgroups only need each other (B <= 8
// workgroups) to become resident at some point. No residency assumption on the grid is made (the
// first version of this kernel had the ~1000 residual workgroups poll for the base values instead:
// it needed the whole grid co-resident, a census launch to prove it, and was 10-25 % slower).
// All spins are bounded by the real-time counter; a workgroup that gives up writes NaN, which the
// non-finite counter reports.
#include "sdh_internal.h"
#include "sdh_base_dev.h"
#include "sdh_delta_dev.h"

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
    unsigned long long * vg;      // [B][L] base granules {value, epoch}: the exchange between base workgroups (wide)
    unsigned long long * ag;      // [n_out] residual granules {row sum, epoch}
    unsigned   epoch;
    int        chunk;             // wide, B >= 2: outputs finished by each base workgroup
    int        ko;                // ceil(chunk / 256): outputs per thread of a base workgroup
};

// Element i of a buffer as (uniform base + 32-bit byte offset): the addressing form that keeps the
// base in scalar registers, so a lane holds one dword per pending access instead of a 64-bit
// address (the buffers here are far below 4 GiB).
template <typename T>
__device__ __forceinline__ T * at32(T * base, int i) {
    using Byte = std::conditional_t<std::is_const_v<T>, const char, char>;
    return reinterpret_cast<T *>(reinterpret_cast<Byte *>(base) + (size_t) ((unsigned) i * (unsigned) sizeof(T)));
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long * base, int i) {
    return __hip_atomic_load(at32(base, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long * base, int i, unsigned epoch, float v) {
    __hip_atomic_store(at32(base, i), ((unsigned long long) epoch << 32) | (unsigned long long) __float_as_uint(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// slots of the wide gather: per finished output B base granules and the residual granule; the host
// checks (B + 1) * ko <= slots (true for every B <= 8 when n_out <= L)
template <int LOG_E> struct FusedSlots { static constexpr int N = (1 << LOG_E) + 8; };

template <bool BLOCK, int STEPS, int LOG_E>
__global__ __launch_bounds__(256, 4)
void decode_fused_kernel(const DeltaParams p, const BasePackDesc pd, const FusedArgs fa) {
#pragma clang fp contract(off)
    using Ly = BaseLay<LOG_E>;
    constexpr int L = Ly::L;
    constexpr int E = Ly::E;
    constexpr int JL = Ly::J - 1;
    __shared__ __align__(16) float lds[L];             // one transpose buffer

    if ((int) blockIdx.x >= p.B) {
        // ---- residual role: stream the rows, publish each row sum, leave ------------------------
#ifdef SDH_RESID_DELAY
        __builtin_amdgcn_s_sleep(SDH_RESID_DELAY);     // experiment: let the base blocks' loads reach the memory system first
#endif
        const int lane = threadIdx.x & 63;
        const int wave = threadIdx.x >> 6;
        const int G = p.G;
        const int l = lane & (G - 1);
        const int o = (((int) blockIdx.x - p.B) * 4 + wave) * (64 / G) + (lane / G);
        const bool row_ok = o < p.n_out;
        SDH_STAMP(10);
        float acc = delta_row_partial<uint16_t, uint16_t, BLOCK, STEPS>(p, p.x, row_ok ? o : 0, row_ok, l, G);
        acc = group_reduce(acc, G);
        SDH_STAMP(11);
        if (row_ok && l == 0) granule_store(fa.ag, o, fa.epoch, acc);
        return;
    }

    // ---- base role: one block of the base transform, then the outputs it owns ---------------------
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(SDH_BASE_PRIO);         // the base blocks are the critical path of the launch
    const float qnan = __uint_as_float(0x7FC00000u);
    float y[E];

    if (p.tall || p.B == 1) {
        // Every output of rows [o0, o0 + L) needs only this block's base value: finish them straight
        // from the last layout (consecutive lanes, consecutive rows).
        const int o0 = p.tall ? b * L : 0;
        float sc[E];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int o = o0 + Ly::idx(JL, tid, r);
            sc[r] = (p.rs && o < p.n_out) ? *at32(p.rs, o) : 1.0f;
        }
        base_transform<LOG_E, false, true>(p.x, p.n_in, p.tall ? 0 : b * L, p.x_vec_ok != 0, pd, b, lds, tid, y);
        SDH_STAMP(6);
        // One pass = every granule load of the thread issued back to back from clamped (always valid)
        // addresses, then the tags: a branch around a load would make hipcc wait for each load in
        // turn (measured: 20 dependent round trips, 7.4 us, instead of one).
        float a[E];
        bool ok;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        do {
            unsigned long long g[E];
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const int o = o0 + Ly::idx(JL, tid, r);
                g[r] = granule_load(fa.ag, o < p.n_out ? o : o0);
            }
            ok = true;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const int o = o0 + Ly::idx(JL, tid, r);
                ok = ok & (((unsigned) (g[r] >> 32) == fa.epoch) | (o >= p.n_out));
                a[r] = __uint_as_float((unsigned) g[r]);
            }
            if (ok) break;
            __builtin_amdgcn_s_sleep(SDH_POLL_SLEEP);
        } while (__builtin_amdgcn_s_memrealtime() - t0 < SPIN_TICKS);
        SDH_STAMP(12);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int o = o0 + Ly::idx(JL, tid, r);
            if (o < p.n_out) {
                const float yh = p.tall ? y[r] : 0.0f + y[r];       // wide: the sum over blocks starts from 0.0f
                float out = yh + a[r];
                if (!ok) out = qnan;                                 // a row sum never arrived: surface it as a NaN
                store_output(p, o, 0, out, sc[r]);
            }
        }
        return;
    }

    // wide, B >= 2: output o is sum_b v_b[o]. Workgroup b finishes outputs [own0, own1). Every base
    // workgroup publishes all of its values as granules and gathers, per output it owns, the B base
    // values (its own included: the round trip is hidden behind the wait for the others and keeps
    // the gather uniform) and the residual sum.
    constexpr int SL = FusedSlots<LOG_E>::N;
    constexpr int KMAX = E / 2;                        // B >= 2: at most L / 2 outputs per workgroup
    const int own0 = b * fa.chunk;
    const int own1 = min(own0 + fa.chunk, p.n_out);
    float rsv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int o = own0 + tid + 256 * k;
        rsv[k] = (p.rs && k < fa.ko && o < own1) ? *at32(p.rs, o) : 1.0f;
    }
    base_transform<LOG_E, false, true>(p.x, p.n_in, b * L, p.x_vec_ok != 0, pd, b, lds, tid, y);
#ifdef SDH_DIAG_TWICE
    // diagnostic: the same transform again, now with warm parameters and a warm instruction cache
    if (threadIdx.x == 0) {
        unsigned long long * d = sdh_diag_buf + (size_t) blockIdx.x * 32;
        for (int k = 1; k <= 5; ++k) d[16 + k] = d[k];
    }
    __syncthreads();
    SDH_STAMP(7);
    base_transform<LOG_E, false, true>(p.x, p.n_in, b * L, p.x_vec_ok != 0, pd, b, lds, tid, y);
    SDH_STAMP(8);
#endif
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int pos = Ly::idx(JL, tid, r);
        if (pos < p.n_out) granule_store(fa.vg, b * L + pos, fa.epoch, y[r]);
    }
    SDH_STAMP(6);
    __syncthreads();                                   // the transform's last LDS readers
#pragma unroll
    for (int k = 0; k < KMAX; ++k) lds[tid + 256 * k] = rsv[k];   // private to this thread from here on

    float val[SL];
    bool ok;
    [[maybe_unused]] unsigned long long polls = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    do {
        ++polls;
        unsigned long long g[SL];
        unsigned dead = 0;                             // bit s: slot s is not in use
        int k = 0, bb = 0;                             // slot -> (output k of this thread, source bb); uniform
#pragma unroll
        for (int s = 0; s < SL; ++s) {                 // loads first, branch-free (see the tall path)
            const int o = own0 + tid + 256 * k;
            const bool live = (k < fa.ko) & (o < own1);
            g[s] = granule_load(fa.vg, live ? bb * L + o : b * L);   // ag follows vg: source B is the residual
            dead |= live ? 0u : (1u << s);
            const bool wrap = bb == p.B;
            bb = wrap ? 0 : bb + 1;
            k += wrap ? 1 : 0;
        }
        ok = true;
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            ok = ok & (((unsigned) (g[s] >> 32) == fa.epoch) | ((dead >> s) & 1u));
            val[s] = __uint_as_float((unsigned) g[s]);
        }
        if (ok) break;
        __builtin_amdgcn_s_sleep(SDH_POLL_SLEEP);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < SPIN_TICKS);
    SDH_STAMP(12);
    {
        int k = 0, bb = 0;
        float yh = 0.0f;                               // block order from 0.0f, then the residual (:1140-1150)
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            if (k < fa.ko) {
                if (bb < p.B) yh += val[s];
                else {
                    const int o = own0 + tid + 256 * k;
                    if (o < own1) store_output(p, o, 0, ok ? yh + val[s] : qnan, lds[tid + 256 * k]);
                    yh = 0.0f;
                }
            }
            if (++bb > p.B) { bb = 0; ++k; }
        }
    }
    SDH_DIAG_VAL(14, polls);
    SDH_STAMP(13);
}

// ---- host side ----------------------------------------------------------------------------------
template <bool BLOCK, int STEPS, int LOG_E>
static int launch_inst(const DeltaParams & p, const BasePackDesc & pd, FusedArgs fa, unsigned grid, hipStream_t s) {
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
    if (!p.tall && p.B >= 2) {
        fa.chunk = (p.n_out + p.B - 1) / p.B;
        fa.ko = (fa.chunk + 255) / 256;
        if ((p.B + 1) * fa.ko > FusedSlots<LOG_E>::N || fa.ko > (1 << LOG_E) / 2) {
            if (dbg) fprintf(stderr, "[sdh] fused decode declined: B=%d ko=%d exceeds the gather slots\n", p.B, fa.ko);
            return 1;
        }
    }
    hipLaunchKernelGGL((decode_fused_kernel<BLOCK, STEPS, LOG_E>), dim3(grid), dim3(256), 0, s, p, pd, fa);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

template <bool BLOCK, int LOG_E>
static int launch_steps(int steps, const DeltaParams & p, const BasePackDesc & pd, const FusedArgs & fa, unsigned grid, hipStream_t s) {
    switch (steps) {
        case 1: return launch_inst<BLOCK, 1, LOG_E>(p, pd, fa, grid, s);
        case 2: return launch_inst<BLOCK, 2, LOG_E>(p, pd, fa, grid, s);
        case 3: return launch_inst<BLOCK, 3, LOG_E>(p, pd, fa, grid, s);
        case 4: return launch_inst<BLOCK, 4, LOG_E>(p, pd, fa, grid, s);
        default: return 1;
    }
}

// Returns SDH_OK if the fused kernel was launched, 1 if this payload / call does not qualify (the
// caller then runs the two-kernel path), <0 on error.
int launch_decode_fused(DeviceWeight & w, const DeltaLaunch & a, hipStream_t s) {
    const bool is_block = (w.kind == SDH_OP_BASE_BLOCK);
    static const bool dbg = getenv("SDH_DEBUG") != nullptr;
#define SDH_DECLINE(why) do { if (dbg) fprintf(stderr, "[sdh] fused decode declined: %s\n", why); return 1; } while (0)
    if (a.T != 1 || a.exact_order) return 1;
    if (w.kind != SDH_OP_BASE_BLOCK && w.kind != SDH_OP_BASE_COO) return 1;
    if (w.idx_type != SDH_TYPE_I16 || w.val_type != SDH_TYPE_F16) SDH_DECLINE("payload is not I16/F16");
    if (w.pack.wide32 || !w.handoff) SDH_DECLINE("base parameters are 32-bit");
    if (!(w.L == 1024 || w.L == 2048 || w.L == 4096)) SDH_DECLINE("L not in {1024,2048,4096}");
    if (!w.tall && w.B > 8) SDH_DECLINE("more than 8 base blocks per output");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) SDH_DECLINE("stream is capturing");

    DeltaParams p = make_delta_params(w, a);
    int G = 1, steps = 0;
    if (!delta_vector_shape(p, is_block, &G, &steps) || steps < 1 || steps > 4) SDH_DECLINE("row shape needs the scalar or looped kernel");
#undef SDH_DECLINE
    p.G = G;
    const int rows_per_block = 4 * (64 / G);
    const unsigned grid = (unsigned) w.B + (unsigned) ((p.n_out + rows_per_block - 1) / rows_per_block);

    if (++w.epoch == 0) w.epoch = 1;
    FusedArgs fa{};
    fa.vg = reinterpret_cast<unsigned long long *>(w.handoff);
    fa.ag = fa.vg + (size_t) w.B * (size_t) w.L;
    fa.epoch = w.epoch;
    switch (w.L) {
        case 1024: return is_block ? launch_steps<true, 2>(steps, p, w.pack, fa, grid, s) : launch_steps<false, 2>(steps, p, w.pack, fa, grid, s);
        case 2048: return is_block ? launch_steps<true, 3>(steps, p, w.pack, fa, grid, s) : launch_steps<false, 3>(steps, p, w.pack, fa, grid, s);
        default:   return is_block ? launch_steps<true, 4>(steps, p, w.pack, fa, grid, s) : launch_steps<false, 4>(steps, p, w.pack, fa, grid, s);
    }
}

} // namespace sdh
