// sdh_stack.hip — a stack of gated SeedDelta FFN blocks at T = 1 as ONE persistent launch.
//
//   x_{l+1} = down_l( act(gate_l(x_l)) * up_l(x_l) ),   l = 0 .. n_layers-1          (build_ffn's node sequence,
//                                                        src/llama-graph.cpp:826-932, one token)
//
// Why. One launch per tensor tops out near 0.21 of the HBM roofline: 2.5 us of launch floor plus a
// cold one-shot stream per 7-13 MB. Here the grid is launched once (one 512-thread workgroup per CU) and walks
// the layers; every hand-off between phases is an 8-byte {value, tag} granule written by ONE agent-scope store and
// read by ONE agent-scope load (cdna_hip_programming.md Guideline 16, form R2 — the form the fused decode launch
// already uses), so there is no grid barrier, no fence and no flag. The payload of the NEXT phase is requested as
// soon as a wave has published its results, i.e. the HBM stream runs underneath the dependency chain
//   x_l -> [gate|up base blocks] -> h_l -> [down base blocks] -> x_{l+1}
// and the per-layer time is that chain's latency, not launch + stream + chain.
//
// Roles (blockIdx):
//   0 .. 3B-1      base workgroups: block b of gate, of up, of down. Per layer: request the block's parameters, wait
//                  for the input vector (x_l, or chunk b of h_l) as granules, run D3 H D2 H P1 D1 (sdh_base_dev.h,
//                  bit-identical to the stand-alone transform), publish L granules.
//   next n_dwg     down-row workgroups: stage h_l whole in LDS (coalesced granule sweep), form their rows' sums, add the B
//                  base terms, publish x_{l+1} (or store y), THEN request the next layer's payload.
//   the rest       gate/up-row workgroups: stage x_l in LDS (granule sweep), form the two residual sums per row from
//                  payload already in registers, pick up the two base terms, write h[o] = act(g)*u as a granule,
//                  THEN request the next layer's payload.
// A wave never has the next phase's payload loads outstanding while it polls for something it needs now: loads
// retire in order, so a poll issued behind a fresh HBM request would wait for it.
//
// Arithmetic: per row exactly the operations of decode_pair_kernel / decode_fused_kernel (dot8 / add_if chains,
// xor-shuffle reduction, base term first for tall, blocks summed from 0.0f for wide), so a stack's output is
// bit-identical to the same layers evaluated node by node on the fast path.
//
// Safety: consumers spin, so all workgroups must be resident together: grid = number of CUs, checked by the
// occupancy query and a one-time census launch. Every spin is bounded by the real-time counter; a consumer that
// gives up counts a timeout, takes NaN and PUBLISHES it with the valid tag, so nothing downstream waits again and
// the grid always drains. Tags are unique per (call, layer); tag 0 is never used. Not capturable into a HIP graph
// (a replay would repeat its tags): the host entry point declines under capture.
#include "sdh_internal.h"
#ifdef SDH_STK_STAMPS
static __shared__ unsigned long long stk_stamp_lds[512];
static __shared__ int stk_stamp_l;
#define SDH_CORE_STAMP(k)                                                                                      \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (threadIdx.x == 0 && stk_stamp_l >= 0) stk_stamp_lds[stk_stamp_l * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#endif
#include "sdh_base_dev.h"
#include "sdh_delta_dev.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace sdh {

constexpr int STK_WAVES   = 8;          // 512-thread workgroups
constexpr int STK_MAXP_GU = 3;          // gate/up row groups per wave and layer (register-resident payload)
constexpr int STK_MAXP_D  = 3;          // down row groups per wave and layer
constexpr int STK_MAXX    = 16;         // x granules staged per thread
constexpr int STK_MAXH    = 32;         // h granules staged per thread of a down workgroup (n_ff <= 16384)
constexpr int STK_MAXB    = 4;          // base blocks per output of the wide (down) tensor
constexpr unsigned long long STK_SPIN_TICKS = 50000;     // 100 MHz ticks = 500 us

struct StackShape { int Ktot, nb, block, block_shift, n_blk_in, G, n_in, n_out; };

struct StackLayer {                     // device memory, one per layer; [0] gate, [1] up, [2] down
    const uint16_t * idx[3];
    const uint16_t * val[3];
    const float *    rs[3];
    const uint8_t *  pack[3];
};

struct StackArgs {
    const StackLayer * layers;
    int n_layers;
    const float * x;
    float * y;
    unsigned long long * XG, * HG, * VGU, * VD;     // granules: x [n_embd], h [n_ff], gate|up base [2][B][L], down base [B][L]
    unsigned tag0;                                  // layer l uses tag0 + l
    int n_embd, n_ff, L, B;
    StackShape gu, d;
    BasePackDesc pd[3];                             // parameter slab layouts (the pack pointer comes from the layer)
    int act;
    const uint16_t * gelu_tab;
    int zero_nf;
    int * nf;                                       // [n_layers][3] non-finite counters
    unsigned * ctr;                                 // [0] census arrivals, [1] census failures
    unsigned * tmo;                                 // hand-off waits of THIS call that timed out (cleared with nf before every launch)
    int n_dwg;                                      // workgroups that own down rows (the rest of the row workgroups own gate/up rows)
    int census;
    unsigned long long * stamps;                    // diagnostic build (-DSDH_STK_STAMPS): [7 samples][n_layers][8] real-time ticks
};

#ifdef SDH_STK_STAMPS
// Stamps go to LDS while the kernel runs and to memory when the sampled workgroup leaves: a global store per stamp
// sits in the wave's vmcnt queue and every later wait for a load then also waits for that store (~1 us each).
#define STK_STAMP(sample, l, k)                                                                              \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if ((sample) >= 0 && threadIdx.x == 0 && (l) < 64) stk_stamp_lds[(l) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#define STK_STAMP_FLUSH(sample)                                                                              \
    do {                                                                                                     \
        if ((sample) >= 0 && threadIdx.x == 0)                                                               \
            for (int i_ = 0; i_ < a.n_layers * 8 && i_ < 512; ++i_) a.stamps[(sample) * a.n_layers * 8 + i_] = stk_stamp_lds[i_]; \
    } while (0)
#else
#define STK_STAMP(sample, l, k) do { } while (0)
#define STK_STAMP_FLUSH(sample) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long stk_gload(const unsigned long long * base, int i) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(base) + (size_t) ((unsigned) i * 8u)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stk_gstore(unsigned long long * base, int i, float v, unsigned tag) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(base) + (size_t) ((unsigned) i * 8u)),
                       ((unsigned long long) tag << 32) | (unsigned long long) __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Waiting is gated per WORKGROUP. While a hand-off is not there yet, ONE lane of the workgroup watches ONE of the
// granules it needs (stk_gate, a load every ~0.3 us) and the other waves sit in s_barrier, which costs no memory
// traffic. Agent-scope granule loads are served by the fabric, not by L1/L2 hits: with every lane of ~250 workgroups
// spinning on its own granules the pollers alone moved several TB/s and a layer's hand-offs took 2-5 us each
// (profiles/r02_b_stack_stamps_before.txt). After the gate, every lane loads all its N granules back to back,
// branch-free, and checks the tags; stragglers are re-polled at a slow cadence. Returns false on timeout; unused
// slots give 0.
__device__ __forceinline__ void stk_gate(const unsigned long long * vg, int i, unsigned tag, bool watch) {
    if (threadIdx.x == 0 && watch) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((unsigned) (stk_gload(vg, i) >> 32) != tag) {
            if (__builtin_amdgcn_s_memrealtime() - t0 >= STK_SPIN_TICKS) break;
            __builtin_amdgcn_s_sleep(10);
        }
    }
    __syncthreads();
}

template <int N>
__device__ __forceinline__ bool stk_poll(const unsigned long long * vg, const int (&gi)[N], const bool (&used)[N], unsigned tag,
                                         float (&vals)[N]) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = false;
    do {
        unsigned long long g[N];
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = stk_gload(vg, gi[i]);
        ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            ok = ok & (((unsigned) (g[i] >> 32) == tag) | !used[i]);
            vals[i] = used[i] ? __uint_as_float((unsigned) g[i]) : 0.0f;
        }
        if (ok) break;
        __builtin_amdgcn_s_sleep(6);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < STK_SPIN_TICKS);
    return ok;
}

// The workgroup's wait for a hand-off (every wave of the workgroup calls it): gate, then load. Loading first and
// gating only when something is missing measured 22.6 against 17.8 us per layer: the first attempt of ~250
// workgroups is itself the polling storm the gate avoids.
template <int N>
__device__ __forceinline__ bool stk_wait(const unsigned long long * vg, const int (&gi)[N], const bool (&used)[N], unsigned tag,
                                         float (&vals)[N]) {
    stk_gate(vg, gi[0], tag, used[0]);
    return stk_poll<N>(vg, gi, used, tag, vals);
}

// ---- one row group's payload in registers -------------------------------------------------------------
template <bool BLOCK, int STEPS>
struct RowRegs {
    Vec8<uint16_t> vv[STEPS];
    Vec8<uint16_t> ii[STEPS];           // COO
    int   blk[STEPS];                   // block scheme
    float scale;
};

template <bool BLOCK, int STEPS>
__device__ __forceinline__ void stk_row_load(RowRegs<BLOCK, STEPS> & r, const StackShape & s, const uint16_t * idx, const uint16_t * val,
                                             const float * rs, int orow, bool row_ok, int l) {
    const uint16_t * vrow = val + (size_t) orow * s.Ktot;
    const uint16_t * irow = idx + (size_t) orow * (BLOCK ? s.nb : s.Ktot);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const int e = (u * s.G + l) * 8;
        const int e0 = (row_ok & (e < s.Ktot)) ? e : 0;
        if constexpr (BLOCK) r.blk[u] = (int) (int16_t) __builtin_nontemporal_load(irow + (e0 >> s.block_shift));
        else r.ii[u].template load<true>(irow + e0);
        r.vv[u].template load<true>(vrow + e0);
    }
    r.scale = (row_ok && l == 0 && rs) ? rs[orow] : 1.0f;
}

template <bool BLOCK, int STEPS>
__device__ __forceinline__ void stk_row_zero(RowRegs<BLOCK, STEPS> & r) {
#pragma unroll
    for (int u = 0; u < STEPS; ++u) { r.vv[u].q = make_uint4(0, 0, 0, 0); r.ii[u].q = make_uint4(0, 0, 0, 0); r.blk[u] = 0; }
    r.scale = 1.0f;
}

// the row's partial sum against activations in LDS: the operations of delta_row_partial, in its order
template <bool BLOCK, int STEPS>
__device__ __forceinline__ float stk_row_dot_lds(const RowRegs<BLOCK, STEPS> & r, const StackShape & s, const float * xs, bool row_ok, int l) {
    float acc = 0.0f;
    if constexpr (BLOCK) {
        float4 xa[STEPS], xb[STEPS];
        bool ok[STEPS];
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            const int e = (u * s.G + l) * 8;
            const int e0 = (row_ok & (e < s.Ktot)) ? e : 0;
            const int raw = (r.blk[u] << s.block_shift) + (e0 & (s.block - 1));
            ok[u] = row_ok & (e < s.Ktot) & ((unsigned) r.blk[u] < (unsigned) s.n_blk_in) & (raw < s.n_in);
            const int in0 = ok[u] ? raw : 0;
            xa[u] = *reinterpret_cast<const float4 *>(xs + in0);
            xb[u] = *reinterpret_cast<const float4 *>(xs + in0 + 4);
        }
#pragma unroll
        for (int u = 0; u < STEPS; ++u) acc = add_if(acc, ok[u], dot8<uint16_t>(r.vv[u], xa[u], xb[u]));
    } else {
        float xv[STEPS][8];
        bool okj[STEPS][8];
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            const int e = (u * s.G + l) * 8;
            const bool ok = row_ok & (e < s.Ktot);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = idx_to_i32<uint16_t>(r.ii[u].get(j));
                okj[u][j] = ok & ((unsigned) i < (unsigned) s.n_in);
                xv[u][j] = xs[okj[u][j] ? i : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = add_if(acc, okj[u][j], __fmul_rn(val_to_f32<uint16_t>(r.vv[u].get(j)), xv[u][j]));
        }
    }
    return acc;
}

// scale (only if != 1, :645), count / optionally zero non-finite outputs (:363-392): finish_output's operations
__device__ __forceinline__ float stk_finish(float y, float scale, int * nf, int zero_nf) {
    float out = (scale != 1.0f) ? (y * scale) : y;
    if (!isfinite(out)) {
        atomicAdd(nf, 1);
        if (zero_nf) out = 0.0f;
    }
    return out;
}

template <bool BLOCK, int STEPS_GU, int STEPS_D, int LOG_E, int LOG_T>
__global__ __launch_bounds__(STK_WAVES * 64)
void ffn_stack_kernel(const StackArgs a) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    constexpr int E = Ly::E;
    extern __shared__ __align__(16) float lds[];        // max(2 L, n_embd, n_ff) floats (the base transform alternates between two buffers)
#ifdef SDH_STK_STAMPS
    if (threadIdx.x == 0) { for (int i_ = 0; i_ < 512; ++i_) stk_stamp_lds[i_] = 0; stk_stamp_l = -1; }
#endif

    if (a.census) {                                     // residency self-test (as decode_fused_kernel)
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&a.ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool all = false;
            while (__builtin_amdgcn_s_memrealtime() - t0 < 2 * STK_SPIN_TICKS) {
                if (__hip_atomic_load(&a.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!all) __hip_atomic_fetch_add(&a.ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }

    const int B = a.B;
    const int n_base_wg = 3 * B;
    const int bid = blockIdx.x;
    const float qnan = __uint_as_float(0x7FC00000u);

    if (bid < n_base_wg) {
        // ---- base role ------------------------------------------------------------------------------------
        const int which = bid / B, b = bid - which * B;
        const int tid = threadIdx.x;
        if (tid >= Ly::THREADS) return;                 // whole extra waves leave: barriers count live waves
        __builtin_amdgcn_s_setprio(3);
        const int n_src = which < 2 ? a.n_embd : a.n_ff;
        const int in0 = which < 2 ? 0 : b * L;          // tall: x padded to L; wide: chunk b of h
        const unsigned long long * src = which < 2 ? a.XG : a.HG;
        unsigned long long * out = which < 2 ? a.VGU + (size_t) (which * B + b) * L : a.VD + (size_t) b * L;
        int gi[E];
        bool used[E];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int i = in0 + E * tid + r;
            used[r] = i < n_src;
            gi[r] = used[r] ? i : 0;
        }
        const int smp = b == 0 ? which : -1;           // stamps: block 0 of gate, up, down
        (void) smp;
        for (int l = 0; l < a.n_layers; ++l) {
            __syncthreads();                            // the previous layer's last LDS reads are done
#ifdef SDH_STK_STAMPS
            if (threadIdx.x == 0) stk_stamp_l = (smp >= 0 && l < 64) ? l : -1;
#endif
            STK_STAMP(smp, l, 0);
#ifdef SDH_STK_STAMPS
            if (smp >= 0 && threadIdx.x == 0 && l < 64) stk_stamp_lds[l * 8 + 7] = __builtin_amdgcn_s_memtime();   // shader clock beside the real-time stamp 0
#endif
            const unsigned tag = a.tag0 + (unsigned) l;
            BasePackDesc pd = a.pd[which];
            pd.pack = a.layers[l].pack[which];
            BasePrm<Ly> q;
            base_prm_load<Ly, false>(q, pd, b, tid);    // requested before the input exists
            float y[E];
            if (which < 2 && l == 0) {
#pragma unroll
                for (int r = 0; r < E; ++r) y[r] = used[r] ? a.x[gi[r]] : 0.0f;
            } else {
                if (!stk_poll<E>(src, gi, used, tag, y)) {     // 3B workgroups: they poll directly
                    atomicAdd(a.tmo, 1u);
#pragma unroll
                    for (int r = 0; r < E; ++r) y[r] = qnan;
                }
            }
            STK_STAMP(smp, l, 1);
            base_transform_core<Ly, false>(y, q, pd, lds, tid);
            STK_STAMP(smp, l, 2);
#pragma unroll
            for (int r = 0; r < E; ++r) stk_gstore(out, Ly::idx(Ly::J - 1, tid, r), y[r], tag);
            STK_STAMP(smp, l, 3);
        }
        STK_STAMP_FLUSH(smp);
        return;
    }

    // ---- row roles --------------------------------------------------------------------------------------------
    // A workgroup owns EITHER gate/up rows OR down rows: a CU then never has its own payload prefetch in its memory
    // queue while one of its waves waits for a hand-off (the price of a hand-off sits in the consumer CU's queue:
    // with both kinds of rows on every CU a layer took 17.4 us, the prefetch of one kind delaying the polls of the other).
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float * const xs = lds;
    const bool is_down_wg = bid < n_base_wg + a.n_dwg;

    if (!is_down_wg) {
        // ---- gate | up rows ---------------------------------------------------------------------------------
        const StackShape & s = a.gu;
        const int G = s.G, l_in = lane & (G - 1), rpp = 64 / G;
        const int n_rw = (int) gridDim.x - n_base_wg - a.n_dwg, rw = bid - n_base_wg - a.n_dwg;
        const int slots = n_rw * STK_WAVES, slot = rw * STK_WAVES + wave;
        const int n_groups = (s.n_out + rpp - 1) / rpp;
        const int tg = threadIdx.x, nt = STK_WAVES * 64; // staging threads: the whole workgroup
        int xi[STK_MAXX];
        bool xu[STK_MAXX];
#pragma unroll
        for (int k = 0; k < STK_MAXX; ++k) {
            const int i = tg + k * nt;
            xu[k] = i < a.n_embd;
            xi[k] = xu[k] ? i : 0;
        }
        int orow[STK_MAXP_GU];
        bool rok[STK_MAXP_GU], act_pass[STK_MAXP_GU];
#pragma unroll
        for (int p = 0; p < STK_MAXP_GU; ++p) {
            const int group = slot + p * slots;
            act_pass[p] = group < n_groups;             // wave-uniform
            const int o = group * rpp + lane / G;
            rok[p] = act_pass[p] && o < s.n_out;
            orow[p] = rok[p] ? o : 0;
        }
        RowRegs<BLOCK, STEPS_GU> rg[STK_MAXP_GU], ru[STK_MAXP_GU];
        auto load_layer = [&](int l) {
            const StackLayer & ly = a.layers[l];
#pragma unroll
            for (int p = 0; p < STK_MAXP_GU; ++p) {
                if (act_pass[p]) {
                    stk_row_load<BLOCK, STEPS_GU>(rg[p], s, ly.idx[0], ly.val[0], ly.rs[0], orow[p], rok[p], l_in);
                    stk_row_load<BLOCK, STEPS_GU>(ru[p], s, ly.idx[1], ly.val[1], ly.rs[1], orow[p], rok[p], l_in);
                } else {
                    stk_row_zero(rg[p]);
                    stk_row_zero(ru[p]);
                }
            }
        };
        load_layer(0);
        const int smp = rw == 0 ? 5 : (rw == n_rw - 1 ? 6 : -1);
        (void) smp;
        for (int l = 0; l < a.n_layers; ++l) {
            const unsigned tag = a.tag0 + (unsigned) l;
            STK_STAMP(smp, l, 0);
            // stage x_l (every gate/up wave of the workgroup takes part)
            if (l == 0) {
#pragma unroll
                for (int k = 0; k < STK_MAXX; ++k) if (xu[k]) xs[xi[k]] = a.x[xi[k]];
            } else {
                float xv[STK_MAXX] = {};
                if (!stk_wait<STK_MAXX>(a.XG, xi, xu, tag, xv)) {
                    atomicAdd(a.tmo, 1u);
#pragma unroll
                    for (int k = 0; k < STK_MAXX; ++k) xv[k] = qnan;
                }
#pragma unroll
                for (int k = 0; k < STK_MAXX; ++k) if (xu[k]) xs[xi[k]] = xv[k];
            }
            __syncthreads();
            STK_STAMP(smp, l, 1);
            float accg[STK_MAXP_GU], accu[STK_MAXP_GU];
#pragma unroll
            for (int p = 0; p < STK_MAXP_GU; ++p) {
                accg[p] = group_reduce(stk_row_dot_lds<BLOCK, STEPS_GU>(rg[p], s, xs, rok[p], l_in), G);
                accu[p] = group_reduce(stk_row_dot_lds<BLOCK, STEPS_GU>(ru[p], s, xs, rok[p], l_in), G);
            }
            STK_STAMP(smp, l, 2);
            // base terms of this wave's rows: granule o of the gate block array, granule B*L + o of the up one
            int vi[2 * STK_MAXP_GU];
            bool vu[2 * STK_MAXP_GU];
            float vv[2 * STK_MAXP_GU] = {};
#pragma unroll
            for (int p = 0; p < STK_MAXP_GU; ++p) {
                const bool leader = rok[p] && l_in == 0;
                vu[2 * p] = vu[2 * p + 1] = leader;
                vi[2 * p] = leader ? orow[p] : 0;
                vi[2 * p + 1] = leader ? B * L + orow[p] : 0;
            }
            const bool ok = stk_wait<2 * STK_MAXP_GU>(a.VGU, vi, vu, tag, vv);
            STK_STAMP(smp, l, 3);
            if (!ok) atomicAdd(a.tmo, 1u);
#pragma unroll
            for (int p = 0; p < STK_MAXP_GU; ++p) {
                if (rok[p] && l_in == 0) {
                    float yg = vv[2 * p] + accg[p], yu = vv[2 * p + 1] + accu[p];
                    if (!ok) yg = yu = qnan;
                    yg = stk_finish(yg, rg[p].scale, a.nf + 3 * l + 0, a.zero_nf);
                    yu = stk_finish(yu, ru[p].scale, a.nf + 3 * l + 1, a.zero_nf);
                    stk_gstore(a.HG, orow[p], glu_value(a.act, yg, yu, a.gelu_tab), tag);
                }
            }
            STK_STAMP(smp, l, 4);
            if (l + 1 < a.n_layers) load_layer(l + 1);  // the HBM stream of the next layer runs under the down phase
            STK_STAMP(smp, l, 5);
        }
        STK_STAMP_FLUSH(smp);
        return;
    }

    // ---- down rows ----------------------------------------------------------------------------------------------
    {
        const StackShape & s = a.d;
        const int G = s.G, l_in = lane & (G - 1), rpp = 64 / G;
        const int slots = a.n_dwg * STK_WAVES, slot = (bid - n_base_wg) * STK_WAVES + wave;
        const int n_groups = (s.n_out + rpp - 1) / rpp;
        int orow[STK_MAXP_D];
        bool rok[STK_MAXP_D], act_pass[STK_MAXP_D];
#pragma unroll
        for (int p = 0; p < STK_MAXP_D; ++p) {
            const int group = slot + p * slots;
            act_pass[p] = group < n_groups;
            const int o = group * rpp + lane / G;
            rok[p] = act_pass[p] && o < s.n_out;
            orow[p] = rok[p] ? o : 0;
        }
        RowRegs<BLOCK, STEPS_D> rd[STK_MAXP_D];
        auto load_layer = [&](int l) {
            const StackLayer & ly = a.layers[l];
#pragma unroll
            for (int p = 0; p < STK_MAXP_D; ++p) {
                if (act_pass[p]) stk_row_load<BLOCK, STEPS_D>(rd[p], s, ly.idx[2], ly.val[2], ly.rs[2], orow[p], rok[p], l_in);
                else stk_row_zero(rd[p]);
            }
        };
        load_layer(0);
        // h_l is staged whole in LDS by a coalesced granule sweep (n_ff / 512 granules per thread, 512 contiguous bytes
        // per wave instruction). Gathering each row's entries straight from the granule array cost 3.2 us per layer:
        // 24 scattered 8-byte loads per lane are 64 address-coalescer cycles per wave instruction, 5 us per CU
        // (tools/gpu/gpu_stack_ablate.sh, profiles/r02_c_stack_ablation.txt).
        int hi[STK_MAXH];
        bool hu[STK_MAXH];
#pragma unroll
        for (int k = 0; k < STK_MAXH; ++k) {
            const int i = (int) threadIdx.x + k * STK_WAVES * 64;
            hu[k] = i < a.n_ff;
            hi[k] = hu[k] ? i : 0;
        }
        const int smp = bid == n_base_wg ? 3 : (bid == n_base_wg + a.n_dwg - 1 ? 4 : -1);
        (void) smp;
        for (int l = 0; l < a.n_layers; ++l) {
            const unsigned tag = a.tag0 + (unsigned) l;
            STK_STAMP(smp, l, 0);
            {
                float hv[STK_MAXH] = {};
                if (!stk_wait<STK_MAXH>(a.HG, hi, hu, tag, hv)) {
                    atomicAdd(a.tmo, 1u);
#pragma unroll
                    for (int k = 0; k < STK_MAXH; ++k) hv[k] = qnan;
                }
#pragma unroll
                for (int k = 0; k < STK_MAXH; ++k) if (hu[k]) xs[hi[k]] = hv[k];
            }
            __syncthreads();
            STK_STAMP(smp, l, 1);
            float acc[STK_MAXP_D];
#pragma unroll
            for (int p = 0; p < STK_MAXP_D; ++p)
                acc[p] = group_reduce(stk_row_dot_lds<BLOCK, STEPS_D>(rd[p], s, xs, rok[p], l_in), G);
            STK_STAMP(smp, l, 2);
            // base terms: sum_b v_b[o], from 0.0f in block order
            int vi[STK_MAXP_D * STK_MAXB];
            bool vu[STK_MAXP_D * STK_MAXB];
            float vv[STK_MAXP_D * STK_MAXB] = {};
#pragma unroll
            for (int p = 0; p < STK_MAXP_D; ++p) {
                const bool leader = rok[p] && l_in == 0;
#pragma unroll
                for (int bb = 0; bb < STK_MAXB; ++bb) {
                    vu[p * STK_MAXB + bb] = leader && bb < B;
                    vi[p * STK_MAXB + bb] = (leader && bb < B) ? bb * L + orow[p] : 0;
                }
            }
            const bool okv = stk_wait<STK_MAXP_D * STK_MAXB>(a.VD, vi, vu, tag, vv);
            STK_STAMP(smp, l, 3);
            if (!okv) atomicAdd(a.tmo, 1u);
            const bool last = l + 1 == a.n_layers;
#pragma unroll
            for (int p = 0; p < STK_MAXP_D; ++p) {
                if (rok[p] && l_in == 0) {
                    float yh = 0.0f;
#pragma unroll
                    for (int bb = 0; bb < STK_MAXB; ++bb) if (bb < B) yh += vv[p * STK_MAXB + bb];
                    float yo = yh + acc[p];
                    if (!okv) yo = qnan;
                    yo = stk_finish(yo, rd[p].scale, a.nf + 3 * l + 2, a.zero_nf);
                    if (last) a.y[orow[p]] = yo;
                    else stk_gstore(a.XG, orow[p], yo, tag + 1u);
                }
            }
            STK_STAMP(smp, l, 4);
            if (!last) load_layer(l + 1);
            STK_STAMP(smp, l, 5);
        }
        STK_STAMP_FLUSH(smp);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
struct FfnStack {
    int n_layers = 0;
    std::vector<DeviceWeight *> w[3];
    StackLayer * dev_layers = nullptr;
    unsigned long long * XG = nullptr, * HG = nullptr, * VGU = nullptr, * VD = nullptr;
    int * nf = nullptr;
    unsigned * ctr = nullptr;
    unsigned long long * stamps = nullptr;
    unsigned tag = 0;                   // last tag handed out
    int n_embd = 0, n_ff = 0, L = 0, B = 0;
    StackShape gu{}, d{};
    bool block = false;
    int steps_gu = 0, steps_d = 0;
    int n_dwg = 0;
    unsigned grid = 0;
    size_t lds = 0;
    bool census_ok = false;
    hipStream_t last_stream = nullptr;
};

typedef void (*stack_kernel_t)(const StackArgs);

template <bool BLOCK, int LOG_E, int LOG_T>
static stack_kernel_t pick_steps(int sgu, int sd) {
    if (sgu == 1 && sd == 1) return ffn_stack_kernel<BLOCK, 1, 1, LOG_E, LOG_T>;
    if (sgu == 2 && sd == 1) return ffn_stack_kernel<BLOCK, 2, 1, LOG_E, LOG_T>;
    if (sgu == 1 && sd == 2) return ffn_stack_kernel<BLOCK, 1, 2, LOG_E, LOG_T>;
    if (sgu == 2 && sd == 2) return ffn_stack_kernel<BLOCK, 2, 2, LOG_E, LOG_T>;
    return nullptr;
}
static stack_kernel_t pick_kernel(const FfnStack & st) {
    switch (st.L) {
        case 4096: return st.block ? pick_steps<true, 3, 9>(st.steps_gu, st.steps_d) : pick_steps<false, 3, 9>(st.steps_gu, st.steps_d);
        case 2048: return st.block ? pick_steps<true, 3, 8>(st.steps_gu, st.steps_d) : pick_steps<false, 3, 8>(st.steps_gu, st.steps_d);
        case 1024: return st.block ? pick_steps<true, 2, 8>(st.steps_gu, st.steps_d) : pick_steps<false, 2, 8>(st.steps_gu, st.steps_d);
        default: return nullptr;
    }
}

static bool same_layout(const BasePackDesc & a, const BasePackDesc & b) {
    return a.blk_stride == b.blk_stride && a.off_p1 == b.off_p1 && a.off_d2 == b.off_d2 && a.off_d1 == b.off_d1 &&
           a.off_d3 == b.off_d3 && a.off_p2 == b.off_p2 && a.wide32 == b.wide32;
}

static StackShape shape_of(const DeviceWeight & w, int G) {
    StackShape s{};
    const bool is_block = w.kind == SDH_OP_BASE_BLOCK;
    s.nb = (int) w.nb; s.block = (int) w.block;
    s.Ktot = is_block ? (int) (w.nb * w.block) : (int) w.K;
    s.n_in = (int) w.n_in; s.n_out = (int) w.n_out;
    s.n_blk_in = is_block ? (int) ((w.n_in + w.block - 1) / w.block) : 0;
    s.block_shift = -1;
    if (is_block && w.block > 0 && (w.block & (w.block - 1)) == 0) { int sft = 0; while ((1 << sft) < w.block) ++sft; s.block_shift = sft; }
    s.G = G;
    return s;
}

void stack_free(FfnStack * st) {
    if (!st) return;
    (void) hipFree(st->dev_layers); (void) hipFree(st->XG); (void) hipFree(st->HG); (void) hipFree(st->VGU); (void) hipFree(st->VD);
    (void) hipFree(st->nf); (void) hipFree(st->ctr); (void) hipFree(st->stamps);
    delete st;
}

// Returns SDH_OK, SDH_ERR_UNSUPPORTED when the layers do not have the one shape this kernel is built for (the caller
// evaluates them with sdh_ffn_forward layer by layer instead), or an error.
int stack_create(const sdh_weight * const * gate, const sdh_weight * const * up, const sdh_weight * const * down, int n_layers,
                 FfnStack ** out) {
    *out = nullptr;
    if (n_layers < 1 || n_layers > 4096) { set_error("ffn stack: n_layers %d out of range", n_layers); return SDH_ERR_INVALID; }
#define STK_DECLINE(why) do { set_error("ffn stack: %s", why); stack_free(st); return SDH_ERR_UNSUPPORTED; } while (0)
    FfnStack * st = new FfnStack();
    st->n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        if (!gate[l] || !up[l] || !down[l]) { set_error("ffn stack: null weight in layer %d", l); stack_free(st); return SDH_ERR_INVALID; }
        st->w[0].push_back(reinterpret_cast<DeviceWeight *>(const_cast<sdh_weight *>(gate[l])));
        st->w[1].push_back(reinterpret_cast<DeviceWeight *>(const_cast<sdh_weight *>(up[l])));
        st->w[2].push_back(reinterpret_cast<DeviceWeight *>(const_cast<sdh_weight *>(down[l])));
    }
    const DeviceWeight & g0 = *st->w[0][0], & d0 = *st->w[2][0];
    if (g0.kind != SDH_OP_BASE_COO && g0.kind != SDH_OP_BASE_BLOCK) STK_DECLINE("payloads without a base");
    st->block = g0.kind == SDH_OP_BASE_BLOCK;
    st->n_embd = (int) g0.n_in; st->n_ff = (int) g0.n_out; st->L = (int) g0.L; st->B = (int) g0.B;
    if (!g0.tall || d0.tall || d0.n_in != g0.n_out || d0.n_out != g0.n_in) STK_DECLINE("not a tall gate/up with a wide down of the transposed shape");
    if (d0.L != g0.L || d0.B != g0.B) STK_DECLINE("gate and down base dims differ");
    if (st->B > STK_MAXB) STK_DECLINE("more than 4 base blocks");
    for (int t = 0; t < 3; ++t)
        for (int l = 0; l < n_layers; ++l) {
            const DeviceWeight & w = *st->w[t][l], & r = *st->w[t == 1 ? 0 : t][0];
            if (w.kind != g0.kind || w.n_in != r.n_in || w.n_out != r.n_out || w.L != r.L || w.B != r.B || w.K != r.K || w.nb != r.nb ||
                w.block != r.block) STK_DECLINE("layers differ in shape");
            if (w.idx_type != SDH_TYPE_I16 || w.val_type != SDH_TYPE_F16 || w.pack.wide32 || !w.pack.pack) STK_DECLINE("payload is not I16/F16 with 16-bit base parameters");
            if (!same_layout(w.pack, st->w[t][0]->pack)) STK_DECLINE("base parameter layouts differ between layers");
        }
    for (int l = 0; l < n_layers; ++l)
        if (st->w[0][l] == st->w[1][l]) STK_DECLINE("gate and up are one handle");
    // row shapes: the vector path of the decode kernels (16-byte payload loads, 8-element runs)
    DeltaLaunch dl{};
    alignas(16) static float dummy_x[4];
    dl.x = dummy_x; dl.ldx = 4; dl.T = 1;
    int G = 1, steps = 0;
    DeltaParams pg = make_delta_params(g0, dl);
    if (!delta_vector_shape(pg, st->block, &G, &steps) || steps < 1 || steps > 2) STK_DECLINE("gate/up rows need the scalar or looped kernel");
    st->gu = shape_of(g0, G); st->steps_gu = steps;
    DeltaParams pdn = make_delta_params(d0, dl);
    if (!delta_vector_shape(pdn, st->block, &G, &steps) || steps < 1 || steps > 2) STK_DECLINE("down rows need the scalar or looped kernel");
    st->d = shape_of(d0, G); st->steps_d = steps;
    if (st->block && (st->n_embd % 8 != 0 || st->n_ff % 8 != 0)) STK_DECLINE("block scheme needs dims that are multiples of 8");
    if (!pick_kernel(*st)) STK_DECLINE("no kernel for this base block length");

    // grid: one workgroup per CU; split the row workgroups' waves between gate/up and down rows
    LibState & ls = lib();
    st->grid = (unsigned) ls.num_cu;
    const int n_rw = (int) st->grid - 3 * st->B;
    if (n_rw < 8) STK_DECLINE("too few CUs");
    const int groups_gu = (st->n_ff + 64 / st->gu.G - 1) / (64 / st->gu.G), groups_d = (st->n_embd + 64 / st->d.G - 1) / (64 / st->d.G);
    // the fewest down workgroups whose waves hold their row groups in MAXP_D passes; the rest own gate/up rows
    const int n_dwg = (groups_d + STK_WAVES * STK_MAXP_D - 1) / (STK_WAVES * STK_MAXP_D);      // more (64, 86) measured the same
    const int n_guwg = n_rw - n_dwg;
    if (n_guwg < 1 || (groups_gu + n_guwg * STK_WAVES - 1) / (n_guwg * STK_WAVES) > STK_MAXP_GU) STK_DECLINE("rows do not fit the register-resident passes");
    st->n_dwg = n_dwg;
    if ((st->n_embd + STK_WAVES * 64 - 1) / (STK_WAVES * 64) > STK_MAXX) STK_DECLINE("n_embd too large for the staging pass");
    st->lds = (size_t) std::max(std::max(2 * st->L, st->n_embd), st->n_ff) * sizeof(float);
    if (st->n_ff > STK_MAXH * STK_WAVES * 64) STK_DECLINE("n_ff too large for the staging pass");
    if (st->lds > 64 * 1024) STK_DECLINE("activation vector too large for LDS");
#undef STK_DECLINE

    auto dmalloc = [&](void ** p, size_t bytes) -> int {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
        if (e != hipSuccess) { set_error("ffn stack: allocation failed: %s", hipGetErrorString(e)); return SDH_ERR_HIP; }
        return SDH_OK;
    };
    int rc;
    const size_t BL = (size_t) st->B * st->L;
    if ((rc = dmalloc((void **) &st->XG, (size_t) st->n_embd * 8)) || (rc = dmalloc((void **) &st->HG, (size_t) st->n_ff * 8)) ||
        (rc = dmalloc((void **) &st->VGU, 2 * BL * 8)) || (rc = dmalloc((void **) &st->VD, BL * 8)) ||
        (rc = dmalloc((void **) &st->nf, ((size_t) n_layers * 3 + 1) * sizeof(int))) || (rc = dmalloc((void **) &st->ctr, 16)) ||
        (rc = dmalloc((void **) &st->stamps, (size_t) 7 * n_layers * 8 * sizeof(unsigned long long))) ||
        (rc = dmalloc((void **) &st->dev_layers, (size_t) n_layers * sizeof(StackLayer)))) { stack_free(st); return rc; }
    std::vector<StackLayer> hl((size_t) n_layers);
    for (int l = 0; l < n_layers; ++l)
        for (int t = 0; t < 3; ++t) {
            const DeviceWeight & w = *st->w[t][l];
            hl[l].idx[t] = (const uint16_t *) w.idx; hl[l].val[t] = (const uint16_t *) w.val; hl[l].rs[t] = w.rs; hl[l].pack[t] = w.pack.pack;
        }
    hipError_t e = hipMemcpy(st->dev_layers, hl.data(), hl.size() * sizeof(StackLayer), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("ffn stack: upload failed: %s", hipGetErrorString(e)); stack_free(st); return SDH_ERR_HIP; }
    *out = st;
    return SDH_OK;
}

int stack_forward(FfnStack * st, int act, const uint16_t * gelu_tab, const float * x, float * y, bool zero_nf, hipStream_t s) {
    LibState & ls = lib();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        set_error("ffn stack: not capturable into a HIP graph (tags are per call); evaluate the layers with sdh_ffn_forward");
        return SDH_ERR_UNSUPPORTED;
    }
    stack_kernel_t kern = pick_kernel(*st);
    StackArgs a{};
    a.layers = st->dev_layers; a.n_layers = st->n_layers;
    a.x = x; a.y = y;
    a.XG = st->XG; a.HG = st->HG; a.VGU = st->VGU; a.VD = st->VD;
    a.n_embd = st->n_embd; a.n_ff = st->n_ff; a.L = st->L; a.B = st->B;
    a.gu = st->gu; a.d = st->d;
    for (int t = 0; t < 3; ++t) a.pd[t] = st->w[t][0]->pack;
    a.act = act; a.gelu_tab = gelu_tab; a.zero_nf = zero_nf ? 1 : 0;
    a.nf = st->nf; a.ctr = st->ctr; a.n_dwg = st->n_dwg; a.stamps = st->stamps;
    a.tmo = (unsigned *) (st->nf + (size_t) st->n_layers * 3);      // behind the per-node counters: one memset clears both
    if (!st->census_ok) {
        int occ = 0;
        SDH_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, STK_WAVES * 64, st->lds));
        if (occ < 1 || (long long) st->grid > (long long) occ * ls.num_cu) { set_error("ffn stack: grid does not fit the device"); return SDH_ERR_UNSUPPORTED; }
        SDH_HIP_CHECK(hipMemsetAsync(st->ctr, 0, 16, s));
        StackArgs ca = a;
        ca.census = 1;
        hipLaunchKernelGGL(kern, dim3(st->grid), dim3(STK_WAVES * 64), st->lds, s, ca);
        SDH_HIP_CHECK(hipGetLastError());
        unsigned h[2] = {0, 0};
        SDH_HIP_CHECK(hipMemcpyAsync(h, st->ctr, sizeof(h), hipMemcpyDeviceToHost, s));
        SDH_HIP_CHECK(hipStreamSynchronize(s));
        if (h[0] != st->grid || h[1] != 0) { set_error("ffn stack: residency census failed (%u of %u workgroups)", h[0], st->grid); return SDH_ERR_UNSUPPORTED; }
        st->census_ok = true;
    }
    // tags tag0 .. tag0 + n_layers - 1, never 0
    if (st->tag > 0xFFFFFFFFu - 2u * (unsigned) st->n_layers - 2u) st->tag = 0;
    a.tag0 = st->tag + 1;
    st->tag += (unsigned) st->n_layers;
    SDH_HIP_CHECK(hipMemsetAsync(st->nf, 0, ((size_t) st->n_layers * 3 + 1) * sizeof(int), s));       // per-node counters and this call's timeouts
    hipLaunchKernelGGL(kern, dim3(st->grid), dim3(STK_WAVES * 64), st->lds, s, a);
    SDH_HIP_CHECK(hipGetLastError());
    st->last_stream = s;
    return SDH_OK;
}

int stack_status(FfnStack * st, int64_t * nonfinite, int64_t * timeouts) {
    SDH_HIP_CHECK(hipStreamSynchronize(st->last_stream));
    std::vector<int> nf((size_t) st->n_layers * 3 + 1);
    SDH_HIP_CHECK(hipMemcpy(nf.data(), st->nf, nf.size() * sizeof(int), hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (size_t i = 0; i + 1 < nf.size(); ++i) n += nf[i];
    if (nonfinite) *nonfinite = n;
    if (timeouts) *timeouts = (unsigned) nf.back();           // of the last call only
    return SDH_OK;
}

int stack_stamps(FfnStack * st, unsigned long long * out) {      // diagnostic: 7 x n_layers x 8 ticks of the last call
    SDH_HIP_CHECK(hipStreamSynchronize(st->last_stream));
    SDH_HIP_CHECK(hipMemcpy(out, st->stamps, (size_t) 7 * st->n_layers * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return SDH_OK;
}

int stack_layers(const FfnStack * st) { return st->n_layers; }
int64_t stack_bytes(const FfnStack * st) {
    int64_t b = 0;
    for (int t = 0; t < 3; ++t) for (const DeviceWeight * w : st->w[t]) b += w->payload_bytes;
    return b;
}

} // namespace sdh
