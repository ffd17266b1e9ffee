// sdh_base.hip — the SeedDelta base term on gfx950: stand-alone kernels (prefill / generic path).
//
//   v_b = D3 * H * [D2 * [P2] * H] * P1 * D1 * x_b          (one length-L block b, one token)
//
// Follows apply_base_block of the reference (src/llama-seeddelta.cpp:497-574). H is the
// unnormalised fast Walsh-Hadamard transform in the reference's stage order (len = 1,2,4,...,
// butterfly (u+v, u-v), :484-493). A butterfly stage touches every element exactly once, so
// running its L/2 butterflies in parallel yields the same fp32 bits as the sequential loop as
// long as the STAGE ORDER is kept: the base term computed here is bit-identical to the CPU op.
//
//   base_apply_kernel<LOG_E, P32>   L = 256 * 2^LOG_E in {1024, 2048, 4096, 8192}: the register /
//                                   LDS-transpose transform of sdh_base_dev.h, parameters read from
//                                   the packed 16-bit slab (one contiguous region per block).
//   base_apply_lds_kernel           any other power-of-two L that fits LDS; one barrier per stage.
//
// Also: the closed-form dense W0 tile generator (integer XOR indexing into G_b = FWHT(d2_b)).
#include "sdh_internal.h"
#include "sdh_base_dev.h"

// Every multiply and add in this file must round separately (the reference TU is built without FMA
// contraction): a fused x*d1 + ... in the first butterfly stage would change the bits.
#pragma clang fp contract(off)

namespace sdh {

// ---------------------------------------------------------------------------------------------
// generic LDS kernel (small or unusual L)
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS)
void base_apply_lds_kernel(const float * __restrict__ x, int64_t ldx, int n_in, int L, int B, int tall,
                           const float * __restrict__ d1, const float * __restrict__ d2,
                           const float * __restrict__ d3, const int * __restrict__ p1,
                           const int * __restrict__ p2, float * __restrict__ v) {
    extern __shared__ float sm[];
    float * y   = sm;
    float * tmp = sm + L;
    const int b = blockIdx.x;
    const int64_t t = blockIdx.y;
    const int tid = threadIdx.x;
    const float * xt = x + t * ldx;
    const int in0 = tall ? 0 : b * L;            // tall: x padded to L; wide: chunk b (:1091-1096, :1136-1145)
    const int64_t pb = (int64_t) b * L;

    for (int i = tid; i < L; i += THREADS) {     // y = D1 * x
        const int src = in0 + i;
        const float xv = (src < n_in) ? xt[src] : 0.0f;
        y[i] = xv * d1[pb + i];
    }
    __syncthreads();
    if (p1) {                                    // y = P1 * y (gather; out-of-range mapped to 0 at upload)
        for (int i = tid; i < L; i += THREADS) tmp[i] = y[p1[pb + i]];
        __syncthreads();
        float * s = y; y = tmp; tmp = s;
    }
    auto fwht = [&](float * a) {
        for (int len = 1; len < L; len <<= 1) {
            for (int k = tid; k < (L >> 1); k += THREADS) {
                const int i = ((k & ~(len - 1)) << 1) | (k & (len - 1));
                const float u = a[i];
                const float w = a[i + len];
                a[i]       = u + w;
                a[i + len] = u - w;
            }
            __syncthreads();
        }
    };
    fwht(y);
    float * out = v + ((size_t) t * B + b) * (size_t) L;
    if (!d2 && !p2) {                            // depth 1 (:537-544)
        for (int i = tid; i < L; i += THREADS) out[i] = d3 ? y[i] * d3[pb + i] : y[i];
        return;
    }
    if (d2) {
        for (int i = tid; i < L; i += THREADS) y[i] *= d2[pb + i];
        __syncthreads();
    }
    if (p2) {
        for (int i = tid; i < L; i += THREADS) tmp[i] = y[p2[pb + i]];
        __syncthreads();
        float * s = y; y = tmp; tmp = s;
    }
    fwht(y);
    for (int i = tid; i < L; i += THREADS) out[i] = d3 ? y[i] * d3[pb + i] : y[i];
}

// ---------------------------------------------------------------------------------------------
// register / LDS-transpose kernel: one workgroup per (base block, token)
// ---------------------------------------------------------------------------------------------
template <int LOG_E, bool P32, int LOG_T = 8>
__global__ __launch_bounds__(1 << LOG_T)
void base_apply_kernel(const float * __restrict__ x, int64_t ldx, int n_in, int B, int tall, int x_vec_ok,
                       const BasePackDesc pd, float * __restrict__ v) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    __shared__ __align__(16) float lds[2 * L];
    const int b = blockIdx.x;
    const int64_t t = blockIdx.y;
    SDH_STAMP(0);
    float * out = v + ((size_t) t * B + b) * (size_t) L;
    float y[1 << LOG_E];
    base_transform<Ly, P32, false>(x + t * ldx, n_in, tall ? 0 : b * L, x_vec_ok != 0, pd, b, lds, threadIdx.x, y);
#pragma unroll
    for (int r = 0; r < (1 << LOG_E); ++r) out[Ly::idx(Ly::J - 1, threadIdx.x, r)] = y[r];
    SDH_STAMP(9);
}

// L = 4096 runs 512 threads x 8 elements (the decode kernels' shape) instead of 256 x 16: shorter per block and no
// slower in bulk. Mistral FFN layer, us, 256- / 512-thread form: T=8 35.8 / 34.0, T=32 62.4 / 60.5, T=512 534 / 529.
// SDH_BASE_WIDE_WG=0 selects the 256-thread form (A/B).
static bool base_wide_wg(int64_t) {
    static const bool off = [] { const char * e = getenv("SDH_BASE_WIDE_WG"); return e && atoi(e) == 0; }();
    return !off;
}

// Two tensors of one shape over the same activations (gate and up of a gated FFN) in one launch: blocks 0..B-1 are the
// first tensor's, B..2B-1 the second's. The transform is base_apply_kernel's.
template <int LOG_E, bool P32, int LOG_T = 8>
__global__ __launch_bounds__(1 << LOG_T)
void base_apply_pair_kernel(const float * __restrict__ x, int64_t ldx, int n_in, int B, int tall, int x_vec_ok,
                            const BasePackDesc pd0, const BasePackDesc pd1, float * __restrict__ v0, float * __restrict__ v1) {
    using Ly = BaseLay<LOG_E, LOG_T>;
    constexpr int L = Ly::L;
    __shared__ __align__(16) float lds[2 * L];
    const bool second = (int) blockIdx.x >= B;
    const int b = (int) blockIdx.x - (second ? B : 0);
    const int64_t t = blockIdx.y;
    float * out = (second ? v1 : v0) + ((size_t) t * B + b) * (size_t) L;
    float y[1 << LOG_E];
    base_transform<Ly, P32, false>(x + t * ldx, n_in, tall ? 0 : b * L, x_vec_ok != 0, second ? pd1 : pd0, b, lds, threadIdx.x, y);
#pragma unroll
    for (int r = 0; r < (1 << LOG_E); ++r) out[Ly::idx(Ly::J - 1, threadIdx.x, r)] = y[r];
}

template <int LOG_E, int LOG_T = 8>
static void launch_reg_pair(const DeviceWeight & w0, const DeviceWeight & w1, const float * x, int64_t ldx, int64_t T,
                            float * v0, float * v1, hipStream_t s) {
    dim3 grid((unsigned) (2 * w0.B), (unsigned) T);
    const int x_vec_ok = (((uintptr_t) x % 16) == 0 && (ldx % 4) == 0) ? 1 : 0;
    if (w0.pack.wide32) {
        hipLaunchKernelGGL((base_apply_pair_kernel<LOG_E, true, LOG_T>), grid, dim3(1 << LOG_T), 0, s,
                           x, ldx, (int) w0.n_in, (int) w0.B, w0.tall ? 1 : 0, x_vec_ok, w0.pack, w1.pack, v0, v1);
    } else {
        hipLaunchKernelGGL((base_apply_pair_kernel<LOG_E, false, LOG_T>), grid, dim3(1 << LOG_T), 0, s,
                           x, ldx, (int) w0.n_in, (int) w0.B, w0.tall ? 1 : 0, x_vec_ok, w0.pack, w1.pack, v0, v1);
    }
}

template <int LOG_E, int LOG_T = 8>
static void launch_reg(const DeviceWeight & w, const float * x, int64_t ldx, int64_t T, float * v, hipStream_t s) {
    dim3 grid((unsigned) w.B, (unsigned) T);
    const int x_vec_ok = (((uintptr_t) x % 16) == 0 && (ldx % 4) == 0) ? 1 : 0;
    if (w.pack.wide32) {
        hipLaunchKernelGGL((base_apply_kernel<LOG_E, true, LOG_T>), grid, dim3(1 << LOG_T), 0, s,
                           x, ldx, (int) w.n_in, (int) w.B, w.tall ? 1 : 0, x_vec_ok, w.pack, v);
    } else {
        hipLaunchKernelGGL((base_apply_kernel<LOG_E, false, LOG_T>), grid, dim3(1 << LOG_T), 0, s,
                           x, ldx, (int) w.n_in, (int) w.B, w.tall ? 1 : 0, x_vec_ok, w.pack, v);
    }
}

// generic launcher on plain f32 / i32 parameter arrays (any power-of-two L that fits LDS); also used
// by the scan pipeline for the freshly fitted base (sdh_scan.hip)
int launch_base_apply_generic(const float * x, int64_t ldx, int n_in, int L, int B, bool tall,
                              const float * d1, const float * d2, const float * d3, const int * p1, const int * p2,
                              int64_t T, float * v, hipStream_t s) {
    if (T <= 0) return SDH_OK;
    if (T > 65535) { set_error("base: token chunk %lld exceeds grid limit", (long long) T); return SDH_ERR_UNSUPPORTED; }
    const size_t lds = (size_t) 2 * L * sizeof(float);
    if (lds > 160 * 1024) {
        set_error("base block length L=%d needs %zu B of LDS (max 163840)", L, lds);
        return SDH_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned) B, (unsigned) T);
    if (L >= 2048) {
        constexpr int TH = 1024;
        static bool attr_set = false;
        if (!attr_set) {
            SDH_HIP_CHECK(hipFuncSetAttribute((const void *) base_apply_lds_kernel<TH>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(base_apply_lds_kernel<TH>, grid, dim3(TH), lds, s, x, ldx, n_in, L, B, tall ? 1 : 0,
                           d1, d2, d3, p1, p2, v);
    } else {
        constexpr int TH = 256;
        hipLaunchKernelGGL(base_apply_lds_kernel<TH>, grid, dim3(TH), lds, s, x, ldx, n_in, L, B, tall ? 1 : 0,
                           d1, d2, d3, p1, p2, v);
    }
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

int launch_base_apply(const DeviceWeight & w, const float * x, int64_t ldx, int64_t T,
                      float * v, hipStream_t s) {
    const int L = (int) w.L;
    if (T <= 0) return SDH_OK;
    if (T > 65535) { set_error("base: token chunk %lld exceeds grid limit", (long long) T); return SDH_ERR_UNSUPPORTED; }
    switch (L) {
        case 1024: launch_reg<2>(w, x, ldx, T, v, s); break;
        case 2048: launch_reg<3>(w, x, ldx, T, v, s); break;
        case 4096: if (base_wide_wg(T * w.B)) launch_reg<3, 9>(w, x, ldx, T, v, s); else launch_reg<4>(w, x, ldx, T, v, s); break;
        case 8192: launch_reg<5>(w, x, ldx, T, v, s); break;
        default:
            return launch_base_apply_generic(x, ldx, (int) w.n_in, L, (int) w.B, w.tall, w.d1, w.d2, w.d3, w.p1, w.p2, T, v, s);
    }
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

// Returns SDH_OK if launched, 1 if the two tensors do not share a shape the register kernels cover.
int launch_base_apply_pair(const DeviceWeight & w0, const DeviceWeight & w1, const float * x, int64_t ldx, int64_t T,
                           float * v0, float * v1, hipStream_t s) {
    if (T <= 0 || T > 65535) return 1;
    if (w0.L != w1.L || w0.B != w1.B || w0.n_in != w1.n_in || w0.tall != w1.tall || w0.pack.wide32 != w1.pack.wide32) return 1;
    switch ((int) w0.L) {
        case 1024: launch_reg_pair<2>(w0, w1, x, ldx, T, v0, v1, s); break;
        case 2048: launch_reg_pair<3>(w0, w1, x, ldx, T, v0, v1, s); break;
        case 4096: if (base_wide_wg(2 * T * w0.B)) launch_reg_pair<3, 9>(w0, w1, x, ldx, T, v0, v1, s); else launch_reg_pair<4>(w0, w1, x, ldx, T, v0, v1, s); break;
        case 8192: launch_reg_pair<5>(w0, w1, x, ldx, T, v0, v1, s); break;
        default: return 1;
    }
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

// ---- closed-form W0 tile ------------------------------------------------------------------------
// tall: W0[o,i] = d3[p,b] * G_b[(p ^ inv_b[i]) & (L-1)] * d1[i,b],  b=o/L, p=o-bL
// wide: W0[o,i] = d3[o,b] * G_b[(o ^ inv_b[q]) & (L-1)] * d1[q,b],  b=i/L, q=i-bL
// Two separately rounded fp32 multiplies, (d3*G)*d1, so the tile is bit-identical to the oracle.
__global__ void w0_tile_kernel(const float * __restrict__ d1, const float * __restrict__ d3,
                               const float * __restrict__ G, const int * __restrict__ inv,
                               int L, int tall, int64_t o0, int64_t o1, int64_t i0, int64_t i1,
                               float * __restrict__ out) {
    const int64_t no = o1 - o0;
    const int64_t n = no * (i1 - i0);
    for (int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (int64_t) gridDim.x * blockDim.x) {
        const int64_t o = o0 + e % no;
        const int64_t i = i0 + e / no;
        int64_t b, p, q;
        if (tall) { b = o / L; p = o - b * L; q = i; }
        else      { b = i / L; q = i - b * L; p = o; }
        const float g  = G[b * L + ((p ^ (int64_t) inv[b * L + q]) & (L - 1))];
        const float s3 = d3 ? d3[b * L + p] : 1.0f;
        const float tt = s3 * g;
        out[e] = tt * d1[b * L + q];
    }
}

int launch_w0_tile(const DeviceWeight & w, int64_t o0, int64_t o1, int64_t i0, int64_t i1,
                   float * out_dev, hipStream_t s) {
    const int64_t n = (o1 - o0) * (i1 - i0);
    if (n <= 0) return SDH_OK;
    const int threads = 256;
    int64_t blocks = (n + threads - 1) / threads;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(w0_tile_kernel, dim3((unsigned) blocks), dim3(threads), 0, s,
                       w.d1, w.d3, w.G, w.inv, (int) w.L, w.tall ? 1 : 0, o0, o1, i0, i1, out_dev);
    SDH_HIP_CHECK(hipGetLastError());
    return SDH_OK;
}

} // namespace sdh
