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
