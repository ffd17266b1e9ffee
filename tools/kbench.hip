// kbench.hip — kernel-tuning microbenchmarks (development tool, not part of the product path).
//
// Measures, on one MI355X, launch-shape ceilings that the SeedDelta decode kernels are compared
// against: an empty launch, a pure streaming read of the cfg3 payload size with the same grid,
// and the library's own forward through the C ABI. All timings: hipEvents over many back-to-back
// launches rotating over NPAY distinct buffers (> 256 MB in total, so the Infinity Cache does not
// serve them).
//
//   hipcc --offload-arch=gfx950 -O3 -o kbench tools/kbench.hip -Iinclude -Lllama.cpp_amd/csrc -lseeddelta_hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <random>
#include <vector>

#include "seeddelta_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty(float * out) { if (out == nullptr && threadIdx.x == 9999) out[0] = 0; }

// one wave per 3072-byte row: 3 x 16 B per lane, reduce, one store per wave
__global__ __launch_bounds__(256) void k_stream(const uint4 * __restrict__ src, float * __restrict__ out, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint4 * p = src + (size_t) row * 192;
    uint4 a = p[lane], b = p[64 + lane], c = p[128 + lane];
    float acc = __uint_as_float(a.x ^ b.y ^ c.z) + __uint_as_float(a.w ^ b.x ^ c.y) + __uint_as_float(a.y ^ a.z ^ b.z ^ b.w ^ c.x ^ c.w);
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[row] = acc;
}

// same, plus the dependent gather: a 2-byte index per 4 lanes, then 32 B of x from L2
__global__ __launch_bounds__(256) void k_stream_gather(const uint4 * __restrict__ src, const short * __restrict__ idx,
                                                       const float * __restrict__ x, float * __restrict__ out, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint4 * p = src + (size_t) row * 192;
    const short * ir = idx + (size_t) row * 48;
    int i0 = ir[lane >> 2], i1 = ir[16 + (lane >> 2)], i2 = ir[32 + (lane >> 2)];
    uint4 a = p[lane], b = p[64 + lane], c = p[128 + lane];
    const float4 * x0 = reinterpret_cast<const float4 *>(x + i0 * 32 + (lane & 3) * 8);
    const float4 * x1 = reinterpret_cast<const float4 *>(x + i1 * 32 + (lane & 3) * 8);
    const float4 * x2 = reinterpret_cast<const float4 *>(x + i2 * 32 + (lane & 3) * 8);
    float4 xa = x0[0], xb = x0[1], xc = x1[0], xd = x1[1], xe = x2[0], xf = x2[1];
    float acc = __uint_as_float(a.x) * xa.x + __uint_as_float(a.y) * xa.y + __uint_as_float(a.z) * xb.z + __uint_as_float(a.w) * xb.w
              + __uint_as_float(b.x) * xc.x + __uint_as_float(b.y) * xc.y + __uint_as_float(b.z) * xd.z + __uint_as_float(b.w) * xd.w
              + __uint_as_float(c.x) * xe.x + __uint_as_float(c.y) * xe.y + __uint_as_float(c.z) * xf.z + __uint_as_float(c.w) * xf.w;
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[row] = acc;
}

// launch-shape sweep: THREADS per workgroup, ROWS_PER_WAVE rows (3072 B each) per wave, optional
// non-temporal loads
template <int THREADS, int RPW, bool NT>
__global__ __launch_bounds__(THREADS) void k_stream_shape(const uint4 * __restrict__ src, float * __restrict__ out, int rows) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    uint4 v[RPW][3];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = wave * RPW + r;
        const uint4 * p = src + (size_t) (row < rows ? row : 0) * 192;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if constexpr (NT) {
                const unsigned * q = reinterpret_cast<const unsigned *>(p + k * 64 + lane);
                v[r][k].x = __builtin_nontemporal_load(q); v[r][k].y = __builtin_nontemporal_load(q + 1);
                v[r][k].z = __builtin_nontemporal_load(q + 2); v[r][k].w = __builtin_nontemporal_load(q + 3);
            } else {
                v[r][k] = p[k * 64 + lane];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = wave * RPW + r;
        float acc = __uint_as_float(v[r][0].x ^ v[r][1].y ^ v[r][2].z) + __uint_as_float(v[r][0].w ^ v[r][1].x ^ v[r][2].y) +
                    __uint_as_float(v[r][0].y ^ v[r][0].z ^ v[r][1].z ^ v[r][1].w ^ v[r][2].x ^ v[r][2].w);
        for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0 && row < rows) out[row] = acc;
    }
}

static float time_loop(hipStream_t s, int iters, const std::function<void(int)> & launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 64; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;   // us per launch
}

static sdh_tensor mk(const void * d, int type, int nd, int64_t n0, int64_t n1, int64_t n2, size_t es) {
    sdh_tensor t{};
    t.data = d; t.type = type; t.n_dims = nd;
    t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = n2; t.ne[3] = 1;
    t.nb[0] = (int64_t) es; t.nb[1] = t.nb[0] * n0; t.nb[2] = t.nb[1] * n1; t.nb[3] = t.nb[2] * n2;
    return t;
}

int main(int argc, char ** argv) {
    const int NPAY = 32, ROWS = 4096, NIN = 14336, NB = 48, BLOCK = 32, L = 4096, B = 4;
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    const bool with_base = argc > 2 ? atoi(argv[2]) != 0 : true;
    CK(hipSetDevice(0));
    if (sdh_init(0) != 0) { fprintf(stderr, "sdh_init: %s\n", sdh_last_error()); return 1; }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    const size_t row_bytes = (size_t) NB * BLOCK * 2;          // 3072
    std::vector<uint4 *> bufs(NPAY);
    std::vector<short *> idxs(NPAY);
    std::vector<float *> xs(NPAY), ys(NPAY);
    std::mt19937 rng(1);
    std::vector<uint16_t> hv((size_t) ROWS * NB * BLOCK);
    std::vector<int16_t> hi((size_t) ROWS * NB);
    std::vector<float> hx(NIN);
    std::vector<sdh_weight *> ws(NPAY);
    std::vector<uint16_t> hrs(ROWS), hd((size_t) L * B);
    std::vector<int16_t> hp((size_t) L * B);
    for (int p = 0; p < NPAY; ++p) {
        for (auto & v : hv) v = (uint16_t) (0x2000 + (rng() & 0x0FFF));     // small positive halfs
        for (int r = 0; r < ROWS; ++r) for (int b = 0; b < NB; ++b) hi[(size_t) r * NB + b] = (int16_t) (rng() % (NIN / BLOCK));
        for (auto & v : hx) v = (float) ((int) (rng() % 2001) - 1000) * 1e-3f;
        for (auto & v : hrs) v = 0x3C00;
        for (auto & v : hd) v = (uint16_t) (0x1000 + (rng() & 0x3FF));
        for (int b = 0; b < B; ++b) { for (int i = 0; i < L; ++i) hp[(size_t) b * L + i] = (int16_t) i; std::shuffle(hp.begin() + (size_t) b * L, hp.begin() + (size_t) (b + 1) * L, rng); }
        CK(hipMalloc((void **) &bufs[p], ROWS * row_bytes));
        CK(hipMemcpy(bufs[p], hv.data(), ROWS * row_bytes, hipMemcpyHostToDevice));
        CK(hipMalloc((void **) &idxs[p], hi.size() * 2));
        CK(hipMemcpy(idxs[p], hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc((void **) &xs[p], NIN * 4));
        CK(hipMemcpy(xs[p], hx.data(), NIN * 4, hipMemcpyHostToDevice));
        CK(hipMalloc((void **) &ys[p], ROWS * 4));
        sdh_weight_desc d{};
        d.n_in = NIN; d.n_out = ROWS;
        d.b_idx = mk(hi.data(), SDH_TYPE_I16, 2, NB, ROWS, 1, 2);
        d.b_val = mk(hv.data(), SDH_TYPE_F16, 3, BLOCK, NB, ROWS, 2);
        d.row_scale = mk(hrs.data(), SDH_TYPE_F16, 1, ROWS, 1, 1, 2);
        if (with_base) {
            std::vector<uint16_t> ones((size_t) L * B, 0x3C00);
            d.base_d1 = mk(ones.data(), SDH_TYPE_F16, 2, L, B, 1, 2);
            d.base_d2 = mk(hd.data(), SDH_TYPE_F16, 2, L, B, 1, 2);
            d.base_d3 = mk(ones.data(), SDH_TYPE_F16, 2, L, B, 1, 2);
            d.base_perm1 = mk(hp.data(), SDH_TYPE_I16, 2, L, B, 1, 2);
            if (sdh_weight_create(&d, &ws[p]) != 0) { fprintf(stderr, "create: %s\n", sdh_last_error()); return 1; }
        } else {
            if (sdh_weight_create(&d, &ws[p]) != 0) { fprintf(stderr, "create: %s\n", sdh_last_error()); return 1; }
        }
    }
    CK(hipDeviceSynchronize());
    const double mb = (double) ROWS * row_bytes / 1e6;

    float t_empty = time_loop(s, iters, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, ys[0]); });
    float t_stream = time_loop(s, iters, [&](int i) { hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); });
    float t_gather = time_loop(s, iters, [&](int i) { int p = i % NPAY; hipLaunchKernelGGL(k_stream_gather, dim3(1024), dim3(256), 0, s, bufs[p], idxs[p], xs[p], ys[p], ROWS); });
    float t_fwd = time_loop(s, iters, [&](int i) { int p = i % NPAY; sdh_forward(ws[p], xs[p], NIN, 1, ys[p], ROWS, 0, s); });
    {
        struct { const char * name; std::function<void(int)> fn; } shapes[] = {
            {"256thr x1row", [&](int i) { hipLaunchKernelGGL((k_stream_shape<256, 1, false>), dim3(1024), dim3(256), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
            {"512thr x1row", [&](int i) { hipLaunchKernelGGL((k_stream_shape<512, 1, false>), dim3(512), dim3(512), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
            {"1024thr x1row", [&](int i) { hipLaunchKernelGGL((k_stream_shape<1024, 1, false>), dim3(256), dim3(1024), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
            {"256thr x2row", [&](int i) { hipLaunchKernelGGL((k_stream_shape<256, 2, false>), dim3(512), dim3(256), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
            {"256thr x4row", [&](int i) { hipLaunchKernelGGL((k_stream_shape<256, 4, false>), dim3(256), dim3(256), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
            {"512thr x2row", [&](int i) { hipLaunchKernelGGL((k_stream_shape<512, 2, false>), dim3(256), dim3(512), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
            {"256thr x1row nt", [&](int i) { hipLaunchKernelGGL((k_stream_shape<256, 1, true>), dim3(1024), dim3(256), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
            {"512thr x2row nt", [&](int i) { hipLaunchKernelGGL((k_stream_shape<512, 2, true>), dim3(256), dim3(512), 0, s, bufs[i % NPAY], ys[i % NPAY], ROWS); }},
        };
        printf("stream-read launch-shape sweep (us per launch, GB/s):\n");
        for (auto & sh : shapes) { float t = time_loop(s, iters, sh.fn); printf("  %-18s %6.2f  %5.0f\n", sh.name, t, mb * 1e6 / t / 1e3); }
    }
    printf("back-to-back launch period, us (payload %.2f MB per launch, %d distinct payloads)\n", mb, NPAY);
    printf("  empty 1024x256           : %7.2f\n", t_empty);
    printf("  stream read              : %7.2f  (%.0f GB/s)\n", t_stream, mb * 1e6 / t_stream / 1e3);
    printf("  stream + dependent gather: %7.2f  (%.0f GB/s)\n", t_gather, mb * 1e6 / t_gather / 1e3);
    printf("  sdh_forward (%s)   : %7.2f  (%.0f GB/s algorithmic)\n", with_base ? "base+block" : "block only", t_fwd, 13.189e6 / t_fwd / 1e3);
    if (with_base) {
        float t_nf = time_loop(s, iters, [&](int i) { int p = i % NPAY; sdh_forward(ws[p], xs[p], NIN, 1, ys[p], ROWS, SDH_FWD_NO_FUSE, s); });
        printf("  sdh_forward two launches   : %7.2f  (%.0f GB/s algorithmic)\n", t_nf, 13.189e6 / t_nf / 1e3);
    }
    return 0;
}
