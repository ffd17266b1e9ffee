// fused_diag.hip — diagnostic build of the fused decode kernel with in-kernel time stamps
// (development tool): when do the base workgroups finish, when do the residual waves finish their
// own work, and how long do they wait for the base term.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/bin/fused_diag tools/fused_diag.hip
#ifndef SDH_DIAG_NO_STAMPS
#define SDH_DIAG_STAMPS 1
#endif
#include "../llama.cpp_amd/csrc/sdh_api.hip"
#include "../llama.cpp_amd/csrc/sdh_base.hip"
#include "../llama.cpp_amd/csrc/sdh_delta.hip"
#include "../llama.cpp_amd/csrc/sdh_decode.hip"
#include "../llama.cpp_amd/csrc/sdh_spmm.hip"
#include "../llama.cpp_amd/csrc/sdh_w0mfma.hip"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static sdh_tensor mk(const void * d, int type, int nd, int64_t n0, int64_t n1, int64_t n2, size_t es) {
    sdh_tensor t{};
    t.data = d; t.type = type; t.n_dims = nd;
    t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = n2; t.ne[3] = 1;
    t.nb[0] = (int64_t) es; t.nb[1] = t.nb[0] * n0; t.nb[2] = t.nb[1] * n1; t.nb[3] = t.nb[2] * n2;
    return t;
}

int main(int argc, char ** argv) {
    const int BURST = argc > 1 ? atoi(argv[1]) : 1;
    const int L = 4096, B = 4, NIN = 14336, NOUT = 4096, NB = 48, BLOCK = 32, NPAY = 32, ITERS = 160;
    const int GRID = B + NOUT / 8 + 16;  // 512-thread workgroups, 8 rows each, plus shadow ids (sdh_decode.hip)
    CK(hipSetDevice(0));
    if (sdh_init(0) != 0) { fprintf(stderr, "init: %s\n", sdh_last_error()); return 1; }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::mt19937 rng(3);
    std::vector<sdh_weight *> ws(NPAY);
    std::vector<float *> xs(NPAY);
    std::vector<uint16_t> ones((size_t) L * B, 0x3C00), hd((size_t) L * B), hrs(NOUT, 0x3C00);
    std::vector<int16_t> hp((size_t) L * B), hi((size_t) NOUT * NB);
    std::vector<uint16_t> hv((size_t) NOUT * NB * BLOCK);
    std::vector<float> hx(NIN);
    float * y;
    CK(hipMalloc((void **) &y, NOUT * 4));
    for (int p = 0; p < NPAY; ++p) {
        for (auto & v : hd) v = (uint16_t) (0x1000 + (rng() & 0x3FF));
        for (auto & v : hv) v = (uint16_t) (0x2000 + (rng() & 0x0FFF));
        for (auto & v : hi) v = (int16_t) (rng() % (NIN / BLOCK));
        for (int b = 0; b < B; ++b) { for (int i = 0; i < L; ++i) hp[(size_t) b * L + i] = (int16_t) i; std::shuffle(hp.begin() + (size_t) b * L, hp.begin() + (size_t) (b + 1) * L, rng); }
        sdh_weight_desc d{};
        d.n_in = NIN; d.n_out = NOUT;
        d.b_idx = mk(hi.data(), SDH_TYPE_I16, 2, NB, NOUT, 1, 2);
        d.b_val = mk(hv.data(), SDH_TYPE_F16, 3, BLOCK, NB, NOUT, 2);
        d.row_scale = mk(hrs.data(), SDH_TYPE_F16, 1, NOUT, 1, 1, 2);
        d.base_d1 = mk(ones.data(), SDH_TYPE_F16, 2, L, B, 1, 2);
        d.base_d2 = mk(hd.data(), SDH_TYPE_F16, 2, L, B, 1, 2);
        d.base_d3 = mk(ones.data(), SDH_TYPE_F16, 2, L, B, 1, 2);
        d.base_perm1 = mk(hp.data(), SDH_TYPE_I16, 2, L, B, 1, 2);
        if (sdh_weight_create(&d, &ws[p]) != 0) { fprintf(stderr, "create: %s\n", sdh_last_error()); return 1; }
        for (auto & f : hx) f = (float) ((int) (rng() % 2001) - 1000) * 1e-3f;
        CK(hipMalloc((void **) &xs[p], NIN * 4)); CK(hipMemcpy(xs[p], hx.data(), NIN * 4, hipMemcpyHostToDevice));
    }
    unsigned long long * dbg;
    const size_t DB = (size_t) GRID * 32 * 8;
    CK(hipMalloc((void **) &dbg, DB));
#ifdef SDH_DIAG_STAMPS
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sdh_diag_buf), &dbg, sizeof(dbg)));
#endif
    std::vector<unsigned long long> h((size_t) GRID * 32);
    double a_base_pub = 0, a_base_done = 0, a_base_x = 0, a_red_mean = 0, a_red_max = 0, a_wait_mean = 0, a_end_max = 0, a_start_max = 0;
    int n = 0;
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    if (argc > 2) {       // fused_diag <burst> <n>: back-to-back launch period from C++ (no stamps read), n launches
        const int N = atoi(argv[2]);
        for (int k = 0; k < 256; ++k) sdh_forward(ws[k % NPAY], xs[k % NPAY], NIN, 1, y, NOUT, 0, s);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(ev0, s));
        for (int k = 0; k < N; ++k) sdh_forward(ws[k % NPAY], xs[k % NPAY], NIN, 1, y, NOUT, 0, s);
        CK(hipEventRecord(ev1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, ev0, ev1));
        printf("back-to-back period over %d launches driven from C++: %.3f us\n", N, ms * 1e3 / N);
        return 0;
    }
    double a_event_us = 0;
    for (int it = 0; it < ITERS; ++it) {
        const int p = it % NPAY;
        CK(hipMemsetAsync(dbg, 0, DB, s));
        // BURST > 1: that many launches back to back on rotating payloads; the stamps read are the last one's
        for (int k = 0; k + 1 < BURST; ++k)
            if (sdh_forward(ws[(p + 1 + k) % NPAY], xs[(p + 1 + k) % NPAY], NIN, 1, y, NOUT, 0, s) != 0) { fprintf(stderr, "fwd: %s\n", sdh_last_error()); return 1; }
        CK(hipEventRecord(ev0, s));
        if (sdh_forward(ws[p], xs[p], NIN, 1, y, NOUT, 0, s) != 0) { fprintf(stderr, "fwd: %s\n", sdh_last_error()); return 1; }
        CK(hipEventRecord(ev1, s));
        CK(hipStreamSynchronize(s));
        if (it < 64) continue;       // census + warm-up
        { float ms = 0; CK(hipEventElapsedTime(&ms, ev0, ev1)); a_event_us += ms * 1e3; }
        CK(hipMemcpy(h.data(), dbg, DB, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        // time zero: the earliest stamp of the launch (a residual workgroup's start; the base role has no stamp before its loads land)
        for (int g = B; g < GRID; ++g) { unsigned long long st = h[(size_t) g * 32 + 10]; if (st && st < t0) t0 = st; }
        for (int b = 0; b < B; ++b) { unsigned long long st = h[(size_t) b * 32 + 1]; if (st && st < t0) t0 = st; }
        double base_pub = 0, base_done = 0, base_x = 0, red_mean = 0, red_max = 0, wait_mean = 0, end_max = 0, start_max = 0;
        for (int b = 0; b < B; ++b) {
            base_done = std::max(base_done, (double) (h[(size_t) b * 32 + 5] - t0) * 0.01);
            base_pub = std::max(base_pub, (double) (h[(size_t) b * 32 + 6] - t0) * 0.01);
            base_x = std::max(base_x, (double) (h[(size_t) b * 32 + 1] - t0) * 0.01);
        }
        int nres = 0;
        for (int g = B; g < GRID; ++g) {
            if (!h[(size_t) g * 32 + 10]) continue;          // shadow or unused id: no stamps
            ++nres;
            const double st = (double) (h[(size_t) g * 32 + 10] - t0) * 0.01, r = (double) (h[(size_t) g * 32 + 11] - t0) * 0.01,
                         e = (double) (h[(size_t) g * 32 + 12] - t0) * 0.01;
            red_mean += r; red_max = std::max(red_max, r); wait_mean += e - r; end_max = std::max(end_max, e); start_max = std::max(start_max, st);
        }
        red_mean /= std::max(nres, 1); wait_mean /= std::max(nres, 1);
        a_base_pub += base_pub; a_base_done += base_done; a_base_x += base_x; a_red_mean += red_mean; a_red_max += red_max; a_wait_mean += wait_mean; a_end_max += end_max; a_start_max += start_max;
        ++n;
    }
    printf("fused decode kernel, cfg3 shape, us since the earliest stamp of the launch (mean over %d launches)\n", n);
    printf("  last residual workgroup starts        %6.2f\n", a_start_max / n);
    printf("  base: x landed (slowest block)        %6.2f\n", a_base_x / n);
    printf("  base: second FWHT done (slowest)      %6.2f\n", a_base_done / n);
    printf("  base: published, flag stored (slowest) %6.2f\n", a_base_pub / n);
    printf("  residual: own sum reduced, mean / max %6.2f / %6.2f\n", a_red_mean / n, a_red_max / n);
    printf("  residual: wait for the base, mean     %6.2f\n", a_wait_mean / n);
    printf("  last residual wave has its base term  %6.2f\n", a_end_max / n);
    printf("  HIP events around the (isolated) launch  %6.2f\n", a_event_us / n);
    {   // base block 0 of the last launch, phase by phase (stamps 1..6 of sdh_base_dev.h / sdh_decode.hip)
        unsigned long long t0 = ~0ull;
        for (int g = 0; g < GRID; ++g) for (int k : {1, 10}) { unsigned long long st = h[(size_t) g * 32 + k]; if (st && st < t0) t0 = st; }
        for (int b = 0; b < B; ++b) {
            printf("  base block %d (last launch): x,d1 landed %.2f | P1 gathered %.2f | FWHT1 %.2f | D2/back %.2f | FWHT2 %.2f | published %.2f\n", b,
                   (double) (h[(size_t) b * 32 + 1] - t0) * 0.01, (double) (h[(size_t) b * 32 + 2] - t0) * 0.01, (double) (h[(size_t) b * 32 + 3] - t0) * 0.01,
                   (double) (h[(size_t) b * 32 + 4] - t0) * 0.01, (double) (h[(size_t) b * 32 + 5] - t0) * 0.01, (double) (h[(size_t) b * 32 + 6] - t0) * 0.01);
        }
    }
    return 0;
}
