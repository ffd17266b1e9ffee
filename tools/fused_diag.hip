// fused_diag.hip — diagnostic build of the fused decode kernel with in-kernel time stamps
// (development tool): when do the base workgroups finish, when do the residual waves finish their
// own work, and how long do they wait for the base term.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/bin/fused_diag tools/fused_diag.hip
#define SDH_DIAG_STAMPS 1
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

int main() {
    const int L = 4096, B = 4, NIN = 14336, NOUT = 4096, NB = 48, BLOCK = 32, NPAY = 32, ITERS = 160;
    const int GRID = B + NOUT / 4;
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
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sdh_diag_buf), &dbg, sizeof(dbg)));
    std::vector<unsigned long long> h((size_t) GRID * 32);
    double acc[16] = {0};
    int n = 0;
    for (int it = 0; it < ITERS; ++it) {
        const int p = it % NPAY;
        CK(hipMemsetAsync(dbg, 0, DB, s));
        if (sdh_forward(ws[p], xs[p], NIN, 1, y, NOUT, 0, s) != 0) { fprintf(stderr, "fwd: %s\n", sdh_last_error()); return 1; }
        CK(hipStreamSynchronize(s));
        if (it < 64) continue;       // warm-up
        CK(hipMemcpy(h.data(), dbg, DB, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int g = B; g < GRID; ++g) { unsigned long long st = h[(size_t) g * 32 + 10]; if (st && st < t0) t0 = st; }
        auto us = [&](int g, int k) { return (double) (long long) (h[(size_t) g * 32 + k] - t0) * 0.01; };
        double bx = 0, bd = 0, bp = 0, bg = 0, be = 0, polls = 0, rs = 0, rm = 0, rx = 0;
        for (int b = 0; b < B; ++b) {
            bx = std::max(bx, us(b, 1)); bd = std::max(bd, us(b, 5)); bp = std::max(bp, us(b, 6));
            bg = std::max(bg, us(b, 12)); be = std::max(be, us(b, 13)); polls = std::max(polls, (double) h[(size_t) b * 32 + 14]);
        }
        for (int g = B; g < GRID; ++g) { rs = std::max(rs, us(g, 10)); rm += us(g, 11); rx = std::max(rx, us(g, 11)); }
        rm /= (GRID - B);
        const double v[9] = {rs, bx, bd, bp, bg, be, polls, rm, rx};
        for (int k = 0; k < 9; ++k) acc[k] += v[k];
        ++n;
    }
    printf("fused decode kernel, cfg3 shape, us since the first residual workgroup started (mean over %d launches)\n", n);
    printf("  last residual workgroup starts          %6.2f\n", acc[0] / n);
    printf("  residual: row sums reduced, mean / max  %6.2f / %6.2f\n", acc[7] / n, acc[8] / n);
    printf("  base: x landed (slowest block)          %6.2f\n", acc[1] / n);
    printf("  base: second FWHT done (slowest)        %6.2f\n", acc[2] / n);
    printf("  base: granules stored (slowest)         %6.2f\n", acc[3] / n);
    printf("  base: gather complete (slowest)         %6.2f   polls (max over blocks) %.1f\n", acc[4] / n, acc[6] / n);
    printf("  base: outputs stored (slowest)          %6.2f\n", acc[5] / n);
#ifdef SDH_DIAG_TWICE
    printf("  base block 0 phases, cold run then warm re-run (last launch):\n");
    {
        auto u0 = [&](int k) { return (double) (long long) (h[k] - h[16 + 1]) * 0.01; };
        printf("    cold: loads landed %.2f  P1 %.2f  FWHT1 %.2f  D2/P2/back %.2f  FWHT2 %.2f (us after loads landed)\n", 0.0, u0(16 + 2), u0(16 + 3), u0(16 + 4), u0(16 + 5));
        auto u1 = [&](int k) { return (double) (long long) (h[k] - h[7]) * 0.01; };
        printf("    warm: loads landed %.2f  P1 %.2f  FWHT1 %.2f  D2/P2/back %.2f  FWHT2 %.2f  end %.2f (us after the re-run started)\n", u1(1), u1(2), u1(3), u1(4), u1(5), u1(8));
    }
#endif
    return 0;
}
