// base_diag.hip — diagnostic build of the library with in-kernel time stamps in the base-term
// kernel (development tool). Prints where a cold base_apply launch spends its time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/bin/base_diag tools/base_diag.hip
#define SDH_DIAG_STAMPS 1
#include "../llama.cpp_amd/csrc/sdh_api.hip"
#include "../llama.cpp_amd/csrc/sdh_base.hip"
#include "../llama.cpp_amd/csrc/sdh_delta.hip"
#include "../llama.cpp_amd/csrc/sdh_decode.hip"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static sdh_tensor mk(const void * d, int type, int nd, int64_t n0, int64_t n1, size_t es) {
    sdh_tensor t{};
    t.data = d; t.type = type; t.n_dims = nd;
    t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = 1; t.ne[3] = 1;
    t.nb[0] = (int64_t) es; t.nb[1] = t.nb[0] * n0; t.nb[2] = t.nb[1] * n1; t.nb[3] = t.nb[2];
    return t;
}

int main(int argc, char ** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 4096;        // base block length: 1024 / 2048 / 4096 (how does one block's chain scale with L?)
    const int B = 4, NIN = L * 7 / 2, NOUT = L, NPAY = 32, ITERS = 256;
    CK(hipSetDevice(0));
    if (sdh_init(0) != 0) { fprintf(stderr, "init: %s\n", sdh_last_error()); return 1; }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::mt19937 rng(3);
    std::vector<sdh_weight *> ws(NPAY);
    std::vector<float *> xs(NPAY);
    std::vector<uint16_t> ones((size_t) L * B, 0x3C00), hd((size_t) L * B);
    std::vector<int16_t> hp((size_t) L * B), hidx((size_t) NOUT * 8, -1);
    std::vector<uint16_t> hval((size_t) NOUT * 8, 0);
    std::vector<float> hx(NIN);
    float * y;
    CK(hipMalloc((void **) &y, NOUT * 4));
    for (int p = 0; p < NPAY; ++p) {
        for (auto & v : hd) v = (uint16_t) (0x1000 + (rng() & 0x3FF));
        for (int b = 0; b < B; ++b) { for (int i = 0; i < L; ++i) hp[(size_t) b * L + i] = (int16_t) i; std::shuffle(hp.begin() + (size_t) b * L, hp.begin() + (size_t) (b + 1) * L, rng); }
        sdh_weight_desc d{};
        d.n_in = NIN; d.n_out = NOUT;
        d.d_idx = mk(hidx.data(), SDH_TYPE_I16, 2, 8, NOUT, 2);
        d.d_val = mk(hval.data(), SDH_TYPE_F16, 2, 8, NOUT, 2);
        d.base_d1 = mk(ones.data(), SDH_TYPE_F16, 2, L, B, 2);
        d.base_d2 = mk(hd.data(), SDH_TYPE_F16, 2, L, B, 2);
        d.base_d3 = mk(ones.data(), SDH_TYPE_F16, 2, L, B, 2);
        d.base_perm1 = mk(hp.data(), SDH_TYPE_I16, 2, L, B, 2);
        if (sdh_weight_create(&d, &ws[p]) != 0) { fprintf(stderr, "create: %s\n", sdh_last_error()); return 1; }
        for (auto & f : hx) f = (float) ((int) (rng() % 2001) - 1000) * 1e-3f;
        CK(hipMalloc((void **) &xs[p], NIN * 4)); CK(hipMemcpy(xs[p], hx.data(), NIN * 4, hipMemcpyHostToDevice));
    }
    // 320 MB of traffic between uses keeps the parameters out of the Infinity Cache
    char * flush; const size_t FL = (size_t) 320 << 20;
    CK(hipMalloc((void **) &flush, FL));
    unsigned long long * dbg;
    CK(hipMalloc((void **) &dbg, B * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sdh_diag_buf), &dbg, sizeof(dbg)));
    double acc[16] = {0};
    unsigned long long hdv[4 * 16];
    int n = 0;
    for (int it = 0; it < ITERS; ++it) {
        const int p = it % NPAY;
        if (p == 0) CK(hipMemsetAsync(flush, it, FL, s));
        if (sdh_forward(ws[p], xs[p], NIN, 1, y, NOUT, SDH_FWD_NO_FUSE, s) != 0) { fprintf(stderr, "fwd: %s\n", sdh_last_error()); return 1; }
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hdv, dbg, sizeof(hdv), hipMemcpyDeviceToHost));
        if (it < 32) continue;
        const int ks[] = {1, 2, 3, 4, 5, 9};
        for (int k : ks) { double m = 0; for (int b = 0; b < B; ++b) m += (double) (hdv[b * 16 + k] - hdv[b * 16]) * 0.01; acc[k] += m / B; }
        ++n;
    }
    printf("base_apply_kernel, L = %d, cold, cumulative us since kernel start (mean over %d launches, %d blocks):\n", L, n, B);
    printf("  x landed (and d1 applied)   %7.2f\n", acc[1] / n);
    printf("  perm1 gather                %7.2f\n", acc[2] / n);
    printf("  first FWHT                  %7.2f\n", acc[3] / n);
    printf("  d2 + back to layout 0       %7.2f\n", acc[4] / n);
    printf("  second FWHT                 %7.2f\n", acc[5] / n);
    printf("  d3 + stores issued          %7.2f\n", acc[9] / n);
    return 0;
}
