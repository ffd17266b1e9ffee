// flaglat.hip — dev tool: how long from an agent-scope 8-byte store in one workgroup until a polling
// agent-scope load in another workgroup sees it, same XCD vs different XCD (s_memrealtime, 10 ns ticks).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/flaglat tools/flaglat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// grid of NB workgroups; workgroup 0 is the producer, all others poll the same granule array (one granule per consumer lane)
__global__ void lat(unsigned long long * gran, unsigned long long * out, unsigned epoch, int n_gran) {
    const int b = blockIdx.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
    if (b == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 500) __builtin_amdgcn_s_sleep(8);     // 5 us: let the pollers settle
        const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
        for (int i = threadIdx.x; i < n_gran; i += blockDim.x)
            __hip_atomic_store(&gran[i], ((unsigned long long) epoch << 32) | (unsigned) i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) { out[0] = ts; out[1] = xcc; }
        return;
    }
    const int i = threadIdx.x % n_gran;
    unsigned long long g, seen = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    do {
        g = __hip_atomic_load(&gran[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned) (g >> 32) == epoch) { seen = __builtin_amdgcn_s_memrealtime(); break; }
        __builtin_amdgcn_s_sleep(1);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < 5000);
    if (threadIdx.x == 0) { out[2 * b] = seen; out[2 * b + 1] = xcc; }
}

int main() {
    CK(hipSetDevice(0));
    const int NB = 64, NG = 256;
    unsigned long long * gran, * out;
    CK(hipMalloc((void **) &gran, NG * 8)); CK(hipMemset(gran, 0, NG * 8));
    CK(hipMalloc((void **) &out, NB * 16));
    std::vector<unsigned long long> h(NB * 2);
    std::vector<double> same, other;
    for (unsigned it = 1; it <= 200; ++it) {
        CK(hipMemset(out, 0, NB * 16));
        hipLaunchKernelGGL(lat, dim3(NB), dim3(256), 0, 0, gran, out, it, NG);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, NB * 16, hipMemcpyDeviceToHost));
        if (it <= 20) continue;
        for (int b = 1; b < NB; ++b) {
            if (!h[2 * b]) continue;
            const double us = (double) (long long) (h[2 * b] - h[0]) * 0.01;
            (h[2 * b + 1] == h[1] ? same : other).push_back(us);
        }
    }
    auto med = [](std::vector<double> & v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto p90 = [](std::vector<double> & v) { if (v.empty()) return -1.0; return v[v.size() * 9 / 10]; };
    printf("store -> seen, consumer on the producer's XCD: median %.2f us, p90 %.2f us (%zu samples)\n", med(same), p90(same), same.size());
    printf("store -> seen, consumer on another XCD:         median %.2f us, p90 %.2f us (%zu samples)\n", med(other), p90(other), other.size());
    return 0;
}
