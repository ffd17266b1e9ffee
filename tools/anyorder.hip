// dev: does hipExtAnyOrderLaunch let two kernels of one stream overlap on this device?
// Kernel A spins ~30 us and stamps its end; kernel B (launched right behind it on the same stream with
// the flag) stamps its start. B.start < A.end means the packets were not serialised.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_kernel(unsigned long long * t, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    t[0] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long) ticks) __builtin_amdgcn_s_sleep(8);
    t[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp_kernel(unsigned long long * t) { t[2] = __builtin_amdgcn_s_memrealtime(); }
int main() {
    unsigned long long * d; CK(hipMalloc((void **) &d, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int flags = 0; flags <= 1; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d, 0, 64, s));
            CK(hipStreamSynchronize(s));
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, d, 3000);
            hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, (unsigned) flags, d);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            unsigned long long h[3]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            printf("flags=%d rep=%d  A: 0 .. %.2f us   B start %.2f us  (%s)\n", flags, rep, (h[1] - h[0]) / 100.0,
                   ((long long) h[2] - (long long) h[0]) / 100.0, h[2] < h[1] ? "OVERLAP" : "serialised");
        }
    }
    return 0;
}
