// placement.hip — dev tool: where does the dispatcher put the workgroups of one launch?
// Every workgroup records its (XCC, SE, CU) and stays resident for ~20 us; prints workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/placement tools/placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void probe(unsigned * out, int lds_floats) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
        if (lds_floats > 0) lds[0] = 1.0f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(8);   // 20 us
}

int main() {
    CK(hipSetDevice(0));
    unsigned * d;
    CK(hipMalloc((void **) &d, 4096 * 8));
    const int cfgs[][3] = {{256, 1024, 16384}, {260, 1024, 16384}, {252 + 4, 1024, 32768}, {516, 512, 16384}, {1028, 256, 16384}};
    for (auto & c : cfgs) {
        std::vector<unsigned> h(c[0] * 2);
        hipLaunchKernelGGL(probe, dim3(c[0]), dim3(c[1]), c[2], 0, d, c[2] / 4);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, int> per_cu;
        std::map<unsigned, int> first4;
        for (int g = 0; g < c[0]; ++g) {
            const unsigned hw = h[g * 2], xcc = h[g * 2 + 1] & 15;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_cu[key]++;
            if (g < 4) first4[key] = g;
        }
        std::map<int, int> hist;
        for (auto & kv : per_cu) hist[kv.second]++;
        printf("grid %d x %d threads, %d B LDS: %zu distinct CUs;", c[0], c[1], c[2], per_cu.size());
        for (auto & kv : hist) printf(" %d CUs hold %d WG(s);", kv.second, kv.first);
        printf(" CUs of workgroups 0-3 hold:");
        for (auto & kv : first4) printf(" %d", per_cu[kv.first]);
        printf("\n");
    }
    return 0;
}
