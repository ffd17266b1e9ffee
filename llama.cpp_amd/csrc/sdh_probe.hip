// sdh_probe.hip — measurement hook: the device's floor for ONE launch that streams `bytes` once.
//
// The decode kernel reads a 13 MB payload once per launch; a launch that small never reaches steady-state HBM
// streaming. This probe measures what the device gives a kernel that does nothing else: a wave per 3072-byte run,
// three 16-byte non-temporal loads per lane, one reduction, one store per wave — launched back to back on the
// caller's stream over `n_buffers` distinct buffers (more than the 256 MB Infinity Cache in total, as bench.py's
// rotation), timed with HIP events on that stream. bench.py reports it as roofline.attainable beside the 8 TB/s
// peak. No counterpart in the reference.
#include "sdh_internal.h"
#include <vector>

namespace sdh {

__global__ __launch_bounds__(256)
void probe_stream_kernel(const uint4 * __restrict__ src, float * __restrict__ out, int rows) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4 * p = reinterpret_cast<const u32x4 *>(src) + (size_t) row * 192;
    const u32x4 a = __builtin_nontemporal_load(p + lane), b = __builtin_nontemporal_load(p + 64 + lane), c = __builtin_nontemporal_load(p + 128 + lane);
    float acc = __uint_as_float(a.x ^ b.y ^ c.z) + __uint_as_float(a.w ^ b.x ^ c.y) + __uint_as_float(a.y ^ a.z ^ b.z ^ b.w ^ c.x ^ c.w);
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[row] = acc;
}

__global__ void probe_empty_kernel(float * out) { if (out == nullptr && threadIdx.x == 9999) out[0] = 0; }

} // namespace sdh

using namespace sdh;

extern "C" int sdh_probe_stream_floor(int64_t bytes, int n_buffers, int reps, void * stream, double * us_stream, double * us_empty) {
    if (!lib().inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    if (bytes < 3072 || n_buffers < 1 || n_buffers > 1024 || reps < 1 || !us_stream) { set_error("sdh_probe_stream_floor: bad argument"); return SDH_ERR_INVALID; }
    hipStream_t s = (hipStream_t) stream;
    const int rows = (int) (bytes / 3072);
    std::vector<void *> bufs((size_t) n_buffers, nullptr);
    float * out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SDH_OK;
    auto cleanup = [&] { for (void * b : bufs) (void) hipFree(b); (void) hipFree(out); if (e0) (void) hipEventDestroy(e0); if (e1) (void) hipEventDestroy(e1); };
    auto fail = [&](hipError_t e, const char * what) { set_error("sdh_probe_stream_floor: %s: %s", what, hipGetErrorString(e)); cleanup(); return SDH_ERR_HIP; };
    hipError_t e;
    for (int i = 0; i < n_buffers; ++i) {
        if ((e = hipMalloc(&bufs[(size_t) i], (size_t) rows * 3072)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMemsetAsync(bufs[(size_t) i], i + 1, (size_t) rows * 3072, s)) != hipSuccess) return fail(e, "hipMemsetAsync");
    }
    if ((e = hipMalloc((void **) &out, (size_t) rows * 4)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) return fail(e, "hipEventCreate");
    const dim3 grid((unsigned) ((rows + 3) / 4));
    for (int pass = 0; pass < 2; ++pass) {              // 0: the streaming launch, 1: an empty launch of the same shape
        if (pass == 1 && !us_empty) break;
        for (int w = 0; w < 2; ++w) {                   // w = 0: warm-up round
            if (w == 1 && (e = hipEventRecord(e0, s)) != hipSuccess) return fail(e, "hipEventRecord");
            for (int r = 0; r < (w ? reps : 2); ++r)
                for (int i = 0; i < n_buffers; ++i) {
                    if (pass == 0) hipLaunchKernelGGL(probe_stream_kernel, grid, dim3(256), 0, s, (const uint4 *) bufs[(size_t) i], out, rows);
                    else           hipLaunchKernelGGL(probe_empty_kernel, grid, dim3(256), 0, s, out);
                }
        }
        if ((e = hipEventRecord(e1, s)) != hipSuccess || (e = hipEventSynchronize(e1)) != hipSuccess) return fail(e, "hipEventSynchronize");
        float ms = 0.0f;
        if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) return fail(e, "hipEventElapsedTime");
        (pass == 0 ? *us_stream : *us_empty) = (double) ms * 1e3 / ((double) reps * n_buffers);
    }
    cleanup();
    return rc;
}
