// sdh_api.hip — the C ABI of include/seeddelta_hip.h: device set-up, payload validation and
// upload, forward dispatch. Host code only (kernels live in sdh_base.hip / sdh_delta.hip).
#include "sdh_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace sdh {

static thread_local std::string g_last_error;

void set_error(const char * fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

LibState & lib() {
    static LibState s;
    return s;
}

std::atomic<long long> g_alloc_calls{0};

static std::mutex g_mu;

// ---- sdh_forward_noalloc: the scratch its kernels use instead of a hipMalloc'ed workspace --------------
// The base term v[T_chunk][B][L] that the base kernel writes and the residual kernel reads. Every GPU the
// library runs on holds its own copy of this array, at its own address, until the process ends. In host code
// the array's NAME is only a stand-in: its device address is taken once per device with hipGetSymbolAddress
// (sdh_init, that device current) and kept in g_scratch_ptr; kernels and hipMemsetAsync only ever see that
// pointer.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t) SDH_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
static void * g_scratch_ptr[MAX_DEVICES] = {};

// Which stream a call runs on. Device I/O: exactly the caller's handle, and NULL is HIP's legacy default
// stream, as in every HIP API — so work the caller queued on that stream (the producer of x, the consumer
// of y) is ordered with ours. Host I/O calls are synchronous (copy in, run, copy out, wait) and own their
// ordering, so NULL there means the library's private non-blocking stream (sdh_library_stream()).
static inline hipStream_t resolve_stream(void * stream, int flags) {
    if ((flags & SDH_FWD_HOST_IO) && !stream) return lib().stream;
    return (hipStream_t) stream;
}

// ---- host-side tensor helpers ---------------------------------------------------------------
static inline bool present(const sdh_tensor & t) { return t.data != nullptr; }

static float half_to_float(uint16_t h) {
    const uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    uint32_t bits;
    if (e == 0x1Fu)      bits = sign | 0x7F800000u | (m << 13);
    else if (e != 0)     bits = sign | ((e + 112u) << 23) | (m << 13);
    else if (m == 0)     bits = sign;
    else {
        uint32_t ee = 113u;
        while ((m & 0x400u) == 0) { m <<= 1; --ee; }
        bits = sign | (ee << 23) | ((m & 0x3FFu) << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline size_t type_size(int type) {
    switch (type) {
        case SDH_TYPE_F32: case SDH_TYPE_I32: return 4;
        case SDH_TYPE_F16: case SDH_TYPE_I16: return 2;
        default: return 0;
    }
}

static inline const uint8_t * elem_ptr(const sdh_tensor & t, int64_t i0, int64_t i1, int64_t i2) {
    return (const uint8_t *) t.data + i0 * t.nb[0] + i1 * t.nb[1] + i2 * t.nb[2];
}

// copy a (possibly strided) tensor into a contiguous byte vector, keeping its element type
static std::vector<uint8_t> pack_raw(const sdh_tensor & t) {
    const size_t es = type_size(t.type);
    const int64_t n0 = t.ne[0], n1 = t.ne[1], n2 = t.ne[2];
    std::vector<uint8_t> out((size_t) (n0 * n1 * n2) * es);
    uint8_t * dst = out.data();
    const bool row_contig = (size_t) t.nb[0] == es;
    for (int64_t i2 = 0; i2 < n2; ++i2)
        for (int64_t i1 = 0; i1 < n1; ++i1) {
            if (row_contig) {
                memcpy(dst, elem_ptr(t, 0, i1, i2), (size_t) n0 * es);
                dst += (size_t) n0 * es;
            } else {
                for (int64_t i0 = 0; i0 < n0; ++i0) { memcpy(dst, elem_ptr(t, i0, i1, i2), es); dst += es; }
            }
        }
    return out;
}

static std::vector<float> pack_f32(const sdh_tensor & t) {     // F16/F32 -> F32, exact
    const int64_t n0 = t.ne[0], n1 = t.ne[1];
    std::vector<float> out((size_t) (n0 * n1));
    for (int64_t i1 = 0; i1 < n1; ++i1)
        for (int64_t i0 = 0; i0 < n0; ++i0) {
            const uint8_t * p = elem_ptr(t, i0, i1, 0);
            float f;
            if (t.type == SDH_TYPE_F16) { uint16_t h; memcpy(&h, p, 2); f = half_to_float(h); }
            else memcpy(&f, p, 4);
            out[(size_t) (i1 * n0 + i0)] = f;
        }
    return out;
}

static std::vector<int32_t> pack_i32(const sdh_tensor & t) {   // I16/I32 -> I32
    const int64_t n0 = t.ne[0], n1 = t.ne[1];
    std::vector<int32_t> out((size_t) (n0 * n1));
    for (int64_t i1 = 0; i1 < n1; ++i1)
        for (int64_t i0 = 0; i0 < n0; ++i0) {
            const uint8_t * p = elem_ptr(t, i0, i1, 0);
            int32_t v;
            if (t.type == SDH_TYPE_I16) { int16_t s; memcpy(&s, p, 2); v = s; }
            else memcpy(&v, p, 4);
            out[(size_t) (i1 * n0 + i0)] = v;
        }
    return out;
}

static void host_fwht(float * a, int64_t n) {      // reference stage order (:484-493)
    for (int64_t len = 1; len < n; len <<= 1)
        for (int64_t i = 0; i < n; i += (len << 1))
            for (int64_t j = 0; j < len; ++j) {
                const float u = a[i + j], v = a[i + j + len];
                a[i + j] = u + v;
                a[i + j + len] = u - v;
            }
}

template <typename T>
static int upload(const std::vector<T> & h, T ** d) {
    *d = nullptr;
    if (h.empty()) return SDH_OK;
    SDH_HIP_CHECK(hipMalloc((void **) d, h.size() * sizeof(T)));
    SDH_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return SDH_OK;
}

static void free_weight(DeviceWeight * w) {
    if (!w) return;
    (void) hipFree(w->idx); (void) hipFree(w->val); (void) hipFree(w->rs);
    (void) hipFree(w->d1); (void) hipFree(w->d2); (void) hipFree(w->d3); (void) hipFree(w->p1); (void) hipFree(w->p2);
    (void) hipFree(w->pack_dev); (void) hipFree(w->handoff);
    (void) hipFree(w->G); (void) hipFree(w->inv); (void) hipFree(w->nf_ring); (void) hipFree(w->ws);
    fold_free(*w);
    direct_free(*w);
    delete w;
}

static int ensure_ws(DeviceWeight * w, size_t bytes) {
    if (w->ws_bytes >= bytes) return SDH_OK;
    if (w->ws) { SDH_HIP_CHECK(hipFree(w->ws)); w->ws = nullptr; w->ws_bytes = 0; }
    SDH_HIP_CHECK(hipMalloc((void **) &w->ws, bytes));
    w->ws_bytes = bytes;
    return SDH_OK;
}

// drain recorded event pairs into the running totals (waits for the last one)
static int prof_flush(LibState & s) {
    for (int i = 0; i < s.prof_used; ++i) {
        SDH_HIP_CHECK(hipEventSynchronize(s.prof_ev[2 * i + 1]));
        float ms = 0.0f;
        SDH_HIP_CHECK(hipEventElapsedTime(&ms, s.prof_ev[2 * i], s.prof_ev[2 * i + 1]));
        s.prof_ms += (double) ms;
        s.prof_n += 1;
    }
    s.prof_used = 0;
    return SDH_OK;
}

} // namespace sdh

using namespace sdh;

// ================================================================================================
extern "C" {

int sdh_abi_version(void) { return SDH_ABI_VERSION; }

const char * sdh_last_error(void) { return g_last_error.c_str(); }

int sdh_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    LibState & s = lib();
    if (s.inited && s.device == device) return SDH_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible");
        return SDH_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (%d visible)", device, n);
        return SDH_ERR_INVALID;
    }
    SDH_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SDH_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (!s.stream) SDH_HIP_CHECK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    if (device < MAX_DEVICES && !g_scratch_ptr[device]) {       // this device's copy of `scratch` (device is current)
        void * ptr = nullptr;
        SDH_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        g_scratch_ptr[device] = ptr;
    }
    s.num_cu = prop.multiProcessorCount;
    s.device = device;
    s.inited = true;
    return SDH_OK;
}

void * sdh_library_stream(void) { return lib().inited ? (void *) lib().stream : nullptr; }

int sdh_device_synchronize(void) {
    if (!lib().inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    SDH_HIP_CHECK(hipDeviceSynchronize());
    return SDH_OK;
}

// Validation follows try_add_seeddelta (src/llama-context.cpp:145-268).
static int sdh_weight_create_impl(const sdh_weight_desc * desc, sdh_weight ** out) {
    if (!desc || !out) { set_error("null argument"); return SDH_ERR_INVALID; }
    *out = nullptr;
    if (!lib().inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    const int64_t n_in = desc->n_in, n_out = desc->n_out;
    if (n_in <= 0 || n_out <= 0 || n_in > INT32_MAX / 2 || n_out > INT32_MAX / 2) {
        set_error("invalid dims n_in=%lld n_out=%lld", (long long) n_in, (long long) n_out);
        return SDH_ERR_INVALID;
    }

    sdh_tensor d_idx = desc->d_idx, d_val = desc->d_val, b_idx = desc->b_idx, b_val = desc->b_val;
    sdh_tensor rs = desc->row_scale;
    sdh_tensor d1 = desc->base_d1, d2 = desc->base_d2, d3 = desc->base_d3;
    sdh_tensor p1 = desc->base_perm1, p2 = desc->base_perm2;
    auto drop = [](sdh_tensor & t) { t.data = nullptr; };
    auto is_idx = [](int t) { return t == SDH_TYPE_I16 || t == SDH_TYPE_I32; };
    auto is_flt = [](int t) { return t == SDH_TYPE_F16 || t == SDH_TYPE_F32; };

    bool have_coo = present(d_idx) && present(d_val);
    bool have_block = present(b_idx) && present(b_val);
    if (have_coo) {                                                      // :162-177
        if (d_idx.n_dims != 2 || d_val.n_dims != 2 || !is_idx(d_idx.type) || !is_flt(d_val.type) ||
            d_idx.ne[0] != d_val.ne[0] || d_idx.ne[1] != d_val.ne[1] || d_idx.ne[1] != n_out) {
            have_coo = false;
        }
    }
    if (have_block) {                                                    // :179-194
        if (b_idx.n_dims != 2 || b_val.n_dims != 3 || !is_idx(b_idx.type) || !is_flt(b_val.type) ||
            b_val.ne[1] != b_idx.ne[0] || b_val.ne[2] != b_idx.ne[1] || b_idx.ne[1] != n_out) {
            have_block = false;
        }
    }
    if (!have_coo && !have_block) {
        set_error("payload has no valid COO or block residual");
        return SDH_ERR_INVALID;
    }
    if (present(rs)) {                                                   // :206-213
        if (rs.n_dims != 1 || !is_flt(rs.type) || rs.ne[0] != n_out) drop(rs);
    }
    const bool any_base = present(d1) || present(d2) || present(d3) || present(p1) || present(p2);
    if (any_base) {                                                      // :218-268
        bool ok = present(d1) && d1.n_dims == 2 && is_flt(d1.type);
        if (ok) {
            const int64_t L = d1.ne[0], B = d1.ne[1];
            const bool pow2 = L > 0 && (L & (L - 1)) == 0;
            const int64_t need_L = n_in < n_out ? n_in : n_out;
            const int64_t need_B = ((n_in > n_out ? n_in : n_out) + L - 1) / L;
            if (!pow2 || L < need_L || B < need_B) ok = false;
        }
        if (!ok) { drop(d1); drop(d2); drop(d3); drop(p1); drop(p2); }
        auto same = [&](sdh_tensor & t, bool perm) {
            if (!present(t) || !present(d1)) return;
            if (t.n_dims != 2 || t.ne[0] != d1.ne[0] || t.ne[1] != d1.ne[1] ||
                (perm ? !is_idx(t.type) : !is_flt(t.type))) drop(t);
        };
        same(d2, false); same(d3, false); same(p1, true); same(p2, true);
    }
    const bool have_base = present(d1);

    DeviceWeight * w = new DeviceWeight();
    w->device = lib().device;
    w->n_in = n_in; w->n_out = n_out;
    w->tall = n_out >= n_in;                                             // src/llama-seeddelta.cpp:1073
    // dispatch priority of src/llama-graph.cpp:681-703
    if (have_block) w->kind = have_base ? SDH_OP_BASE_BLOCK : SDH_OP_BLOCK;
    else            w->kind = have_base ? SDH_OP_BASE_COO : SDH_OP_COO;

    int rc = SDH_OK;
    auto fail = [&](int code) { free_weight(w); return code; };

    if (have_block) {
        w->nb = b_idx.ne[0]; w->block = b_val.ne[0];
        w->idx_type = b_idx.type; w->val_type = b_val.type;
        if (w->nb * w->block > INT32_MAX / 4) { set_error("row too long"); return fail(SDH_ERR_UNSUPPORTED); }
        std::vector<uint8_t> hi = pack_raw(b_idx), hv = pack_raw(b_val);
        w->payload_bytes += (int64_t) hi.size() + (int64_t) hv.size();
        if ((rc = upload(hi, (uint8_t **) &w->idx)) != SDH_OK) return fail(rc);
        if ((rc = upload(hv, (uint8_t **) &w->val)) != SDH_OK) return fail(rc);
    } else {
        w->K = d_idx.ne[0];
        w->idx_type = d_idx.type; w->val_type = d_val.type;
        if (w->K > INT32_MAX / 4) { set_error("row too long"); return fail(SDH_ERR_UNSUPPORTED); }
        std::vector<uint8_t> hi = pack_raw(d_idx), hv = pack_raw(d_val);
        w->payload_bytes += (int64_t) hi.size() + (int64_t) hv.size();
        if ((rc = upload(hi, (uint8_t **) &w->idx)) != SDH_OK) return fail(rc);
        if ((rc = upload(hv, (uint8_t **) &w->val)) != SDH_OK) return fail(rc);
    }
    if (present(rs)) {
        sdh_tensor r2 = rs; r2.ne[1] = 1;
        std::vector<float> h = pack_f32(r2);
        w->payload_bytes += (int64_t) n_out * (int64_t) type_size(rs.type);
        if ((rc = upload(h, &w->rs)) != SDH_OK) return fail(rc);
    }
    if (have_base) {
        const int64_t L = d1.ne[0], B = d1.ne[1];
        w->L = L; w->B = B;
        w->logL = 0; while ((1LL << w->logL) < L) ++w->logL;
        const size_t n = (size_t) (L * B);
        std::vector<float> h1 = pack_f32(d1), h2, h3;
        w->payload_bytes += (int64_t) n * (int64_t) type_size(d1.type);
        if ((rc = upload(h1, &w->d1)) != SDH_OK) return fail(rc);
        if (present(d2)) {
            h2 = pack_f32(d2);
            w->payload_bytes += (int64_t) n * (int64_t) type_size(d2.type);
            if ((rc = upload(h2, &w->d2)) != SDH_OK) return fail(rc);
        }
        if (present(d3)) {
            h3 = pack_f32(d3);
            w->payload_bytes += (int64_t) n * (int64_t) type_size(d3.type);
            if ((rc = upload(h3, &w->d3)) != SDH_OK) return fail(rc);
        }
        std::vector<int32_t> hp1, hp2;
        auto clamp_perm = [&](std::vector<int32_t> & p) {          // out of range -> 0 (:524-527)
            for (auto & s : p) if (s < 0 || s >= (int32_t) L) s = 0;
        };
        if (present(p1)) {
            hp1 = pack_i32(p1); clamp_perm(hp1);
            w->payload_bytes += (int64_t) n * (int64_t) type_size(p1.type);
            if ((rc = upload(hp1, &w->p1)) != SDH_OK) return fail(rc);
        }
        if (present(p2)) {
            hp2 = pack_i32(p2); clamp_perm(hp2);
            w->payload_bytes += (int64_t) n * (int64_t) type_size(p2.type);
            if ((rc = upload(hp2, &w->p2)) != SDH_OK) return fail(rc);
        }
        // closed-form tables: G_b = FWHT(d2_b) (d2 absent => all-ones), inv_b = perm1_b^-1
        std::vector<float> G(n);
        std::vector<int32_t> inv(n, -1);
        bool bijective = true;
        for (int64_t b = 0; b < B; ++b) {
            for (int64_t i = 0; i < L; ++i) G[(size_t) (b * L + i)] = h2.empty() ? 1.0f : h2[(size_t) (b * L + i)];
            host_fwht(G.data() + b * L, L);
            for (int64_t i = 0; i < L; ++i) {
                const int32_t src = hp1.empty() ? (int32_t) i : hp1[(size_t) (b * L + i)];
                if (inv[(size_t) (b * L + src)] != -1) bijective = false;
                inv[(size_t) (b * L + src)] = (int32_t) i;
            }
        }
        if (!bijective) for (auto & v : inv) if (v < 0) v = 0;
        w->w0_closed_form = bijective && present(d2) && !present(p2);
        if ((rc = upload(G, &w->G)) != SDH_OK) return fail(rc);
        if ((rc = upload(inv, &w->inv)) != SDH_OK) return fail(rc);

        // packed hot-path slab: per base block [perm1][d2][d1][d3][perm2], 16-bit when the GGUF
        // stores F16 / I16 (what the builder writes), else widened to 32-bit. d1 / d3 that are all
        // ones (what the builder writes) are elided.
        auto all_ones = [](const std::vector<float> & v) {
            for (float f : v) if (f != 1.0f) return false;
            return true;
        };
        const bool keep_d1 = !all_ones(h1);
        const bool keep_d3 = present(d3) && !all_ones(h3);
        bool narrow = L <= 32768 && (!keep_d1 || d1.type == SDH_TYPE_F16) &&
                      (!present(d2) || d2.type == SDH_TYPE_F16) && (!keep_d3 || d3.type == SDH_TYPE_F16) &&
                      (!present(p1) || p1.type == SDH_TYPE_I16) && (!present(p2) || p2.type == SDH_TYPE_I16);
        const size_t es = narrow ? 2 : 4;
        BasePackDesc & pd = w->pack;
        pd.wide32 = narrow ? 0 : 1;
        int off = 0;
        auto region = [&](bool on) { int o = -1; if (on) { o = off; off += (int) (((size_t) L * es + 15) / 16 * 16); } return o; };
        pd.off_p1 = region(present(p1));
        pd.off_d2 = region(present(d2));
        pd.off_d1 = region(keep_d1);
        pd.off_d3 = region(keep_d3);
        pd.off_p2 = region(present(p2));
        pd.blk_stride = off;
        if (off > 0) {
            // behind the parameters: room for the direct one-token launch's tables (sdh_direct.hip: B L floats and B L 16-bit
            // offsets, filled on the device below): no allocation of their own.
            const size_t par_bytes = ((size_t) off * (size_t) B + 255) / 256 * 256;
            const bool tab_room = L >= 256 && L <= 16384 && (L & (L - 1)) == 0;
            std::vector<uint8_t> slab(par_bytes + (tab_room ? (size_t) L * (size_t) B * 6 : 0), 0);
            auto put_f = [&](int o, const sdh_tensor & src, const std::vector<float> & f32) {
                if (o < 0) return;
                for (int64_t b = 0; b < B; ++b) {
                    uint8_t * dst = slab.data() + (size_t) b * off + o;
                    if (narrow) { for (int64_t i = 0; i < L; ++i) memcpy(dst + 2 * i, elem_ptr(src, i, b, 0), 2); }
                    else        { memcpy(dst, f32.data() + b * L, (size_t) L * 4); }
                }
            };
            auto put_i = [&](int o, const std::vector<int32_t> & i32) {     // already clamped to [0, L)
                if (o < 0) return;
                for (int64_t b = 0; b < B; ++b) {
                    uint8_t * dst = slab.data() + (size_t) b * off + o;
                    for (int64_t i = 0; i < L; ++i) {
                        const int32_t v = i32[(size_t) (b * L + i)];
                        if (narrow) { const uint16_t u = (uint16_t) v; memcpy(dst + 2 * i, &u, 2); }
                        else memcpy(dst + 4 * i, &v, 4);
                    }
                }
            };
            put_i(pd.off_p1, hp1);
            put_f(pd.off_d2, d2, h2);
            put_f(pd.off_d1, d1, h1);
            put_f(pd.off_d3, d3, h3);
            put_i(pd.off_p2, hp2);
            if ((rc = upload(slab, &w->pack_dev)) != SDH_OK) return fail(rc);
            pd.pack = w->pack_dev;
            if (tab_room) w->dir_space = w->pack_dev + par_bytes;
        }
        // hand-off buffer of the fused decode kernel: [B][L] granules {value, epoch} (tag 0 = never
        // written; epochs start at 1)
        {
            const size_t gb = (size_t) MAX_FUSED_TOKENS * (size_t) L * (size_t) B * sizeof(unsigned long long);
            hipError_t e = hipMalloc((void **) &w->handoff, gb);
            if (e == hipSuccess) e = hipMemset(w->handoff, 0, gb);
            if (e != hipSuccess) { set_error("hand-off alloc failed: %s", hipGetErrorString(e)); return fail(SDH_ERR_HIP); }
        }
    }
    if (w->kind == SDH_OP_BASE_COO || w->kind == SDH_OP_BASE_BLOCK) {   // tables of the direct one-token launch (payloads that qualify)
        LibState & ls = lib();
        int rc = direct_build_tables(*w, ls.stream);
        if (rc == SDH_OK && hipStreamSynchronize(ls.stream) != hipSuccess) { set_error("direct decode: table kernels failed"); rc = SDH_ERR_HIP; }
        if (rc != SDH_OK) return fail(rc);
    }
    {
        hipError_t e = hipMalloc((void **) &w->nf_ring, DeviceWeight::NF_SLOTS * sizeof(int));
        if (e == hipSuccess) e = hipMemset(w->nf_ring, 0, DeviceWeight::NF_SLOTS * sizeof(int));
        if (e != hipSuccess) { set_error("nf ring alloc failed: %s", hipGetErrorString(e)); return fail(SDH_ERR_HIP); }
    }
    *out = reinterpret_cast<sdh_weight *>(w);
    return SDH_OK;
}

void sdh_weight_destroy(sdh_weight * h) {
    if (!h) return;
    (void) hipDeviceSynchronize();
    free_weight(reinterpret_cast<DeviceWeight *>(h));
}

int sdh_weight_op_kind(const sdh_weight * h) {
    if (!h) { set_error("null weight"); return SDH_ERR_INVALID; }
    return reinterpret_cast<const DeviceWeight *>(h)->kind;
}

int64_t sdh_weight_payload_bytes(const sdh_weight * h) {
    if (!h) { set_error("null weight"); return SDH_ERR_INVALID; }
    return reinterpret_cast<const DeviceWeight *>(h)->payload_bytes;
}

// host -> device of an [n_tokens][n] F32 activation with row stride ld (elements): through the pinned
// bounce buffer when it fits, else a pageable 2-D copy
static int host_to_device(LibState & L, float * dst, const float * src, int64_t ld, int64_t n, int64_t n_tokens, hipStream_t s) {
    const size_t bytes = (size_t) n_tokens * (size_t) n * 4;
    if (bytes <= LibState::PIN_BYTES) {
        if (!L.pin_x) SDH_HIP_CHECK(hipHostMalloc((void **) &L.pin_x, LibState::PIN_BYTES, hipHostMallocDefault));
        for (int64_t t = 0; t < n_tokens; ++t) memcpy(L.pin_x + t * n, src + t * ld, (size_t) n * 4);
        SDH_HIP_CHECK(hipMemcpyAsync(dst, L.pin_x, bytes, hipMemcpyHostToDevice, s));
        return SDH_OK;
    }
    SDH_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t) n * 4, src, (size_t) ld * 4, (size_t) n * 4, (size_t) n_tokens, hipMemcpyHostToDevice, s));
    return SDH_OK;
}
// device -> host, then wait for the stream
static int device_to_host_sync(LibState & L, float * dst, int64_t ld, const float * src, int64_t n, int64_t n_tokens, hipStream_t s) {
    const size_t bytes = (size_t) n_tokens * (size_t) n * 4;
    if (bytes <= LibState::PIN_BYTES) {
        if (!L.pin_y) SDH_HIP_CHECK(hipHostMalloc((void **) &L.pin_y, LibState::PIN_BYTES, hipHostMallocDefault));
        SDH_HIP_CHECK(hipMemcpyAsync(L.pin_y, src, bytes, hipMemcpyDeviceToHost, s));
        SDH_HIP_CHECK(hipStreamSynchronize(s));
        for (int64_t t = 0; t < n_tokens; ++t) memcpy(dst + t * ld, L.pin_y + t * n, (size_t) n * 4);
        return SDH_OK;
    }
    SDH_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t) ld * 4, src, (size_t) n * 4, (size_t) n * 4, (size_t) n_tokens, hipMemcpyDeviceToHost, s));
    SDH_HIP_CHECK(hipStreamSynchronize(s));
    return SDH_OK;
}

// per-call non-finite counter: ring slot, the second half of the ring is re-zeroed while the
// first half is in use (one small memset every NF_SLOTS/2 calls, none on the other calls)
static int nf_slot(DeviceWeight * w, hipStream_t s, int * slot_out) {
    constexpr int NS = DeviceWeight::NF_SLOTS;
    const int slot = (int) (w->seq % NS);
    if (w->seq > 0 && slot % (NS / 2) == 0) {
        SDH_HIP_CHECK(hipMemsetAsync(w->nf_ring + slot, 0, (NS / 2) * sizeof(int), s));
    }
    *slot_out = slot;
    return SDH_OK;
}

struct GluEpilogue { const float * gate; int64_t ldg; int act; const uint16_t * tab; };

// From how many tokens on sdh_forward takes the folded f16 contraction (sdh_w0fold.hip) by itself — just above the measured
// crossovers against the fastest fp32 form of each payload class on Mistral-7B tensors (tools/gpu/r03_fold_T.sh, us per call,
// fold / others: gate COO K=128 57.5 / 55.4 at T=128, 61 / 80 at 192; gate block16 63 / 63 at 192, 64 / 83 at 256; down COO
// 60 / 49 at 128, 66 / 80 at 192; down block16 68 / 64 at 256, 91 / 92 at 384; block 32 with K=1536 62 / 58 at 64, 64 / 86
// at 96). The folded call costs one round of workgroups: ~57 us up to 128 tokens, ~63 up to 256 (token tiles of 128), ~88 at 512.
// INT64_MAX: never by default (no closed-form base; wide block payloads in q order).
static int64_t fold_min_tokens(const DeviceWeight * w) {
    const bool has_base = w->kind == SDH_OP_BASE_COO || w->kind == SDH_OP_BASE_BLOCK;
    if (!has_base || !w->w0_closed_form || w->L < 128 || (w->L % 128) != 0) return INT64_MAX;
    const bool block = w->kind == SDH_OP_BASE_BLOCK;
    const int64_t row_entries = block ? w->nb * w->block : w->K;
    const bool perm = w0_fold_permuted(*w);
    if (row_entries >= 512) return perm ? 80 : 160;
    if (perm) return block ? (w->tall ? 224 : 320) : 160;
    // the q-order kernel (d1 / d3 present, or blocks much longer than their inputs — Gemma's 1152 / 2560 inputs in blocks of
    // 2048 / 4096) costs ~1.5 us per slab: Mistral tensors 111-127 us at T=512 against 143-189 (fp32; down block16: 125, a tie)
    // and ~100 against 63-106 at 256; a Gemma-3 4B gate (COO) 58 / 68 at 256 and 62 / 119 at 512; a Gemma-3 1B gate
    // (block16, K=64) 32 / 27 at 256 and 34 / 49 at 512, its down 43 / 35 at 512
    if (block) return w->tall ? 448 : INT64_MAX;
    return 320;
}
// ... and the gated FFN block (three tensors, activation in the up node's epilogue): us per Mistral layer, fold / others:
// COO 176 / 180 at T=192, 177 / 194 at 224; block16 177 / 173 at 192, 176 / 207 at 224 (tools/ffn_tokens.py)
constexpr int64_t FFN_FOLD_MIN_TOKENS = 224;
constexpr int FOLD_EXCLUDES = SDH_FWD_NO_MFMA | SDH_FWD_W0_MFMA | SDH_FWD_NO_SPMM | SDH_FWD_EXACT_ORDER;

static int sdh_forward_impl(sdh_weight * h, const float * x, int64_t ldx, int64_t n_tokens,
                float * y, int64_t ldy, int flags, void * stream,
                float * ext_ws = nullptr, size_t ext_ws_bytes = 0, const GluEpilogue * glu = nullptr) {
    // ext_ws: the caller-provided base-term workspace of sdh_forward_noalloc; with it this function makes no
    // allocating call (no workspace growth, no host-I/O staging, no fused launch: its first use allocates the
    // census counter and waits for the device)
    const bool noalloc = ext_ws != nullptr;
    if (!h || !x || !y) { set_error("null argument"); return SDH_ERR_INVALID; }
    if (!lib().inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    DeviceWeight * w = reinterpret_cast<DeviceWeight *>(h);
    if (n_tokens < 0 || ldx < w->n_in || ldy < w->n_out) {
        set_error("bad shape: T=%lld ldx=%lld (n_in=%lld) ldy=%lld (n_out=%lld)", (long long) n_tokens,
                  (long long) ldx, (long long) w->n_in, (long long) ldy, (long long) w->n_out);
        return SDH_ERR_INVALID;
    }
    if (n_tokens == 0) return SDH_OK;
    hipStream_t s = resolve_stream(stream, flags);
    const bool host_io = (flags & SDH_FWD_HOST_IO) != 0;
    const bool has_base = (w->kind == SDH_OP_BASE_COO || w->kind == SDH_OP_BASE_BLOCK);

    const float * dx = x;
    float * dy = y;
    int64_t dldx = ldx, dldy = ldy;
    if (host_io) {
        LibState & L = lib();
        const size_t xb = (size_t) n_tokens * (size_t) w->n_in * 4, yb = (size_t) n_tokens * (size_t) w->n_out * 4;
        if (L.hx_bytes < xb) {
            if (L.hx) SDH_HIP_CHECK(hipFree(L.hx));
            L.hx = nullptr; L.hx_bytes = 0;
            SDH_HIP_CHECK(hipMalloc((void **) &L.hx, xb)); L.hx_bytes = xb;
        }
        if (L.hy_bytes < yb) {
            if (L.hy) SDH_HIP_CHECK(hipFree(L.hy));
            L.hy = nullptr; L.hy_bytes = 0;
            SDH_HIP_CHECK(hipMalloc((void **) &L.hy, yb)); L.hy_bytes = yb;
        }
        { int rc = host_to_device(L, L.hx, x, ldx, w->n_in, n_tokens, s); if (rc != SDH_OK) return rc; }
        dx = L.hx; dy = L.hy; dldx = w->n_in; dldy = w->n_out;
    }

    int slot = 0;
    { int rc = nf_slot(w, s, &slot); if (rc != SDH_OK) return rc; }

    // token chunking bounds the base-term workspace of the two-kernel path
    int64_t chunk = n_tokens;
    size_t per_tok = has_base ? (size_t) w->B * (size_t) w->L * 4 : 0;
    // the tiled MFMA form of the base term (SDH_FWD_W0_MFMA): tall payloads write Y0 straight into y and need only
    // the f16 image of x as workspace; wide ones keep the [T][B][L] partial products as well. SDH_W0_MFMA_V1 (A/B):
    // the first form, one fragment per lane from global memory. Not on the no-alloc entry point (its scratch is 8 MB).
    static const bool mfma_v1 = getenv("SDH_W0_MFMA_V1") != nullptr;
    const bool mfma_tiled = has_base && (flags & SDH_FWD_W0_MFMA) && !(flags & SDH_FWD_EXACT_ORDER) && !noalloc && !mfma_v1 &&
                            w->w0_closed_form && w->L >= 128 && n_tokens > MAX_FUSED_TOKENS;
    size_t x16_per_tok = 0;
    if (mfma_tiled) {
        x16_per_tok = ((size_t) (w->n_in + 63) / 64 * 64 + 64) * 2;
        if (w->tall) per_tok = 0;
    }
    if (has_base) {
        const size_t cap = noalloc ? ext_ws_bytes : (size_t) 512 << 20;
        if (noalloc && per_tok > cap) {
            set_error("sdh_forward_noalloc: one token's base term needs B*L*4 = %zu bytes, the static scratch holds %zu",
                      per_tok, cap);
            return SDH_ERR_TOO_LARGE;
        }
        if ((per_tok + x16_per_tok) * (size_t) chunk > cap) chunk = (int64_t) (cap / (per_tok + x16_per_tok));
        if (mfma_tiled && chunk < n_tokens) chunk = std::max<int64_t>(256, chunk / 256 * 256);      // whole token tiles
        if (chunk < 1) chunk = 1;
    }
    LibState & ls = lib();
    // Many tokens, one dense contraction per tensor: the residual folded into the W0 tile in LDS (SDH_FWD_MFMA_FOLD;
    // f16 operands, fp32 accumulation). The call is complete after it: row scale, non-finite census, everything.
    // Taken on request, and by default from the payload class's crossover on (fold_min_tokens). SDH_FWD_NO_MFMA keeps the
    // fp32 forms (bit-exact SpMM / rows kernels), as do SDH_FWD_NO_SPMM and SDH_FWD_EXACT_ORDER.
    const bool fold_default = n_tokens >= fold_min_tokens(w) && !(flags & (FOLD_EXCLUDES | SDH_FWD_INTERNAL_ROWS));
    if (((flags & SDH_FWD_MFMA_FOLD) || fold_default) && has_base && !(flags & SDH_FWD_EXACT_ORDER) && !noalloc &&
        w->w0_closed_form && w->L >= 128 && n_tokens > MAX_FUSED_TOKENS) {
        const int64_t fchunk = std::min<int64_t>(n_tokens, 2048);
        int rc = ensure_ws(w, w0_fold_ws_bytes(*w, fchunk));
        if (rc != SDH_OK) return rc;
        int pair = -1;
        if (ls.prof_on) {
            if (ls.prof_used == LibState::PROF_PAIRS) { rc = prof_flush(ls); if (rc != SDH_OK) return rc; }
            pair = ls.prof_used;
            SDH_HIP_CHECK(hipEventRecord(ls.prof_ev[2 * pair], s));
        }
        for (int64_t t0 = 0; t0 < n_tokens && rc == SDH_OK; t0 += fchunk) {
            const int64_t tc = std::min<int64_t>(fchunk, n_tokens - t0);
            rc = launch_w0_fold(*w, dx + t0 * dldx, dldx, tc, w->ws, dy + t0 * dldy, dldy, w->nf_ring + slot,
                                (flags & SDH_FWD_ZERO_NONFINITE) != 0, s,
                                glu ? glu->gate + t0 * glu->ldg : nullptr, glu ? glu->ldg : 0, glu ? glu->act : 0, glu ? glu->tab : nullptr);
        }
        if (rc < 0) return rc;
        if (rc == SDH_OK) {
            if (pair >= 0) { SDH_HIP_CHECK(hipEventRecord(ls.prof_ev[2 * pair + 1], s)); ls.prof_used = pair + 1; }
            w->seq++;
            w->last_stream = s;
            ls.span_forwards++;
            if (host_io) return device_to_host_sync(ls, y, ldy, dy, w->n_out, n_tokens, s);
            return SDH_OK;
        }
        // rc == 1: the payload does not fit the scheme (only possible before anything was launched)
    }
    for (int64_t t0 = 0; t0 < n_tokens; t0 += chunk) {
        const int64_t tc = (n_tokens - t0 < chunk) ? (n_tokens - t0) : chunk;
        const float * xc = dx + t0 * dldx;
        DeltaLaunch a{};
        a.x = xc; a.ldx = dldx; a.T = tc;
        a.y = dy + t0 * dldy; a.ldy = dldy;
        a.v = nullptr;
        a.nf_counter = w->nf_ring + slot;
        a.zero_nonfinite = (flags & SDH_FWD_ZERO_NONFINITE) != 0;
        a.exact_order = (flags & SDH_FWD_EXACT_ORDER) != 0;
        if (glu) { a.glu_gate = glu->gate + t0 * glu->ldg; a.glu_ldg = glu->ldg; a.glu_act = glu->act; a.glu_tab = glu->tab; }
        if (tc == n_tokens) a.x4 = w->x4_hint;

        auto prof_begin = [&](int & pair) -> int {
            pair = -1;
            if (!ls.prof_on) return SDH_OK;
            if (ls.prof_used == LibState::PROF_PAIRS) { int rcf = prof_flush(ls); if (rcf != SDH_OK) return rcf; }
            pair = ls.prof_used;
            SDH_HIP_CHECK(hipEventRecord(ls.prof_ev[2 * pair], s));
            return SDH_OK;
        };
        auto prof_end = [&](int pair) -> int {
            if (pair < 0) return SDH_OK;
            SDH_HIP_CHECK(hipEventRecord(ls.prof_ev[2 * pair + 1], s));
            ls.prof_used = pair + 1;
            return SDH_OK;
        };

        // decode with a base, 1..4 tokens: one fused launch when the payload qualifies
        if (has_base && tc >= 1 && tc <= MAX_FUSED_TOKENS && !(flags & SDH_FWD_NO_FUSE) && !noalloc) {
            int pair, rc = prof_begin(pair);
            if (rc != SDH_OK) return rc;
            rc = (tc == 1 && !(flags & SDH_FWD_NO_DIRECT)) ? launch_decode_direct(*w, a, s) : 1;
            if (tc == 1) ls.last_decode_kind = rc == SDH_OK ? 1 : 3;
            if (rc == 1) {
                rc = launch_decode_fused(*w, a, s);
                if (tc == 1 && rc == SDH_OK) ls.last_decode_kind = 2;
            }
            if (rc < 0) return rc;
            if (rc == SDH_OK) {
                rc = prof_end(pair);
                if (rc != SDH_OK) return rc;
                continue;
            }
            // rc == 1: not applicable; the opening event is simply re-recorded below
        }
        if (mfma_tiled) {
            const size_t part_bytes = per_tok * (size_t) chunk;                 // wide: [chunk][B][L] partial products
            int rc = ensure_ws(w, part_bytes + w0_mfma_x16_bytes(*w, chunk));
            if (rc != SDH_OK) return rc;
            _Float16 * x16 = (_Float16 *) ((unsigned char *) w->ws + part_bytes);
            if (w->tall) { rc = launch_w0_mfma_tiled(*w, xc, dldx, tc, x16, a.y, a.ldy, s); a.v = a.y; a.v_ld = a.ldy; }
            else         { rc = launch_w0_mfma_tiled(*w, xc, dldx, tc, x16, w->ws, (int64_t) w->B * w->L, s); a.v = w->ws; }
            if (rc != SDH_OK) { if (rc == 1) set_error("w0 mfma: payload does not fit the tile scheme"); return rc == 1 ? SDH_ERR_UNSUPPORTED : rc; }
        } else if (has_base) {
            int rc = noalloc ? SDH_OK : ensure_ws(w, per_tok * (size_t) chunk);
            if (rc != SDH_OK) return rc;
            float * const ws = noalloc ? ext_ws : w->ws;
            rc = 1;
            if ((flags & SDH_FWD_W0_MFMA) && !a.exact_order) {
                rc = launch_w0_mfma(*w, xc, dldx, tc, ws, s);
                if (rc == SDH_OK && !w->tall) a.v_blocks = 1;
            }
            if (rc == 1 && w->base_ready && !noalloc && tc == n_tokens) rc = SDH_OK;   // the FFN path's gate|up base launch wrote ws
            if (rc == 1) rc = launch_base_apply(*w, xc, dldx, tc, ws, s);
            if (rc != SDH_OK) return rc;
            a.v = ws;
        }
        int pair, rc = prof_begin(pair);
        if (rc != SDH_OK) return rc;
        // many tokens: token-tiled sparse x dense product (bit-exact order); few: per-token row kernels
        static const int64_t spmm_min_t = [] { const char * e = getenv("SDH_SPMM_MIN_T"); return e ? (int64_t) atoll(e) : (int64_t) 6; }();   // Mistral FFN layer, us: T=6 62.8 per-token vs 59 tiled, T=5 equal
        rc = 1;
        // the choice follows the CALL's token count, not the chunk's: a short last chunk must not switch a call to
        // the per-token kernels (same values within tolerance, but not the same bits as its other chunks)
        static const int64_t rows_max_t = [] { const char * e = getenv("SDH_FFN_ROWS_MAX_T"); return e ? (int64_t) atoll(e) : (int64_t) 32; }();
        // COO: the LDS rows kernel beats the tiled path at every token count measured (Mistral layer, us: T=64 78 against 86,
        // T=256 247 / 290, T=512 482 / 548, T=1024 864 / 953); block payloads: up to 32 tokens (T=40 69 against 63)
        const bool coo_kind = w->kind == SDH_OP_COO || w->kind == SDH_OP_BASE_COO;
        // block payloads whose input is so long that only two tokens fit the LDS (Mistral down): the rows kernel as well
        // (T=512 ubatch: 0.086 against 0.126 ms tiled), where the eight-token tile of gate / up favours the tiled path
        const bool two_token_tile = (size_t) w->n_in * 16 > (size_t) 152 * 1024;
        const bool prefer_rows = (flags & SDH_FWD_INTERNAL_ROWS) && (coo_kind || two_token_tile || n_tokens <= rows_max_t);
        if (n_tokens >= spmm_min_t && !a.exact_order && !(flags & SDH_FWD_NO_SPMM) && !prefer_rows) rc = launch_delta_spmm(*w, a, s);
        if (rc == 1) rc = launch_delta(*w, a, s);
        if (rc != SDH_OK) return rc;
        rc = prof_end(pair);
        if (rc != SDH_OK) return rc;
    }
    w->seq++;
    w->last_stream = s;
    ls.span_forwards++;

    if (host_io) {
        int rc = device_to_host_sync(ls, y, ldy, dy, w->n_out, n_tokens, s);
        if (rc != SDH_OK) return rc;
    }
    return SDH_OK;
}

// ---- gated FFN block ----------------------------------------------------------------------------------
// f16 GELU table exactly as ggml builds it (ggml/src/ggml-cpu/ggml-cpu.c:3660-3667 with
// ggml_gelu_f32, vec.h:976): entry i = f16( 0.5*x*(1 + tanhf(sqrt(2/pi)*x*(1 + 0.044715*x*x))) ),
// x = f32(f16 bits i), evaluated on the host in float with this machine's libm, as the reference does.
static int ensure_gelu_table(LibState & ls) {
    if (ls.gelu_tab) return SDH_OK;
    std::vector<uint16_t> tab(1 << 16);
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    for (uint32_t i = 0; i < (1u << 16); ++i) {
        const float f = half_to_float((uint16_t) i);
        const float v = 0.5f * f * (1.0f + tanhf(SQRT_2_OVER_PI * f * (1.0f + GELU_COEF_A * f * f)));
        tab[i] = float_to_half(v);
    }
    SDH_HIP_CHECK(hipMalloc((void **) &ls.gelu_tab, tab.size() * 2));
    SDH_HIP_CHECK(hipMemcpy(ls.gelu_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    return SDH_OK;
}

int sdh_glu(int act, const float * g, const float * u, float * y, int64_t n, void * stream) {
    LibState & ls = lib();
    if (!ls.inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    if (!g || !u || !y || n < 0 || (act != SDH_ACT_SILU && act != SDH_ACT_GELU)) { set_error("sdh_glu: bad argument"); return SDH_ERR_INVALID; }
    if (act == SDH_ACT_GELU) { int rc = ensure_gelu_table(ls); if (rc != SDH_OK) return rc; }
    return launch_glu(act, ls.gelu_tab, g, u, y, n, (hipStream_t) stream);
}

// FFN workspace: sections of `floats` each, 16-byte aligned, grown on demand
static int ffn_workspace(LibState & ls, hipStream_t s, const size_t * floats, int n, float ** out) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (floats[i] + 3) & ~(size_t) 3;
    const size_t need = total * sizeof(float) + 64;
    if (ls.ffn_ws_bytes < need) {
        if (ls.ffn_ws) { SDH_HIP_CHECK(hipStreamSynchronize(s)); SDH_HIP_CHECK(hipFree(ls.ffn_ws)); }
        ls.ffn_ws = nullptr; ls.ffn_ws_bytes = 0;
        SDH_HIP_CHECK(hipMalloc((void **) &ls.ffn_ws, need)); ls.ffn_ws_bytes = need;
    }
    float * p = ls.ffn_ws;
    for (int i = 0; i < n; ++i) { out[i] = p; p += (floats[i] + 3) & ~(size_t) 3; }
    return SDH_OK;
}

static int ffn_check(sdh_weight * hg, sdh_weight * hu, int act, const void * x, const void * y) {
    if (!hg || !hu || !x || !y) { set_error("null argument"); return SDH_ERR_INVALID; }
    if (!lib().inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    if (act != SDH_ACT_SILU && act != SDH_ACT_GELU) { set_error("unknown activation %d", act); return SDH_ERR_INVALID; }
    const DeviceWeight * g = reinterpret_cast<const DeviceWeight *>(hg);
    const DeviceWeight * u = reinterpret_cast<const DeviceWeight *>(hu);
    if (u->n_in != g->n_in || u->n_out != g->n_out) {
        set_error("gate %lldx%lld and up %lldx%lld differ in shape", (long long) g->n_out, (long long) g->n_in,
                  (long long) u->n_out, (long long) u->n_in);
        return SDH_ERR_INVALID;
    }
    return SDH_OK;
}

// From FFN_FOLD_MIN_TOKENS on the block runs as folded contractions by itself, if every tensor has the permuted form (gate and
// up tall: the activation epilogue) and no flag asks for the fp32 forms
static bool ffn_fold_default(const DeviceWeight * g, const DeviceWeight * u, const DeviceWeight * d, int64_t T, int flags) {
    if (T < FFN_FOLD_MIN_TOKENS || (flags & (FOLD_EXCLUDES | SDH_FWD_NO_FUSE))) return false;
    auto ok = [](const DeviceWeight * w) { return fold_min_tokens(w) != INT64_MAX && w0_fold_permuted(*w); };
    return ok(g) && ok(u) && g->tall && u->tall && (!d || ok(d));
}

// h[t] = act(gate(x[t])) * up(x[t]) on device buffers; gbuf / ubuf: [T][n_ff] scratch for the node-by-node path
static int ffn_gate_up(sdh_weight * hg, sdh_weight * hu, int act, const float * dx, int64_t dldx, int64_t T,
                       float * hbuf, float * gbuf, float * ubuf, int flags, hipStream_t s, float * h4buf = nullptr) {
    LibState & ls = lib();
    DeviceWeight * g = reinterpret_cast<DeviceWeight *>(hg);
    DeviceWeight * u = reinterpret_cast<DeviceWeight *>(hu);
    const int64_t n_ff = g->n_out;
    // base_ready ("the workspace already holds this call's base terms") lives only inside this function: whatever
    // way it is left, both handles have it cleared, so an unrelated sdh_forward can never see a stale claim
    struct ClearBaseReady { DeviceWeight * a, * b; ~ClearBaseReady() { a->base_ready = b->base_ready = false; } } clear_ready{g, u};
    if (act == SDH_ACT_GELU) { int rc = ensure_gelu_table(ls); if (rc != SDH_OK) return rc; }
    ls.ffn_last_fused = 0;
    int rc = 1;
    if (T >= 1 && T <= MAX_FUSED_TOKENS && !(flags & (SDH_FWD_NO_FUSE | SDH_FWD_EXACT_ORDER))) {
        int sg = 0, su = 0;
        if ((rc = nf_slot(g, s, &sg)) != SDH_OK) return rc;
        if ((rc = nf_slot(u, s, &su)) != SDH_OK) return rc;
        rc = launch_decode_pair(*g, *u, act, ls.gelu_tab, dx, dldx, (int) T, hbuf, n_ff, h4buf, g->nf_ring + sg, u->nf_ring + su,
                                (flags & SDH_FWD_ZERO_NONFINITE) != 0, s);
        if (rc < 0) return rc;
        if (rc == SDH_OK) {
            g->seq++; u->seq++;
            g->last_stream = u->last_stream = s;
            ls.span_forwards += 2;
            ls.ffn_last_fused = 1;
            return SDH_OK;
        }
    }
    // node by node on the device
    // the gate node, then the up node with the activation in its epilogue: h = act(g) * up(x) is written by the up
    // launch itself (no separate activation pass over three [T][n_ff] buffers; ubuf is no longer used)
    (void) ubuf;
    // few tokens: the base terms of both tensors from ONE launch (2B blocks per token) into the tensors' workspaces
    const size_t per_tok = (size_t) g->B * (size_t) g->L * 4;
    const auto with_base = [](const DeviceWeight * w) { return w->kind == SDH_OP_BASE_COO || w->kind == SDH_OP_BASE_BLOCK; };
    // (SDH_FWD_MFMA_FOLD: both nodes as folded contractions instead, the activation in the up node's epilogue)
    if (T > MAX_FUSED_TOKENS && T <= 1024 && g != u && with_base(g) && with_base(u) && g->tall && u->tall &&
        !(flags & (SDH_FWD_EXACT_ORDER | SDH_FWD_W0_MFMA | SDH_FWD_MFMA_FOLD)) && per_tok * (size_t) T <= ((size_t) 64 << 20)) {
        if ((rc = ensure_ws(g, per_tok * (size_t) T)) != SDH_OK) return rc;
        if ((rc = ensure_ws(u, per_tok * (size_t) T)) != SDH_OK) return rc;
        rc = launch_base_apply_pair(*g, *u, dx, dldx, T, g->ws, u->ws, s);
        if (rc < 0) return rc;
        if (rc == SDH_OK) g->base_ready = u->base_ready = true;
        // ... and, for COO payloads, the rows of both tensors from one launch over one tile of x, writing h directly
        if (rc == SDH_OK && !(flags & (SDH_FWD_NO_FUSE | SDH_FWD_NO_SPMM))) {
            int sg = 0, su = 0;
            if ((rc = nf_slot(g, s, &sg)) != SDH_OK) return rc;
            if ((rc = nf_slot(u, s, &su)) != SDH_OK) return rc;
            DeltaLaunch ag{};
            ag.x = dx; ag.ldx = dldx; ag.T = T; ag.y = hbuf; ag.ldy = n_ff;
            ag.zero_nonfinite = (flags & SDH_FWD_ZERO_NONFINITE) != 0;
            DeltaLaunch au = ag;
            ag.v = g->ws; ag.nf_counter = g->nf_ring + sg;
            au.v = u->ws; au.nf_counter = u->nf_ring + su;
            rc = launch_delta_rows_pair(*g, *u, ag, au, act, ls.gelu_tab, hbuf, n_ff, s);
            if (rc < 0) { g->base_ready = u->base_ready = false; return rc; }
            if (rc == SDH_OK) {
                g->base_ready = u->base_ready = false;
                g->seq++; u->seq++;
                g->last_stream = u->last_stream = s;
                ls.span_forwards += 2;
                return SDH_OK;
            }
        }
    }
    // many tokens as folded contractions: gate and up in ONE two-pass launch that writes act(gate) * up (sdh_w0fold.hip)
    static const bool no_pair = getenv("SDH_FOLD_NO_PAIR") != nullptr;            // A/B switch
    if ((flags & SDH_FWD_MFMA_FOLD) && T > 256 && g != u && !(flags & SDH_FWD_EXACT_ORDER) && !no_pair) {
        const int64_t fchunk = std::min<int64_t>(T, 2048);
        if ((rc = ensure_ws(g, w0_fold_ws_bytes(*g, fchunk))) != SDH_OK) return rc;
        if ((rc = ensure_ws(u, w0_fold_ws_bytes(*u, fchunk))) != SDH_OK) return rc;
        int sg = 0, su = 0;
        if ((rc = nf_slot(g, s, &sg)) != SDH_OK) return rc;
        if ((rc = nf_slot(u, s, &su)) != SDH_OK) return rc;
        rc = SDH_OK;
        for (int64_t t0 = 0; t0 < T && rc == SDH_OK; t0 += fchunk) {
            const int64_t tc = std::min<int64_t>(fchunk, T - t0);
            rc = launch_w0_fold_pair(*g, *u, dx + t0 * dldx, dldx, tc, hbuf + t0 * n_ff, n_ff, g->nf_ring + sg, u->nf_ring + su,
                                     (flags & SDH_FWD_ZERO_NONFINITE) != 0, act, ls.gelu_tab, s);
        }
        if (rc < 0) return rc;
        if (rc == SDH_OK) {
            g->seq++; u->seq++;
            g->last_stream = u->last_stream = s;
            ls.span_forwards += 2;
            return SDH_OK;
        }
        // rc == 1: the pair does not fit (only possible before anything was launched): node by node
    }
    rc = sdh_forward(hg, dx, dldx, T, gbuf, n_ff, flags, s);
    g->base_ready = false;
    if (rc != SDH_OK) { u->base_ready = false; return rc; }
    const GluEpilogue ge{gbuf, n_ff, act, ls.gelu_tab};
    rc = sdh_forward_impl(hu, dx, dldx, T, hbuf, n_ff, flags, s, nullptr, 0, &ge);
    u->base_ready = false;
    return rc;
}

int sdh_ffn_last_fused(void) { return lib().ffn_last_fused; }

static int sdh_ffn_gate_up_impl(sdh_weight * hg, sdh_weight * hu, int act, const float * x, int64_t ldx, int64_t n_tokens,
                    float * h, int64_t ldh, int flags, void * stream) {
    int rc = ffn_check(hg, hu, act, x, h);
    if (rc != SDH_OK) return rc;
    LibState & ls = lib();
    DeviceWeight * g = reinterpret_cast<DeviceWeight *>(hg);
    const int64_t n_embd = g->n_in, n_ff = g->n_out;
    if (n_tokens < 0 || ldx < n_embd || ldh < n_ff) { set_error("bad shape"); return SDH_ERR_INVALID; }
    if (n_tokens == 0) return SDH_OK;
    hipStream_t s = resolve_stream(stream, flags);
    const bool host_io = (flags & SDH_FWD_HOST_IO) != 0;
    const size_t T = (size_t) n_tokens;
    const bool dense_h = !host_io && ldh == n_ff;                // h can be written in place
    const size_t sec[4] = {host_io ? T * (size_t) n_embd : 0, dense_h ? 0 : T * (size_t) n_ff, T * (size_t) n_ff, T * (size_t) n_ff};
    float * ws[4];
    if ((rc = ffn_workspace(ls, s, sec, 4, ws)) != SDH_OK) return rc;
    const float * dx = x; int64_t dldx = ldx;
    if (host_io) {
        if ((rc = host_to_device(ls, ws[0], x, ldx, n_embd, n_tokens, s)) != SDH_OK) return rc;
        dx = ws[0]; dldx = n_embd;
    }
    float * hbuf = dense_h ? h : ws[1];
    int sub = (flags & ~SDH_FWD_HOST_IO) | (n_tokens > MAX_FUSED_TOKENS ? SDH_FWD_INTERNAL_ROWS : 0);
    if (ffn_fold_default(g, reinterpret_cast<DeviceWeight *>(hu), nullptr, n_tokens, flags)) sub |= SDH_FWD_MFMA_FOLD;
    if ((rc = ffn_gate_up(hg, hu, act, dx, dldx, n_tokens, hbuf, ws[2], ws[3], sub, s)) != SDH_OK) return rc;
    if (host_io) return device_to_host_sync(ls, h, ldh, hbuf, n_ff, n_tokens, s);
    if (!dense_h) {
        SDH_HIP_CHECK(hipMemcpy2DAsync(h, (size_t) ldh * 4, hbuf, (size_t) n_ff * 4, (size_t) n_ff * 4, T, hipMemcpyDeviceToDevice, s));
    }
    return SDH_OK;
}

static int sdh_ffn_forward_impl(sdh_weight * hg, sdh_weight * hu, sdh_weight * hd, int act,
                    const float * x, int64_t ldx, int64_t n_tokens, float * y, int64_t ldy,
                    int flags, void * stream) {
    int rc = ffn_check(hg, hu, act, x, y);
    if (rc != SDH_OK) return rc;
    if (!hd) { set_error("null argument"); return SDH_ERR_INVALID; }
    LibState & ls = lib();
    DeviceWeight * g = reinterpret_cast<DeviceWeight *>(hg);
    DeviceWeight * d = reinterpret_cast<DeviceWeight *>(hd);
    const int64_t n_embd = g->n_in, n_ff = g->n_out;
    if (d->n_in != n_ff) {
        set_error("down %lldx%lld does not take the gate/up output (%lld)", (long long) d->n_out, (long long) d->n_in, (long long) n_ff);
        return SDH_ERR_INVALID;
    }
    if (n_tokens < 0 || ldx < n_embd || ldy < d->n_out) { set_error("bad shape"); return SDH_ERR_INVALID; }
    if (n_tokens == 0) return SDH_OK;
    hipStream_t s = resolve_stream(stream, flags);
    const bool host_io = (flags & SDH_FWD_HOST_IO) != 0;
    const size_t T = (size_t) n_tokens;
    const bool few = n_tokens > 1 && n_tokens <= MAX_FUSED_TOKENS;
    const size_t sec[6] = {host_io ? T * (size_t) n_embd : 0, host_io ? T * (size_t) d->n_out : 0,
                           T * (size_t) n_ff, T * (size_t) n_ff, T * (size_t) n_ff, few ? 4 * (size_t) n_ff : 0};
    float * ws[6];
    if ((rc = ffn_workspace(ls, s, sec, 6, ws)) != SDH_OK) return rc;
    const float * dx = x; int64_t dldx = ldx;
    float * dy = y; int64_t dldy = ldy;
    if (host_io) {
        if ((rc = host_to_device(ls, ws[0], x, ldx, n_embd, n_tokens, s)) != SDH_OK) return rc;
        dx = ws[0]; dldx = n_embd; dy = ws[1]; dldy = d->n_out;
    }
    int sub = (flags & ~SDH_FWD_HOST_IO) | (n_tokens > MAX_FUSED_TOKENS ? SDH_FWD_INTERNAL_ROWS : 0);
    if (ffn_fold_default(g, reinterpret_cast<DeviceWeight *>(hu), d, n_tokens, flags)) sub |= SDH_FWD_MFMA_FOLD;
    // 5 tokens and more: node by node, the residual through the LDS rows kernels (SDH_FWD_INTERNAL_ROWS: COO payloads at
    // every token count, block payloads up to 32 tokens) and, up to 64 tokens, the base terms of gate and up from one
    // launch. Mistral layer, us, COO / block16: T=8 31 / 30 (56 / 58 through the tiled SpMM path, 49 / 37 as two groups of
    // four tokens through the single-launch forms — tried, dropped), T=16 38 / 38 (59 / 59), T=32 52 / 53 (77 / 69), T=512
    // 482 / 401 (548 COO tiled). On the rows kernels a token's result is its one-token result.
    if ((rc = ffn_gate_up(hg, hu, act, dx, dldx, n_tokens, ws[2], ws[3], ws[4], sub, s, few ? ws[5] : nullptr)) != SDH_OK) return rc;
    // 2..4 tokens through the gate|up launch: it left h token-interleaved as well, which the down launch gathers from
    if (few && ls.ffn_last_fused) d->x4_hint = ws[5];
    rc = sdh_forward(hd, ws[2], n_ff, n_tokens, dy, dldy, sub, s);
    d->x4_hint = nullptr;
    if (rc != SDH_OK) return rc;
    if (host_io) return device_to_host_sync(ls, y, ldy, dy, d->n_out, n_tokens, s);
    return SDH_OK;
}

// ---- persistent FFN stack ---------------------------------------------------------------------------------------
int sdh_ffn_stack_create(sdh_weight * const * gate, sdh_weight * const * up, sdh_weight * const * down, int n_layers,
                         sdh_ffn_stack ** out) {
    if (!gate || !up || !down || !out) { set_error("null argument"); return SDH_ERR_INVALID; }
    *out = nullptr;
    if (!lib().inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    FfnStack * st = nullptr;
    auto run = [&]() -> int { return stack_create(gate, up, down, n_layers, &st); };
    int rc;
    try { rc = run(); } catch (...) { set_error("internal error"); return SDH_ERR_HIP; }
    if (rc == SDH_OK) *out = reinterpret_cast<sdh_ffn_stack *>(st);
    return rc;
}

void sdh_ffn_stack_destroy(sdh_ffn_stack * h) {
    if (!h) return;
    (void) hipDeviceSynchronize();
    stack_free(reinterpret_cast<FfnStack *>(h));
}

int sdh_ffn_stack_forward(sdh_ffn_stack * h, int act, const float * x, float * y, int flags, void * stream) {
    if (!h || !x || !y) { set_error("null argument"); return SDH_ERR_INVALID; }
    LibState & ls = lib();
    if (!ls.inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    if (act != SDH_ACT_SILU && act != SDH_ACT_GELU) { set_error("unknown activation %d", act); return SDH_ERR_INVALID; }
    if (flags & ~SDH_FWD_ZERO_NONFINITE) { set_error("sdh_ffn_stack_forward: device pointers, fast path only (flags 0x%x)", flags); return SDH_ERR_INVALID; }
    if (act == SDH_ACT_GELU) { int rc = ensure_gelu_table(ls); if (rc != SDH_OK) return rc; }
    return stack_forward(reinterpret_cast<FfnStack *>(h), act, ls.gelu_tab, x, y, (flags & SDH_FWD_ZERO_NONFINITE) != 0, (hipStream_t) stream);
}

int sdh_ffn_stack_status(sdh_ffn_stack * h, int64_t * nonfinite, int64_t * timeouts) {
    if (!h) { set_error("null argument"); return SDH_ERR_INVALID; }
    return stack_status(reinterpret_cast<FfnStack *>(h), nonfinite, timeouts);
}

// diagnostic builds only (-DSDH_STK_STAMPS fills the buffer): not declared in the public header
int sdh_ffn_stack_debug_stamps(sdh_ffn_stack * h, unsigned long long * out) {
    if (!h || !out) return SDH_ERR_INVALID;
    return stack_stamps(reinterpret_cast<FfnStack *>(h), out);
}

int64_t sdh_ffn_stack_payload_bytes(const sdh_ffn_stack * h) {
    if (!h) { set_error("null argument"); return SDH_ERR_INVALID; }
    return stack_bytes(reinterpret_cast<const FfnStack *>(h));
}

int64_t sdh_nonfinite_count(sdh_weight * h) {
    if (!h) { set_error("null weight"); return SDH_ERR_INVALID; }
    DeviceWeight * w = reinterpret_cast<DeviceWeight *>(h);
    if (w->seq == 0) return 0;
    const int slot = (int) ((w->seq - 1) % DeviceWeight::NF_SLOTS);
    int v = 0;
    if (hipStreamSynchronize(w->last_stream) != hipSuccess ||
        hipMemcpy(&v, w->nf_ring + slot, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("reading non-finite counter failed");
        return SDH_ERR_HIP;
    }
    return v;
}

int sdh_profile_enable(int on) {
    LibState & s = lib();
    if (!s.inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    if (on && !s.prof_ev) {
        s.prof_ev = new hipEvent_t[2 * LibState::PROF_PAIRS];
        for (int i = 0; i < 2 * LibState::PROF_PAIRS; ++i) SDH_HIP_CHECK(hipEventCreate(&s.prof_ev[i]));
    }
    if (!on && s.prof_on) { int rc = prof_flush(s); if (rc != SDH_OK) return rc; }
    s.prof_on = on != 0;
    return SDH_OK;
}

int sdh_profile_span_begin(void * stream) {
    LibState & s = lib();
    if (!s.inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    if (!s.span_ev[0]) {
        SDH_HIP_CHECK(hipEventCreate(&s.span_ev[0]));
        SDH_HIP_CHECK(hipEventCreate(&s.span_ev[1]));
    }
    s.span_forwards = 0;
    SDH_HIP_CHECK(hipEventRecord(s.span_ev[0], (hipStream_t) stream));
    return SDH_OK;
}

int sdh_profile_span_end(void * stream, double * elapsed_ms, int64_t * forwards) {
    LibState & s = lib();
    if (!s.inited || !s.span_ev[0]) { set_error("sdh_profile_span_begin not called"); return SDH_ERR_INVALID; }
    SDH_HIP_CHECK(hipEventRecord(s.span_ev[1], (hipStream_t) stream));
    SDH_HIP_CHECK(hipEventSynchronize(s.span_ev[1]));
    float ms = 0.0f;
    SDH_HIP_CHECK(hipEventElapsedTime(&ms, s.span_ev[0], s.span_ev[1]));
    if (elapsed_ms) *elapsed_ms = (double) ms;
    if (forwards) *forwards = s.span_forwards;
    return SDH_OK;
}

int sdh_profile_read(double * total_ms, int64_t * launches) {
    LibState & s = lib();
    if (!s.inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    int rc = prof_flush(s);
    if (rc != SDH_OK) return rc;
    if (total_ms) *total_ms = s.prof_ms;
    if (launches) *launches = s.prof_n;
    s.prof_ms = 0.0;
    s.prof_n = 0;
    return SDH_OK;
}

static int sdh_w0_tile_impl(sdh_weight * h, int64_t o0, int64_t o1, int64_t i0, int64_t i1,
                float * out, int flags, void * stream) {
    if (!h || !out) { set_error("null argument"); return SDH_ERR_INVALID; }
    DeviceWeight * w = reinterpret_cast<DeviceWeight *>(h);
    if (!w->d1) { set_error("payload has no base"); return SDH_ERR_INVALID; }
    if (!w->w0_closed_form) {
        set_error("closed-form W0 needs depth 2, no perm2 and a bijective perm1");
        return SDH_ERR_UNSUPPORTED;
    }
    if (o0 < 0 || o1 > w->n_out || i0 < 0 || i1 > w->n_in || o0 > o1 || i0 > i1) {
        set_error("tile out of range");
        return SDH_ERR_INVALID;
    }
    hipStream_t s = resolve_stream(stream, flags);
    const size_t bytes = (size_t) (o1 - o0) * (size_t) (i1 - i0) * 4;
    if (bytes == 0) return SDH_OK;
    if (flags & SDH_FWD_HOST_IO) {
        float * d = nullptr;
        SDH_HIP_CHECK(hipMalloc((void **) &d, bytes));
        int rc = launch_w0_tile(*w, o0, o1, i0, i1, d, s);
        if (rc == SDH_OK) {
            hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { set_error("w0 tile copy failed: %s", hipGetErrorString(e)); rc = SDH_ERR_HIP; }
        }
        (void) hipFree(d);
        return rc;
    }
    return launch_w0_tile(*w, o0, o1, i0, i1, out, s);
}

// ---- the ABI is C: nothing is thrown across it (host allocations can fail) ---------------------------
int sdh_weight_create(const sdh_weight_desc * desc, sdh_weight ** out) {
    SDH_NO_THROW(sdh_weight_create_impl(desc, out));
}

int sdh_forward(sdh_weight * h, const float * x, int64_t ldx, int64_t n_tokens,
                float * y, int64_t ldy, int flags, void * stream) {
    SDH_NO_THROW(sdh_forward_impl(h, x, ldx, n_tokens, y, ldy, flags, stream));
}

// No hipMalloc / hipFree / allocating routine on this path: device I/O only, the base term goes through the
// static `scratch` array (this device's address of it, g_scratch_ptr), tokens are processed in chunks that fit it.
static int sdh_forward_noalloc_impl(sdh_weight * h, const float * x, int64_t ldx, int64_t n_tokens,
                                    float * y, int64_t ldy, int flags, void * stream) {
    if (!lib().inited) { set_error("sdh_init not called"); return SDH_ERR_NO_DEVICE; }
    if (flags & (SDH_FWD_HOST_IO | SDH_FWD_W0_MFMA)) {
        set_error("sdh_forward_noalloc: device pointers only, FWHT base only (flags 0x%x)", flags);
        return SDH_ERR_INVALID;
    }
    const int dev = lib().device;
    if (dev < 0 || dev >= MAX_DEVICES || !g_scratch_ptr[dev]) { set_error("sdh_forward_noalloc: no scratch on device %d", dev); return SDH_ERR_NO_DEVICE; }
    return sdh_forward_impl(h, x, ldx, n_tokens, y, ldy, flags, stream,
                            reinterpret_cast<float *>(g_scratch_ptr[dev]), NOALLOC_SCRATCH_BYTES);
}

int sdh_forward_noalloc(sdh_weight * h, const float * x, int64_t ldx, int64_t n_tokens,
                        float * y, int64_t ldy, int flags, void * stream) {
    SDH_NO_THROW(sdh_forward_noalloc_impl(h, x, ldx, n_tokens, y, ldy, flags, stream));
}

int64_t sdh_noalloc_scratch_bytes(void) { return (int64_t) NOALLOC_SCRATCH_BYTES; }
int64_t sdh_debug_alloc_calls(void) { return (int64_t) sdh::g_alloc_calls.load(); }
int sdh_debug_last_decode_kind(void) { return lib().last_decode_kind; }
int sdh_debug_fold_layout(sdh_weight * h) {
    if (!h) return 0;
    const DeviceWeight * w = reinterpret_cast<const DeviceWeight *>(h);
    if (!w->fold_ent) return 0;
    return 1 | (w->fold_perm ? 2 : 0) | (w->fold_dup ? 4 : 0) | (w->fold_over ? 8 : 0);
}

int sdh_ffn_gate_up(sdh_weight * hg, sdh_weight * hu, int act, const float * x, int64_t ldx, int64_t n_tokens,
                    float * h, int64_t ldh, int flags, void * stream) {
    SDH_NO_THROW(sdh_ffn_gate_up_impl(hg, hu, act, x, ldx, n_tokens, h, ldh, flags, stream));
}

int sdh_ffn_forward(sdh_weight * hg, sdh_weight * hu, sdh_weight * hd, int act,
                    const float * x, int64_t ldx, int64_t n_tokens, float * y, int64_t ldy,
                    int flags, void * stream) {
    SDH_NO_THROW(sdh_ffn_forward_impl(hg, hu, hd, act, x, ldx, n_tokens, y, ldy, flags, stream));
}

int sdh_w0_tile(sdh_weight * h, int64_t o0, int64_t o1, int64_t i0, int64_t i1,
                float * out, int flags, void * stream) {
    SDH_NO_THROW(sdh_w0_tile_impl(h, o0, o1, i0, i1, out, flags, stream));
}

} // extern "C"
