// ref_build_harness.cpp — drives the REFERENCE's own builder evaluation code
// (/root/reference/tools/seeddelta-build/{sd_eval,sd_cost,sd_ffn_proxy}.cpp, compiled unmodified
// from where they lie; see oracle/Makefile target `buildref`) on caller-supplied buffers.
// Test infrastructure only: it pins oracle/sd_scan_oracle.c and generates tests/golden/scan_*.npz.
//
// sd_build.cpp itself (the run_trial lambda, :463-800) is NOT built: it includes
// seeddelta_policy.h, which GCC 11 rejects (incomplete type in unordered_map), and the rules forbid
// building modified copies. Its column loop is therefore restated in the oracle only; the pieces it
// calls (topk_abs_weighted, topk_blocks_energy_weighted, fit_base_xor_circulant, the eval_* family,
// estimate_cost, eval_ffn_proxy_coo_replace_one) are all reachable here.
#include "sd_oracle.h"          // sdo_tensor (argument layout only)

#include "ggml.h"
#include "sd_cost.h"
#include "sd_eval.h"
#include "sd_ffn_proxy.h"
#include "sd_utils.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

namespace {

struct wrapped {
    ggml_context * ctx = nullptr;
    ggml_tensor * t = nullptr;
    ~wrapped() { if (ctx) ggml_free(ctx); }
};

// W as the builder sees it: 2-D [n_in, n_out], any ggml type with to_float
bool wrap2d(wrapped & w, const sdo_tensor * s) {
    if (!s || !s->data) return true;
    ggml_init_params ip = { 4 * ggml_tensor_overhead() + 1024, nullptr, true };
    w.ctx = ggml_init(ip);
    if (!w.ctx) return false;
    w.t = ggml_new_tensor_2d(w.ctx, (ggml_type) s->type, s->ne[0], s->ne[1]);
    w.t->data = (void *) s->data;
    w.t->nb[0] = (size_t) s->nb[0];
    w.t->nb[1] = (size_t) s->nb[1];
    w.t->nb[2] = w.t->nb[3] = w.t->nb[1] * (size_t) s->ne[1];
    return true;
}

std::vector<float> vec_or_empty(const float * p, size_t n) {
    return p ? std::vector<float>(p, p + n) : std::vector<float>();
}

struct base_arrays {       // [B][L] each; h2/perm1_inv are what the evals read, d1..perm1 what apply reads
    int64_t L, B;
    const float * d1, * d2, * d3, * h2;
    const int32_t * perm1, * perm1_inv;
};

base_fit to_fit(const base_arrays * a) {
    base_fit f;
    if (!a || a->L <= 0) return f;
    const size_t n = (size_t) a->L * (size_t) a->B;
    f.L = a->L; f.B = a->B;
    f.d1 = vec_or_empty(a->d1, n); f.d2 = vec_or_empty(a->d2, n); f.d3 = vec_or_empty(a->d3, n);
    f.h2 = vec_or_empty(a->h2, n);
    if (a->perm1)     f.perm1.assign(a->perm1, a->perm1 + n);
    if (a->perm1_inv) f.perm1_inv.assign(a->perm1_inv, a->perm1_inv + n);
    return f;
}

void put(double * out, const eval_metrics & m) {
    out[0] = m.rel_l2_mean; out[1] = m.rel_l2_p95; out[2] = m.cos_mean; out[3] = m.cos_p05; out[4] = m.norm_ratio_mean;
}

} // namespace

extern "C" {

// ---- libstdc++ RNG, exactly as the reference consumes it -----------------------------------------
void * sdbref_rng_new(uint32_t seed) { return new std::mt19937(seed); }
void   sdbref_rng_free(void * h) { delete (std::mt19937 *) h; }
uint32_t sdbref_rng_u32(void * h) { return (uint32_t) (*(std::mt19937 *) h)(); }
void sdbref_rng_shuffle_iota_i64(void * h, int64_t n, int64_t first, int64_t * out) {
    std::vector<int64_t> v((size_t) n);
    std::iota(v.begin(), v.end(), first);
    std::shuffle(v.begin(), v.end(), *(std::mt19937 *) h);
    std::memcpy(out, v.data(), (size_t) n * sizeof(int64_t));
}
void sdbref_rng_shuffle_iota_i32(void * h, int64_t n, int32_t * out) {
    std::vector<int32_t> v((size_t) n);
    std::iota(v.begin(), v.end(), 0);
    std::shuffle(v.begin(), v.end(), *(std::mt19937 *) h);
    std::memcpy(out, v.data(), (size_t) n * sizeof(int32_t));
}
// one normal_distribution<float> object for the whole run of n draws (it caches the second variate)
void sdbref_rng_normal_f32(void * h, int64_t n, float * out) {
    std::normal_distribution<float> nd(0.0f, 1.0f);
    for (int64_t i = 0; i < n; ++i) out[i] = nd(*(std::mt19937 *) h);
}

// ---- sd_eval.cpp ------------------------------------------------------------------------------------
void sdbref_read_row(const sdo_tensor * W, int64_t row, float * out) {
    wrapped w; wrap2d(w, W);
    std::vector<float> v;
    read_column_f32(w.t, row, v);
    std::memcpy(out, v.data(), v.size() * sizeof(float));
}

int64_t sdbref_topk_abs(const float * v, const float * w_scale, int64_t n, int64_t K, int32_t * out) {
    std::vector<float> vv(v, v + n), ws = vec_or_empty(w_scale, (size_t) n);
    std::vector<int32_t> idx;
    topk_abs_weighted(vv, w_scale ? &ws : nullptr, K, idx);
    std::memcpy(out, idx.data(), idx.size() * sizeof(int32_t));
    return (int64_t) idx.size();
}

int64_t sdbref_topk_blocks(const float * v, const float * w_scale, int64_t n, int64_t block, int64_t keep, int32_t * out) {
    std::vector<float> vv(v, v + n), ws = vec_or_empty(w_scale, (size_t) n);
    std::vector<int32_t> idx;
    topk_blocks_energy_weighted(vv, w_scale ? &ws : nullptr, block, keep, idx);
    std::memcpy(out, idx.data(), idx.size() * sizeof(int32_t));
    return (int64_t) idx.size();
}

// outputs sized [B][L] by the caller with L = next_pow2(min(n_in,n_out)), B = ceil(max/L)
void sdbref_fit_base(const sdo_tensor * W, int64_t max_samples, const float * w_scale, int perm_trials, void * rng,
                     int64_t * L, int64_t * B, float * d2, float * h2, int32_t * perm1, int32_t * perm1_inv) {
    wrapped w; wrap2d(w, W);
    std::vector<float> ws = vec_or_empty(w_scale, (size_t) W->ne[0]);
    base_fit f = fit_base_xor_circulant(w.t, max_samples, w_scale ? &ws : nullptr, perm_trials, *(std::mt19937 *) rng);
    *L = f.L; *B = f.B;
    const size_t n = (size_t) f.L * (size_t) f.B;
    std::memcpy(d2, f.d2.data(), n * sizeof(float));
    std::memcpy(h2, f.h2.data(), n * sizeof(float));
    std::memcpy(perm1, f.perm1.data(), n * sizeof(int32_t));
    std::memcpy(perm1_inv, f.perm1_inv.data(), n * sizeof(int32_t));
}

void sdbref_apply_base_block(const base_arrays * a, int64_t b, const float * x, float * y) {
    base_fit f = to_fit(a);
    std::vector<float> tmp((size_t) f.L);
    apply_base_block_f32(x, y, tmp.data(), f, b);
}

// scheme 0 = COO (idx [n_out][K], val [n_out][K]); 1 = block (idx [n_out][keep], val [n_out][keep][block])
void sdbref_eval_w(const sdo_tensor * W, const base_arrays * base, int scheme, const int32_t * idx, const float * val,
                   int64_t K_or_keep, int64_t block, const float * row_scale, const float * w_scale,
                   int64_t eval_cols, void * rng, double * out5) {
    wrapped w; wrap2d(w, W);
    const int64_t n_in = W->ne[0], n_out = W->ne[1];
    std::vector<float> rs = vec_or_empty(row_scale, (size_t) n_out), ws = vec_or_empty(w_scale, (size_t) n_in);
    std::mt19937 & g = *(std::mt19937 *) rng;
    eval_metrics m;
    if (scheme == 0) {
        const size_t n = (size_t) K_or_keep * (size_t) n_out;
        std::vector<int32_t> di(idx, idx + n);
        std::vector<float> dv(val, val + n);
        if (base) {
            base_fit f = to_fit(base);
            m = eval_seeddelta_base_residual(w.t, f, di, dv, row_scale ? &rs : nullptr, w_scale ? &ws : nullptr, K_or_keep, eval_cols, g);
        } else {
            m = eval_sparse_residual(w.t, di, dv, row_scale ? &rs : nullptr, w_scale ? &ws : nullptr, K_or_keep, eval_cols, g);
        }
    } else {
        const size_t n = (size_t) K_or_keep * (size_t) n_out;
        std::vector<int32_t> bi(idx, idx + n);
        std::vector<float> bv(val, val + n * (size_t) block);
        if (base) {
            base_fit f = to_fit(base);
            m = eval_seeddelta_base_block_residual(w.t, f, bi, bv, block, K_or_keep, row_scale ? &rs : nullptr, w_scale ? &ws : nullptr, eval_cols, g);
        } else {
            m = eval_block_residual(w.t, bi, bv, block, K_or_keep, row_scale ? &rs : nullptr, w_scale ? &ws : nullptr, eval_cols, g);
        }
    }
    put(out5, m);
}

void sdbref_eval_x(const sdo_tensor * W, const base_arrays * base, int scheme, const int32_t * idx, const float * val,
                   int64_t K_or_keep, int64_t block, const float * row_scale, const float * x_scale,
                   int64_t eval_cols, int64_t eval_x, void * rng, double * out5) {
    wrapped w; wrap2d(w, W);
    const int64_t n_in = W->ne[0], n_out = W->ne[1];
    std::vector<float> rs = vec_or_empty(row_scale, (size_t) n_out), xs = vec_or_empty(x_scale, (size_t) n_in);
    std::mt19937 & g = *(std::mt19937 *) rng;
    base_fit f = to_fit(base);
    const size_t n = (size_t) K_or_keep * (size_t) n_out;
    std::vector<int32_t> di(idx, idx + n);
    eval_metrics m;
    if (scheme == 0) {
        std::vector<float> dv(val, val + n);
        m = eval_seeddelta_x(w.t, base ? &f : nullptr, di, dv, row_scale ? &rs : nullptr, x_scale ? &xs : nullptr, K_or_keep, eval_cols, eval_x, g);
    } else {
        std::vector<float> bv(val, val + n * (size_t) block);
        m = eval_seeddelta_x_block(w.t, base ? &f : nullptr, di, bv, block, K_or_keep, row_scale ? &rs : nullptr, x_scale ? &xs : nullptr, eval_cols, eval_x, g);
    }
    put(out5, m);
}

// ---- sd_ffn_proxy.cpp / sd_cost.cpp -----------------------------------------------------------------
// out8 = cos_mean, cos_p05, l2_mean, l2_p95, lnr_mean, lnr_p95, eval_x, eval_out; returns 1 if available
int sdbref_ffn_proxy(int kind /*0 gate,1 up,2 down*/, const sdo_tensor * Wg, const sdo_tensor * Wu, const sdo_tensor * Wd,
                     const int32_t * idx, const float * val, const base_arrays * base, int64_t K,
                     int64_t eval_x, int64_t eval_cols, int seed, double * out8) {
    wrapped g, u, d; wrap2d(g, Wg); wrap2d(u, Wu); wrap2d(d, Wd);
    const int64_t n_rows = kind == 2 ? Wd->ne[1] : Wg->ne[1];
    const size_t n = (size_t) K * (size_t) n_rows;
    std::vector<int32_t> di(idx, idx + n);
    std::vector<float> dv(val, val + n);
    base_fit f = to_fit(base);
    ffn_proxy_metrics m;
    const char * names[3] = { "ffn_gate", "ffn_up", "ffn_down" };
    const bool ok = eval_ffn_proxy_coo_replace_one(names[kind], g.t, u.t, d.t, di, dv, base ? &f : nullptr, base != nullptr,
                                                   K, eval_x, eval_cols, seed, m);
    out8[0] = m.cos_mean; out8[1] = m.cos_p05; out8[2] = m.l2_mean; out8[3] = m.l2_p95;
    out8[4] = m.log_norm_ratio_mean; out8[5] = m.log_norm_ratio_p95; out8[6] = (double) m.eval_x; out8[7] = (double) m.eval_out;
    return ok ? 1 : 0;
}

// sd_make_imatrix_sqrt_scale (sd_utils.cpp:254-295) on one importance-matrix row
int sdbref_imatrix_scale(const float * imatrix, int64_t n, int64_t n_in, float eps, float power, float * out) {
    std::unordered_map<std::string, std::vector<float>> data;
    data["w"] = std::vector<float>(imatrix, imatrix + n);
    std::vector<float> sc;
    if (!sd_make_imatrix_sqrt_scale(data, "w", n_in, eps, power, sc)) return 0;
    std::memcpy(out, sc.data(), sc.size() * sizeof(float));
    return 1;
}

// out8 = L, B, ops_dense, ops_base, ops_delta, ops_row_scale, ops_total, ops_ratio
void sdbref_cost(int64_t L, int64_t B, int64_t n_in, int64_t n_out, int64_t K, int row_scale, double * out8) {
    base_fit f; f.L = L; f.B = B;
    cost_estimate c = estimate_cost(L > 0 ? &f : nullptr, n_in, n_out, K, row_scale != 0);
    out8[0] = (double) c.L; out8[1] = (double) c.B; out8[2] = c.ops_dense; out8[3] = c.ops_base;
    out8[4] = c.ops_delta; out8[5] = c.ops_row_scale; out8[6] = c.ops_total; out8[7] = c.ops_ratio;
}

} // extern "C"
