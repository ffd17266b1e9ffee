/*
 * sd_scan_oracle.c — see sd_scan_oracle.h. TEST INFRASTRUCTURE ONLY.
 *
 * Restates, in one place and without the reference's four-fold duplication:
 *   read_column_f32                      tools/seeddelta-build/sd_eval.cpp:20-40
 *   topk_abs_weighted / topk_blocks_…    sd_eval.cpp:42-116
 *   apply_base_block_f32                 sd_eval.cpp:370-425
 *   fit_base_xor_circulant               sd_eval.cpp:431-654
 *   eval_{sparse,block,seeddelta_base,seeddelta_base_block}_residual   sd_eval.cpp:118-330, 656-995
 *   eval_seeddelta_x / _x_block          sd_eval.cpp:997-1419
 *   run_trial column loop                sd_build.cpp:506-703
 *   eval_ffn_proxy_coo_replace_one       sd_ffn_proxy.cpp:20-247
 *   estimate_cost                        sd_cost.cpp:7-40
 * and the libstdc++ 11 algorithms the reference draws its samples with (std::mt19937, std::shuffle
 * with the two-swaps-per-draw path and Lemire's bounded draw, std::normal_distribution<float> as
 * Marsaglia polar over generate_canonical<float,24>): bits/stl_algo.h, bits/uniform_int_dist.h,
 * bits/random.tcc of GCC 11.4 (not vendored by the reference; pinned by tests against the
 * container's libstdc++ through oracle/_ref/libsdbuildref.so).
 *
 * Every double accumulation keeps the reference's order and expression shape: the compiled
 * reference and this file agree bit-for-bit (built -O3 -ffp-contract=off).
 */
#define _GNU_SOURCE
#include "sd_scan_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ===== random numbers ================================================================== */
void sdso_rng_seed(sdso_rng * g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) {
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t) i;
    }
    g->pos = 624;
    g->have_saved = 0;
    g->saved = 0.0f;
}

uint32_t sdso_rng_u32(sdso_rng * g) {
    if (g->pos >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->pos = 0;
    }
    uint32_t y = g->mt[g->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* uniform_int_distribution<unsigned long>{0, range-1} on a 32-bit engine: Lemire's method */
static uint32_t bounded(sdso_rng * g, uint32_t range) {
    uint64_t prod = (uint64_t) sdso_rng_u32(g) * (uint64_t) range;
    uint32_t low = (uint32_t) prod;
    if (low < range) {
        const uint32_t threshold = (uint32_t) (0u - range) % range;
        while (low < threshold) {
            prod = (uint64_t) sdso_rng_u32(g) * (uint64_t) range;
            low = (uint32_t) prod;
        }
    }
    return (uint32_t) (prod >> 32);
}

#define SDSO_SHUFFLE(NAME, T)                                                            \
    void NAME(sdso_rng * g, T * v, int64_t n) {                                          \
        if (n <= 0) return;                                                              \
        const uint64_t un = (uint64_t) n;                                                \
        int64_t i = 1;                                                                   \
        if (0xffffffffull / un >= un) {          /* two swap positions per draw */       \
            if ((un % 2) == 0) {                                                         \
                const int64_t j = (int64_t) bounded(g, 2);                               \
                T t = v[i]; v[i] = v[j]; v[j] = t; ++i;                                  \
            }                                                                            \
            while (i < n) {                                                              \
                const uint64_t r = (uint64_t) i + 1;                                     \
                const uint64_t x = bounded(g, (uint32_t) (r * (r + 1)));                 \
                const int64_t a = (int64_t) (x / (r + 1)), b = (int64_t) (x % (r + 1));  \
                T t = v[i]; v[i] = v[a]; v[a] = t; ++i;                                  \
                t = v[i]; v[i] = v[b]; v[b] = t; ++i;                                    \
            }                                                                            \
            return;                                                                      \
        }                                                                                \
        for (; i < n; ++i) {                                                             \
            const int64_t j = (int64_t) bounded(g, (uint32_t) (i + 1));                  \
            T t = v[i]; v[i] = v[j]; v[j] = t;                                           \
        }                                                                                \
    }
SDSO_SHUFFLE(sdso_shuffle_i64, int64_t)
SDSO_SHUFFLE(sdso_shuffle_i32, int32_t)

/* generate_canonical<float, 24>(mt19937): one 32-bit draw, rounded to float, divided by 2^32 */
static float canonical_f32(sdso_rng * g) {
    const float sum = (float) sdso_rng_u32(g);
    float r = sum / 4294967296.0f;
    if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
    return r;
}

static float normal_draw(sdso_rng * g) {
    if (g->have_saved) {
        g->have_saved = 0;
        return g->saved;          /* * stddev(1) + mean(0): exact */
    }
    float x, y, r2;
    do {
        x = 2.0f * canonical_f32(g) - 1.0f;
        y = 2.0f * canonical_f32(g) - 1.0f;
        r2 = x * x + y * y;
    } while (r2 > 1.0f || r2 == 0.0f);
    const float mult = sqrtf(-2 * logf(r2) / r2);
    g->saved = x * mult;
    g->have_saved = 1;
    return y * mult;
}

void sdso_normal_f32(sdso_rng * g, int64_t n, float * out) {
    g->have_saved = 0;            /* a fresh distribution object */
    for (int64_t i = 0; i < n; ++i) out[i] = normal_draw(g);
    g->have_saved = 0;
}

/* ===== rows of W =========================================================================== */
void sdso_read_row(const sdo_tensor * W, int64_t row, float * out) {
    const int64_t n_in = W->ne[0];
    const uint8_t * p = (const uint8_t *) W->data + row * W->nb[1];
    if (W->type == SDO_F32) {
        memcpy(out, p, (size_t) n_in * sizeof(float));
    } else if (W->type == SDO_F16) {
        const uint16_t * h = (const uint16_t *) p;
        for (int64_t i = 0; i < n_in; ++i) out[i] = sdo_fp16_to_fp32(h[i]);
    } else {
        sdo_dequant_row_q8_0(p, out, n_in);
    }
}

/* ===== selection =========================================================================== */
typedef struct { double s; int32_t i; } keyed;
static int by_score_desc(const void * a, const void * b) {
    const keyed * x = (const keyed *) a, * y = (const keyed *) b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int64_t sdso_topk_abs(const float * v, const float * w_scale, int64_t n, int64_t K, int32_t * out) {
    if (K > n) K = n;
    keyed * k = (keyed *) malloc((size_t) (n > 0 ? n : 1) * sizeof(keyed));
    for (int64_t i = 0; i < n; ++i) {
        const float s = fabsf(v[i]);
        k[i].s = (double) (w_scale ? s * w_scale[i] : s);     /* score is a float product */
        k[i].i = (int32_t) i;
    }
    qsort(k, (size_t) n, sizeof(keyed), by_score_desc);
    for (int64_t r = 0; r < K; ++r) out[r] = k[r].i;
    free(k);
    return K;
}

int64_t sdso_topk_blocks(const float * v, const float * w_scale, int64_t n, int64_t block, int64_t keep, int32_t * out) {
    if (n <= 0 || block <= 0 || keep <= 0) return 0;
    const int64_t nblk = (n + block - 1) / block;
    if (keep > nblk) keep = nblk;
    keyed * k = (keyed *) malloc((size_t) nblk * sizeof(keyed));
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t i0 = b * block, i1 = (i0 + block < n) ? i0 + block : n;
        double e = 0.0;
        for (int64_t i = i0; i < i1; ++i) {
            const double x = (double) v[i] * (w_scale ? (double) w_scale[i] : 1.0);
            e += x * x;
        }
        k[b].s = e;
        k[b].i = (int32_t) b;
    }
    qsort(k, (size_t) nblk, sizeof(keyed), by_score_desc);
    for (int64_t r = 0; r < keep; ++r) out[r] = k[r].i;
    free(k);
    return keep;
}

/* ===== base ================================================================================ */
void sdso_fit_dims(int64_t n_in, int64_t n_out, int64_t * L, int64_t * B) {
    const int64_t lo = n_in < n_out ? n_in : n_out, hi = n_in < n_out ? n_out : n_in;
    int64_t p = 1;
    while (p < lo) p <<= 1;
    *L = p;
    *B = (hi + p - 1) / p;
}

void sdso_apply_base_block(const sdso_base * a, int64_t b, const float * x, float * y) {
    const int64_t L = a->L;
    const float * d1 = a->d1 ? a->d1 + b * L : NULL, * d2 = a->d2 ? a->d2 + b * L : NULL, * d3 = a->d3 ? a->d3 + b * L : NULL;
    const int32_t * p1 = a->perm1 ? a->perm1 + b * L : NULL;
    float * tmp = (float *) malloc((size_t) L * sizeof(float));
    for (int64_t i = 0; i < L; ++i) y[i] = d1 ? x[i] * d1[i] : x[i];
    if (p1) {
        for (int64_t i = 0; i < L; ++i) {
            int32_t s = p1[i];
            if (s < 0 || s >= (int32_t) L) s = 0;
            tmp[i] = y[s];
        }
        memcpy(y, tmp, (size_t) L * sizeof(float));
    }
    sdo_hadamard(y, L);
    if (d2) for (int64_t i = 0; i < L; ++i) y[i] *= d2[i];
    sdo_hadamard(y, L);
    if (d3) for (int64_t i = 0; i < L; ++i) y[i] *= d3[i];
    free(tmp);
}

/* one perm trial for one base block: least-squares XOR-circulant projection (sd_eval.cpp:486-555) */
static double fit_trial(const sdo_tensor * W, int tall, int64_t b, int64_t L, const int64_t * rows, int64_t S,
                        const int32_t * perm, const float * w_scale, float * wrow,
                        double * sum, double * sum2, double * wgt, float * h2, float * d2) {
    const int64_t n_in = W->ne[0];
    memset(sum, 0, (size_t) L * sizeof(double));
    memset(sum2, 0, (size_t) L * sizeof(double));
    memset(wgt, 0, (size_t) L * sizeof(double));
    const int64_t in0 = tall ? 0 : b * L;
    const int64_t in1 = tall ? n_in : (in0 + L < n_in ? in0 + L : n_in);
    for (int64_t si = 0; si < S; ++si) {
        const int64_t row = rows[si];
        const int64_t p = tall ? row - b * L : row;
        sdso_read_row(W, row, wrow);
        for (int64_t j = 0; j < L; ++j) {
            const int64_t src = in0 + (int64_t) perm[j];
            if (src < in0 || src >= in1) continue;
            const double ws = w_scale ? (double) w_scale[src] : 1.0;
            const double ww = ws * ws;
            const double v = (double) wrow[src];
            const int64_t u = (p ^ j) & (L - 1);
            sum[u]  += ww * v;
            sum2[u] += ww * v * v;
            wgt[u]  += ww;
        }
    }
    double sse = 0.0;
    for (int64_t u = 0; u < L; ++u) {
        h2[u] = 0.0f;
        if (wgt[u] > 0.0) {
            h2[u] = (float) (sum[u] / wgt[u]);
            const double e = sum2[u] - (sum[u] * sum[u]) / wgt[u];
            sse += e > 0.0 ? e : 0.0;
        }
    }
    memcpy(d2, h2, (size_t) L * sizeof(float));
    sdo_hadamard(d2, L);
    const float inv_L = 1.0f / (float) L;
    for (int64_t u = 0; u < L; ++u) d2[u] *= inv_L;
    return sse;
}

void sdso_fit_base(const sdo_tensor * W, int64_t max_samples, const float * w_scale, int perm_trials, sdso_rng * g,
                   float * d2_out, float * h2_out, int32_t * perm1, int32_t * perm1_inv) {
    const int64_t n_in = W->ne[0], n_out = W->ne[1];
    const int tall = n_out >= n_in;
    int64_t L, B;
    sdso_fit_dims(n_in, n_out, &L, &B);
    if (perm_trials < 1) perm_trials = 1;
    memset(d2_out, 0, (size_t) (L * B) * sizeof(float));
    memset(h2_out, 0, (size_t) (L * B) * sizeof(float));
    memset(perm1, 0, (size_t) (L * B) * sizeof(int32_t));
    memset(perm1_inv, 0, (size_t) (L * B) * sizeof(int32_t));

    float * wrow = (float *) malloc((size_t) n_in * sizeof(float));
    double * acc = (double *) malloc((size_t) (3 * L) * sizeof(double));
    float * h2 = (float *) malloc((size_t) (2 * L) * sizeof(float)), * d2 = h2 + L;
    int32_t * perm = (int32_t *) malloc((size_t) L * sizeof(int32_t));
    const int64_t max_rows = tall ? L : (n_out < L ? n_out : L);
    int64_t * rows = (int64_t *) malloc((size_t) max_rows * sizeof(int64_t));
    int64_t n_rows = 0, S = 0;

    if (!tall) {                       /* wide: one sample of output rows shared by all blocks */
        n_rows = max_rows;
        for (int64_t i = 0; i < n_rows; ++i) rows[i] = i;
        sdso_shuffle_i64(g, rows, n_rows);
        S = max_samples <= 0 ? n_rows : (n_rows < max_samples ? n_rows : max_samples);
    }
    for (int64_t b = 0; b < B; ++b) {
        if (tall) {
            const int64_t o0 = b * L, o1 = (b + 1) * L < n_out ? (b + 1) * L : n_out;
            n_rows = o1 - o0;
            if (n_rows <= 0) continue;
            for (int64_t i = 0; i < n_rows; ++i) rows[i] = o0 + i;
            sdso_shuffle_i64(g, rows, n_rows);
            S = max_samples <= 0 ? n_rows : (n_rows < max_samples ? n_rows : max_samples);
        } else if (((b * L + L < n_in) ? L : n_in - b * L) <= 0) {
            continue;
        }
        double best = INFINITY;
        for (int t = 0; t < perm_trials; ++t) {
            for (int64_t i = 0; i < L; ++i) perm[i] = (int32_t) i;
            if (t > 0) sdso_shuffle_i32(g, perm, L);
            const double sse = fit_trial(W, tall, b, L, rows, S, perm, w_scale, wrow, acc, acc + L, acc + 2 * L, h2, d2);
            if (sse < best) {
                best = sse;
                memcpy(h2_out + b * L, h2, (size_t) L * sizeof(float));
                memcpy(d2_out + b * L, d2, (size_t) L * sizeof(float));
                memcpy(perm1 + b * L, perm, (size_t) L * sizeof(int32_t));
            }
        }
        /* NB: when every trial's SSE is NaN the reference keeps an empty best_perm; not reproduced */
        for (int64_t i = 0; i < L; ++i) {
            const int32_t s = perm1[b * L + i];
            if (s >= 0 && s < (int32_t) L) perm1_inv[b * L + s] = (int32_t) i;
        }
    }
    free(rows); free(perm); free(h2); free(acc); free(wrow);
}

/* base weight the evals and run_trial read: h2_b[(p ^ inv_b[q]) & (L-1)] */
static inline float base_weight(const sdso_base * a, int tall, int64_t row, int64_t i) {
    const int64_t L = a->L;
    const int64_t b = tall ? row / L : i / L;
    const int64_t p = tall ? row - b * L : row;
    const int64_t q = tall ? i : i - b * L;
    const int64_t j = a->perm1_inv ? (int64_t) a->perm1_inv[b * L + q] : q;
    return a->h2[b * L + ((p ^ j) & (L - 1))];
}

/* ===== run_trial column loop (sd_build.cpp:527-695) ======================================== */
typedef struct {
    const sdo_tensor * W; const sdso_base * base; const float * ws; int scheme; int64_t K, block;
    int32_t * idx; float * val; float * rs; int64_t r0, r1;
    const int64_t * rows;          /* NULL: rows r0..r1-1; else rows[r0..r1-1] (outputs still stored at the row's own slot) */
} select_job;

static void * select_worker(void * arg) {
    const select_job * j = (const select_job *) arg;
    const int64_t n_in = j->W->ne[0], n_out = j->W->ne[1];
    const int tall = n_out >= n_in;
    float * w = (float *) malloc((size_t) n_in * sizeof(float));
    float * r = (float *) malloc((size_t) n_in * sizeof(float));
    int32_t * top = (int32_t *) malloc((size_t) (j->K > 0 ? j->K : 1) * sizeof(int32_t));
    for (int64_t rr0 = j->r0; rr0 < j->r1; ++rr0) {
        const int64_t row = j->rows ? j->rows[rr0] : rr0;
        sdso_read_row(j->W, row, w);
        double dot = 0.0, nn = 0.0;
        const float * src = w;
        if (j->base) {
            /* tall rows whose base block lies beyond B cannot occur for a fit from this W */
            for (int64_t i = 0; i < n_in; ++i) {
                const float bw = base_weight(j->base, tall, row, i);
                r[i] = w[i] - bw;
                if (j->rs) {
                    const double s = j->ws ? (double) j->ws[i] : 1.0;
                    dot += (double) w[i] * (double) bw * s * s;
                    nn  += (double) bw * (double) bw * s * s;
                }
            }
            src = r;
        }
        if (j->scheme == 1) {
            const int64_t got = sdso_topk_blocks(src, j->ws, n_in, j->block, j->K, top);
            if (j->rs) {
                for (int64_t bi = 0; bi < got; ++bi) {
                    const int64_t in0 = (int64_t) top[bi] * j->block, in1 = in0 + j->block < n_in ? in0 + j->block : n_in;
                    for (int64_t i = in0; i < in1; ++i) {
                        const double wi = (double) w[i], di = (double) src[i], s = j->ws ? (double) j->ws[i] : 1.0;
                        if (j->base) { dot += wi * di * s * s; nn += (2.0 * wi * di - di * di) * s * s; }
                        else         { dot += wi * wi * s * s; nn += wi * wi * s * s; }
                    }
                }
                j->rs[row] = nn > 1e-30 ? (float) (dot / nn) : 1.0f;
            }
            for (int64_t bi = 0; bi < j->K; ++bi) {
                const int32_t blk = bi < got ? top[bi] : -1;
                j->idx[row * j->K + bi] = blk;
                for (int64_t t = 0; t < j->block; ++t) {
                    const int64_t i = (int64_t) blk * j->block + t;
                    j->val[(row * j->K + bi) * j->block + t] = (blk >= 0 && i < n_in) ? src[i] : 0.0f;
                }
            }
        } else {
            const int64_t got = sdso_topk_abs(src, j->ws, n_in, j->K, top);
            if (j->rs) {
                for (int64_t rr = 0; rr < got; ++rr) {
                    const int64_t i = top[rr];
                    const double wi = (double) w[i], di = (double) src[i], s = j->ws ? (double) j->ws[i] : 1.0;
                    if (j->base) { dot += wi * di * s * s; nn += (2.0 * wi * di - di * di) * s * s; }
                    else         { dot += wi * wi * s * s; nn += wi * wi * s * s; }
                }
                j->rs[row] = nn > 1e-30 ? (float) (dot / nn) : 1.0f;
            }
            for (int64_t rr = 0; rr < j->K; ++rr) {
                j->idx[row * j->K + rr] = rr < got ? top[rr] : -1;
                j->val[row * j->K + rr] = rr < got ? src[top[rr]] : 0.0f;
            }
        }
    }
    free(top); free(r); free(w);
    return NULL;
}

void sdso_select_rows(const sdo_tensor * W, const sdso_base * base, const float * w_scale, int scheme,
                      int64_t K_or_keep, int64_t block, int32_t * idx, float * val, float * row_scale, int nth) {
    const int64_t n_out = W->ne[1];
    if (nth < 1) nth = 1;
    if (nth > 64) nth = 64;
    select_job jobs[64];
    pthread_t th[64];
    for (int t = 0; t < nth; ++t) {
        select_job j = { W, base, w_scale, scheme, K_or_keep, block, idx, val, row_scale,
                         n_out * t / nth, n_out * (t + 1) / nth, NULL };
        jobs[t] = j;
        if (nth == 1) select_worker(&jobs[0]);
        else pthread_create(&th[t], NULL, select_worker, &jobs[t]);
    }
    if (nth > 1) for (int t = 0; t < nth; ++t) pthread_join(th[t], NULL);
}

/* The same column loop for a LIST of rows (each one independent of the others in run_trial): outputs go to the
 * rows' own slots of the [n_out] arrays, other slots are left untouched. Lets a full-size parity test restate only
 * the rows the metric passes sample (sd_eval.cpp:138, :1052). */
void sdso_select_rows_list(const sdo_tensor * W, const sdso_base * base, const float * w_scale, int scheme,
                           int64_t K_or_keep, int64_t block, int32_t * idx, float * val, float * row_scale,
                           const int64_t * rows, int64_t n_rows) {
    select_job j = { W, base, w_scale, scheme, K_or_keep, block, idx, val, row_scale, 0, n_rows, rows };
    select_worker(&j);
}

/* ===== metrics ============================================================================= */
static int cmp_double(const void * a, const void * b) {
    const double x = *(const double *) a, y = *(const double *) b;
    return (x > y) - (x < y);
}

double sdso_percentile(const double * v, int64_t n, double p) {
    if (n <= 0) return 0.0;
    double * s = (double *) malloc((size_t) n * sizeof(double));
    memcpy(s, v, (size_t) n * sizeof(double));
    qsort(s, (size_t) n, sizeof(double), cmp_double);
    const double x = p * (double) (n - 1);
    const size_t i = (size_t) x;
    const size_t j = i + 1 < (size_t) n - 1 ? i + 1 : (size_t) n - 1;
    const double a = x - (double) i;
    const double r = s[i] * (1.0 - a) + s[j] * a;
    free(s);
    return r;
}

static void summarise(const double * rel, const double * cs, const double * nr, int64_t n, double * out5) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int64_t i = 0; i < n; ++i) a += rel[i];
    for (int64_t i = 0; i < n; ++i) b += cs[i];
    for (int64_t i = 0; i < n; ++i) c += nr[i];
    out5[0] = n ? a / (double) n : 0.0;
    out5[1] = sdso_percentile(rel, n, 0.95);
    out5[2] = n ? b / (double) n : 0.0;
    out5[3] = sdso_percentile(cs, n, 0.05);
    out5[4] = n ? c / (double) n : 0.0;
}

static void triple_to_metrics(double n_true, double n_hat, double dot, double * rel, double * cs, double * nr) {
    double err2 = n_true + n_hat - 2.0 * dot;
    if (err2 < 0.0) err2 = 0.0;
    const double dt = sqrt(n_true > 1e-20 ? n_true : 1e-20), dh = sqrt(n_hat > 1e-20 ? n_hat : 1e-20);
    *rel = sqrt(err2) / dt;
    *cs = dot / (dt * dh);
    *nr = dh / dt;
}

void sdso_eval_w_rows(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                      int64_t K, int64_t block, const float * row_scale, const float * w_scale,
                      const int64_t * rows, int64_t n_rows, double * per_row, double * out5) {
    const int64_t n_in = W->ne[0], n_out = W->ne[1];
    const int tall = n_out >= n_in;
    float * w = (float *) malloc((size_t) n_in * sizeof(float));
    double * m = (double *) malloc((size_t) (3 * (n_rows > 0 ? n_rows : 1)) * sizeof(double));
    double * rel = m, * cs = m + n_rows, * nr = m + 2 * n_rows;
    for (int64_t ci = 0; ci < n_rows; ++ci) {
        const int64_t row = rows[ci];
        sdso_read_row(W, row, w);
        const double scale = row_scale ? (double) row_scale[row] : 1.0;
        double w_norm2 = 0.0, hat2 = 0.0, dot = 0.0;
        for (int64_t i = 0; i < n_in; ++i) {
            const double ws = w_scale ? (double) w_scale[i] : 1.0;
            const double wi = (double) w[i] * ws;
            w_norm2 += wi * wi;
            if (base) {
                const double bs = (double) base_weight(base, tall, row, i) * ws;
                hat2 += bs * bs;
                dot += wi * bs;
            }
        }
        const int64_t n_ent = scheme == 1 ? K * block : K;
        for (int64_t e = 0; e < n_ent; ++e) {
            int64_t i;
            float dv;
            if (scheme == 1) {
                const int32_t blk = idx[row * K + e / block];
                if (blk < 0) continue;
                i = (int64_t) blk * block + e % block;
                if (i >= n_in) continue;
                dv = val[(row * K + e / block) * block + e % block];
            } else {
                const int32_t ii = idx[row * K + e];
                if (ii < 0 || ii >= (int32_t) n_in) continue;
                i = ii;
                dv = val[row * K + e];
            }
            const double ws = w_scale ? (double) w_scale[i] : 1.0;
            if (base) {
                const double wi = (double) w[i] * ws;
                const double ds = (double) dv * ws;
                const double bs = (double) base_weight(base, tall, row, i) * ws;
                dot += wi * ds;
                hat2 += 2.0 * bs * ds + ds * ds;
            } else {
                const double wh = scale * (double) dv;
                const double whs = wh * ws;
                hat2 += whs * whs;
                dot += (double) w[i] * ws * whs;
            }
        }
        if (base) {                 /* the base variants scale once at the end (sd_eval.cpp:782-783) */
            dot = scale * dot;
            hat2 = scale * scale * hat2;
        }
        if (per_row) { per_row[3 * ci] = w_norm2; per_row[3 * ci + 1] = hat2; per_row[3 * ci + 2] = dot; }
        triple_to_metrics(w_norm2, hat2, dot, &rel[ci], &cs[ci], &nr[ci]);
    }
    summarise(rel, cs, nr, n_rows, out5);
    free(m); free(w);
}

static int64_t draw_rows(sdso_rng * g, int64_t n_out, int64_t want, int64_t ** rows) {
    int64_t * all = (int64_t *) malloc((size_t) n_out * sizeof(int64_t));
    for (int64_t i = 0; i < n_out; ++i) all[i] = i;
    sdso_shuffle_i64(g, all, n_out);
    *rows = all;
    return want < n_out ? want : n_out;
}

void sdso_eval_w(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                 int64_t K, int64_t block, const float * row_scale, const float * w_scale,
                 int64_t eval_cols, sdso_rng * g, double * out5) {
    memset(out5, 0, 5 * sizeof(double));
    if (eval_cols <= 0 || W->ne[0] <= 0 || W->ne[1] <= 0 || K <= 0) return;
    int64_t * rows;
    const int64_t n = draw_rows(g, W->ne[1], eval_cols, &rows);
    sdso_eval_w_rows(W, base, scheme, idx, val, K, block, row_scale, w_scale, rows, n, NULL, out5);
    free(rows);
}

/* base output for the sampled rows, fp32 exactly as the builder computes it (sd_eval.cpp:1102-1136) */
static void base_outputs(const sdso_base * a, int tall, int64_t n_in, const float * x,
                         const int64_t * rows, int64_t n_rows, double * y_hat) {
    const int64_t L = a->L, B = a->B;
    float * xin = (float *) calloc((size_t) L, sizeof(float));
    float * v = (float *) malloc((size_t) L * sizeof(float));
    if (tall) {
        memcpy(xin, x, (size_t) n_in * sizeof(float));
        for (int64_t b = 0; b < B; ++b) {
            int used = 0;
            for (int64_t ci = 0; ci < n_rows && !used; ++ci) used = rows[ci] / L == b;
            if (!used) continue;
            sdso_apply_base_block(a, b, xin, v);
            for (int64_t ci = 0; ci < n_rows; ++ci) {
                if (rows[ci] / L == b) y_hat[ci] = (double) v[rows[ci] - b * L];
            }
        }
    } else {
        float * acc = (float *) calloc((size_t) L, sizeof(float));
        for (int64_t b = 0; b < B; ++b) {
            const int64_t in0 = b * L, n = (in0 + L < n_in ? L : n_in - in0);
            memset(xin, 0, (size_t) L * sizeof(float));
            if (n > 0) memcpy(xin, x + in0, (size_t) n * sizeof(float));
            sdso_apply_base_block(a, b, xin, v);
            for (int64_t i = 0; i < L; ++i) acc[i] += v[i];
        }
        for (int64_t ci = 0; ci < n_rows; ++ci) y_hat[ci] = rows[ci] < L ? (double) acc[rows[ci]] : 0.0;
        free(acc);
    }
    free(v); free(xin);
}

void sdso_eval_x_rows(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                      int64_t K, int64_t block, const float * row_scale,
                      const int64_t * rows, int64_t n_rows, const float * xs, int64_t n_x, double * y_out, double * out5) {
    const int64_t n_in = W->ne[0], n_out = W->ne[1];
    const int tall = n_out >= n_in;
    float * wrows = (float *) malloc((size_t) (n_rows * n_in) * sizeof(float));
    for (int64_t ci = 0; ci < n_rows; ++ci) sdso_read_row(W, rows[ci], wrows + ci * n_in);
    double * y_true = (double *) malloc((size_t) (2 * n_rows) * sizeof(double)), * y_hat = y_true + n_rows;
    double * m = (double *) malloc((size_t) (3 * (n_x > 0 ? n_x : 1)) * sizeof(double));
    double * rel = m, * cs = m + n_x, * nr = m + 2 * n_x;
    for (int64_t xi = 0; xi < n_x; ++xi) {
        const float * x = xs + xi * n_in;
        for (int64_t ci = 0; ci < n_rows; ++ci) {
            const float * w = wrows + ci * n_in;
            double acc = 0.0;
            for (int64_t i = 0; i < n_in; ++i) acc += (double) w[i] * (double) x[i];
            y_true[ci] = acc;
            y_hat[ci] = 0.0;
        }
        if (base && base->L > 0 && base->B > 0) base_outputs(base, tall, n_in, x, rows, n_rows, y_hat);
        for (int64_t ci = 0; ci < n_rows; ++ci) {
            const int64_t row = rows[ci];
            double y = y_hat[ci];
            if (scheme == 1) {
                for (int64_t bi = 0; bi < K; ++bi) {
                    const int32_t blk = idx[row * K + bi];
                    if (blk < 0) continue;
                    const int64_t in0 = (int64_t) blk * block, in1 = in0 + block < n_in ? in0 + block : n_in;
                    for (int64_t i = in0; i < in1; ++i) y += (double) val[(row * K + bi) * block + (i - in0)] * (double) x[i];
                }
            } else {
                for (int64_t r = 0; r < K; ++r) {
                    const int32_t ii = idx[row * K + r];
                    if (ii < 0 || ii >= (int32_t) n_in) continue;
                    y += (double) val[row * K + r] * (double) x[ii];
                }
            }
            y_hat[ci] = y * (row_scale ? (double) row_scale[row] : 1.0);
        }
        double n1 = 0.0, n2 = 0.0, dot = 0.0;
        for (int64_t ci = 0; ci < n_rows; ++ci) {
            n1 += y_true[ci] * y_true[ci];
            n2 += y_hat[ci] * y_hat[ci];
            dot += y_true[ci] * y_hat[ci];
            if (y_out) { y_out[(xi * n_rows + ci) * 2] = y_true[ci]; y_out[(xi * n_rows + ci) * 2 + 1] = y_hat[ci]; }
        }
        triple_to_metrics(n1, n2, dot, &rel[xi], &cs[xi], &nr[xi]);
    }
    summarise(rel, cs, nr, n_x, out5);
    free(m); free(y_true); free(wrows);
}

void sdso_eval_x(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                 int64_t K, int64_t block, const float * row_scale, const float * x_scale,
                 int64_t eval_cols, int64_t eval_x, sdso_rng * g, double * out5) {
    memset(out5, 0, 5 * sizeof(double));
    const int64_t n_in = W->ne[0];
    if (eval_cols <= 0 || eval_x <= 0 || n_in <= 0 || W->ne[1] <= 0 || K <= 0) return;
    int64_t * rows;
    const int64_t n = draw_rows(g, W->ne[1], eval_cols, &rows);
    /* nothing else draws between the x vectors, so they can be drawn up front */
    float * xs = (float *) malloc((size_t) (eval_x * n_in) * sizeof(float));
    sdso_normal_f32(g, eval_x * n_in, xs);
    if (x_scale) for (int64_t k = 0; k < eval_x * n_in; ++k) xs[k] = xs[k] * x_scale[k % n_in];
    sdso_eval_x_rows(W, base, scheme, idx, val, K, block, row_scale, rows, n, xs, eval_x, NULL, out5);
    free(xs); free(rows);
}

/* ===== FFN proxy (sd_ffn_proxy.cpp:20-247) ================================================= */
static double dense_dot(const sdo_tensor * W, int64_t row, const float * x, float * buf) {
    sdso_read_row(W, row, buf);
    double acc = 0.0;
    for (int64_t i = 0; i < W->ne[0]; ++i) acc += (double) buf[i] * (double) x[i];
    return acc;
}

int sdso_ffn_proxy(int kind, const sdo_tensor * Wg, const sdo_tensor * Wu, const sdo_tensor * Wd,
                   const int32_t * idx, const float * val, const sdso_base * base, int64_t K,
                   int64_t eval_x, int64_t eval_cols, int seed, double * out8) {
    memset(out8, 0, 8 * sizeof(double));
    if (!Wg || !Wu || !Wd || !Wg->data || !Wu->data || !Wd->data || eval_x <= 0 || kind == 2) return 0;
    const int64_t n_in = Wg->ne[0], n_hidden = Wg->ne[1], n_out = Wd->ne[1];
    const int64_t n_x = eval_x < 128 ? eval_x : 128;
    const int64_t cap = eval_cols > 0 ? eval_cols : 128;
    const int64_t n_hs = n_hidden < cap ? n_hidden : cap, n_os = n_out < cap ? n_out : cap;
    const int have_base = base && base->L > 0 && base->B > 0;
    const int tall = n_out >= n_in;          /* sic: the proxy compares W_down's rows with n_in (:55) */

    sdso_rng g;
    sdso_rng_seed(&g, (uint32_t) seed);
    int64_t * hid, * outs;
    draw_rows(&g, n_hidden, n_hs, &hid);
    draw_rows(&g, n_out, n_os, &outs);

    float * x = (float *) malloc((size_t) n_in * sizeof(float));
    const int64_t down_in = Wd->ne[0];
    float * buf = (float *) malloc((size_t) (n_in > down_in ? n_in : down_in) * sizeof(float));
    double * gt = (double *) malloc((size_t) (4 * n_hs) * sizeof(double)), * gh = gt + n_hs, * ut = gh + n_hs, * uh = ut + n_hs;
    double * cs = (double *) malloc((size_t) (3 * n_x) * sizeof(double)), * l2 = cs + n_x, * lnr = l2 + n_x;
    int64_t n_lnr = 0;
    g.have_saved = 0;
    for (int64_t xi = 0; xi < n_x; ++xi) {
        for (int64_t i = 0; i < n_in; ++i) x[i] = normal_draw(&g);   /* one distribution object for all x */
        for (int64_t hi = 0; hi < n_hs; ++hi) {
            const int64_t h = hid[hi];
            gt[hi] = dense_dot(Wg, h, x, buf);
            ut[hi] = dense_dot(Wu, h, x, buf);
            gh[hi] = gt[hi];
            uh[hi] = ut[hi];
            if (kind == 0 || kind == 1) {
                double hat = 0.0;
                if (have_base) base_outputs(base, tall, n_in, x, &h, 1, &hat);
                double acc = 0.0;
                for (int64_t r = 0; r < K; ++r) {
                    const int32_t ii = idx[h * K + r];
                    if (ii < 0 || ii >= (int32_t) n_in) continue;
                    acc += (double) val[h * K + r] * (double) x[ii];
                }
                hat = hat + acc;
                if (kind == 0) gh[hi] = hat; else uh[hi] = hat;
            }
        }
        double dot = 0.0, n1 = 0.0, n2 = 0.0, d2 = 0.0;
        for (int64_t oi = 0; oi < n_os; ++oi) {
            sdso_read_row(Wd, outs[oi], buf);
            double yt = 0.0, yh = 0.0;
            for (int64_t hi = 0; hi < n_hs; ++hi) {
                const double at = gt[hi] / (1.0 + exp(-gt[hi])), ah = gh[hi] / (1.0 + exp(-gh[hi]));
                const double w = hid[hi] < down_in ? (double) buf[hid[hi]] : 0.0;
                yt += at * ut[hi] * w;
                yh += ah * uh[hi] * w;
            }
            dot += yt * yh; n1 += yt * yt; n2 += yh * yh;
            const double d = yt - yh;
            d2 += d * d;
        }
        const double den = sqrt(n1 > 1e-20 ? n1 : 1e-20) * sqrt(n2 > 1e-20 ? n2 : 1e-20);
        cs[xi] = den > 0.0 ? dot / den : 0.0;
        l2[xi] = sqrt(d2 > 0.0 ? d2 : 0.0);
        const double nr = n1 > 0.0 ? sqrt(n2 > 1e-20 ? n2 : 1e-20) / sqrt(n1 > 1e-20 ? n1 : 1e-20) : 0.0;
        if (nr > 0.0) lnr[n_lnr++] = log(nr);
    }
    double a = 0.0, b = 0.0, c = 0.0;
    for (int64_t i = 0; i < n_x; ++i) a += cs[i];
    for (int64_t i = 0; i < n_x; ++i) b += l2[i];
    for (int64_t i = 0; i < n_lnr; ++i) c += lnr[i];
    out8[0] = n_x ? a / (double) n_x : 0.0;
    out8[1] = sdso_percentile(cs, n_x, 0.05);
    out8[2] = n_x ? b / (double) n_x : 0.0;
    out8[3] = sdso_percentile(l2, n_x, 0.95);
    out8[4] = n_lnr ? c / (double) n_lnr : 0.0;
    out8[5] = sdso_percentile(lnr, n_lnr, 0.95);
    out8[6] = (double) n_x;
    out8[7] = (double) n_os;
    free(cs); free(gt); free(buf); free(x); free(outs); free(hid);
    return 1;
}

/* ===== cost model (sd_cost.cpp:7-40) ======================================================= */
void sdso_cost(int64_t L, int64_t B, int64_t n_in, int64_t n_out, int64_t K, int row_scale, double * o) {
    memset(o, 0, 8 * sizeof(double));
    if (n_in <= 0 || n_out <= 0 || K <= 0) return;
    o[2] = 2.0 * (double) n_in * (double) n_out;
    o[4] = 2.0 * (double) K * (double) n_out;
    o[5] = row_scale ? (double) n_out : 0.0;
    if (L > 0 && B > 0) {
        o[0] = (double) L; o[1] = (double) B;
        const double l = (double) L, lg = log2(l > 2.0 ? l : 2.0);
        o[3] = (double) B * (2.0 * l * lg + 3.0 * l + (n_out >= n_in ? 0.0 : l));
    }
    o[6] = o[3] + o[4] + o[5];
    o[7] = o[2] > 0.0 ? o[6] / o[2] : 0.0;
}

/* ---- --delta-norm-clamp-down (see sd_scan_oracle.h) --------------------------------------------- */
double sdso_tensor_l2_norm(const sdo_tensor * W) {
    const int64_t n_in = W->ne[0], n_out = W->ne[1];
    if (n_in <= 0 || n_out <= 0) return 0.0;
    float * col = (float *) malloc((size_t) n_in * sizeof(float));
    double sum = 0.0;
    for (int64_t j = 0; j < n_out; ++j) {
        sdso_read_row(W, j, col);
        for (int64_t i = 0; i < n_in; ++i) {
            const double dv = (double) col[i];
            sum += dv * dv;
        }
    }
    free(col);
    return sum > 0.0 ? sqrt(sum) : 0.0;
}

double sdso_delta_clamp(float * val, int64_t n, double w_l2_norm, double clamp) {
    if (!(clamp > 0.0) || n <= 0 || !(w_l2_norm > 0.0)) return 1.0;
    double delta_norm = 0.0;
    for (int64_t i = 0; i < n; ++i) delta_norm += (double) val[i] * (double) val[i];
    delta_norm = delta_norm > 0.0 ? sqrt(delta_norm) : 0.0;
    const double limit = clamp * w_l2_norm;
    if (limit > 0.0 && delta_norm > limit) {
        const double scale = limit / delta_norm;
        for (int64_t i = 0; i < n; ++i) val[i] = (float) (val[i] * scale);
        return scale;
    }
    return 1.0;
}
