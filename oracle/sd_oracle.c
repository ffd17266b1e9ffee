/*
 * sd_oracle.c — CPU oracle (test infrastructure; see sd_oracle.h for the rules).
 *
 * Restates, in plain C, what frudas24/llama.cpp src/llama-seeddelta.cpp computes. Every
 * function names the reference lines it follows. The arithmetic is kept in the reference's
 * order (fp32 multiply then fp32 add, sequential over r / (bi, j), base term first) so the
 * results are bit-identical to the reference TU built with its stock flags.
 */
#include "sd_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* fp16 <-> fp32. Value-identical to ggml_compute_fp16_to_fp32 / _fp32_to_fp16            */
/* (ggml/src/ggml-impl.h:378-425); written from the IEEE-754 binary16 definition.         */
/* ------------------------------------------------------------------------------------ */
float sdo_fp16_to_fp32(uint16_t h) {
    const uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
    const uint32_t e    = (h >> 10) & 0x1Fu;
    uint32_t       m    = h & 0x3FFu;
    uint32_t bits;
    if (e == 0x1Fu) {                       /* inf / nan */
        bits = sign | 0x7F800000u | (m << 13);
    } else if (e != 0) {                    /* normal */
        bits = sign | ((e + 112u) << 23) | (m << 13);
    } else if (m == 0) {                    /* signed zero */
        bits = sign;
    } else {                                /* subnormal: renormalise */
        uint32_t ee = 113u;
        while ((m & 0x400u) == 0) { m <<= 1; --ee; }
        bits = sign | (ee << 23) | ((m & 0x3FFu) << 13);
    }
    float f;
    memcpy(&f, &bits, sizeof f);
    return f;
}

uint16_t sdo_fp32_to_fp16(float f) {
    /* round-to-nearest-even, NaN -> 0x7E00 with sign (ggml-impl.h:401-425 behaviour) */
    uint32_t x;
    memcpy(&x, &f, sizeof x);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000u);
    const uint32_t ax   = x & 0x7FFFFFFFu;
    if (ax > 0x7F800000u) return (uint16_t) (sign | 0x7E00u);       /* nan */
    if (ax >= 0x47800000u) return (uint16_t) (sign | 0x7C00u);      /* >= 65536 -> inf (65520 rounds below) */
    if (ax >= 0x38800000u) {                                        /* normal half range */
        uint32_t mant = ax & 0x7FFFFFu;
        uint32_t e    = (ax >> 23) - 112u;
        uint32_t half = (e << 10) | (mant >> 13);
        const uint32_t rem = mant & 0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half; /* may carry into exponent / inf */
        return (uint16_t) (sign | half);
    }
    if (ax < 0x33000000u) return sign;                              /* < 2^-25 -> 0 */
    {                                                               /* subnormal half */
        const uint32_t e     = ax >> 23;                            /* 102..112 */
        const uint32_t mant  = (ax & 0x7FFFFFu) | 0x800000u;
        const uint32_t shift = 126u - e;                            /* 14..24 */
        uint32_t half = mant >> shift;
        const uint32_t rem  = mant & ((1u << shift) - 1u);
        const uint32_t mid  = 1u << (shift - 1u);
        if (rem > mid || (rem == mid && (half & 1u))) ++half;
        return (uint16_t) (sign | half);
    }
}

/* ------------------------------------------------------------------------------------ */
/* strided readers (read_f16_or_f32*, read_idx_i16_i32, read_base_*; :22-74)             */
/* ------------------------------------------------------------------------------------ */
static inline float rd_f(const sdo_tensor * t, const uint8_t * base, int64_t i0, int64_t i1) {
    const uint8_t * p = base + i0 * t->nb[0] + i1 * t->nb[1];
    if (t->type == SDO_F16) {
        uint16_t h;
        memcpy(&h, p, 2);
        return sdo_fp16_to_fp32(h);
    }
    float v;
    memcpy(&v, p, 4);
    return v;
}

static inline int32_t rd_i(const sdo_tensor * t, const uint8_t * base, int64_t i0, int64_t i1) {
    const uint8_t * p = base + i0 * t->nb[0] + i1 * t->nb[1];
    if (t->type == SDO_I16) {
        int16_t v;
        memcpy(&v, p, 2);
        return (int32_t) v;
    }
    int32_t v;
    memcpy(&v, p, 4);
    return v;
}

static inline int present(const sdo_tensor * t) { return t && t->data != NULL; }

/* ggml_n_dims (ggml/src/ggml.c): trailing dimensions of extent 1 do not count, so a single-token
 * x [n_in, 1] is 1-D. The reference's dense fallback therefore refuses to run at T = 1 and the
 * NaN guard writes 0.0 there (src/llama-seeddelta.cpp:323-325) — restated as is. */
static inline int ggml_like_n_dims(const sdo_tensor * t) {
    for (int i = 3; i >= 1; --i) {
        if (t->ne[i] > 1) return i + 1;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------ */
/* NaN guard + dense fallback (:294-392)                                                  */
/* ------------------------------------------------------------------------------------ */
static int64_t g_nan_count = 0;
static int     g_nan_warned = 0;

int64_t sdo_nan_count(void) { return __atomic_load_n(&g_nan_count, __ATOMIC_RELAXED); }
void    sdo_nan_count_reset(void) {
    __atomic_store_n(&g_nan_count, 0, __ATOMIC_RELAXED);
    __atomic_store_n(&g_nan_warned, 0, __ATOMIC_RELAXED);
}

void sdo_dequant_row_q8_0(const void * row, float * y, int64_t k) {
    /* block_q8_0 = { fp16 d; int8 qs[32]; } (ggml/src/ggml-common.h:219-224) */
    const uint8_t * p = (const uint8_t *) row;
    const int64_t nblk = k / 32;
    for (int64_t ib = 0; ib < nblk; ++ib) {
        uint16_t dh;
        memcpy(&dh, p + ib * 34, 2);
        const float d = sdo_fp16_to_fp32(dh);
        const int8_t * qs = (const int8_t *) (p + ib * 34 + 2);
        for (int j = 0; j < 32; ++j) {
            y[ib * 32 + j] = (float) qs[j] * d;
        }
    }
}

/* one row of the dense weight as fp32 (llama_seeddelta_read_weight_col_f32, :234-292) */
static int read_weight_row(const sdo_tensor * w, int64_t col, float * out) {
    if (!present(w) || ggml_like_n_dims(w) != 2) return 0;
    const int64_t n_in = w->ne[0], n_out = w->ne[1];
    if (n_in <= 0 || col < 0 || col >= n_out) return 0;
    const uint8_t * base = (const uint8_t *) w->data + col * w->nb[1];
    if (w->type == SDO_F32) {
        for (int64_t i = 0; i < n_in; ++i) memcpy(&out[i], base + i * w->nb[0], 4);
        return 1;
    }
    if (w->type == SDO_F16) {
        for (int64_t i = 0; i < n_in; ++i) {
            uint16_t h;
            memcpy(&h, base + i * w->nb[0], 2);
            out[i] = sdo_fp16_to_fp32(h);
        }
        return 1;
    }
    if (w->type == SDO_Q8_0) {
        sdo_dequant_row_q8_0(base, out, n_in);
        return 1;
    }
    return 0;
}

/* llama_seeddelta_dense_fallback (:294-361): y = sum_i w[i,o]*x[i,t] accumulated in double */
static int dense_fallback(const sdo_tensor * w, const sdo_tensor * x, int64_t o, int64_t t, float * y_out) {
    if (!present(w) || !present(x)) return 0;
    if (ggml_like_n_dims(x) < 2 || ggml_like_n_dims(w) != 2) return 0;
    const int64_t n_in = x->ne[0];
    if (w->ne[0] != n_in) return 0;
    if (o < 0 || o >= w->ne[1]) return 0;
    if (t < 0 || t >= x->ne[1]) return 0;
    float * row = (float *) malloc((size_t) n_in * sizeof(float));
    if (!row) return 0;
    if (!read_weight_row(w, o, row)) { free(row); return 0; }
    const uint8_t * xd = (const uint8_t *) x->data + t * x->nb[1];
    double acc = 0.0;
    for (int64_t i = 0; i < n_in; ++i) {
        const float xv = rd_f(x, xd, i, 0);
        acc += (double) row[i] * (double) xv;
    }
    free(row);
    *y_out = (float) acc;
    return isfinite(*y_out) ? 1 : 0;
}

/* llama_seeddelta_nan_guard (:363-392) */
static inline float nan_guard(const sdo_op_args * a, int64_t o, int64_t t, float y) {
    if (isfinite(y)) return y;
    float y_fb = 0.0f;
    if (dense_fallback(&a->w_ref, &a->x, o, t, &y_fb)) {
        y = y_fb;
    } else {
        y = 0.0f;
    }
    __atomic_fetch_add(&g_nan_count, 1, __ATOMIC_RELAXED);
    if (!__atomic_exchange_n(&g_nan_warned, 1, __ATOMIC_RELAXED)) {
        fprintf(stderr, "sd_oracle: NaN/Inf detected in op output (repaired)\n");
    }
    return y;
}

/* ------------------------------------------------------------------------------------ */
/* base transform                                                                         */
/* ------------------------------------------------------------------------------------ */
void sdo_hadamard(float * data, int64_t n) {
    /* :484-493 — stages len = 1,2,4,...; butterfly (u+v, u-v) */
    for (int64_t len = 1; len < n; len <<= 1) {
        for (int64_t i = 0; i < n; i += (len << 1)) {
            for (int64_t j = 0; j < len; ++j) {
                const float u = data[i + j];
                const float v = data[i + j + len];
                data[i + j]       = u + v;
                data[i + j + len] = u - v;
            }
        }
    }
}

static void gather_perm(const sdo_tensor * p, float * y, float * tmp, int64_t L, int64_t b) {
    /* :522-531 / :554-563 — tmp[i] = y[perm[i]], out-of-range entries read element 0 */
    const uint8_t * pd = (const uint8_t *) p->data;
    for (int64_t i = 0; i < L; ++i) {
        int32_t src = rd_i(p, pd, i, b);
        if (src < 0 || src >= (int32_t) L) src = 0;
        tmp[i] = y[src];
    }
    memcpy(y, tmp, (size_t) L * sizeof(float));
}

static void scale_diag(const sdo_tensor * d, float * y, int64_t L, int64_t b) {
    const uint8_t * dd = (const uint8_t *) d->data;
    for (int64_t i = 0; i < L; ++i) y[i] *= rd_f(d, dd, i, b);
}

void sdo_apply_base_block(const float * x_in, float * y_out, float * tmp, int64_t L, int64_t b,
                          const sdo_tensor * d1, const sdo_tensor * d2, const sdo_tensor * d3,
                          const sdo_tensor * p1, const sdo_tensor * p2) {
    /* :497-574 — v = D3 * H * [D2 * [P2] * H] * P1 * D1 * x */
    memcpy(y_out, x_in, (size_t) L * sizeof(float));
    scale_diag(d1, y_out, L, b);                       /* d1 is required by the callers */
    if (present(p1)) gather_perm(p1, y_out, tmp, L, b);
    sdo_hadamard(y_out, L);
    if (!present(d2) && !present(p2)) {                /* depth 1 (:537-544) */
        if (present(d3)) scale_diag(d3, y_out, L, b);
        return;
    }
    if (present(d2)) scale_diag(d2, y_out, L, b);
    if (present(p2)) gather_perm(p2, y_out, tmp, L, b);
    sdo_hadamard(y_out, L);
    if (present(d3)) scale_diag(d3, y_out, L, b);
}

/* ------------------------------------------------------------------------------------ */
/* residual dot products                                                                  */
/* ------------------------------------------------------------------------------------ */
/* COO row: y += val[r,o] * xv[idx[r,o]] for r = 0..K-1, invalid idx skipped (:691-699).
 * xcol is either the strided x column (x_is_tensor) or a staged fp32 vector.           */
static inline float coo_row(const sdo_op_args * a, int64_t o, float y,
                            const float * x_staged, int64_t t) {
    const sdo_tensor * idx = &a->idx;
    const sdo_tensor * val = &a->val;
    const int64_t K = idx->ne[0];
    const int64_t n_in = a->x.ne[0];
    const uint8_t * idx_col = (const uint8_t *) idx->data + o * idx->nb[1];
    const uint8_t * val_col = (const uint8_t *) val->data + o * val->nb[1];
    const uint8_t * xd = (const uint8_t *) a->x.data;
    for (int64_t r = 0; r < K; ++r) {
        const int32_t ii = rd_i(idx, idx_col, r, 0);
        if (ii < 0 || ii >= (int32_t) n_in) continue;
        const float vv = rd_f(val, val_col, r, 0);
        const float xv = x_staged ? x_staged[ii] : rd_f(&a->x, xd, ii, t);
        y += vv * xv;
    }
    return y;
}

/* block row: blocks with blk<0 or blk*block>=n_in skipped, last block clipped (:927-944) */
static inline float block_row(const sdo_op_args * a, int64_t o, float y,
                              const float * x_staged, int64_t t) {
    const sdo_tensor * idx = &a->idx;
    const sdo_tensor * val = &a->val;
    const int64_t nb = idx->ne[0];
    const int64_t block = val->ne[0];
    const int64_t n_in = a->x.ne[0];
    const uint8_t * idx_col = (const uint8_t *) idx->data + o * idx->nb[1];
    const uint8_t * val_col = (const uint8_t *) val->data + o * val->nb[2];
    const uint8_t * xd = (const uint8_t *) a->x.data;
    for (int64_t bi = 0; bi < nb; ++bi) {
        const int32_t blk = rd_i(idx, idx_col, bi, 0);
        if (blk < 0) continue;
        const int64_t in0 = (int64_t) blk * block;
        if (in0 < 0 || in0 >= n_in) continue;
        const int64_t in1 = (in0 + block < n_in) ? in0 + block : n_in;
        const uint8_t * val_blk = val_col + bi * val->nb[1];
        for (int64_t ii = in0; ii < in1; ++ii) {
            const float vv = rd_f(val, val_blk, ii - in0, 0);
            const float xv = x_staged ? x_staged[ii] : rd_f(&a->x, xd, ii, t);
            y += vv * xv;
        }
    }
    return y;
}

static inline void store_out(const sdo_op_args * a, int64_t o, int64_t t, float y) {
    /* scale only if != 1 (:645), then NaN guard, then store */
    float scale = 1.0f;
    if (present(&a->row_scale)) scale = rd_f(&a->row_scale, (const uint8_t *) a->row_scale.data, o, 0);
    float out = (scale != 1.0f) ? (y * scale) : y;
    out = nan_guard(a, o, t, out);
    memcpy((uint8_t *) a->dst.data + o * a->dst.nb[0] + t * a->dst.nb[1], &out, 4);
}

/* ------------------------------------------------------------------------------------ */
/* the four ops                                                                           */
/* ------------------------------------------------------------------------------------ */
void sdo_coo_op(const sdo_op_args * a, int ith, int nth) {
    /* :582-706; rows split across threads (:610-611) */
    const int64_t n_out = a->idx.ne[1], T = a->x.ne[1];
    const int64_t o0 = (n_out * ith) / nth, o1 = (n_out * (ith + 1)) / nth;
    for (int64_t o = o0; o < o1; ++o)
        for (int64_t t = 0; t < T; ++t)
            store_out(a, o, t, coo_row(a, o, 0.0f, NULL, t));
}

void sdo_block_op(const sdo_op_args * a, int ith, int nth) {
    /* :714-951; the F32, F16xF16 and generic paths are arithmetically identical */
    const int64_t n_out = a->idx.ne[1], T = a->x.ne[1];
    const int64_t o0 = (n_out * ith) / nth, o1 = (n_out * (ith + 1)) / nth;
    for (int64_t o = o0; o < o1; ++o)
        for (int64_t t = 0; t < T; ++t)
            store_out(a, o, t, block_row(a, o, 0.0f, NULL, t));
}

static void base_common(const sdo_op_args * a, int ith, int nth, int is_block) {
    /* :1014-1176 (COO) and :1178-1360 (block): identical structure */
    const int64_t n_in = a->x.ne[0], T = a->x.ne[1];
    const int64_t n_out = a->idx.ne[1];
    const int64_t L = a->d1.ne[0], B = a->d1.ne[1];
    const uint8_t * xd = (const uint8_t *) a->x.data;
    const int is_tall = n_out >= n_in;                            /* :1073 */

    float * v   = (float *) malloc((size_t) L * 4);
    float * tmp = (float *) malloc((size_t) L * 4);
    if (is_tall) {
        const int64_t o0 = (n_out * ith) / nth, o1 = (n_out * (ith + 1)) / nth;
        float * x_hat = (float *) malloc((size_t) L * 4);
        for (int64_t t = 0; t < T; ++t) {
            for (int64_t i = 0; i < n_in; ++i) x_hat[i] = rd_f(&a->x, xd, i, t);   /* :1091-1096 */
            for (int64_t i = n_in; i < L; ++i) x_hat[i] = 0.0f;
            if (o1 <= o0) continue;
            const int64_t b0 = o0 / L, b1 = (o1 - 1) / L;
            for (int64_t b = b0; b <= b1 && b < B; ++b) {                             /* :1100 */
                sdo_apply_base_block(x_hat, v, tmp, L, b, &a->d1, &a->d2, &a->d3, &a->p1, &a->p2);
                const int64_t bo0 = o0 > b * L ? o0 : b * L;
                const int64_t bo1 = o1 < (b + 1) * L ? o1 : (b + 1) * L;
                for (int64_t o = bo0; o < bo1; ++o) {
                    float y = v[o - b * L];                                           /* base term first */
                    y = is_block ? block_row(a, o, y, x_hat, t) : coo_row(a, o, y, x_hat, t);
                    store_out(a, o, t, y);
                }
            }
        }
        free(x_hat);
    } else {
        const int64_t t0 = (T * ith) / nth, t1 = (T * (ith + 1)) / nth;             /* :1129-1130 */
        float * x_chunk = (float *) malloc((size_t) L * 4);
        float * x_full  = (float *) malloc((size_t) n_in * 4);
        float * y_hat   = (float *) malloc((size_t) L * 4);
        for (int64_t t = t0; t < t1; ++t) {
            for (int64_t i = 0; i < L; ++i) y_hat[i] = 0.0f;
            for (int64_t b = 0; b < B; ++b) {
                const int64_t in0 = b * L;
                const int64_t in1 = (in0 + L < n_in) ? in0 + L : n_in;
                int64_t i = 0;
                for (; i < in1 - in0; ++i) {
                    const float xv = rd_f(&a->x, xd, in0 + i, t);
                    x_chunk[i] = xv;
                    x_full[in0 + i] = xv;
                }
                for (i = (in1 - in0 > 0 ? in1 - in0 : 0); i < L; ++i) x_chunk[i] = 0.0f;
                sdo_apply_base_block(x_chunk, v, tmp, L, b, &a->d1, &a->d2, &a->d3, &a->p1, &a->p2);
                for (i = 0; i < L; ++i) y_hat[i] += v[i];                             /* :1148-1150 */
            }
            for (int64_t o = 0; o < n_out; ++o) {
                float y = y_hat[o];
                y = is_block ? block_row(a, o, y, x_full, t) : coo_row(a, o, y, x_full, t);
                store_out(a, o, t, y);
            }
        }
        free(x_chunk); free(x_full); free(y_hat);
    }
    free(v); free(tmp);
}

void sdo_base_op(const sdo_op_args * a, int ith, int nth)       { base_common(a, ith, nth, 0); }
void sdo_base_block_op(const sdo_op_args * a, int ith, int nth) { base_common(a, ith, nth, 1); }

/* ------------------------------------------------------------------------------------ */
/* threaded runner (what ggml-cpu's threadpool does for GGML_OP_CUSTOM)                   */
/* ------------------------------------------------------------------------------------ */
typedef struct { int kind; const sdo_op_args * a; int ith, nth; } sdo_job;

static void run_slice(int kind, const sdo_op_args * a, int ith, int nth) {
    switch (kind) {
        case 0: sdo_coo_op(a, ith, nth); break;
        case 1: sdo_block_op(a, ith, nth); break;
        case 2: sdo_base_op(a, ith, nth); break;
        default: sdo_base_block_op(a, ith, nth); break;
    }
}

static void * job_main(void * p) {
    sdo_job * j = (sdo_job *) p;
    run_slice(j->kind, j->a, j->ith, j->nth);
    return NULL;
}

void sdo_run(int kind, const sdo_op_args * a, int nth) {
    if (nth <= 1) { run_slice(kind, a, 0, 1); return; }
    pthread_t * th = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) nth);
    sdo_job * jobs = (sdo_job *) malloc(sizeof(sdo_job) * (size_t) nth);
    for (int i = 0; i < nth; ++i) {
        jobs[i].kind = kind; jobs[i].a = a; jobs[i].ith = i; jobs[i].nth = nth;
        pthread_create(&th[i], NULL, job_main, &jobs[i]);
    }
    for (int i = 0; i < nth; ++i) pthread_join(th[i], NULL);
    free(th); free(jobs);
}

/* ------------------------------------------------------------------------------------ */
/* closed-form W0                                                        */
/* ------------------------------------------------------------------------------------ */
void sdo_w0_tables(const sdo_tensor * d2, const sdo_tensor * p1, int64_t L, int64_t B,
                   float * G, int32_t * inv) {
    for (int64_t b = 0; b < B; ++b) {
        float * g = G + b * L;
        for (int64_t i = 0; i < L; ++i) g[i] = present(d2) ? rd_f(d2, (const uint8_t *) d2->data, i, b) : 1.0f;
        sdo_hadamard(g, L);
        int32_t * iv = inv + b * L;
        for (int64_t i = 0; i < L; ++i) iv[i] = -1;
        for (int64_t i = 0; i < L; ++i) {
            int32_t src = present(p1) ? rd_i(p1, (const uint8_t *) p1->data, i, b) : (int32_t) i;
            if (src < 0 || src >= (int32_t) L) src = 0;
            iv[src] = (int32_t) i;    /* (P1 z)[i] = z[perm[i]]  =>  inv[perm[i]] = i */
        }
    }
}

float sdo_w0_elem(const sdo_tensor * d1, const sdo_tensor * d3, const float * G, const int32_t * inv,
                  int64_t L, int64_t n_in, int64_t n_out, int64_t o, int64_t i) {
    int64_t b, p, q;
    if (n_out >= n_in) { b = o / L; p = o - b * L; q = i; }
    else               { b = i / L; q = i - b * L; p = o; }
    const float g  = G[b * L + ((p ^ (int64_t) inv[b * L + q]) & (L - 1))];
    const float s3 = present(d3) ? rd_f(d3, (const uint8_t *) d3->data, p, b) : 1.0f;
    const float s1 = rd_f(d1, (const uint8_t *) d1->data, q, b);
    const float t  = s3 * g;
    return t * s1;
}

/* ---- gate activation (see sd_oracle.h) ------------------------------------------------------- */
void sdo_gelu_table(uint16_t * table) {
    static const float GELU_COEF_A = 0.044715f;
    static const float SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    for (uint32_t i = 0; i < (1u << 16); ++i) {
        const float x = sdo_fp16_to_fp32((uint16_t) i);
        table[i] = sdo_fp32_to_fp16(0.5f*x*(1.0f + tanhf(SQRT_2_OVER_PI*x*(1.0f + GELU_COEF_A*x*x))));
    }
}

void sdo_glu(int act, int64_t n, const float * g, const float * u, float * y, const uint16_t * gelu_table) {
    for (int64_t i = 0; i < n; ++i) {
        if (act == 0) {
            y[i] = (g[i] / (1.0f + expf(-g[i]))) * u[i];
        } else if (g[i] <= -10.0f) {
            y[i] = 0.0f;
        } else if (g[i] >= 10.0f) {
            y[i] = g[i] * u[i];
        } else {
            y[i] = sdo_fp16_to_fp32(gelu_table[sdo_fp32_to_fp16(g[i])]) * u[i];
        }
    }
}
