/*
 * sd_scan_oracle.h — CPU restatement of the builder's candidate-evaluation loop
 * (tools/seeddelta-build of the reference). TEST INFRASTRUCTURE ONLY: used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product never links it.
 *
 * Pinned against the reference's own sd_eval.cpp / sd_cost.cpp / sd_ffn_proxy.cpp compiled
 * unmodified into oracle/_ref/libsdbuildref.so (tests/test_scan_oracle.py), and against libstdc++ 11
 * for the RNG restatement. The run_trial column loop (sd_build.cpp:463-703) cannot be compiled here
 * (see oracle/ref_build_harness.cpp): it is restated from the source, its selection calls and the
 * evals that consume its outputs are pinned, the row-scale least squares itself is parity-unpinned.
 */
#ifndef SD_SCAN_ORACLE_H
#define SD_SCAN_ORACLE_H

#include <stdint.h>
#include "sd_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- libstdc++-compatible random numbers ---------------------------------------------------- */
typedef struct sdso_rng { uint32_t mt[624]; int32_t pos; int32_t have_saved; float saved; } sdso_rng;
void     sdso_rng_seed(sdso_rng * g, uint32_t seed);                 /* std::mt19937(seed)            */
uint32_t sdso_rng_u32(sdso_rng * g);                                 /* operator()                    */
void     sdso_shuffle_i64(sdso_rng * g, int64_t * v, int64_t n);     /* std::shuffle (GCC 11)         */
void     sdso_shuffle_i32(sdso_rng * g, int32_t * v, int64_t n);
/* n draws of ONE std::normal_distribution<float>(0,1) object (the pair cache is reset first) */
void     sdso_normal_f32(sdso_rng * g, int64_t n, float * out);

/* ---- base fit: [B][L] arrays, same field order as base_arrays in ref_build_harness.cpp --------- */
typedef struct sdso_base {
    int64_t L, B;
    const float * d1, * d2, * d3, * h2;
    const int32_t * perm1, * perm1_inv;
} sdso_base;

void    sdso_read_row(const sdo_tensor * W, int64_t row, float * out);
/* selection: K largest by score, descending; equal scores ordered by ascending index (the
 * reference leaves the order of ties to std::sort — see tests for how ties are compared) */
int64_t sdso_topk_abs(const float * v, const float * w_scale, int64_t n, int64_t K, int32_t * out);
int64_t sdso_topk_blocks(const float * v, const float * w_scale, int64_t n, int64_t block, int64_t keep, int32_t * out);
void    sdso_fit_dims(int64_t n_in, int64_t n_out, int64_t * L, int64_t * B);
void    sdso_fit_base(const sdo_tensor * W, int64_t max_samples, const float * w_scale, int perm_trials, sdso_rng * g,
                      float * d2, float * h2, int32_t * perm1, int32_t * perm1_inv);
void    sdso_apply_base_block(const sdso_base * a, int64_t b, const float * x, float * y);
/* run_trial column loop. scheme 0: idx/val [n_out][K]; 1: idx [n_out][keep], val [n_out][keep][block] */
void    sdso_select_rows(const sdo_tensor * W, const sdso_base * base, const float * w_scale, int scheme,
                         int64_t K_or_keep, int64_t block, int32_t * idx, float * val, float * row_scale /* or NULL */,
                         int nth);
/* the same loop for a list of rows; outputs land in the rows' own slots of the [n_out] arrays */
void    sdso_select_rows_list(const sdo_tensor * W, const sdso_base * base, const float * w_scale, int scheme,
                              int64_t K_or_keep, int64_t block, int32_t * idx, float * val, float * row_scale /* or NULL */,
                              const int64_t * rows, int64_t n_rows);
/* weight-space metrics on the given rows; per_row (optional) receives w_norm2, wh_norm2, dot per row;
 * out5 = rel_l2_mean, rel_l2_p95, cos_mean, cos_p05, norm_ratio_mean */
void    sdso_eval_w_rows(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                         int64_t K_or_keep, int64_t block, const float * row_scale, const float * w_scale,
                         const int64_t * rows, int64_t n_rows, double * per_row, double * out5);
void    sdso_eval_w(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                    int64_t K_or_keep, int64_t block, const float * row_scale, const float * w_scale,
                    int64_t eval_cols, sdso_rng * g, double * out5);
/* activation-space metrics; xs = [n_x][n_in] already drawn; y (optional) receives y_true,y_hat [n_x][n_rows][2] */
void    sdso_eval_x_rows(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                         int64_t K_or_keep, int64_t block, const float * row_scale,
                         const int64_t * rows, int64_t n_rows, const float * xs, int64_t n_x, double * y, double * out5);
void    sdso_eval_x(const sdo_tensor * W, const sdso_base * base, int scheme, const int32_t * idx, const float * val,
                    int64_t K_or_keep, int64_t block, const float * row_scale, const float * x_scale,
                    int64_t eval_cols, int64_t eval_x, sdso_rng * g, double * out5);
/* FFN proxy, COO only; out8 = cos_mean, cos_p05, l2_mean, l2_p95, lnr_mean, lnr_p95, eval_x, eval_out */
int     sdso_ffn_proxy(int kind, const sdo_tensor * Wg, const sdo_tensor * Wu, const sdo_tensor * Wd,
                       const int32_t * idx, const float * val, const sdso_base * base, int64_t K,
                       int64_t eval_x, int64_t eval_cols, int seed, double * out8);
void    sdso_cost(int64_t L, int64_t B, int64_t n_in, int64_t n_out, int64_t K, int row_scale, double * out8);
double  sdso_percentile(const double * v, int64_t n, double p);
/* --delta-norm-clamp-down: ||W||_2 as sd_tensor_l2_norm (sd_build.cpp:29-46: one running double sum over
 * rows and elements), then the clamp of the residual values (:705-720). Returns the scale applied (1.0
 * if none). PARITY UNPINNED against a compiled reference: sd_build.cpp does not build here. */
double  sdso_tensor_l2_norm(const sdo_tensor * W);
double  sdso_delta_clamp(float * val, int64_t n, double w_l2_norm, double clamp);

#ifdef __cplusplus
}
#endif
#endif
