/*
 * sd_oracle.h — CPU oracle for the SeedDelta FFN hot path.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C restatement of the reference's CPU algorithm
 * (frudas24/llama.cpp, src/llama-seeddelta.cpp). It exists so the HIP path can be checked
 * against it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it. Nothing in the product path (llama.cpp_amd/, include/) links or calls it.
 *
 * Parity status: PINNED — checked bit-for-bit against the reference's own translation unit
 * compiled from /root/reference (oracle/_ref, see oracle/Makefile + ref_harness.cpp) and
 * against the golden vectors that build produced (tests/golden/, made by
 * oracle/make_golden.py). The reference ships no golden vectors of its own for this path
 * (SURVEY.md §4), only the analytical identities A1/A2/A4 which tests/ also checks.
 *
 * Build: gcc -O3 -ffp-contract=off (the reference TU is built -O3 without -march, so no FMA
 * contraction; -ffp-contract=off states that explicitly).
 */
#ifndef SD_ORACLE_H
#define SD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* type codes = enum ggml_type (ggml/include/ggml.h:385-411) */
enum { SDO_F32 = 0, SDO_F16 = 1, SDO_Q8_0 = 8, SDO_I16 = 25, SDO_I32 = 26 };

typedef struct sdo_tensor {
    void *  data;      /* NULL => tensor absent */
    int32_t type;
    int32_t n_dims;
    int64_t ne[4];
    int64_t nb[4];
} sdo_tensor;

/* Arguments of one op evaluation: the sources the reference reads from dst->src[0..8]
 * (src/llama-seeddelta.cpp:585-588, :1017-1025), the destination and the dense weight
 * passed as userdata (w_ref). For the COO ops idx/val are d_idx/d_val; for the block ops
 * they are b_idx/b_val. */
typedef struct sdo_op_args {
    sdo_tensor x;          /* [n_in, n_tokens] F32/F16 */
    sdo_tensor idx;
    sdo_tensor val;
    sdo_tensor row_scale;  /* optional */
    sdo_tensor d1, d2, d3; /* base, [L,B]; d1 required for base ops */
    sdo_tensor p1, p2;     /* base perms, optional */
    sdo_tensor w_ref;      /* dense weight [n_in, n_out] F32/F16/Q8_0, optional (NaN repair) */
    sdo_tensor dst;        /* [n_out, n_tokens] F32 */
} sdo_op_args;

/* fp16 <-> fp32 (ggml/src/ggml-impl.h:378-425; values identical, code independent) */
float    sdo_fp16_to_fp32(uint16_t h);
uint16_t sdo_fp32_to_fp16(float f);

/* Gate activation of a parallel-gated FFN, y[i] = act(g[i]) * u[i], as build_ffn emits it between the
 * SeedDelta nodes (src/llama-graph.cpp:867-881: ggml_swiglu_split / ggml_geglu_split).
 *   act 0: ggml_vec_swiglu_f32 scalar form (ggml/src/ggml-cpu/vec.cpp:447-449) with ggml_silu_f32
 *          (vec.h:1061): g/(1+expf(-g)) * u. (The reference's x86 SIMD bodies use a polynomial expf:
 *          equal to this within ~2e-7 relative; tests carry that tolerance.)
 *   act 1: ggml_vec_geglu_f32, GGML_GELU_FP16 form (vec.h:1430-1443) over the f16 table of
 *          ggml-cpu.c:3657-3667. sdo_gelu_table fills the 65536-entry table. */
void sdo_gelu_table(uint16_t * table);
void sdo_glu(int act, int64_t n, const float * g, const float * u, float * y, const uint16_t * gelu_table);

/* hadamard_transform, scalar path (src/llama-seeddelta.cpp:484-493) */
void sdo_hadamard(float * data, int64_t n);

/* apply_base_block (src/llama-seeddelta.cpp:497-574). tmp: L floats of scratch. */
void sdo_apply_base_block(const float * x_in, float * y_out, float * tmp, int64_t L, int64_t b,
                          const sdo_tensor * d1, const sdo_tensor * d2, const sdo_tensor * d3,
                          const sdo_tensor * p1, const sdo_tensor * p2);

/* The four custom-op callbacks, one (ith, nth) slice each, exactly as ggml-cpu would invoke
 * them (ggml/src/ggml-cpu/ops.cpp:10141-10149). Results do not depend on nth. */
void sdo_coo_op       (const sdo_op_args * a, int ith, int nth);  /* :582  */
void sdo_block_op     (const sdo_op_args * a, int ith, int nth);  /* :714  */
void sdo_base_op      (const sdo_op_args * a, int ith, int nth);  /* :1014 */
void sdo_base_block_op(const sdo_op_args * a, int ith, int nth);  /* :1178 */

/* Convenience: run one op over all slices with `nth` POSIX threads (nth<=1: inline).
 * kind: 0 coo, 1 block, 2 base+coo, 3 base+block. */
void sdo_run(int kind, const sdo_op_args * a, int nth);

/* count of NaN/Inf outputs repaired since the last reset (llama_seeddelta_nan_count, :77) */
int64_t sdo_nan_count(void);
void    sdo_nan_count_reset(void);

/* Dense W0 element implied by the base parameters (depth 2, no perm2, bijective perm1):
 * tall: W0[o,i] = d3[p,b] * G_b[(p ^ inv_b[i]) & (L-1)] * d1[i,b],  b=o/L, p=o-bL
 * wide: W0[o,i] = d3[o,b] * G_b[(o ^ inv_b[q]) & (L-1)] * d1[q,b],  b=i/L, q=i-bL
 * with G_b = hadamard(d2_b) (reference butterfly order) and inv_b = perm1_b^-1.
 * Product order: (d3 * G) * d1, each multiply rounded to fp32.
 * (builder form tools/seeddelta-build/sd_eval.cpp:506,528,714-716.)
 * G: [L*B] floats, inv: [L*B] int32, as produced by sdo_w0_tables. */
void  sdo_w0_tables(const sdo_tensor * d2, const sdo_tensor * p1, int64_t L, int64_t B,
                    float * G, int32_t * inv);
float sdo_w0_elem(const sdo_tensor * d1, const sdo_tensor * d3, const float * G, const int32_t * inv,
                  int64_t L, int64_t n_in, int64_t n_out, int64_t o, int64_t i);

/* dequantize one Q8_0 row (ggml/src/ggml-quants.c:401-414): y[j] = qs[j] * fp32(d) */
void sdo_dequant_row_q8_0(const void * row, float * y, int64_t k);

#ifdef __cplusplus
}
#endif
#endif
