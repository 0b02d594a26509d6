/*
 * oracle.h -- CPU restatement of the PhantomFHE RNS polynomial-arithmetic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there
 * only as the checker.  The product (phantom-fhe_amd/) has its own, independent host precompute
 * and its own HIP kernels and never links or imports this code.
 *
 * PARITY STATUS: "parity unpinned against an executed reference".  The reference's host code
 * cannot be compiled in this image without writing stand-in CUDA headers
 * (include/host/defines.h:34 includes <cuda_runtime_api.h>), the reference's tests contain no
 * golden vectors (test/ntt_test.cu:71-122 is a round trip on constants), and upstream SEAL is
 * absent.  The oracle is pinned instead by (i) the literal constants the reference embeds
 * (src/host/globals.cu:71 default primes), (ii) the values SURVEY.md 8(c) records from the
 * survey's run of the reference host code, (iii) the mathematical definition of each operation
 * (direct O(N^2) evaluation, CRT recomposition with Python big integers), since every stored
 * output of the path is a canonical residue in [0,q) and therefore unique.
 *
 * Every function cites the reference file:line whose semantics it restates.
 */
#ifndef PHANTOM_ORACLE_H
#define PHANTOM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- number theory / host precompute (src/host/numth.cu, modulus.cu, ntt.cu) ---- */
int orc_is_prime(uint64_t v);
/* get_primes: src/host/numth.cu:207-233. Returns 0 on success. */
int orc_get_primes(uint64_t ntt_size, int bit_size, size_t count, uint64_t *out);
/* CoeffModulus::Create: src/host/modulus.cu:82-111 (same-size primes handed out from the back). */
int orc_coeff_modulus_create(uint64_t n, const int *bit_sizes, size_t count, uint64_t *out);
/* Modulus::set_value const_ratio = floor(2^128/q): src/host/modulus.cu:15-48 */
void orc_const_ratio(uint64_t q, uint64_t ratio[2]);
/* try_minimal_primitive_root: src/host/numth.cu:309-331 */
int orc_minimal_primitive_root(uint64_t degree, uint64_t q, uint64_t *root);
uint64_t orc_compute_shoup(uint64_t w, uint64_t q); /* include/host/uintarithsmallmod.h:119 */
uint64_t orc_invmod(uint64_t a, uint64_t q);
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q);
/* NTT::NTT tables: src/host/ntt.cu:11-56 (itw[1] is pre-multiplied by n^-1, :53-55). */
int orc_ntt_tables(int log_n, uint64_t q, uint64_t *tw, uint64_t *tw_shoup, uint64_t *itw,
                   uint64_t *itw_shoup, uint64_t *n_inv, uint64_t *n_inv_shoup);

/* ---- transforms (src/ntt/fntt_2d.cu:9-198,620-653; src/ntt/intt_2d.cu:9-207,724-757) ---- */
void orc_ntt_forward(uint64_t *x, int log_n, uint64_t q, const uint64_t *tw, const uint64_t *tw_shoup);
void orc_ntt_inverse(uint64_t *x, int log_n, uint64_t q, const uint64_t *itw, const uint64_t *itw_shoup,
                     uint64_t n_inv, uint64_t n_inv_shoup);

/* ---- RNS context: all tables for a QP chain (include/ntt.cuh:34-129 DNTTTable) ---- */
typedef struct orc_ctx orc_ctx;
orc_ctx *orc_ctx_create(int log_n, const uint64_t *primes_qp, size_t size_qp, size_t size_p);
void orc_ctx_destroy(orc_ctx *c);
size_t orc_ctx_n(const orc_ctx *c);
const uint64_t *orc_ctx_twiddle(const orc_ctx *c, size_t prime_idx, int which); /* 0 tw,1 tw_shoup,2 itw,3 itw_shoup */
uint64_t orc_ctx_n_inv(const orc_ctx *c, size_t prime_idx);

/* threads used by the limb-parallel loops (0 = all cores); test / baseline knob only */
void orc_set_threads(int n);
/* limb-batched transforms; prime index of data limb i is start_idx+i (nwt_2d_radix8_forward_inplace
 * fntt_2d.cu:620-653 / backward intt_2d.cu:724-757) */
void orc_nwt_forward(const orc_ctx *c, uint64_t *data, size_t limbs, size_t start_idx);
void orc_nwt_backward(const orc_ctx *c, uint64_t *data, size_t limbs, size_t start_idx);
/* explicit limb -> prime map (covers the *_include_special_mod remap, fntt_2d.cu:434-437) */
void orc_nwt_forward_map(const orc_ctx *c, uint64_t *data, const uint32_t *prime_idx, size_t limbs);
void orc_nwt_backward_map(const orc_ctx *c, uint64_t *data, const uint32_t *prime_idx, size_t limbs);

/* ---- dyadic kernels (src/polymath.cu) ; limb i uses prime start_idx+i ---- */
void orc_add_rns_poly(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *r, size_t limbs, size_t start_idx);      /* :41-56 */
void orc_sub_rns_poly(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *r, size_t limbs, size_t start_idx);      /* :110-124 */
void orc_negate_rns_poly(const orc_ctx *c, const uint64_t *a, uint64_t *r, size_t limbs, size_t start_idx);                      /* :17-32 */
void orc_multiply_rns_poly(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *r, size_t limbs, size_t start_idx); /* :156-172 */
void orc_multiply_and_add_rns_poly(const orc_ctx *c, const uint64_t *a, const uint64_t *b, const uint64_t *d, uint64_t *r, size_t limbs, size_t start_idx); /* :215-232 */
void orc_multiply_scalar_rns_poly(const orc_ctx *c, const uint64_t *a, const uint64_t *scalar, uint64_t *r, size_t limbs, size_t start_idx); /* :197-213 */
/* tensor_prod_2x2_rns_poly :463-496 ; op = [2][limbs][N], res = [3][limbs][N] */
void orc_tensor_prod_2x2(const orc_ctx *c, const uint64_t *op1, const uint64_t *op2, uint64_t *res, size_t limbs);
void orc_tensor_square_2x2(const orc_ctx *c, const uint64_t *op, uint64_t *res, size_t limbs); /* :498-529 */

/* ---- generic fast base conversion (src/rns_bconv.cu:22-60,109-229; src/host/rns.cu:282-337,438-497) ----
 * ibase/obase are lists of primes; src [ibase][N] -> dst [obase][N]  (DBaseConverter::bConv_BEHZ) */
void orc_bconv(const uint64_t *ibase, size_t isz, const uint64_t *obase, size_t osz, const uint64_t *src,
               uint64_t *dst, size_t n);

/* DBaseConverter::bConv_BEHZ_var1 (src/rns_bconv.cu:231-246; constants src/host/rns.cu:469-496): the quotient-style conversion */
void orc_bconv_behz_var1(const uint64_t *ibase, size_t isz, const uint64_t *obase, size_t osz, const uint64_t *src,
                         uint64_t *dst, size_t n);
/* DBaseConverter::exact_convert_array (src/rns_bconv.cu:374-431): src [ibase][N] -> dst [N] modulo ONE output modulus t */
void orc_exact_convert_array(const uint64_t *ibase, size_t isz, uint64_t t, const uint64_t *src, uint64_t *dst, size_t n);

/* ---- DRNSTool at level size_ql (src/rns.cu:11-200) ---- */
typedef struct orc_tool orc_tool;
orc_tool *orc_tool_create(const orc_ctx *c, size_t size_ql);
void orc_tool_destroy(orc_tool *t);
size_t orc_tool_beta(const orc_tool *t);
/* plain-modulus (BGV) constants of the tool, rns.cu:196-285; returns 0 on success, -1 when t is not invertible */
int orc_tool_set_plain_modulus(orc_tool *t, uint64_t plain_t);
/* scheme: 1 = bfv, 2 = ckks (include/host/encryptionparams.h:19-27 uses bfv=1,ckks=2,bgv=3) */
enum { ORC_BFV = 1, ORC_CKKS = 2, ORC_BGV = 3 };
/* DRNSTool::modup rns_bconv.cu:530-627: cks [size_ql][N] -> dst [beta][size_ql+alpha][N] */
void orc_modup(const orc_tool *t, uint64_t *dst, const uint64_t *cks, int scheme);
/* key_switch_inner_prod eval_key_switch.cu:14-92: evk[i] = [2][size_QP][N]; cx = [2][size_ql+alpha][N] */
void orc_key_switch_inner_prod(const orc_tool *t, uint64_t *cx, const uint64_t *t_mod_up, const uint64_t *const *evks);
/* DRNSTool::moddown_from_NTT rns_bconv.cu:776-828: cx [size_ql+alpha][N] (clobbered) -> ct [size_ql][N] */
void orc_moddown_from_ntt(const orc_tool *t, uint64_t *ct, uint64_t *cx, int scheme);
/* DRNSTool::moddown rns_bconv.cu:712-761: as above, but BFV input already in coefficient form and no alpha = 1 special case */
void orc_moddown(const orc_tool *t, uint64_t *ct, uint64_t *cx, int scheme);
/* keyswitch_inplace eval_key_switch.cu:95-182: ct=[2][size_ql][N] += KS(c2) */
void orc_keyswitch_inplace(const orc_tool *t, uint64_t *ct, const uint64_t *c2, const uint64_t *const *evks, int scheme);
/* hoisting_inplace evaluate.cu:1670-1866: ct=[2][size_ql][N] <- sum_e rotate_e(ct); glk[e][digit] = key [2][size_QP][N] */
/* DBaseConverter::bConv_HPS (src/rns_bconv.cu:248-372): exact conversion, overflow count from an fma chain of doubles */
void orc_bconv_hps(const uint64_t *ibase, size_t isz, const uint64_t *obase, size_t osz, const uint64_t *src, uint64_t *dst, size_t n);
/* ---- ciphertext (+|-|*) plaintext: src/scalingvariant.cu:10-60, src/evaluate.cu:1150-1154, 1256-1300 ---- */
void orc_bfv_add_plain(const orc_ctx *c, size_t size_ql, uint64_t *ct, const uint64_t *plain, uint64_t t, int subtract);
void orc_bgv_lift_plain(const orc_ctx *c, size_t size_ql, const uint64_t *plain, uint64_t *out);
void orc_bfv_multiply_plain(const orc_ctx *c, size_t size_ql, uint64_t *ct, size_t cipher_size, const uint64_t *plain, uint64_t t);
/* ---- BFV multiply, hps_overq (src/evaluate.cu:674-818 overq branches; rns.cu:792-885, 1748-1796; rns_bconv.cu:231-246) ---- */
typedef struct orc_hpsq orc_hpsq;
orc_hpsq *orc_hpsq_create(const orc_ctx *c, uint64_t plain_t);
/* hps_overq_leveled with size_Q - size_ql levels dropped (src/rns.cu:897-975); size_ql = size_Q is orc_hpsq_create */
orc_hpsq *orc_hpsq_create_level(const orc_ctx *c, uint64_t plain_t, size_t size_ql);
void orc_hps_scale_q_ql(const orc_hpsq *h, const uint64_t *src, uint64_t *dst);   /* scaleAndRound_HPS_Q_Ql rns.cu:1798-1808 */
void orc_hps_expand_ql_q(const orc_hpsq *h, const uint64_t *src, uint64_t *dst);  /* ExpandCRTBasis_Ql_Q rns.cu:1810-1836 */
/* bfv_mul_relin_hps with levels dropped (evaluate.cu:822-1027): c2 never leaves level l; dst [2][Q][N] */
void orc_bfv_mul_relin_hps_overq_leveled(const orc_tool *t, const orc_hpsq *h, const uint64_t *ct1, const uint64_t *ct2,
                                         const uint64_t *const *evks, uint64_t *dst);
/* BFV key switch with levels dropped (eval_key_switch.cu:142-147,170-175); t = orc_tool of that level */
void orc_keyswitch_bfv_leveled(const orc_tool *t, const orc_hpsq *h, uint64_t *ct, const uint64_t *c2, const uint64_t *const *evks);
void orc_hpsq_scale_round_qlrl_ql(const orc_hpsq *h, const uint64_t *src, uint64_t *dst);  /* scaleAndRound_HPS_QlRl_Ql rns.cu:1748-1796 */
void orc_hpsq_expand_add_to_ct(const orc_hpsq *h, const uint64_t *src, uint64_t *dst);      /* ExpandCRTBasis_Ql_Q_add_to_ct rns.cu:1838-1858 */
void orc_hpsq_destroy(orc_hpsq *h);
size_t orc_hpsq_r_size(const orc_hpsq *h);
void orc_hpsq_base(const orc_hpsq *h, uint64_t *out);
/* ct2 == ct1 (same pointer) selects the squaring path, as in the reference */
void orc_bfv_multiply_hps_overq(const orc_hpsq *h, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst);
void orc_hoisting(const orc_tool *t, uint64_t *ct, const uint32_t *galois_elts, size_t n_elts,
                  const uint64_t *const *const *glk, int scheme);
/* build-defined: sum_e w_e (.) rotate_e(ct), weights over [Q_l || P] in NTT form (BASELINE config 5) */
void orc_hoisting_weighted(const orc_tool *t, uint64_t *ct, const uint32_t *galois_elts, size_t n_elts,
                           const uint64_t *const *const *glk, const uint64_t *const *weights, int scheme);
/* divide_and_round_q_last_ntt rns.cu:1128-1184: src [cipher][size_ql][N] (clobbered) -> dst [cipher][size_ql-1][N] */
void orc_rescale_ntt(const orc_tool *t, uint64_t *src, size_t cipher_size, uint64_t *dst);
/* mod_t_and_divide_q_last_ntt rns.cu:1186-1236 (BGV modulus switch; needs orc_tool_set_plain_modulus) */
void orc_mod_t_divide_q_last_ntt(const orc_tool *t, uint64_t *src, size_t cipher_size, uint64_t *dst);
/* divide_and_round_q_last rns.cu:1082-1126 (BFV coefficient-domain mod switch) */
void orc_divide_and_round_q_last(const orc_tool *t, const uint64_t *src, size_t cipher_size, uint64_t *dst);

/* ---- BFV multiply, BEHZ (src/evaluate.cu:404-548; src/rns.cu:392-560,1249-1510); base q = all data primes ---- */
typedef struct orc_behz orc_behz;
orc_behz *orc_behz_create(const orc_ctx *c, uint64_t plain_t);
void orc_behz_destroy(orc_behz *b);
size_t orc_behz_bsk_size(const orc_behz *b);
void orc_behz_base(const orc_behz *b, uint64_t *bsk_out);  /* B primes then m_sk */
/* the four DRNSTool steps of the multiply, one polynomial each (include/rns.cuh:186-200) */
void orc_behz_fastbconv_m_tilde(const orc_behz *b, const uint64_t *src, uint64_t *dst);                       /* rns.cu:1249-1278: [Q][N] -> [Bsk+1][N] */
void orc_behz_sm_mrq(const orc_behz *b, const uint64_t *src, uint64_t *dst);                                  /* rns.cu:1290-1338: [Bsk+1][N] -> [Bsk][N] */
void orc_behz_fast_floor(const orc_behz *b, const uint64_t *in_q, const uint64_t *in_bsk, uint64_t *out_bsk); /* rns.cu:1343-1419 */
void orc_behz_fastbconv_sk(const orc_behz *b, const uint64_t *in_bsk, uint64_t *out_q);                       /* rns.cu:1421-1510 */
/* ct1, ct2 [2][Q][N] in coefficient form -> dst [3][Q][N] in coefficient form */
void orc_bfv_multiply_behz(const orc_behz *b, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst);

/* ---- BFV multiply, HPS (mul_tech hps; src/evaluate.cu:674-818, src/rns_bconv.cu:248-372, src/rns.cu:687-790,1700-1746) ---- */
typedef struct orc_hps orc_hps;
orc_hps *orc_hps_create(const orc_ctx *c, uint64_t plain_t);
void orc_hps_destroy(orc_hps *h);
size_t orc_hps_r_size(const orc_hps *h);
void orc_hps_base(const orc_hps *h, uint64_t *r_out);
void orc_hps_scale_round_qr_r(const orc_hps *h, const uint64_t *src, uint64_t *dst);   /* scaleAndRound_HPS_QR_R rns.cu:1700-1746: [Q+R][N] -> [R][N] */
void orc_bfv_multiply_hps(const orc_hps *h, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst);

/* ---- batched modular GEMM (benchmark/matmul_bench.cu:215-541), one modulus: row-major A [m][k], B [k][n], C [m][n] ---- */
void orc_gemm_mod(uint64_t q, const uint64_t *A, const uint64_t *B, uint64_t *C, size_t m, size_t n, size_t k);
void orc_gemm_mod_ref_quirk(uint64_t q, const uint64_t *A, const uint64_t *B, uint64_t *C, size_t m, size_t n, size_t k);

/* ---- Galois (include/galois.cuh:98-130, src/galois.cu:11-39) ---- */
void orc_galois_ntt_table(int log_n, uint32_t galois_elt, uint32_t *table);
void orc_apply_galois_ntt(const uint64_t *src, uint64_t *dst, const uint32_t *table, size_t n, size_t limbs);
void orc_apply_galois_coeff(const orc_ctx *c, const uint64_t *src, uint64_t *dst, uint32_t galois_elt, size_t limbs, size_t start_idx);

/* ---- evaluation-key generation for functional tests (src/secretkey.cu:232-341, polymath.cu:318-338) ----
 * sk_ntt [size_QP][N] (NTT form), new_key_ntt [size_Q][N] (NTT form, e.g. s^2 or galois(s)),
 * a_ntt/e_ntt [dnum][size_QP][N] caller-provided randomness (NTT form); evk [dnum][2][size_QP][N]. */
void orc_gen_kswitch_key(const orc_ctx *c, const uint64_t *sk_ntt, const uint64_t *new_key_ntt,
                         const uint64_t *a_ntt, const uint64_t *e_ntt, uint64_t *evk);

#ifdef __cplusplus
}
#endif
#endif
