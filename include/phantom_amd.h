/*
 * phantom_amd.h -- C ABI of the MI355X-native RNS polynomial-arithmetic core (libphantom_amd.so).
 *
 * The reference (encryptorion-lab/phantom-fhe) has no FFI layer: its boundary for this path is the
 * C++ launchers in include/ntt.cuh:157-226, the DRNSTool / DBaseConverter methods
 * (include/rns.cuh:156-205, include/rns_bconv.cuh:62-68) and phantom::key_switch_inner_prod /
 * keyswitch_inplace (include/evaluate.cuh:18-32).  Each entry point below replaces exactly one of
 * those and keeps its argument meaning; the only changes are (i) `const DNTTTable&` / `DRNSTool&`
 * become an opaque context handle plus the level (number of live data limbs), (ii) `cudaStream_t`
 * becomes `void*` (a hipStream_t; NULL = the legacy default stream), (iii) C++ exceptions become
 * an int status (0 = PHA_OK) with pha_last_error() carrying the what() text.
 *
 * All `uint64_t*` data arguments are DEVICE pointers owned by the caller, limb-major contiguous
 * `data[limb * N + coeff]`, values canonical in [0, q_limb).  No entry point
 * allocates caller-visible memory; scratch comes from a per-context, per-stream arena.
 * PRECONDITION on every operand word, including caller-supplied constants (keys, plaintext diagonals / weights, scale and P^-1
 * vectors): canonical, i.e. below its limb's modulus.  The reference's Barrett-128 kernels happen to tolerate lazy or unreduced
 * words (src/polymath.cu:463-496 multiplies an unreduced c0 + c1; include/uintmodmath.cuh:96-136 reduces any 128-bit value); this
 * library does not -- on limbs below 2^50 the dyadic, inner-product, hoisting and epilogue kernels compute in FP64 (exact only for
 * words below 2^52), so a non-canonical word there gives wrong residues WITHOUT an error.  Every word the library writes is
 * canonical, so chains of its own calls keep the precondition.  It is CHECKABLE (r06): pha_check_canonical / _keys count the
 * offending words of a buffer, and strict mode (PHA_STRICT=1 in the environment, or pha_set_strict) makes every entry point that
 * takes caller-supplied operands count first and return status -1 naming the operand instead of computing.
 * Capturing calls into a hipGraph: warm the same call up once first (same level and batch: an arena grows on demand, and growth is
 * an allocation), and capture on an EXPLICIT stream -- the arenas behind NULL / hipStreamPerThread belong to the calling host thread
 * and are released when that thread exits, so a graph captured on them must not be replayed after the thread is gone.
 * Nothing here touches torch types.
 */
#ifndef PHANTOM_AMD_H
#define PHANTOM_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pha_context *pha_context_t;

enum { PHA_OK = 0, PHA_ERR_INVALID_ARGUMENT = -1, PHA_ERR_LOGIC = -2, PHA_ERR_RUNTIME = -3 };
/* scheme_type values of include/host/encryptionparams.h:19-27 */
enum { PHA_SCHEME_BFV = 1, PHA_SCHEME_CKKS = 2, PHA_SCHEME_BGV = 3 };

/* what() of the last failing call on this thread (std::invalid_argument / logic_error /
 * "HIP Runtime Error" -- mirrors include/cuda_wrapper.cuh:19-43). */
const char *pha_last_error(void);

/* ---- the canonical-operand precondition, checkable (csrc/pha_check.hip).  Both calls SYNCHRONISE the stream (they return a
 *      count): debugging and input validation, not the hot path, not inside a stream capture.
 * pha_check_canonical: data [polys][coeff_modulus_size][N] (polys spaced poly_stride words; 0 with polys = 1), limb i checked
 *   (data points at the FIRST checked limb) against table row start_modulus_idx + i -- or, for the last size_P_tail limbs of a [Q_l || P] buffer, against the special rows
 *   (the remap of nwt_2d_radix8_forward_inplace_include_special_mod, src/ntt/fntt_2d.cu:434-437); *bad_words = how many words are
 *   >= their modulus.
 * pha_check_canonical_keys: the limbs a key switch at level size_Ql READS of n_keys keys [2][size_QP][N] (rows 0 .. size_Ql - 1
 *   and the special rows), keys = DEVICE array of device pointers (PhantomRelinKey::public_keys_ptr()).
 * pha_set_strict(on): strict mode on / off for the process, returns the previous state.  Default: on iff PHA_STRICT=1 was in the
 *   environment when the library was first used.  In strict mode the dyadic, tensor, mod-up / inner-product / mod-down, key-switch
 *   and hoisting entries check ct / c2 / t_mod_up / keys / weights before they compute (one synchronising pass per operand) and
 *   fail with status -1 ("PHA_STRICT: <operand> holds k word(s) >= their limb's modulus"); results are unchanged otherwise. ---- */
int pha_check_canonical(pha_context_t ctx, const uint64_t *data, size_t coeff_modulus_size, size_t start_modulus_idx,
                        size_t size_P_tail, size_t polys, size_t poly_stride, uint64_t *bad_words, void *stream);
int pha_check_canonical_keys(pha_context_t ctx, size_t size_Ql, const uint64_t *const *keys, size_t n_keys, uint64_t *bad_words,
                             void *stream);
int pha_set_strict(int on);

/* ---- host precompute (replaces src/host/modulus.cu:82-111 CoeffModulus::Create and the
 *      hard-coded default table src/host/globals.cu:30-120) ---- */
int pha_coeff_modulus_create(uint64_t poly_modulus_degree, const int *bit_sizes, size_t count, uint64_t *out);

/* ---- context: replaces PhantomContext's hot-path state (src/context.cu:121-232): one DNTTTable for
 *      all QP primes (include/ntt.cuh:34-129) plus a DRNSTool per level (src/rns.cu:11-200), built
 *      lazily and cached.  primes_qp = [Q primes..., P primes...]; size_p = special_modulus_size. ---- */
int pha_context_create(pha_context_t *out, uint32_t log_n, const uint64_t *primes_qp, uint32_t size_qp,
                       uint32_t size_p, int device_id);
void pha_context_destroy(pha_context_t ctx);
/* EncryptionParameters::set_plain_modulus as seen by DRNSTool's constructor (argument `t`, src/rns.cu:11-26;
 * BGV constants :196-285).  0 = none.  Call before the first evaluation call: cached per-level tools are
 * dropped and rebuilt.  Errors: plain_modulus == 1 or >= 2^60 -> invalid_argument; not coprime with
 * the chain -> logic_error ("invalid rns bases", rns.cu:207-208,274-275). */
int pha_context_set_plain_modulus(pha_context_t ctx, uint64_t plain_modulus);
uint32_t pha_context_log_n(pha_context_t ctx);
uint32_t pha_context_size_qp(pha_context_t ctx);
uint32_t pha_context_size_p(pha_context_t ctx);
/* host copies of per-prime constants, for callers that mirror DNTTTable getters; rows 0 .. size_QP - 1 are the chain, the rows
 * after them the auxiliary BFV bases once they exist (first use of a BFV multiply entry or pha_tool_aux_sizes): Bsk then m_tilde
 * (root 0: no tables), R.  An index past the last row is status -1. */
int pha_context_prime_info(pha_context_t ctx, uint32_t prime_idx, uint64_t *value, uint64_t const_ratio[2],
                           uint64_t *root, uint64_t *n_inv);
/* download one twiddle row (host buffers of N words): which = 0 twiddle, 1 twiddle_shoup,
 * 2 itwiddle, 3 itwiddle_shoup -- same content as DNTTTable::twiddle()/itwiddle() rows
 * (itwiddle[1] carries N^-1 exactly like src/host/ntt.cu:53-55). */
int pha_context_download_twiddle(pha_context_t ctx, uint32_t prime_idx, int which, uint64_t *host_out);
/* level tool queries (DRNSTool): beta = number of key-switch digits at this level */
int pha_tool_beta(pha_context_t ctx, uint32_t size_ql, uint32_t *beta);

/* ---- NTT launchers (include/ntt.cuh:178-226).  Processed limbs are
 *      [start_modulus_idx, start_modulus_idx + coeff_modulus_size) of the buffer. ---- */
int pha_nwt_2d_radix8_forward_inplace(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size,
                                      size_t start_modulus_idx, void *stream);
int pha_nwt_2d_radix8_forward_inplace_include_special_mod(pha_context_t ctx, uint64_t *inout,
                                                          size_t coeff_modulus_size, size_t start_modulus_idx,
                                                          size_t size_QP, size_t size_P, void *stream);
int pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(
    pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size, size_t start_modulus_idx, size_t size_QP,
    size_t size_P, size_t excluded_range_start, size_t excluded_range_end, void *stream);
int pha_nwt_2d_radix8_forward_inplace_fuse_moddown(pha_context_t ctx, uint64_t *ct, const uint64_t *cx,
                                                   const uint64_t *bigPInv_mod_q,
                                                   const uint64_t *bigPInv_mod_q_shoup, uint64_t *delta,
                                                   size_t coeff_modulus_size, size_t start_modulus_idx,
                                                   void *stream);
int pha_nwt_2d_radix8_backward_inplace(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size,
                                       size_t start_modulus_idx, void *stream);
int pha_nwt_2d_radix8_backward(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t coeff_modulus_size,
                               size_t start_modulus_idx, void *stream);
int pha_nwt_2d_radix8_backward_scale(pha_context_t ctx, uint64_t *out, const uint64_t *in,
                                     size_t coeff_modulus_size, size_t start_modulus_idx, const uint64_t *scale,
                                     const uint64_t *scale_shoup, void *stream);
int pha_nwt_2d_radix8_backward_inplace_scale(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size,
                                             size_t start_modulus_idx, const uint64_t *scale,
                                             const uint64_t *scale_shoup, void *stream);
int pha_nwt_2d_radix8_backward_inplace_include_special_mod(pha_context_t ctx, uint64_t *inout,
                                                           size_t coeff_modulus_size, size_t start_modulus_idx,
                                                           size_t size_QP, size_t size_P, void *stream);

/* nwt_2d_radix8_forward_inplace_include_temp_mod (include/ntt.cuh:182-185, src/ntt/fntt_2d.cu:200-405,655-693) and
 * nwt_2d_radix8_backward_inplace_include_temp_mod_scale (include/ntt.cuh:222-226, src/ntt/intt_2d.cu:313-409,836-873):
 * transforms over the BEHZ base Bsk = B u {m_sk} (callers src/evaluate.cu:434,528).  The reference passes
 * rns_tool.gpu_Bsk_tables(); here the context owns those primes (pha_context_set_plain_modulus must have been
 * called), so coeff_modulus_size must be |Bsk|, start_modulus_idx 0 and total_modulus_size |Bsk| + 1. */
int pha_nwt_2d_radix8_forward_inplace_include_temp_mod(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size,
                                                       size_t start_modulus_idx, size_t total_modulus_size,
                                                       void *stream);
int pha_nwt_2d_radix8_backward_inplace_include_temp_mod_scale(pha_context_t ctx, uint64_t *inout,
                                                              size_t coeff_modulus_size, size_t start_modulus_idx,
                                                              size_t total_modulus_size, const uint64_t *scale,
                                                              const uint64_t *scale_shoup, void *stream);
/* nwt_2d_radix8_forward_modup_fuse (include/ntt.cuh:199-201, src/ntt/ntt_keyswitch_old.cu:10-265): out[limb] = NTT
 * modulo q_{modulus_index} of in[limb], limbs [start, start + coeff_modulus_size) -- how the reference lifts a
 * plaintext modulo t into RNS limb modulus_index (src/evaluate.cu:1152,1210,1321).  Out of place. */
int pha_nwt_2d_radix8_forward_modup_fuse(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t modulus_index,
                                         size_t coeff_modulus_size, size_t start_modulus_idx, void *stream);
/* fnwt_1d / fnwt_1d_opt / inwt_1d / inwt_1d_opt (include/ntt.cuh:157-171, src/ntt/ntt_1d.cu:17-292): single-workgroup
 * transforms for dim <= 2048 over caller-built tables (the reference's NTT test and benchmark use them:
 * test/ntt_test.cu:9-69, benchmark/ntt_bench.cu:8-79).  twiddles[limb * dim + k] / twiddles_shoup likewise; modulus is
 * an array of {value, const_ratio[0], const_ratio[1]} triples (DModulus); the inverse multiplies the FIRST half of
 * its outputs by scalar[limb] (the second half gets N^-1 through slot 1 of the inverse table, src/host/ntt.cu:53-55).
 * fnwt_1d_opt ignores start_modulus_idx, as the reference's kernel does (ntt_1d.cu:92-93).  No context needed. */
int pha_fnwt_1d(uint64_t *inout, const uint64_t *twiddles, const uint64_t *twiddles_shoup, const uint64_t *modulus,
                size_t dim, size_t coeff_modulus_size, size_t start_modulus_idx, void *stream);
int pha_fnwt_1d_opt(uint64_t *inout, const uint64_t *twiddles, const uint64_t *twiddles_shoup, const uint64_t *modulus,
                    size_t dim, size_t coeff_modulus_size, size_t start_modulus_idx, void *stream);
int pha_inwt_1d(uint64_t *inout, const uint64_t *itwiddles, const uint64_t *itwiddles_shoup, const uint64_t *modulus,
                const uint64_t *scalar, const uint64_t *scalar_shoup, size_t dim, size_t coeff_modulus_size,
                size_t start_modulus_idx, void *stream);
int pha_inwt_1d_opt(uint64_t *inout, const uint64_t *itwiddles, const uint64_t *itwiddles_shoup, const uint64_t *modulus,
                    const uint64_t *scalar, const uint64_t *scalar_shoup, size_t dim, size_t coeff_modulus_size,
                    size_t start_modulus_idx, void *stream);
/* Extension (no reference counterpart): the same limbs [start, start+size) of `batch` polynomials that
 * lie `poly_stride` elements apart (e.g. the polynomials of a ciphertext) in ONE launch. */
int pha_nwt_2d_radix8_forward_inplace_batched(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size,
                                              size_t start_modulus_idx, size_t batch, size_t poly_stride,
                                              void *stream);
int pha_nwt_2d_radix8_backward_inplace_batched(pha_context_t ctx, uint64_t *inout, size_t coeff_modulus_size,
                                               size_t start_modulus_idx, size_t batch, size_t poly_stride,
                                               void *stream);

/* ---- dyadic kernels (include/polymath.cuh:6-307, launched <<<N*L/128,128>>> by evaluate.cu).
 *      The reference passes `const DModulus *modulus` (a row of the QP table); here that is
 *      (ctx, mod_start_idx).  Buffers hold coeff_mod_size limbs. ---- */
int pha_add_rns_poly(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2, uint64_t *result,
                     size_t coeff_mod_size, size_t mod_start_idx, void *stream);
int pha_sub_rns_poly(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2, uint64_t *result,
                     size_t coeff_mod_size, size_t mod_start_idx, void *stream);
int pha_negate_rns_poly(pha_context_t ctx, const uint64_t *op, uint64_t *result, size_t coeff_mod_size,
                        size_t mod_start_idx, void *stream);
int pha_multiply_rns_poly(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2, uint64_t *result,
                          size_t coeff_mod_size, size_t mod_start_idx, void *stream);
int pha_multiply_and_add_rns_poly(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2,
                                  const uint64_t *op3, uint64_t *result, size_t coeff_mod_size,
                                  size_t mod_start_idx, void *stream);
/* Shoup overload polymath.cuh:97-103: scalar / scalar_shoup are device arrays, one entry per limb */
int pha_multiply_scalar_rns_poly(pha_context_t ctx, const uint64_t *op, const uint64_t *scalar,
                                 const uint64_t *scalar_shoup, uint64_t *result, size_t coeff_mod_size,
                                 size_t mod_start_idx, void *stream);
/* tensor_prod_2x2_rns_poly / tensor_square_2x2_rns_poly: operands [2][L][N], result [3][L][N];
 * result may alias operand1 (evaluate.cu:377-383 calls it in place). */
int pha_tensor_prod_2x2_rns_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2,
                                 uint64_t *result, size_t coeff_mod_size, void *stream);
int pha_tensor_square_2x2_rns_poly(pha_context_t ctx, const uint64_t *operand, uint64_t *result,
                                   size_t coeff_mod_size, void *stream);
/* the same two kernels with the reference's `modulus` pointer argument as a first table row (polymath.cu:463-529 take
 * `const DModulus *modulus`; bfv_multiply_behz / _hps run them over base Bsk and base R: src/evaluate.cu:489-497, :763-777):
 * limb i uses row mod_start + i, which may be an auxiliary row (pha_context_prime_info lists them).  pha_add / sub / multiply /
 * multiply_and_add / multiply_scalar_rns_poly accept auxiliary rows through their mod_start in the same way. */
int pha_tensor_prod_2x2_rns_poly_at(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2, uint64_t *res, size_t cms,
                                    size_t mod_start, void *stream);
int pha_tensor_square_2x2_rns_poly_at(pha_context_t ctx, const uint64_t *op, uint64_t *res, size_t cms, size_t mod_start,
                                      void *stream);
/* add_to_ct_kernel (src/rns_bconv.cu:763-769) */
int pha_add_to_ct(pha_context_t ctx, uint64_t *ct, const uint64_t *cx, size_t size_Ql, void *stream);

/* ---- the rest of include/polymath.cuh:6-307 (src/polymath.cu): residue-wise kernels the reference's encryption,
 *      decryption and plaintext layers launch around the hot path.  Same conventions: [limb][coeff] buffers,
 *      table rows [mod_start, mod_start + coeff_mod_size) where a mod_start is given, canonical outputs. ---- */
/* add_std_cipher :56-73 -- both polynomials of two size-2 ciphertexts */
int pha_add_std_cipher(pha_context_t ctx, const uint64_t *cipher1, const uint64_t *cipher2, uint64_t *result,
                       size_t coeff_mod_size, void *stream);
/* add_and_negate_rns_poly :82-98 -- -(a + b) */
int pha_add_and_negate_rns_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2, uint64_t *result,
                                size_t coeff_mod_size, size_t mod_start_idx, void *stream);
/* add_many_rns_poly :126-147 -- result[poly_index] = sum of operands[e][poly_index]; operands is a HOST array of
 * add_size DEVICE pointers (the reference takes the same table on the device) */
int pha_add_many_rns_poly(pha_context_t ctx, const uint64_t *const *operands, size_t add_size, uint64_t *result,
                          size_t poly_index, size_t coeff_mod_size, void *stream);
/* multiply_scalar_rns_poly, the overload with ONE scalar for every limb :181-196 (Barrett) */
int pha_multiply_uniform_scalar_rns_poly(pha_context_t ctx, const uint64_t *operand, uint64_t scale, uint64_t *result,
                                         size_t coeff_mod_size, size_t mod_start_idx, void *stream);
/* multiply_scalar_and_add_rns_poly :246-264 -- operand1 + operand2 * scalar; _and_sub :266-283 -- operand1 - operand2 * scalar */
int pha_multiply_scalar_and_add_rns_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2,
                                         uint64_t scalar, uint64_t *result, size_t coeff_mod_size, size_t mod_start_idx,
                                         void *stream);
int pha_multiply_scalar_and_sub_rns_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2,
                                         uint64_t scalar, uint64_t *result, size_t coeff_mod_size, size_t mod_start_idx,
                                         void *stream);
/* multiply_and_scale_add_rns_poly :294-315 -- operand1 * operand2 + operand3 * scale */
int pha_multiply_and_scale_add_rns_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2,
                                        const uint64_t *operand3, uint64_t scale, uint64_t *result,
                                        size_t coeff_mod_size, size_t mod_start_idx, void *stream);
/* multiply_and_add_negate_rns_poly :350-371 -- -(operand1 * operand2 + operand3) (the b half of an RLWE sample) */
int pha_multiply_and_add_negate_rns_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2,
                                         const uint64_t *operand3, uint64_t *result, size_t coeff_mod_size,
                                         size_t mod_start_idx, void *stream);
/* sub_and_scale_rns_poly :392-411 -- (operand1 - operand2) * scale[limb]; _single_mod_poly :374-390 -- one limb, explicit modulus */
int pha_sub_and_scale_rns_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2,
                               const uint64_t *scale, const uint64_t *scale_shoup, uint64_t *result,
                               size_t coeff_mod_size, size_t mod_start_idx, void *stream);
int pha_sub_and_scale_single_mod_poly(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2,
                                      uint64_t scale, uint64_t scale_shoup, uint64_t modulus, uint64_t *result,
                                      void *stream);
/* bfv_add_timesQ_overt_kernel / bfv_sub_timesQ_overt_kernel :413-461 -- ct[limb] +- (pt * (-Ql mod t) mod t) * t^-1 mod q_limb,
 * pt a single limb of N coefficients below t */
int pha_bfv_add_timesQ_overt(pha_context_t ctx, uint64_t *ct, const uint64_t *pt, uint64_t negQl_mod_t,
                             uint64_t negQl_mod_t_shoup, const uint64_t *tInv_mod_q, const uint64_t *tInv_mod_q_shoup,
                             uint64_t t, size_t size_Ql, void *stream);
int pha_bfv_sub_timesQ_overt(pha_context_t ctx, uint64_t *ct, const uint64_t *pt, uint64_t negQl_mod_t,
                             uint64_t negQl_mod_t_shoup, const uint64_t *tInv_mod_q, const uint64_t *tInv_mod_q_shoup,
                             uint64_t t, size_t size_Ql, void *stream);
/* abs_plain_rns_poly :645-664 -- result[limb][i] = operand[i] (+ increment[limb] when operand[i] >= threshold) */
int pha_abs_plain_rns_poly(pha_context_t ctx, const uint64_t *operand, uint64_t plain_upper_half_threshold,
                           const uint64_t *plain_upper_half_increment, uint64_t *result, size_t coeff_mod_size,
                           void *stream);
/* tensor_prod_mxn_rns_poly :546-592 -- ciphertext product of sizes m x n (1..8 polynomials each; the operands sit in
 * registers instead of the reference's per-thread device new[]) */
int pha_tensor_prod_mxn_rns_poly(pha_context_t ctx, const uint64_t *operand1, size_t op1_size, const uint64_t *operand2,
                                 size_t op2_size, uint64_t *result, size_t res_size, size_t coeff_mod_size,
                                 void *stream);
/* multiply_and_negated_add_rns_poly :606-634 -- BEHZ FastBconvSK fix-up: operand3 - alpha_sk * prod(B) with alpha_sk
 * (one limb, modulo m_sk) taken centred */
int pha_multiply_and_negated_add_rns_poly(pha_context_t ctx, const uint64_t *alpha_sk, uint64_t m_sk,
                                          const uint64_t *prod_B_mod_q, const uint64_t *operand3, uint64_t *result,
                                          size_t coeff_mod_size, void *stream);

/* ---- RNS tool at level size_Ql (DRNSTool of context_data(chain) with size_Ql data limbs) ---- */
/* DBaseConverter (include/rns_bconv.cuh:13-87) between two bases given as rows of the context's prime table
 * (0 .. size_QP-1 = the Q primes then the P primes; the auxiliary BFV bases follow once a plain modulus is set).
 * bConv_BEHZ (src/rns_bconv.cu:212-229): dst[j] = sum_i (x_i * qhat_i^-1 mod q_i) * (qhat_i mod p_j) mod p_j, the
 * approximate conversion of the hot path.  bConv_HPS (:248-372): the exact conversion of the HPS multiply (overflow
 * count from a double-precision sum, fused multiply-adds as nvcc builds it).  src [ibase][N] -> dst [obase][N]. */
typedef struct pha_base_converter *pha_base_converter_t;
int pha_base_converter_create(pha_context_t ctx, const uint32_t *ibase, size_t ibase_size, const uint32_t *obase,
                              size_t obase_size, pha_base_converter_t *out);
void pha_base_converter_destroy(pha_base_converter_t conv);
int pha_bConv_BEHZ(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream);
int pha_bConv_HPS(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream);
/* DBaseConverter::bConv_BEHZ_var1 (include/rns_bconv.cuh:64, src/rns_bconv.cu:231-246; constants src/host/rns.cu:469-496): the
 * quotient-style conversion dst[j] = sum_i (x_i * (-P qhat_i^-1) mod q_i) * (q_i^-1 mod p_j) mod p_j with P = prod(obase); callers
 * src/evaluate.cu:747-749, :911-913.  The var1 constants are built on first use. */
int pha_bConv_BEHZ_var1(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream);
/* DBaseConverter::exact_convert_array (include/rns_bconv.cuh:68, src/rns_bconv.cu:374-431): src [ibase][N] -> dst [N] modulo the
 * converter's ONE output modulus (status -1 "out base in exact_convert_array must be one." otherwise, as :423-425), exactly:
 * inner product minus round(sum_i y_i / q_i) * (Q mod t).  The reference's only user converts to the plain modulus t, which is no
 * row of the prime table: pha_base_converter_create_modulus makes a converter from table rows to ONE raw modulus (2 .. 2^61 - 1);
 * such a converter serves exact_convert_array only. */
int pha_base_converter_create_modulus(pha_context_t ctx, const uint32_t *ibase, size_t ibase_size, uint64_t out_modulus,
                                      pha_base_converter_t *out);
int pha_exact_convert_array(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream);
/* DBaseConverter::bConv_BEHZ for base_P_to_Ql_conv (rns_bconv.cu:212-229): src [P][N] -> dst [Ql][N] */
int pha_bconv_P_to_Ql(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
/* DRNSTool::modup (rns_bconv.cu:530-627): cks [Ql][N] -> dst [beta][Ql+P][N] */
int pha_modup(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *cks, int scheme, void *stream);
/* phantom::key_switch_inner_prod (eval_key_switch.cu:71-92): rlk = DEVICE array of beta device
 * pointers, each key [2][size_QP][N]; p_cx [2][Ql+P][N] */
int pha_key_switch_inner_prod(pha_context_t ctx, size_t size_Ql, uint64_t *p_cx, const uint64_t *p_t_mod_up,
                              const uint64_t *const *rlk, void *stream);
/* DRNSTool::moddown_from_NTT (rns_bconv.cu:776-828): cx_i [Ql+P][N] (ckks: P limbs clobbered; bfv / bgv:
 * every limb left in coefficient form and the first P limb overwritten -- scratch, exactly as in the
 * reference) -> ct_i [Ql][N]; ct_i may alias cx_i.  bgv needs pha_context_set_plain_modulus. */
int pha_moddown_from_NTT(pha_context_t ctx, size_t size_Ql, uint64_t *ct_i, uint64_t *cx_i, int scheme,
                         void *stream);
/* phantom::keyswitch_inplace (eval_key_switch.cu:95-182) on raw buffers: ct [2][Ql][N] += KS(c2) */
/* DRNSTool::moddown (include/rns.cuh:159-160, src/rns_bconv.cu:712-761): cx_i [Ql + alpha][N] -> ct_i [Ql][N], the mod-down whose
 * BFV input is ALREADY in coefficient form (moddown_from_NTT transforms it first); ckks / bgv inputs in NTT form as there.  cx_i
 * is scratch afterwards (its P limbs are left in coefficient form, scaled for ckks with alpha > 1). */
int pha_moddown(pha_context_t ctx, size_t size_Ql, uint64_t *ct_i, uint64_t *cx_i, int scheme, void *stream);
int pha_keyswitch_inplace(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *c2,
                          const uint64_t *const *rlk, int scheme, void *stream);
/* Extension (the reference loops over ciphertexts): `batch` independent ciphertexts through ONE set of launches.
 * ct [batch][2][Ql][N] += KS(c2[b]) with c2 [batch][Ql][N]; the key limbs are read once per launch, the NTT and
 * base-conversion launches are batch times larger (the throughput regime of the kernels). */
int pha_keyswitch_inplace_batched(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *c2, size_t batch,
                                  const uint64_t *const *rlk, int scheme, void *stream);

/* Build-defined fusion (no reference launcher): key switch followed by the CKKS rescale, i.e. the relinearize -> rescale_to_next
 * pair of src/evaluate.cu:1029-1075,1376-1427 (keyswitch_inplace eval_key_switch.cu:95-182 then divide_and_round_q_last_ntt
 * rns.cu:1160-1184) as ONE call: dst [2][Ql-1][N] = rescale(ct + keyswitch(c2)), bit-identical to the two calls.  ct [2][Ql][N]
 * (NTT form) and c2 [Ql][N] are only read; dst must not overlap them.  The mod-down's forward transform over 2 x Ql limbs and
 * the rescale's own last-limb inverse disappear (NTT is linear: both subtractions ride on one forward transform).  ckks only. */
int pha_keyswitch_rescale(pha_context_t ctx, size_t size_Ql, const uint64_t *ct, const uint64_t *c2,
                          const uint64_t *const *rlk, uint64_t *dst, void *stream);
/* the same for `batch` ciphertexts ct [batch][2][Ql][N], c2 [batch][Ql][N] -> dst [batch][2][Ql-1][N] against one key */
int pha_keyswitch_rescale_batched(pha_context_t ctx, size_t size_Ql, const uint64_t *ct, const uint64_t *c2, size_t batch,
                                  const uint64_t *const *rlk, uint64_t *dst, void *stream);
/* tensor_prod_2x2_rns_poly for `batch` ciphertext pairs in the layout above: operands [batch][2][L][N],
 * res01 [batch][2][L][N] receives (c0, c1), res2 [batch][L][N] receives c2; res01 may alias operand1 */
int pha_tensor_prod_2x2_batched(pha_context_t ctx, const uint64_t *operand1, const uint64_t *operand2, uint64_t *res01,
                                uint64_t *res2, size_t coeff_mod_size, size_t batch, void *stream);
/* phantom::hoisting_inplace (include/evaluate.cuh:233-241, src/evaluate.cu:1670-1866) on raw buffers:
 * ct [2][Ql][N] <- sum over the n_elts Galois elements of rotate(ct).  galois_elts is a HOST array;
 * glk is a HOST array of n_elts DEVICE pointer tables (PhantomRelinKey::public_keys_ptr() of each
 * element's key).  One mod-up, one fused gather + inner-product kernel, one pair of mod-downs. */
int pha_hoisting(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint32_t *galois_elts, size_t n_elts,
                 const uint64_t *const *const *glk, int scheme, void *stream);
/* Weighted hoisted rotations -- BUILD-DEFINED (BASELINE config 5 "encrypted matmul"; the reference has no such
 * entry point: its building blocks are hoisting_inplace src/evaluate.cu:1670-1866, multiply_plain_inplace :1297-1340
 * and add_inplace :116-198):  ct [2][Ql][N] (NTT form, ckks / bgv) <- sum_e w_e (.) rotate_e(ct), i.e. the
 * diagonal form of a plaintext-matrix x encrypted-vector product.  The weights are multiplied in before the one
 * shared mod-down, so weights[e] (HOST array of DEVICE pointers) is the plaintext over [Q_l || P], NTT form,
 * [Ql + size_P][N].  Galois element 1 (main diagonal) takes no key: glk[e] may be NULL there. */
int pha_hoisting_weighted(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint32_t *galois_elts, size_t n_elts,
                          const uint64_t *const *const *glk, const uint64_t *const *weights, int scheme, void *stream);

/* Baby-step / giant-step form of the same sum (build-defined, BASELINE config 5): with d = n_giant * n_baby diagonals
 *     ct <- sum_i rot_{giant_elts[i]}( sum_j weights[i * n_baby + j] (.) rot_{baby_elts[j]}(ct) )
 * from n_baby - 1 + n_giant - 1 Galois keys instead of d - 1 (element 1 = no rotation, no key; a null weight = no such term).
 * The baby rotations share one mod-up and stay in the extended base ("double hoisting"): giant step i is exactly
 * pha_hoisting_weighted(ct, baby_elts, baby_glk, weights[i * n_baby ..]) -- B_i, one mod-down each, all in one batched launch set --
 * and the giant rotations share ONE mod-down of the sum of their key-switch inner products:
 *     ct <- (sum_i perm_i(B_i0), sum_{identity i} B_i1) + moddown( sum_{keyed i} <modup(perm_i(B_i1)), giant_glk[i]> ).
 * weights[.] are device buffers [Ql + P][N] (NTT form) like pha_hoisting_weighted's, the tables are HOST arrays. ckks / bgv.
 * At most 63 keyed baby steps and 63 / beta keyed giant steps per call. */
int pha_hoisting_weighted_bsgs(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint32_t *baby_elts, size_t n_baby,
                               const uint64_t *const *const *baby_glk, const uint32_t *giant_elts, size_t n_giant,
                               const uint64_t *const *const *giant_glk, const uint64_t *const *weights, int scheme, void *stream);
/* the same for n_blocks row blocks that share the input ciphertext and every Galois key (the rows of a matrix-vector product):
 * weights [n_blocks][n_giant][n_baby], out [n_blocks][2][Ql][N] (must not overlap ct, which is only read).  The fused baby-step
 * kernel serves the (block, giant step) pairs eight at a time, so the baby keys are streamed once per eight of them; each block's
 * result is bit-identical to a one-block call. */
int pha_hoisting_weighted_bsgs_blocks(pha_context_t ctx, size_t size_Ql, const uint64_t *ct, size_t n_blocks, const uint32_t *baby_elts,
                                      size_t n_baby, const uint64_t *const *const *baby_glk, const uint32_t *giant_elts, size_t n_giant,
                                      const uint64_t *const *const *giant_glk, const uint64_t *const *weights, uint64_t *out, int scheme,
                                      void *stream);
/* PhantomSecretKey::generate_one_kswitch_key (src/secretkey.cu:297-341 with encrypt_zero_symmetric :232-295),
 * arithmetic part; the randomness comes from the caller because the PRNG (sample_uniform_poly /
 * sample_error_poly, src/prng.cu) is outside the accelerated path.  All buffers on the device:
 *   sk_ntt [QP][N] secret key, NTT form;  new_key_ntt [Q][N] key to switch from (s^2 or galois(s)), NTT form;
 *   a [dnum][QP][N] uniform residues (used as the NTT-form c1, as the reference samples it);
 *   e [dnum][QP][N] noise residues in COEFFICIENT form -- clobbered (scaled by t for bgv, then NTT in place);
 *   evk: device array of dnum device pointers, key d receives [2][QP][N] = (-(a_d s + e_d) + P new_key on the
 *   digit's limbs, a_d), the layout key_switch_inner_prod consumes. */
int pha_generate_one_kswitch_key(pha_context_t ctx, const uint64_t *sk_ntt, const uint64_t *new_key_ntt,
                                 const uint64_t *a, uint64_t *e, uint64_t *const *evk, int scheme, void *stream);
/* bfv_multiply_behz (src/evaluate.cu:447-548), the 2 x 2 case at the top data level: ct1, ct2 [2][Q][N] in
 * coefficient form -> dst [3][Q][N] in coefficient form (ct1 == ct2 takes the squaring kernels, like the
 * reference).  Needs pha_context_set_plain_modulus; the auxiliary base Bsk u {m_tilde} (src/rns.cu:392-560) and
 * its NTT tables are built on first use.  dst must not alias the inputs. */
int pha_bfv_multiply_behz(pha_context_t ctx, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst, void *stream);
/* bfv_multiply_hps with mul_tech_type::hps (src/evaluate.cu:674-818; bConv_HPS src/rns_bconv.cu:248-372;
 * scaleAndRound_HPS_QR_R src/rns.cu:1700-1746): same shapes as pha_bfv_multiply_behz.  The base R (|Q| + 1 primes
 * below the smallest q_i, src/rns.cu:687-693) and its tables are built on first use.  The double-precision
 * sums are fused multiply-add chains, as nvcc builds the reference's kernels by default. */
int pha_bfv_multiply_hps(pha_context_t ctx, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst, void *stream);
/* bfv_multiply_hps with mul_tech_type::hps_overq, no levels dropped (src/evaluate.cu:674-818, overq branches :745-751,
 * :790-792; bConv_BEHZ_var1 src/rns_bconv.cu:231-246; scaleAndRound_HPS_QlRl_Ql src/rns.cu:1748-1796).  Same shapes as
 * pha_bfv_multiply_hps.  ct1 == ct2 (the same pointer) takes the reference's squaring shortcut, whose result is Q / Rl
 * times the product of two separate objects -- kept as the reference has it. */
int pha_bfv_multiply_hps_overq(pha_context_t ctx, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst, void *stream);
/* hps_overq_leveled with size_Q - size_Ql levels dropped (mul_tech_type::hps_overq_leveled; constants src/rns.cu:897-975).
 * How many levels to drop is host arithmetic (FindLevelsToDrop, src/evaluate.cu:551-647; phantom::detail in the host
 * mirror).  All ciphertext buffers stay over the FULL base Q ([.][Q][N]); size_Ql = size_Q is plain hps_overq.
 *   pha_bfv_multiply_hps_overq_leveled -- bfv_multiply_hps, leveled branches src/evaluate.cu:709-711, :747-748, :794-795
 *   pha_scaleAndRound_HPS_Q_Ql         -- DRNSTool::scaleAndRound_HPS_Q_Ql src/rns.cu:1798-1808: [Q][N] -> [Ql][N]
 *   pha_ExpandCRTBasis_Ql_Q            -- DRNSTool::ExpandCRTBasis_Ql_Q src/rns.cu:1810-1836: [Ql][N] -> [Q][N] (in place allowed)
 *   pha_keyswitch_inplace_bfv_leveled  -- keyswitch_inplace, leveled branches src/eval_key_switch.cu:142-147, :170-175 */
int pha_bfv_multiply_hps_overq_leveled(pha_context_t ctx, size_t size_Ql, const uint64_t *ct1, const uint64_t *ct2,
                                       uint64_t *dst, void *stream);
/* bfv_mul_relin_hps with levels dropped (src/evaluate.cu:822-1027): multiply + relinearize in one call; the product's c2
 * never leaves level l, so the result differs (in its noise) from multiply followed by the leveled key switch. dst [2][Q][N] */
int pha_bfv_mul_relin_hps_overq_leveled(pha_context_t ctx, size_t size_Ql, const uint64_t *ct1, const uint64_t *ct2,
                                        const uint64_t *const *rlk, uint64_t *dst, void *stream);
int pha_scaleAndRound_HPS_Q_Ql(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
int pha_ExpandCRTBasis_Ql_Q(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
int pha_keyswitch_inplace_bfv_leveled(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *c2,
                                      const uint64_t *const *rlk, void *stream);
/* ---- the DRNSTool steps of the BFV multiplies, ONE polynomial per call (include/rns.cuh:167-200; callers bfv_multiply_behz
 *      src/evaluate.cu:447-548, bfv_multiply_hps :674-818, bfv_mul_relin_hps :822-1027), so that code written against the
 *      reference's DRNSTool links step by step.  All buffers in coefficient form, limb-major, canonical.  size_Ql names the
 *      level's tool as everywhere else; the BEHZ base Bsk and the HPS base R exist at the top data level only (size_Ql = |Q|,
 *      status -1 otherwise), the hps_overq base Rl at every level.  pha_tool_aux_sizes reports |Bsk|, |R|, |Rl| (0 where the
 *      base does not exist at that level); each pointer may be NULL.
 *   pha_fastbconv_m_tilde             rns.cu:1249-1278  src [Q][N]        -> dst [Bsk + 1][N]  (last limb modulo m_tilde = 2^32)
 *   pha_sm_mrq                        rns.cu:1290-1338  src [Bsk + 1][N]  -> dst [Bsk][N]
 *   pha_fast_floor                    rns.cu:1343-1419  (input_base_q [Q][N], input_base_Bsk [Bsk][N]) -> out_base_Bsk [Bsk][N]
 *   pha_fastbconv_sk                  rns.cu:1421-1510  input_base_Bsk [Bsk][N] -> out_base_q [Q][N]
 *   pha_scaleAndRound_HPS_QR_R        rns.cu:1700-1746  src [Q + R][N]    -> dst [R][N]
 *   pha_scaleAndRound_HPS_QlRl_Ql     rns.cu:1748-1796  src [Ql + Rl][N]  -> dst [Ql][N]
 *   pha_ExpandCRTBasis_Ql_Q_add_to_ct rns.cu:1838-1858  dst [Ql limbs] += src [Ql][N] * prod(dropped primes); size_Ql < |Q| */
int pha_tool_aux_sizes(pha_context_t ctx, size_t size_Ql, uint32_t *size_Bsk, uint32_t *size_R, uint32_t *size_Rl);
int pha_fastbconv_m_tilde(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
int pha_sm_mrq(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
int pha_fast_floor(pha_context_t ctx, size_t size_Ql, const uint64_t *input_base_q, const uint64_t *input_base_Bsk,
                   uint64_t *out_base_Bsk, void *stream);
int pha_fastbconv_sk(pha_context_t ctx, size_t size_Ql, const uint64_t *input_base_Bsk, uint64_t *out_base_q, void *stream);
int pha_scaleAndRound_HPS_QR_R(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
int pha_scaleAndRound_HPS_QlRl_Ql(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
int pha_ExpandCRTBasis_Ql_Q_add_to_ct(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream);
/* Batched modular GEMM (benchmark/matmul_bench.cu:215-541): for z in [0, batch): C[z] = A[z] * B[z] mod q, q = the
 * context prime mod_start_idx + z; row-major A [batch][m][lda], B [batch][k][ldb], C [batch][m][ldc], inputs
 * canonical (below q; q <= 60 bits).  Exact (the reference's benchmark kernels lose the carries of the low product word,
 * :231-232).  Runs on the matrix cores (i8 digit planes, csrc/pha_gemm.hip).  Limits: k <= 16384, batch <= 65535, leading
 * dimensions below 2^23; moduli above 2^50 anywhere in the batch take the 8-digit path for all of it (k in runs of 128,
 * one launch per run). */
int pha_batched_modular_gemm(pha_context_t ctx, uint64_t *C, size_t ldc, const uint64_t *A, size_t lda, const uint64_t *B,
                             size_t ldb, size_t m, size_t n, size_t k, size_t batch, size_t mod_start_idx, void *stream);
/* DRNSTool::mod_t_and_divide_q_last_ntt (rns.cu:1210-1236), the BGV modulus switch: src [cipher][Ql][N]
 * in NTT form (left in coefficient form, as in the reference) -> dst [cipher][Ql-1][N] in NTT form */
int pha_mod_t_and_divide_q_last_ntt(pha_context_t ctx, size_t size_Ql, uint64_t *src, size_t cipher_size,
                                    uint64_t *dst, void *stream);
/* DRNSTool::divide_and_round_q_last_ntt (rns.cu:1160-1184): src [cipher][Ql][N] (last limb is
 * clobbered, as in the reference) -> dst [cipher][Ql-1][N] */
int pha_divide_and_round_q_last_ntt(pha_context_t ctx, size_t size_Ql, uint64_t *src, size_t cipher_size,
                                    uint64_t *dst, void *stream);
/* DRNSTool::divide_and_round_q_last (rns.cu:1113-1126), BFV coefficient-domain mod switch */
int pha_divide_and_round_q_last(pha_context_t ctx, size_t size_Ql, const uint64_t *src, size_t cipher_size,
                                uint64_t *dst, void *stream);

/* ---- ciphertext (+|-|*) plaintext (src/evaluate.cu:1105-1340): the device work of add_plain_inplace,
 *      sub_plain_inplace and multiply_plain_inplace, with DRNSTool's per-level constants (src/rns.cu:292-324) kept by
 *      the context (pha_context_set_plain_modulus).  CKKS needs nothing beyond pha_add/sub/multiply_rns_poly. ---- */
/* multiply_add_plain_with_scaling_variant / multiply_sub_plain_with_scaling_variant (src/scalingvariant.cu:10-60):
 * ct[0] (coefficient form, [Ql][N]) +- round-free (Ql / t) * plain, plain = N coefficients below t */
int pha_bfv_add_plain(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *plain, int subtract, void *stream);
/* multiply_plain_normal (src/evaluate.cu:1256-1300): every polynomial of ct [cipher_size][Ql][N] (coefficient form)
 * times the centred lift of plain, through the NTT */
int pha_bfv_multiply_plain(pha_context_t ctx, size_t size_Ql, uint64_t *ct, size_t cipher_size, const uint64_t *plain,
                           void *stream);
/* BGV: out [Ql][N] = NTT of plain modulo every q_i -- the nwt_2d_radix8_forward_modup_fuse loop of
 * src/evaluate.cu:1150-1154, 1208-1212, 1319-1323 as one launch; the caller then applies
 * multiply_scalar_and_add/sub (correction factor) or multiply_rns_poly */
int pha_bgv_lift_plain(pha_context_t ctx, size_t size_Ql, const uint64_t *plain, uint64_t *out, void *stream);

/* ---- Galois (include/galois.cuh:98-130, src/galois.cu:11-39,67-102) ---- */
int pha_apply_galois_ntt(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt,
                         size_t coeff_mod_size, void *stream);
int pha_apply_galois(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt,
                     size_t coeff_mod_size, size_t mod_start_idx, void *stream);
/* the same permutation on `polys` polynomials [polys][coeff_mod_size][N] in one launch (ntt_form != 0: the NTT-domain
 * table permutation, else the coefficient-domain one with its sign) */
int pha_apply_galois_batched(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt,
                             size_t coeff_mod_size, size_t polys, int ntt_form, void *stream);
/* Build-defined helper for rotate on batches (BASELINE config 4): the automorphism of `batch` size-2 ciphertexts
   src [batch][2][Ql][N], written in the layout the key switch that follows needs (rotate_internal / apply_galois_inplace,
   src/evaluate.cu:1567-1624): dst_ct [batch][2][Ql][N] receives (galois(c0), 0), dst_c2 [batch][Ql][N] receives galois(c1).
   ntt_form != 0: NTT-domain permutation (ckks / bgv); 0: coefficient-domain with sign (bfv). */
int pha_apply_galois_for_keyswitch(pha_context_t ctx, const uint64_t *src, uint64_t *dst_ct, uint64_t *dst_c2,
                                   uint32_t galois_elt, size_t size_Ql, size_t batch, int ntt_form, void *stream);

/* BASELINE config 4 as one call (build-defined composition of relinearize_inplace src/evaluate.cu:1028-1077 and
 * apply_galois_inplace :1567-1624 over a batch, no reference launcher): out [batch][2][Ql][N] =
 * rotate_{galois_elt}(relinearize(ct3 [batch][3][Ql][N])), i.e. per ciphertext keyswitch_inplace with rlk, the automorphism, and
 * keyswitch_inplace with glk -- bit-identical to those calls.  ct3 is only read (out must not overlap it); the batch runs
 * `chunk` ciphertexts per set of launches (0 = sized so that a set's mod-up digits stay within the 256 MiB MALL, and the sets
 * alternate between two streams the context owns, forked from and joined back into `stream` with events: warm the call up once
 * before capturing it into a graph); no copies. */
int pha_relinearize_rotate_batched(pha_context_t ctx, size_t size_Ql, const uint64_t *ct3, size_t batch,
                                   const uint64_t *const *rlk, const uint64_t *const *glk, uint32_t galois_elt, int scheme,
                                   uint64_t *out, size_t chunk, void *stream);

/* ---- multi-GPU (the reference has none): one-time RCCL broadcast of evaluation / Galois keys.  keys[i] (HOST array
 *      of n_keys DEVICE buffers, words_per_key uint64 words each -- for a PhantomRelinKey the dnum buffers of 2 * #QP * N words,
 *      include/secretkey.h:102-165) are sent from rank `root` of `nccl_comm` (an ncclComm_t the caller created for its ranks, one
 *      per GPU) to every other rank, in place, as one RCCL group on `stream`.  No collective exists on the data path: after this
 *      call every rank key-switches its own ciphertexts.  RCCL is resolved with dlopen at first use (no link-time dependency). ---- */
int pha_broadcast_keys(pha_context_t ctx, uint64_t *const *keys, size_t n_keys, size_t words_per_key, int root, void *nccl_comm,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif
