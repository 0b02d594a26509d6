// pha_plain.hip -- ciphertext (+|-|*) plaintext on raw buffers: the device work of add_plain_inplace /
// sub_plain_inplace / multiply_plain_inplace (src/evaluate.cu:1105-1226, 1228-1340, src/scalingvariant.cu:10-60) with
// the per-level constants of DRNSTool (src/rns.cu:292-324) taken from the context.  Compositions of the kernels in
// pha_poly_ext.hip / pha_ntt.hip; the host mirror (host/phantom.h) keeps the reference's checks and metadata.
#include "../../include/phantom_amd.h"
#include "pha_internal.h"
#include "pha_ntt_core.h"

using namespace pha;

static void need(const void *p) {
    if (!p) throw std::invalid_argument("null device pointer");
}
static void rc(int status) {   // an inner ABI call failed: re-raise with its own category
    if (status == 0) return;
    const std::string what = pha_last_error();
    if (status == -1) throw std::invalid_argument(what);
    if (status == -2) throw std::logic_error(what);
    throw std::runtime_error(what);
}
static Tool &plain_tool(Context &c, size_t size_Ql) {
    if (size_Ql < 1 || size_Ql > c.size_q) throw std::invalid_argument("RNSBase is invalid");
    if (!c.plain_t) throw std::invalid_argument("the context has no plain modulus (pha_context_set_plain_modulus)");
    return c.tool((uint32_t)size_Ql);
}

extern "C" {

int pha_bfv_add_plain(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *plain, int subtract, void *stream) {
    PHA_CTX_BEGIN(ctx)   // multiply_add_plain_with_scaling_variant / multiply_sub_plain_with_scaling_variant
    need(ct); need(plain);
    Context &c = ctx->c;
    Tool &t = plain_tool(c, size_Ql);
    if (subtract)
        rc(pha_bfv_sub_timesQ_overt(ctx, ct, plain, t.neg_ql_mod_t.x, t.neg_ql_mod_t.y, t.t_inv_mod_q.p,
                                    t.t_inv_mod_q_shoup.p, c.plain_t, size_Ql, stream));
    else
        rc(pha_bfv_add_timesQ_overt(ctx, ct, plain, t.neg_ql_mod_t.x, t.neg_ql_mod_t.y, t.t_inv_mod_q.p,
                                    t.t_inv_mod_q_shoup.p, c.plain_t, size_Ql, stream));
    PHA_API_END
}

int pha_bfv_multiply_plain(pha_context_t ctx, size_t size_Ql, uint64_t *ct, size_t cipher_size, const uint64_t *plain,
                           void *stream) {
    PHA_CTX_BEGIN(ctx)   // multiply_plain_normal evaluate.cu:1256-1300
    need(ct); need(plain);
    Context &c = ctx->c;
    Tool &t = plain_tool(c, size_Ql);
    for (uint32_t i = 0; i < size_Ql; i++)
        if (c.primes[i] <= c.plain_t) throw std::invalid_argument("the plain lift needs t below every q_i");
    if (cipher_size == 0) return 0;
    if (cipher_size > 65535) throw std::invalid_argument("cipher_size out of range");
    const size_t ln = size_Ql * c.n;
    u64 *temp = c.scratch(stream, ln);
    // centred lift of the plaintext into every limb (:1283-1285), then NTT
    rc(pha_abs_plain_rns_poly(ctx, plain, (c.plain_t + 1) >> 1, t.plain_upper_half_increment.p, temp, size_Ql, stream));
    rc(pha_nwt_2d_radix8_forward_inplace(ctx, temp, size_Ql, 0, stream));
    // (c_i * pt): NTT, pointwise product, inverse NTT -- all polynomials per launch
    rc(pha_nwt_2d_radix8_forward_inplace_batched(ctx, ct, size_Ql, 0, cipher_size, ln, stream));
    for (size_t i = 0; i < cipher_size; i++) rc(pha_multiply_rns_poly(ctx, ct + i * ln, temp, ct + i * ln, size_Ql, 0, stream));
    rc(pha_nwt_2d_radix8_backward_inplace_batched(ctx, ct, size_Ql, 0, cipher_size, ln, stream));
    PHA_API_END
}

int pha_bgv_lift_plain(pha_context_t ctx, size_t size_Ql, const uint64_t *plain, uint64_t *out, void *stream) {
    PHA_CTX_BEGIN(ctx)   // the modup_fuse loop of evaluate.cu:1150-1154 / 1208-1212 / 1319-1323, every limb in one launch
    need(plain); need(out);
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql > c.size_q) throw std::invalid_argument("RNSBase is invalid");
    NttExtra x;
    x.pro_src = plain;   // each limb transforms the same N coefficients, reduced modulo its own prime on load
    ntt_forward(c, out, out, out, plain_sel(0, size_Ql), EPI_FWD_CANON, x, as_stream(stream));
    PHA_API_END
}

}  // extern "C"
