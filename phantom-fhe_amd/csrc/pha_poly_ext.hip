// pha_poly_ext.hip -- the rest of the reference's residue-wise kernels (include/polymath.cuh:6-307, src/polymath.cu):
// the ones its encryption / decryption / plaintext layers launch around the hot path.  Same conventions as
// pha_poly.hip: limb = blockIdx.y (modulus in SGPRs), one coefficient per thread; results are canonical residues,
// equal to the reference's for the inputs it accepts.
#include "../../include/phantom_amd.h"
#include "pha_internal.h"

namespace pha {

struct XArgs {
    const u64 *a, *b, *d;
    u64 *r;
    const u64 *s0, *s1;           // per-limb constants
    const u64 *const *many;       // add_many: device array of operand pointers
    const DModulus *mod;
    u64 c0, c1, c2;               // uniform constants
    uint32_t n, mod_start, count, m1, m2;
    size_t poly_stride;           // elements between the polynomials of a ciphertext
};

enum XOp { X_ADDNEG, X_ADDMANY, X_MULUNI, X_MULS_ADD, X_MULS_SUB, X_MUL_SCALEADD, X_MULADD_NEG, X_SUBSCALE, X_SUBSCALE_1,
           X_BFV_ADD, X_BFV_SUB, X_ABS_PLAIN, X_MXN, X_MUL_NEGADD };

constexpr int kMaxCipher = 8;     // tensor_prod_mxn: polynomials per operand kept in registers

template <int OP>
__global__ __launch_bounds__(256) void x_kernel(const XArgs k) {
    const uint32_t limb = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)limb * k.n + i;
    DModulus m = OP == X_SUBSCALE_1 ? DModulus{k.c2, 0, 0} : k.mod[k.mod_start + limb];
    const u64 q = m.value;
    if (OP == X_ADDNEG) {                 // add_and_negate_rns_poly polymath.cu:82-98
        k.r[id] = neg_mod(add_mod(k.a[id], k.b[id], q), q);
    } else if (OP == X_ADDMANY) {         // add_many_rns_poly :126-147 (sum, Barrett-64 after every term)
        u64 t = k.many[0][id + k.poly_stride];
        for (uint32_t e = 1; e < k.count; e++) t = barrett64(t + k.many[e][id + k.poly_stride], q, m.ratio1);
        k.r[id + k.poly_stride] = t;
    } else if (OP == X_MULUNI) {          // multiply_scalar_rns_poly, one scalar for all limbs :181-196
        k.r[id] = mul_mod(k.a[id], k.c0, m);
    } else if (OP == X_MULS_ADD) {        // multiply_scalar_and_add_rns_poly :246-264: a + b * scalar
        u64 lo, hi;
        mul128(k.b[id], k.c0, lo, hi);
        const u64 x = k.a[id];
        lo += x; hi += (lo < x);
        k.r[id] = barrett128(lo, hi, m);
    } else if (OP == X_MULS_SUB) {        // multiply_scalar_and_sub_rns_poly :266-283: a - b * scalar
        k.r[id] = sub_mod(k.a[id], mul_mod(k.b[id], k.c0, m), q);
    } else if (OP == X_MUL_SCALEADD) {    // multiply_and_scale_add_rns_poly :294-315: a * b + d * scale
        u64 lo, hi;
        mul128(k.a[id], k.b[id], lo, hi);
        mac128(k.d[id], k.c0, lo, hi);
        k.r[id] = barrett128(lo, hi, m);
    } else if (OP == X_MULADD_NEG) {      // multiply_and_add_negate_rns_poly :350-371: -(a * b + d)
        u64 lo, hi;
        mul128(k.a[id], k.b[id], lo, hi);
        const u64 x = k.d[id];
        lo += x; hi += (lo < x);
        k.r[id] = neg_mod(barrett128(lo, hi, m), q);
    } else if (OP == X_SUBSCALE) {        // sub_and_scale_rns_poly :392-411: (a - b) * scale_limb
        k.r[id] = shoup(k.a[id] + (q - k.b[id]), u64x2{k.s0[limb], k.s1[limb]}, q);
    } else if (OP == X_SUBSCALE_1) {      // sub_and_scale_single_mod_poly :374-390, explicit modulus
        k.r[i] = shoup(k.a[i] + (q - k.b[i]), u64x2{k.c0, k.c1}, q);
    } else if (OP == X_BFV_ADD || OP == X_BFV_SUB) {   // bfv_{add,sub}_timesQ_overt_kernel :413-461
        const u64 t = k.c2;
        const u64 mq = shoup(k.b[i], u64x2{k.c0, k.c1}, t);                       // m * (-Ql mod t) mod t
        const u64 v = shoup(mq, u64x2{k.s0[limb], k.s1[limb]}, q);               // * t^-1 mod q_i
        k.r[id] = OP == X_BFV_ADD ? add_mod(k.r[id], v, q) : sub_mod(k.r[id], v, q);
    } else if (OP == X_ABS_PLAIN) {       // abs_plain_rns_poly :645-664
        u64 v = k.a[i];
        if (v >= k.c0) v += k.s0[limb];
        k.r[id] = v;
    } else if (OP == X_MXN) {             // tensor_prod_mxn_rns_poly :546-592 (registers instead of device new[])
        u64 c1[kMaxCipher], c2[kMaxCipher];
        for (uint32_t e = 0; e < kMaxCipher; e++) {
            c1[e] = e < k.m1 ? k.a[id + e * k.poly_stride] : 0;
            c2[e] = e < k.m2 ? k.b[id + e * k.poly_stride] : 0;
        }
        for (uint32_t j = 0; j < k.count; j++) {
            const uint32_t last1 = j < k.m1 - 1 ? j : k.m1 - 1, first2 = j < k.m2 - 1 ? j : k.m2 - 1, first1 = j - first2;
            u64 lo = 0, hi = 0;
            for (uint32_t e = 0; e + first1 <= last1; e++) mac128(c1[first1 + e], c2[first2 - e], lo, hi);
            k.r[id + j * k.poly_stride] = barrett128(lo, hi, m);
        }
    } else if (OP == X_MUL_NEGADD) {      // multiply_and_negated_add_rns_poly :606-634 (BEHZ FastBconvSK fix-up)
        u64 op1 = k.a[i], pb = k.s0[limb];
        if (op1 > (k.c0 >> 1)) op1 = k.c0 - op1;
        else pb = q - pb;
        k.r[id] = add_mod(k.d[id], mul_mod(op1, pb, m), q);
    }
}

template <int OP>
static void launch_x(Context &c, XArgs k, size_t limbs, size_t mod_start, hipStream_t s) {
    if (limbs == 0) return;
    if (OP != X_SUBSCALE_1 && mod_start + limbs > c.rows) throw std::invalid_argument("modulus index out of range");
    k.mod = c.d_mod.p;
    k.n = (uint32_t)c.n;
    k.mod_start = (uint32_t)mod_start;
    hipLaunchKernelGGL((x_kernel<OP>), dim3((unsigned)(c.n / 256), (unsigned)limbs), dim3(256), 0, s, k);
    check_launch();
}

}  // namespace pha

using namespace pha;

static void need(const void *p) {
    if (!p) throw std::invalid_argument("null device pointer");
}

extern "C" {

int pha_add_std_cipher(pha_context_t ctx, const uint64_t *cipher1, const uint64_t *cipher2, uint64_t *result, size_t cms,
                       void *stream) {
    PHA_CTX_BEGIN(ctx)   // add_std_cipher polymath.cu:56-73: both polynomials of a size-2 ciphertext
    need(cipher1); need(cipher2); need(result);
    const size_t ln = cms * ctx->c.n;
    launch_add(ctx->c, cipher1, cipher2, result, cms, 0, as_stream(stream));
    launch_add(ctx->c, cipher1 + ln, cipher2 + ln, result + ln, cms, 0, as_stream(stream));
    PHA_API_END
}

int pha_add_and_negate_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, uint64_t *r, size_t cms,
                                size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(r);
    XArgs k{};
    k.a = a; k.b = b; k.r = r;
    launch_x<X_ADDNEG>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}

int pha_add_many_rns_poly(pha_context_t ctx, const uint64_t *const *operands, size_t add_size, uint64_t *result,
                          size_t poly_index, size_t cms, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(operands); need(result);
    if (add_size == 0 || add_size > 65535) throw std::invalid_argument("add_size out of range");
    for (size_t e = 0; e < add_size; e++) need(operands[e]);
    u64 *tab = ctx->c.scratch(stream, add_size);
    PHA_HIP(hipMemcpyAsync(tab, operands, add_size * sizeof(void *), hipMemcpyHostToDevice, as_stream(stream)));
    XArgs k{};
    k.many = reinterpret_cast<const u64 *const *>(tab);
    k.r = result;
    k.count = (uint32_t)add_size;
    k.poly_stride = poly_index * cms * ctx->c.n;
    launch_x<X_ADDMANY>(ctx->c, k, cms, 0, as_stream(stream));
    PHA_API_END
}

int pha_multiply_uniform_scalar_rns_poly(pha_context_t ctx, const uint64_t *a, uint64_t scale, uint64_t *r, size_t cms,
                                         size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(r);
    XArgs k{};
    k.a = a; k.r = r; k.c0 = scale;
    launch_x<X_MULUNI>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}

int pha_multiply_scalar_and_add_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, uint64_t scalar,
                                         uint64_t *r, size_t cms, size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(r);
    XArgs k{};
    k.a = a; k.b = b; k.r = r; k.c0 = scalar;
    launch_x<X_MULS_ADD>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}

int pha_multiply_scalar_and_sub_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, uint64_t scalar,
                                         uint64_t *r, size_t cms, size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(r);
    XArgs k{};
    k.a = a; k.b = b; k.r = r; k.c0 = scalar;
    launch_x<X_MULS_SUB>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}

int pha_multiply_and_scale_add_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, const uint64_t *d,
                                        uint64_t scale, uint64_t *r, size_t cms, size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(d); need(r);
    XArgs k{};
    k.a = a; k.b = b; k.d = d; k.r = r; k.c0 = scale;
    launch_x<X_MUL_SCALEADD>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}

int pha_multiply_and_add_negate_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, const uint64_t *d,
                                         uint64_t *r, size_t cms, size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(d); need(r);
    XArgs k{};
    k.a = a; k.b = b; k.d = d; k.r = r;
    launch_x<X_MULADD_NEG>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}

int pha_sub_and_scale_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, const uint64_t *scale,
                               const uint64_t *scale_shoup, uint64_t *r, size_t cms, size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(scale); need(scale_shoup); need(r);
    XArgs k{};
    k.a = a; k.b = b; k.s0 = scale; k.s1 = scale_shoup; k.r = r;
    launch_x<X_SUBSCALE>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}

int pha_sub_and_scale_single_mod_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, uint64_t scale,
                                      uint64_t scale_shoup, uint64_t modulus, uint64_t *r, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(r);
    if (modulus < 2 || modulus >> 62) throw std::invalid_argument("modulus out of range");
    XArgs k{};
    k.a = a; k.b = b; k.r = r; k.c0 = scale; k.c1 = scale_shoup; k.c2 = modulus;
    launch_x<X_SUBSCALE_1>(ctx->c, k, 1, 0, as_stream(stream));
    PHA_API_END
}

static void bfv_times_q_over_t(pha_context_t ctx, bool add, uint64_t *ct, const uint64_t *pt, uint64_t neg_ql_mod_t,
                               uint64_t neg_ql_mod_t_shoup, const uint64_t *t_inv_mod_q,
                               const uint64_t *t_inv_mod_q_shoup, uint64_t t, size_t size_Ql, void *stream) {
    need(ct); need(pt); need(t_inv_mod_q); need(t_inv_mod_q_shoup);
    if (t < 2 || t >> 62) throw std::invalid_argument("plain modulus out of range");
    XArgs k{};
    k.b = pt; k.r = ct; k.c0 = neg_ql_mod_t; k.c1 = neg_ql_mod_t_shoup; k.c2 = t; k.s0 = t_inv_mod_q; k.s1 = t_inv_mod_q_shoup;
    if (add) launch_x<X_BFV_ADD>(ctx->c, k, size_Ql, 0, as_stream(stream));
    else launch_x<X_BFV_SUB>(ctx->c, k, size_Ql, 0, as_stream(stream));
}

int pha_bfv_add_timesQ_overt(pha_context_t ctx, uint64_t *ct, const uint64_t *pt, uint64_t neg_ql_mod_t,
                             uint64_t neg_ql_mod_t_shoup, const uint64_t *t_inv_mod_q, const uint64_t *t_inv_mod_q_shoup,
                             uint64_t t, size_t size_Ql, void *stream) {
    PHA_CTX_BEGIN(ctx)
    bfv_times_q_over_t(ctx, true, ct, pt, neg_ql_mod_t, neg_ql_mod_t_shoup, t_inv_mod_q, t_inv_mod_q_shoup, t, size_Ql, stream);
    PHA_API_END
}

int pha_bfv_sub_timesQ_overt(pha_context_t ctx, uint64_t *ct, const uint64_t *pt, uint64_t neg_ql_mod_t,
                             uint64_t neg_ql_mod_t_shoup, const uint64_t *t_inv_mod_q, const uint64_t *t_inv_mod_q_shoup,
                             uint64_t t, size_t size_Ql, void *stream) {
    PHA_CTX_BEGIN(ctx)
    bfv_times_q_over_t(ctx, false, ct, pt, neg_ql_mod_t, neg_ql_mod_t_shoup, t_inv_mod_q, t_inv_mod_q_shoup, t, size_Ql, stream);
    PHA_API_END
}

int pha_abs_plain_rns_poly(pha_context_t ctx, const uint64_t *operand, uint64_t plain_upper_half_threshold,
                           const uint64_t *plain_upper_half_increment, uint64_t *result, size_t cms, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(operand); need(plain_upper_half_increment); need(result);
    XArgs k{};
    k.a = operand; k.r = result; k.c0 = plain_upper_half_threshold; k.s0 = plain_upper_half_increment;
    launch_x<X_ABS_PLAIN>(ctx->c, k, cms, 0, as_stream(stream));
    PHA_API_END
}

int pha_tensor_prod_mxn_rns_poly(pha_context_t ctx, const uint64_t *op1, size_t op1_size, const uint64_t *op2,
                                 size_t op2_size, uint64_t *result, size_t res_size, size_t cms, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(op1); need(op2); need(result);
    if (op1_size == 0 || op2_size == 0 || op1_size > kMaxCipher || op2_size > kMaxCipher)
        throw std::invalid_argument("ciphertext size out of range (1..8 polynomials)");
    if (res_size != op1_size + op2_size - 1) throw std::invalid_argument("res_size must be op1_size + op2_size - 1");
    XArgs k{};
    k.a = op1; k.b = op2; k.r = result;
    k.m1 = (uint32_t)op1_size; k.m2 = (uint32_t)op2_size; k.count = (uint32_t)res_size;
    k.poly_stride = cms * ctx->c.n;
    launch_x<X_MXN>(ctx->c, k, cms, 0, as_stream(stream));
    PHA_API_END
}

int pha_multiply_and_negated_add_rns_poly(pha_context_t ctx, const uint64_t *alpha_sk, uint64_t m_sk,
                                          const uint64_t *prod_B_mod_q, const uint64_t *operand3, uint64_t *result,
                                          size_t cms, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(alpha_sk); need(prod_B_mod_q); need(operand3); need(result);
    XArgs k{};
    k.a = alpha_sk; k.c0 = m_sk; k.s0 = prod_B_mod_q; k.d = operand3; k.r = result;
    launch_x<X_MUL_NEGADD>(ctx->c, k, cms, 0, as_stream(stream));
    PHA_API_END
}

}  // extern "C"
