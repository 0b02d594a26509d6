// pha_poly.hip -- residue-wise (dyadic) kernels over [limb][coeff] buffers.
//
// Reference: src/polymath.cu (one coefficient per thread, `twr = tid / N`, DModulus re-read per
// thread).  Here the limb is blockIdx.y, so the modulus and its Barrett ratio sit in SGPRs, and
// every thread moves two adjacent coefficients with 16-byte accesses (pure HBM streaming).
#include "../../include/phantom_amd.h"
#include "pha_internal.h"

namespace pha {

constexpr int kEwThreads = 256;
constexpr int kEwPerThread = 2;

struct EwArgs {
    const u64 *a, *b, *d;
    u64 *r;
    u64 *r2;             // tensor product: where c2 goes (null = r + 2 * limbs * n, the reference layout)
    const u64 *s0, *s1;  // per-limb scalars (Shoup pair)
    const DModulus *mod;
    uint32_t n, limbs, mod_start;
    uint32_t poly_limbs;  // limbs between two polynomials of a ciphertext (0 = limbs)
    size_t za, zb, zr, zr2;   // blockIdx.z (batched tensor product): elements between consecutive ciphertexts of a / b / r / r2
    const FpInfo *fpinfo;     // [prime]: limbs below 2^50 form the tensor product in FP64 (r04)
};

// EW_TENSOR_NT (r06): the tensor product of a BATCH of ciphertexts: every word is read once and written once and the batch is larger
// than the MALL from two ciphertexts on (161 MB per pair at C3), so its loads and stores are nontemporal -- they stream past the L2 / MALL
// contents the neighbouring kernels live on.  One box, batched HomMul + relinearize + rescale with this and the batched inner product's
// streams nontemporal: 279.3 / 277.9 -> 275.5 / 274.8 us per op at B = 32, 283.8 / 281.7 -> 277.9 / 275.6 at B = 8
// (profiles/r06_experiments.md section 7).  A single ciphertext keeps the default policy: its c2 is read back from the MALL right away.
enum EwOp { EW_ADD, EW_SUB, EW_NEG, EW_MUL, EW_MULADD, EW_MULSCALAR, EW_TENSOR, EW_SQUARE, EW_TENSOR_NT };

typedef unsigned long long ew_v2 __attribute__((ext_vector_type(2)));
template <bool NT = false>
__device__ __forceinline__ u64x2 ld2(const u64 *p) {
    if (NT) {
        const ew_v2 t = __builtin_nontemporal_load(reinterpret_cast<const ew_v2 *>(p));
        return u64x2{t.x, t.y};
    }
    return *reinterpret_cast<const u64x2 *>(p);
}
template <bool NT = false>
__device__ __forceinline__ void st2(u64 *p, u64x2 v) {
    if (NT) {
        __builtin_nontemporal_store(ew_v2{v.x, v.y}, reinterpret_cast<ew_v2 *>(p));
        return;
    }
    *reinterpret_cast<u64x2 *>(p) = v;
}

template <int OP>
__global__ __launch_bounds__(kEwThreads) void ew_kernel(const EwArgs k) {
    const uint32_t limb = blockIdx.y;
    const DModulus m = k.mod[k.mod_start + limb];
    const u64 q = m.value;
    const size_t idx = (size_t)limb * k.n + ((size_t)blockIdx.x * kEwThreads + threadIdx.x) * kEwPerThread;
    const size_t rc = (size_t)(k.poly_limbs ? k.poly_limbs : k.limbs) * k.n;  // stride between the polynomials of a ciphertext

    if (OP == EW_ADD) {  // add_rns_poly polymath.cu:41-56
        u64x2 x = ld2(k.a + idx), y = ld2(k.b + idx);
        st2(k.r + idx, u64x2{add_mod(x.x, y.x, q), add_mod(x.y, y.y, q)});
    } else if (OP == EW_SUB) {  // sub_rns_poly :109-124
        u64x2 x = ld2(k.a + idx), y = ld2(k.b + idx);
        st2(k.r + idx, u64x2{sub_mod(x.x, y.x, q), sub_mod(x.y, y.y, q)});
    } else if (OP == EW_NEG) {  // negate_rns_poly :17-32
        u64x2 x = ld2(k.a + idx);
        st2(k.r + idx, u64x2{neg_mod(x.x, q), neg_mod(x.y, q)});
    } else if (OP == EW_MUL) {  // multiply_rns_poly :156-172
        u64x2 x = ld2(k.a + idx), y = ld2(k.b + idx);
        st2(k.r + idx, u64x2{mul_mod(x.x, y.x, m), mul_mod(x.y, y.y, m)});
    } else if (OP == EW_MULADD) {  // multiply_and_add_rns_poly :225-244 (128-bit sum, one Barrett)
        u64x2 x = ld2(k.a + idx), y = ld2(k.b + idx), z = ld2(k.d + idx);
        u64 lo, hi, lo2, hi2;
        mul128(x.x, y.x, lo, hi);
        lo += z.x; hi += (lo < z.x);
        mul128(x.y, y.y, lo2, hi2);
        lo2 += z.y; hi2 += (lo2 < z.y);
        st2(k.r + idx, u64x2{barrett128(lo, hi, m), barrett128(lo2, hi2, m)});
    } else if (OP == EW_MULSCALAR) {  // multiply_scalar_rns_poly (Shoup) :198-213
        const u64x2 w{k.s0[limb], k.s1[limb]};
        u64x2 x = ld2(k.a + idx);
        st2(k.r + idx, u64x2{shoup(x.x, w, q), shoup(x.y, w, q)});
    } else if (OP == EW_TENSOR || OP == EW_TENSOR_NT) {  // tensor_prod_2x2_rns_poly :463-496; blockIdx.z walks a batch of ciphertexts (r04: one launch)
        const size_t z = blockIdx.z;
        const u64 *ka = k.a + z * k.za, *kb = k.b + z * k.zb;
        u64 *kr = k.r + z * k.zr, *kr2 = k.r2 ? k.r2 + z * k.zr2 : nullptr;
        constexpr bool NT = OP == EW_TENSOR_NT;
        u64x2 c00 = ld2<NT>(ka + idx), c01 = ld2<NT>(ka + idx + rc), c10 = ld2<NT>(kb + idx), c11 = ld2<NT>(kb + idx + rc);
        u64x2 d0, d1, d2;
        if (k.fpinfo && k.fpinfo[k.mod_start + limb].ok) {   // (uniform) r04: three exact FP64 products (fp_tensor_2x2, pha_arith.h) instead of
            // three Barrett-128 multiplies on 32-bit halves (~115 vector instructions per coefficient -> ~50); the same residues as :487-:494
            const FpInfo fi = k.fpinfo[k.mod_start + limb];
            const FpMod fm{fi.q, fi.qinv, false, false};
            u64 e[6];
            fp_tensor_2x2(c00.x, c01.x, c10.x, c11.x, fm, e[0], e[1], e[2]);
            fp_tensor_2x2(c00.y, c01.y, c10.y, c11.y, fm, e[3], e[4], e[5]);
            d0 = u64x2{e[0], e[3]}; d1 = u64x2{e[1], e[4]}; d2 = u64x2{e[2], e[5]};
            st2<NT>(kr + idx, d0);
            st2<NT>(kr + idx + rc, d1);
            st2<NT>(kr2 ? kr2 + idx : kr + idx + 2 * rc, d2);
            return;
        }
        d0.x = mul_mod(c00.x, c10.x, m); d0.y = mul_mod(c00.y, c10.y, m);
        d2.x = mul_mod(c01.x, c11.x, m); d2.y = mul_mod(c01.y, c11.y, m);
        // (c0 + c1) is not reduced before the multiply (q < 2^61), exactly like :487
        d1.x = mul_mod(c00.x + c01.x, c10.x + c11.x, m);
        d1.y = mul_mod(c00.y + c01.y, c10.y + c11.y, m);
        d1.x = csub(csub(d1.x + 2 * q - d0.x - d2.x, q), q);
        d1.y = csub(csub(d1.y + 2 * q - d0.y - d2.y, q), q);
        st2<NT>(kr + idx, d0);
        st2<NT>(kr + idx + rc, d1);
        st2<NT>(kr2 ? kr2 + idx : kr + idx + 2 * rc, d2);
    } else if (OP == EW_SQUARE) {  // tensor_square_2x2_rns_poly :500-529
        u64x2 c0 = ld2(k.a + idx), c1 = ld2(k.a + idx + rc);
        u64x2 d0, d1, d2;
        if (k.fpinfo && k.fpinfo[k.mod_start + limb].ok) {   // (uniform) the FP64 form: c0^2, 2 c0 c1, c1^2 (fp_square_2x2)
            const FpInfo fi = k.fpinfo[k.mod_start + limb];
            const FpMod fm{fi.q, fi.qinv, false, false};
            u64 e[6];
            fp_square_2x2(c0.x, c1.x, fm, e[0], e[1], e[2]);
            fp_square_2x2(c0.y, c1.y, fm, e[3], e[4], e[5]);
            d0 = u64x2{e[0], e[3]}; d1 = u64x2{e[1], e[4]}; d2 = u64x2{e[2], e[5]};
            st2(k.r + idx, d0);
            st2(k.r + idx + rc, d1);
            st2(k.r + idx + 2 * rc, d2);
            return;
        }
        d0.x = mul_mod(c0.x, c0.x, m); d0.y = mul_mod(c0.y, c0.y, m);
        u64 lo, hi;
        mul128(c0.x, c1.x, lo, hi);
        d1.x = barrett128(lo << 1, (hi << 1) | (lo >> 63), m);
        mul128(c0.y, c1.y, lo, hi);
        d1.y = barrett128(lo << 1, (hi << 1) | (lo >> 63), m);
        d2.x = mul_mod(c1.x, c1.x, m); d2.y = mul_mod(c1.y, c1.y, m);
        st2(k.r + idx, d0);
        st2(k.r + idx + rc, d1);
        st2(k.r + idx + 2 * rc, d2);
    }
}

template <int OP>
static void launch_ew(Context &c, EwArgs k, size_t limbs, size_t mod_start, hipStream_t s, size_t batch = 1) {
    if (limbs == 0 || batch == 0) return;
    // rows past size_qp are the auxiliary BFV bases (the reference's callers hand these kernels base_Bsk / base_Rl moduli,
    // src/evaluate.cu:489-497); they exist once a BFV multiply entry or pha_tool_aux_sizes has built them
    if (mod_start + limbs > c.rows) throw std::invalid_argument("modulus index out of range");
    k.mod = c.d_mod.p;
    k.fpinfo = c.d_fpinfo.p;
    k.n = (uint32_t)c.n;
    k.limbs = (uint32_t)limbs;
    k.mod_start = (uint32_t)mod_start;
    dim3 grid((unsigned)(c.n / (kEwThreads * kEwPerThread)), (unsigned)limbs, (unsigned)batch);
    hipLaunchKernelGGL((ew_kernel<OP>), grid, dim3(kEwThreads), 0, s, k);
    check_launch();
}

// tensor product over table rows [mod_start, mod_start + limbs) (pha_behz.hip: base q and base Bsk)
void launch_tensor(Context &c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t mod_start, bool square,
                   hipStream_t s, size_t poly_limbs) {
    EwArgs k{};
    k.a = a; k.b = b; k.r = r;
    k.poly_limbs = (uint32_t)poly_limbs;
    if (limbs == 0) return;
    if (mod_start + limbs > c.rows) throw std::invalid_argument("modulus index out of range");
    k.mod = c.d_mod.p;
    k.n = (uint32_t)c.n;
    k.limbs = (uint32_t)limbs;
    k.mod_start = (uint32_t)mod_start;
    dim3 grid((unsigned)(c.n / (kEwThreads * kEwPerThread)), (unsigned)limbs);
    if (square) hipLaunchKernelGGL((ew_kernel<EW_SQUARE>), grid, dim3(kEwThreads), 0, s, k);
    else hipLaunchKernelGGL((ew_kernel<EW_TENSOR>), grid, dim3(kEwThreads), 0, s, k);
    check_launch();
}

// used by pha_rns.hip
void launch_add(Context &c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t mod_start, hipStream_t s) {
    EwArgs k{};
    k.a = a; k.b = b; k.r = r;
    launch_ew<EW_ADD>(c, k, limbs, mod_start, s);
}

}  // namespace pha

using namespace pha;

static void need(const void *p) {
    if (!p) throw std::invalid_argument("null device pointer");
}

extern "C" {

int pha_add_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, uint64_t *r, size_t cms,
                     size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(r);
    strict_operand(ctx->c, "operand a", a, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    strict_operand(ctx->c, "operand b", b, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    launch_add(ctx->c, a, b, r, cms, mod_start, as_stream(stream));
    PHA_API_END
}
int pha_sub_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, uint64_t *r, size_t cms,
                     size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(r);
    strict_operand(ctx->c, "operand a", a, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    strict_operand(ctx->c, "operand b", b, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    EwArgs k{};
    k.a = a; k.b = b; k.r = r;
    launch_ew<EW_SUB>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}
int pha_negate_rns_poly(pha_context_t ctx, const uint64_t *a, uint64_t *r, size_t cms, size_t mod_start,
                        void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(r);
    strict_operand(ctx->c, "operand a", a, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    EwArgs k{};
    k.a = a; k.r = r;
    launch_ew<EW_NEG>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}
int pha_multiply_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, uint64_t *r, size_t cms,
                          size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(r);
    strict_operand(ctx->c, "operand a", a, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    strict_operand(ctx->c, "operand b", b, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    EwArgs k{};
    k.a = a; k.b = b; k.r = r;
    launch_ew<EW_MUL>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}
int pha_multiply_and_add_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *b, const uint64_t *d,
                                  uint64_t *r, size_t cms, size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(b); need(d); need(r);
    strict_operand(ctx->c, "operand a", a, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    strict_operand(ctx->c, "operand b", b, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    strict_operand(ctx->c, "operand d", d, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    EwArgs k{};
    k.a = a; k.b = b; k.d = d; k.r = r;
    launch_ew<EW_MULADD>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}
int pha_multiply_scalar_rns_poly(pha_context_t ctx, const uint64_t *a, const uint64_t *scalar,
                                 const uint64_t *scalar_shoup, uint64_t *r, size_t cms, size_t mod_start,
                                 void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(a); need(scalar); need(scalar_shoup); need(r);
    strict_operand(ctx->c, "operand a", a, rows_plain(mod_start, cms), 1, 0, as_stream(stream));
    EwArgs k{};
    k.a = a; k.s0 = scalar; k.s1 = scalar_shoup; k.r = r;
    launch_ew<EW_MULSCALAR>(ctx->c, k, cms, mod_start, as_stream(stream));
    PHA_API_END
}
int pha_tensor_prod_2x2_rns_poly(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2, uint64_t *res,
                                 size_t cms, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(op1); need(op2); need(res);
    strict_operand(ctx->c, "tensor_prod_2x2 operand1", op1, rows_plain(0, cms), 2, cms * ctx->c.n, as_stream(stream));
    strict_operand(ctx->c, "tensor_prod_2x2 operand2", op2, rows_plain(0, cms), 2, cms * ctx->c.n, as_stream(stream));
    EwArgs k{};
    k.a = op1; k.b = op2; k.r = res;
    launch_ew<EW_TENSOR>(ctx->c, k, cms, 0, as_stream(stream));
    PHA_API_END
}
int pha_tensor_prod_2x2_batched(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2, uint64_t *res01,
                                uint64_t *res2, size_t cms, size_t batch, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(op1); need(op2); need(res01); need(res2);
    const size_t ln = cms * ctx->c.n;
    if (batch > 65535) throw std::invalid_argument("batch out of range");
    if (2 * batch <= 65535) {
        strict_operand(ctx->c, "tensor_prod_2x2 operand1", op1, rows_plain(0, cms), (uint32_t)(2 * batch), ln, as_stream(stream));
        strict_operand(ctx->c, "tensor_prod_2x2 operand2", op2, rows_plain(0, cms), (uint32_t)(2 * batch), ln, as_stream(stream));
    }
    // one launch over the batch (blockIdx.z): r04 trace of 8 ciphertexts -- 8 launches of 35.7 us each against 30.7 us for a lone one
    EwArgs k{};
    k.a = op1; k.b = op2; k.r = res01; k.r2 = res2;
    k.za = k.zb = k.zr = 2 * ln;
    k.zr2 = ln;
    if (batch >= 2) launch_ew<EW_TENSOR_NT>(ctx->c, k, cms, 0, as_stream(stream), batch);
    else launch_ew<EW_TENSOR>(ctx->c, k, cms, 0, as_stream(stream), batch);
    PHA_API_END
}
int pha_tensor_square_2x2_rns_poly(pha_context_t ctx, const uint64_t *op, uint64_t *res, size_t cms,
                                   void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(op); need(res);
    strict_operand(ctx->c, "tensor_square_2x2 operand", op, rows_plain(0, cms), 2, cms * ctx->c.n, as_stream(stream));
    EwArgs k{};
    k.a = op; k.r = res;
    launch_ew<EW_SQUARE>(ctx->c, k, cms, 0, as_stream(stream));
    PHA_API_END
}
// tensor_prod_2x2_rns_poly / tensor_square_2x2_rns_poly with the reference's `modulus` pointer argument (polymath.cu:463-529) as a
// first table row: the BEHZ / HPS callers run them over base Bsk and base R (src/evaluate.cu:489-497, :763-777)
int pha_tensor_prod_2x2_rns_poly_at(pha_context_t ctx, const uint64_t *op1, const uint64_t *op2, uint64_t *res, size_t cms,
                                    size_t mod_start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(op1); need(op2); need(res);
    strict_operand(ctx->c, "tensor_prod_2x2 operand1", op1, rows_plain(mod_start, cms), 2, cms * ctx->c.n, as_stream(stream));
    strict_operand(ctx->c, "tensor_prod_2x2 operand2", op2, rows_plain(mod_start, cms), 2, cms * ctx->c.n, as_stream(stream));
    launch_tensor(ctx->c, op1, op2, res, cms, mod_start, false, as_stream(stream), cms);
    PHA_API_END
}
int pha_tensor_square_2x2_rns_poly_at(pha_context_t ctx, const uint64_t *op, uint64_t *res, size_t cms, size_t mod_start,
                                      void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(op); need(res);
    strict_operand(ctx->c, "tensor_square_2x2 operand", op, rows_plain(mod_start, cms), 2, cms * ctx->c.n, as_stream(stream));
    launch_tensor(ctx->c, op, op, res, cms, mod_start, true, as_stream(stream), cms);
    PHA_API_END
}
int pha_add_to_ct(pha_context_t ctx, uint64_t *ct, const uint64_t *cx, size_t size_Ql, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(cx);
    strict_operand(ctx->c, "ct", ct, rows_plain(0, size_Ql), 1, 0, as_stream(stream));
    strict_operand(ctx->c, "cx", cx, rows_plain(0, size_Ql), 1, 0, as_stream(stream));
    launch_add(ctx->c, ct, cx, ct, size_Ql, 0, as_stream(stream));  // add_to_ct_kernel rns_bconv.cu:763-769
    PHA_API_END
}

}  // extern "C"
