// pha_behz.hip -- BFV multiply, BEHZ variant, on gfx950.
//
// Reference: bfv_multiply_behz src/evaluate.cu:404-548; DRNSTool::fastbconv_m_tilde / sm_mrq / fast_floor /
// fastbconv_sk src/rns.cu:1249-1510; multiply_and_negated_add_rns_poly src/polymath.cu:606-634; constants
// src/rns.cu:392-560 (built in pha_context.hip, Context::behz).  The base conversions reuse the one-thread-per-
// coefficient converter of pha_rns.hip (its 128-bit accumulate form: the auxiliary primes are 61 bits wide), the
// transforms reuse the NTT kernels over the auxiliary table rows, and what remains are three element-wise kernels.
#include "../../include/phantom_amd.h"
#include "pha_internal.h"
#include "pha_ntt_core.h"

namespace pha {

// sm_mrq_kernel rns.cu:1290-1320: (x + q * r) * m_tilde^-1 mod Bsk with r = -x_mtilde * q^-1 mod m_tilde, centred
struct MrqArgs {
    u64 *dst;                // [polys][Bsk][N]
    const u64 *src;          // [polys][Bsk + 1][N], last limb modulo m_tilde
    const DModulus *mod;     // table base; Bsk limb j is row aux0 + j
    const u64 *prod_q_mod_bsk;
    const u64x2 *inv_mt_mod_bsk;
    u64x2 neg_inv_prod_q_mod_mt;
    uint32_t aux0, size_bsk, n;
};
__global__ __launch_bounds__(256) void sm_mrq_kernel(const MrqArgs k) {
    const uint32_t j = blockIdx.y, coeff = blockIdx.x * 256 + threadIdx.x;
    const DModulus m = k.mod[k.aux0 + j];
    const u64 mt = (u64)1 << 32;
    const u64 *in = k.src + (size_t)blockIdx.z * (k.size_bsk + 1) * k.n;
    u64 r = shoup(in[(size_t)k.size_bsk * k.n + coeff], k.neg_inv_prod_q_mod_mt, mt);
    if (r >= (mt >> 1)) r += m.value - mt;
    u64 lo, hi;
    mul128(r, k.prod_q_mod_bsk[j], lo, hi);
    const u64 x = in[(size_t)j * k.n + coeff];
    lo += x;
    hi += lo < x;
    k.dst[((size_t)blockIdx.z * k.size_bsk + j) * k.n + coeff] = shoup(barrett128(lo, hi, m), k.inv_mt_mod_bsk[j], m.value);
}

// second half of bconv_fuse_sub_mul_unroll2_kernel rns.cu:1343-1386: (x_Bsk + (p - conv)) * prod(q)^-1 mod p
struct FloorArgs {
    u64 *dst;
    const u64 *x_bsk, *conv;
    const DModulus *mod;
    const u64x2 *scale;
    uint32_t aux0, n;
    size_t dst_stride = 0, x_stride = 0, conv_stride = 0;   // polynomial blockIdx.z
};
__global__ __launch_bounds__(256) void fast_floor_kernel(const FloorArgs k) {
    const uint32_t j = blockIdx.y, coeff = blockIdx.x * 256 + threadIdx.x;
    const u64 p = k.mod[k.aux0 + j].value;
    const size_t id = (size_t)j * k.n + coeff, z = blockIdx.z;
    // the sum is not reduced first (:1376-1378)
    k.dst[z * k.dst_stride + id] = shoup(k.x_bsk[z * k.x_stride + id] + (p - k.conv[z * k.conv_stride + id]), k.scale[j], p);
}

// alpha_sk (bconv_fuse_sub_mul_single_unroll2_kernel rns.cu:1421-1464) and the Shenoy-Kumaresan correction
// (multiply_and_negated_add_rns_poly polymath.cu:606-634) in one kernel over the q limbs
struct SkArgs {
    u64 *out_q;              // [Q][N]: FastBconv(x, B -> q) on entry, the result on exit
    const u64 *conv_msk;     // [N]   FastBconv(x, B -> {m_sk})
    const u64 *x_msk;        // [N]   the m_sk limb of the Bsk polynomial
    const DModulus *mod;
    const u64 *prod_b_mod_q;
    u64x2 inv_prod_b_mod_msk;
    u64 m_sk;
    uint32_t n;
    size_t out_stride = 0, conv_stride = 0, x_stride = 0;   // polynomial blockIdx.z
};
__global__ __launch_bounds__(256) void sk_fix_kernel(SkArgs k) {
    const uint32_t i = blockIdx.y, coeff = blockIdx.x * 256 + threadIdx.x;
    k.out_q += blockIdx.z * k.out_stride;
    k.conv_msk += blockIdx.z * k.conv_stride;
    k.x_msk += blockIdx.z * k.x_stride;
    const DModulus m = k.mod[i];
    u64 a = shoup(k.conv_msk[coeff] + (k.m_sk - k.x_msk[coeff]), k.inv_prod_b_mod_msk, k.m_sk);
    u64 pb = k.prod_b_mod_q[i];
    if (a > (k.m_sk >> 1)) a = k.m_sk - a;   // alpha_sk in [-m_sk/2, 0): -alpha * B is positive
    else pb = m.value - pb;                  // alpha_sk in [0, m_sk/2):  -alpha * B = alpha * (-B mod q)
    const size_t id = (size_t)i * k.n + coeff;
    k.out_q[id] = add_mod(k.out_q[id], mul_mod(a, pb, m), m.value);
}


static LimbSel aux_sel(uint32_t count, uint32_t aux0) {  // a buffer of `count` limbs whose table rows start at aux0
    LimbSel s = plain_sel(0, count);
    s.remap_from = 0;
    s.remap_add = aux0;
    return s;
}

// BEHZ_mul_1 evaluate.cu:404-441 for the two polynomials of one ciphertext
static void behz_lift(Context &c, Behz &b, const u64 *ct, u64 *out_q, u64 *out_bsk, u64 *tmp, hipStream_t s) {
    const uint32_t n = (uint32_t)c.n, sq = b.size_q, sk = b.size_bsk;
    const size_t qn = (size_t)sq * n;
    NttExtra xq;
    xq.batch = 2;
    xq.poly_stride = qn;
    ntt_forward(c, ct, out_q, out_q, plain_sel(0, sq), EPI_FWD_CANON, xq, s);      // out of place: no copy of ct first
    // (1) q -> Bsk u {m_tilde}: phase 1 with m_tilde * qhat^-1 (the converter's own scale-in factors, Context::behz), one
    // conversion for all Bsk + 1 outputs
    u64 *lift = tmp + 2 * qn;  // lift [2][Bsk + 1][N]
    launch_bconv(c, b.d_q_to_bskmt.p, 0, 2, sq, sk + 1, b.q_to_bskmt.split_kind, lift, (size_t)(sk + 1) * n, ct, qn, nullptr, true, s);
    // (2) small Montgomery reduction modulo q, switching to base Bsk
    MrqArgs ma{out_bsk, lift, c.d_mod.p, b.prod_q_mod_bsk.p, b.inv_mt_mod_bsk.p, b.neg_inv_prod_q_mod_mt, b.aux0, sk, n};
    hipLaunchKernelGGL(sm_mrq_kernel, dim3(n / 256, sk, 2), dim3(256), 0, s, ma);
    check_launch();
    NttExtra xb;
    xb.batch = 2;
    xb.poly_stride = (size_t)sk * n;
    ntt_forward(c, out_bsk, out_bsk, out_bsk, aux_sel(sk, b.aux0), EPI_FWD_CANON, xb, s);
}

// ---- HPS variant (mul_tech_type::hps): bConv_HPS src/rns_bconv.cu:248-372, scaleAndRound_HPS_QR_R src/rns.cu:1700-1746.
// Floating point: the reference is built by nvcc with its default -fmad=true, so every `acc += double(x) * c` of
// its kernels is ONE fused multiply-add; the same fma chains are issued here (this file is compiled with
// -ffp-contract=off, the fusions are explicit).
struct HpsFixArgs {
    u64 *out;                // [osz][N]: FastBconv(y) on entry, minus v * prod(ibase) mod p_j on exit
    const u64 *y;            // [isz][N]  x_i * hat_i^-1 mod q_i
    const double *inv;       // [isz]     1 / q_i
    const u64 *alpha_mod;    // [isz + 1][osz]
    const DModulus *mod;
    const uint32_t *oprime;  // table rows of the output primes
    uint32_t isz, osz, n;
};
__global__ __launch_bounds__(256) void hps_fix_kernel(const HpsFixArgs k) {
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    double frac = 0.0;
    for (uint32_t i = 0; i < k.isz; i++) frac = __builtin_fma((double)k.y[(size_t)i * k.n + coeff], k.inv[i], frac);
    const size_t v = (size_t)llround(frac);
    for (uint32_t j = 0; j < k.osz; j++) {
        const u64 p = k.mod[k.oprime[j]].value;
        const size_t id = (size_t)j * k.n + coeff;
        k.out[id] = sub_mod(k.out[id], k.alpha_mod[v * k.osz + j], p);
    }
}

// sum_i x_i * tab[i] (i < count <= PAD) as a 128-bit number, carry-free: x_i arrives cut at bit SX (x0, x1), the uniform table
// entries are cut at bit SM on the way, partial products stay below 2^61 (halves of at most 31 and 30 bits) and the four partial
// sums are folded into the total every 8 terms -- four v_mad_u64_u32 per term instead of the nine of a 128-bit multiply-add.
template <int PAD, int SX, int SM>
__device__ __forceinline__ void split_dot(const u32 (&x0)[PAD], const u32 (&x1)[PAD], const u64 *tab, uint32_t count, u64 &lo, u64 &hi) {
    static_assert(SX + SM == 61, "halves of 30 and 31 bits");
#pragma unroll
    for (int base = 0; base < PAD; base += 8) {
        u64 ll = 0, lh = 0, hl = 0, hh = 0;
#pragma unroll
        for (int i = base; i < base + 8 && i < PAD; i++) {
            const u64 m = i < (int)count ? tab[i] : 0;
            const u32 m0 = (u32)m & ((1u << SM) - 1), m1 = (u32)(m >> SM);
            ll = (u64)x0[i] * m0 + ll;
            lh = (u64)x0[i] * m1 + lh;
            hl = (u64)x1[i] * m0 + hl;
            hh = (u64)x1[i] * m1 + hh;
        }
        lo += ll;
        hi += lo < ll;
        u64 t = lh << SM;
        lo += t;
        hi += (lo < t) + (lh >> (64 - SM));
        t = hl << SX;
        lo += t;
        hi += (lo < t) + (hl >> (64 - SX));
        t = hh << (SX + SM);
        lo += t;
        hi += (lo < t) + (hh >> (64 - SX - SM));
    }
}

struct ScaleRoundArgs {
    u64 *dst;                // [R][N]
    const u64 *src;          // [Q + R][N] coefficient form
    const double *frac;      // [Q]
    const u64 *tab;          // [R][Q + 1]
    const DModulus *mod;
    uint32_t size_q, size_r, aux0, n;
    size_t dst_stride = 0, src_stride = 0;   // polynomial blockIdx.y
};
// QPAD >= size_q: the Q residues of the coefficient stay in registers across the R limbs.  SPLIT: Q primes <= 60 bits and R
// primes <= 62: the carry-free dot product (residues cut at 30 bits, table entries at 31)
template <int QPAD, bool SPLIT>
__global__ __launch_bounds__(256) void hps_scale_round_kernel(ScaleRoundArgs k) {
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    k.src += blockIdx.y * k.src_stride;
    k.dst += blockIdx.y * k.dst_stride;
    u64 x[SPLIT ? 1 : QPAD];
    u32 x0[SPLIT ? QPAD : 1], x1[SPLIT ? QPAD : 1];
    double nu = 0.5;
#pragma unroll
    for (int i = 0; i < QPAD; i++) {
        u64 v = 0;
        if (i < (int)k.size_q) {
            v = k.src[(size_t)i * k.n + coeff];
            nu = __builtin_fma((double)v, k.frac[i], nu);
        }
        if (SPLIT) {
            x0[i] = (u32)v & 0x3fffffffu;
            x1[i] = (u32)(v >> 30);
        } else {
            x[i] = v;
        }
    }
    u64 alpha = (u64)nu;
    for (uint32_t j = 0; j < k.size_r; j++) {
        const DModulus m = k.mod[k.aux0 + j];
        const u64 *tab = k.tab + (size_t)j * (k.size_q + 1);
        u64 lo = 0, hi = 0;
        if constexpr (SPLIT) {
            split_dot<QPAD, 30, 31>(x0, x1, tab, k.size_q, lo, hi);
        } else {
#pragma unroll
            for (int i = 0; i < QPAD; i++)
                if (i < (int)k.size_q) mac128(x[i], tab[i], lo, hi);
        }
        mac128(k.src[(size_t)(k.size_q + j) * k.n + coeff], tab[k.size_q], lo, hi);
        const u64 v = barrett128(lo, hi, m);
        alpha = barrett64(alpha, m.value, m.ratio1);   // reduced IN PLACE across the R limbs, as rns.cu:1733 does
        k.dst[(size_t)j * k.n + coeff] = add_mod(v, alpha, m.value);
    }
}
__global__ __launch_bounds__(256) void hps_scale_round_wide_kernel(ScaleRoundArgs k) {  // size_q > 32
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    k.src += blockIdx.y * k.src_stride;
    k.dst += blockIdx.y * k.dst_stride;
    double nu = 0.5;
    for (uint32_t i = 0; i < k.size_q; i++) nu = __builtin_fma((double)k.src[(size_t)i * k.n + coeff], k.frac[i], nu);
    u64 alpha = (u64)nu;
    for (uint32_t j = 0; j < k.size_r; j++) {
        const DModulus m = k.mod[k.aux0 + j];
        const u64 *tab = k.tab + (size_t)j * (k.size_q + 1);
        u64 lo = 0, hi = 0;
        for (uint32_t i = 0; i < k.size_q; i++) mac128(k.src[(size_t)i * k.n + coeff], tab[i], lo, hi);
        mac128(k.src[(size_t)(k.size_q + j) * k.n + coeff], tab[k.size_q], lo, hi);
        const u64 v = barrett128(lo, hi, m);
        alpha = barrett64(alpha, m.value, m.ratio1);
        k.dst[(size_t)j * k.n + coeff] = add_mod(v, alpha, m.value);
    }
}

// scaleAndRound_HPS_QlRl_Ql (src/rns.cu:1748-1796): [Q || Rl] coefficient form -> base Q, scaled by t / Rl and rounded;
// the carry alpha is reduced in place across the Q limbs exactly as :1783 does.
struct ScaleRoundQArgs {
    u64 *dst;                // [Q][N]
    const u64 *src;          // [Q + Rl][N]
    const double *frac;      // [Rl]
    const u64 *tab;          // [Q][Rl + 1]
    const DModulus *mod;
    uint32_t size_q, size_r, n;
    size_t dst_stride = 0, src_stride = 0;   // polynomial blockIdx.y
};
// RPAD >= size_r: the Rl residues of the coefficient stay in registers across the Q limbs (0 = re-read).  SPLIT (RPAD > 0): R
// primes <= 62 bits and Q primes <= 60: the carry-free dot product (residues cut at 31 bits, table entries at 30)
template <int RPAD, bool SPLIT>
__global__ __launch_bounds__(256) void hps_scale_round_q_kernel(ScaleRoundQArgs k) {
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    k.src += blockIdx.y * k.src_stride;
    k.dst += blockIdx.y * k.dst_stride;
    const u64 *src_r = k.src + (size_t)k.size_q * k.n + coeff;
    u64 x[RPAD && !SPLIT ? RPAD : 1];
    u32 x0[SPLIT ? RPAD : 1], x1[SPLIT ? RPAD : 1];
    double nu = 0.5;
    if (RPAD) {
#pragma unroll
        for (int j = 0; j < RPAD; j++) {
            u64 v = 0;
            if (j < (int)k.size_r) {
                v = src_r[(size_t)j * k.n];
                nu = __builtin_fma((double)v, k.frac[j], nu);
            }
            if (SPLIT) {
                x0[j] = (u32)v & 0x7fffffffu;
                x1[j] = (u32)(v >> 31);
            } else {
                x[j] = v;
            }
        }
    } else {
        for (uint32_t j = 0; j < k.size_r; j++) nu = __builtin_fma((double)src_r[(size_t)j * k.n], k.frac[j], nu);
    }
    u64 alpha = (u64)nu;
    for (uint32_t i = 0; i < k.size_q; i++) {
        const DModulus m = k.mod[i];
        const u64 *tab = k.tab + (size_t)i * (k.size_r + 1);
        u64 lo = 0, hi = 0;
        if constexpr (SPLIT) {
            split_dot<RPAD, 31, 30>(x0, x1, tab, k.size_r, lo, hi);
        } else if (RPAD) {
#pragma unroll
            for (int j = 0; j < RPAD; j++)
                if (j < (int)k.size_r) mac128(x[j], tab[j], lo, hi);
        } else {
            for (uint32_t j = 0; j < k.size_r; j++) mac128(src_r[(size_t)j * k.n], tab[j], lo, hi);
        }
        mac128(k.src[(size_t)i * k.n + coeff], tab[k.size_r], lo, hi);
        const u64 v = barrett128(lo, hi, m);
        alpha = barrett64(alpha, m.value, m.ratio1);
        k.dst[(size_t)i * k.n + coeff] = add_mod(v, alpha, m.value);
    }
}

static void launch_scale_round(Context &c, const ScaleRoundArgs &ka, hipStream_t s, uint32_t batch = 1) {
    // the carry-free dot product needs the residues (Q primes) below 2^60 and the table entries (R primes) below 2^62
    bool split = true;
    for (uint32_t i = 0; i < ka.size_q; i++) split = split && !(c.primes[i] >> 60);
    for (uint32_t j = 0; j < ka.size_r; j++) split = split && !(c.primes[ka.aux0 + j] >> 62);
    const dim3 grid(ka.n / 256, batch), block(256);
#define PHA_SR(P)                                                                                    \
    do {                                                                                             \
        if (split) hipLaunchKernelGGL((hps_scale_round_kernel<P, true>), grid, block, 0, s, ka);      \
        else hipLaunchKernelGGL((hps_scale_round_kernel<P, false>), grid, block, 0, s, ka);           \
    } while (0)
    if (ka.size_q <= 8) PHA_SR(8);
    else if (ka.size_q <= 16) PHA_SR(16);
    else if (ka.size_q <= 32) PHA_SR(32);
    else hipLaunchKernelGGL(hps_scale_round_wide_kernel, grid, block, 0, s, ka);
#undef PHA_SR
    check_launch();
}

void launch_bconv_phase1(Context &c, const BConv &conv, u64 *dst, const u64 *src, hipStream_t s);

// DBaseConverter::bConv_HPS on `batch` polynomials (strides in words): dst [osz][N] <- src [isz][N].  Up to 32 inputs the
// correction term rides in the conversion kernel (one launch for all the polynomials); wider bases: conversion, phase 1 and
// fix-up per polynomial, y [isz][N] scratch.
static void bconv_hps(Context &c, const BConv &conv, const BConvDev *d_conv, const double *inv, const u64 *alpha_mod,
                      u64 *dst, const u64 *src, u64 *y, hipStream_t s, uint32_t batch = 1, size_t dst_stride = 0,
                      size_t src_stride = 0) {
    const uint32_t n = (uint32_t)c.n;
    if (conv.isz <= 32) {
        BConvEpilogue e{};
        e.hps_inv = inv;
        e.hps_alpha = alpha_mod;
        launch_bconv(c, d_conv, 0, batch, conv.isz, conv.osz, conv.split_kind, dst, dst_stride, src, src_stride, nullptr, true, s, 0,
                     0, &e);
        return;
    }
    for (uint32_t b = 1; b < batch; b++) bconv_hps(c, conv, d_conv, inv, alpha_mod, dst + b * dst_stride, src + b * src_stride, y, s);
    launch_bconv(c, d_conv, 0, 1, conv.isz, conv.osz, conv.split_kind, dst, 0, src, 0, nullptr, true, s);
    launch_bconv_phase1(c, conv, y, src, s);   // the fix-up needs the phase-1 values themselves
    HpsFixArgs fa{dst, y, inv, alpha_mod, c.d_mod.p, conv.d_oprime.p, conv.isz, conv.osz, n};
    hipLaunchKernelGGL(hps_fix_kernel, dim3(n / 256), dim3(256), 0, s, fa);
    check_launch();
}

// y_i = x_i * hat_i^-1 mod q_i for the input limbs of a converter (bconv_mult_kernel rns_bconv.cu:22-33)
struct Phase1Args {
    u64 *dst;
    const u64 *src;
    const DModulus *mod;
    const uint32_t *iprime;
    const u64x2 *hat_inv;
    uint32_t n;
};
__global__ __launch_bounds__(256) void bconv_phase1_kernel(const Phase1Args k) {
    const uint32_t i = blockIdx.y, coeff = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)i * k.n + coeff;
    k.dst[id] = shoup(k.src[id], k.hat_inv[i], k.mod[k.iprime[i]].value);
}
void launch_bconv_phase1(Context &c, const BConv &conv, u64 *dst, const u64 *src, hipStream_t s) {
    Phase1Args k{dst, src, c.d_mod.p, conv.d_iprime.p, conv.hat_inv.p, (uint32_t)c.n};
    hipLaunchKernelGGL(bconv_phase1_kernel, dim3((unsigned)(c.n / 256), conv.isz), dim3(256), 0, s, k);
    check_launch();
}

static LimbSel qr_sel(uint32_t size_q, uint32_t size_r, uint32_t aux0) {  // [Q limbs || R limbs] buffers
    LimbSel s = plain_sel(0, size_q + size_r);
    s.remap_from = size_q;
    s.remap_add = aux0 - size_q;
    return s;
}

}  // namespace pha

using namespace pha;

extern "C" int pha_bfv_multiply_hps(pha_context_t ctx, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst,
                                    void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ct1 || !ct2 || !dst) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    Hps &h = c.hps();
    hipStream_t s = as_stream(stream);
    const uint32_t n = (uint32_t)c.n, sq = h.size_q, sr = h.size_r, sqr = sq + sr;
    const size_t qn = (size_t)sq * n, rn = (size_t)sr * n, qrn = (size_t)sqr * n;
    const bool square = ct1 == ct2;
    // scratch: x1 [3][Q+R] | x2 [2][Q+R] | y [max(Q, R)] | tmp [3][R]
    u64 *base = c.scratch(stream, 5 * qrn + std::max(qn, rn) + 3 * rn);
    u64 *x1 = base, *x2 = x1 + 3 * qrn, *y = x2 + 2 * qrn, *tmp = y + std::max(qn, rn);
    // lift every input polynomial from base Q to Q || R (evaluate.cu:702-716, :733-748)
    for (int w = 0; w < (square ? 1 : 2); w++) {
        const u64 *ct = w ? ct2 : ct1;
        u64 *x = w ? x2 : x1;
        for (uint32_t p = 0; p < 2; p++)
            PHA_HIP(hipMemcpyAsync(x + p * qrn, ct + p * qn, qn * sizeof(u64), hipMemcpyDeviceToDevice, s));
        bconv_hps(c, h.q_to_r, h.d_q_to_r.p, h.q_inv.p, h.alpha_q_mod_r.p, x + qn, ct, y, s, 2, qrn, qn);   // both polynomials
        NttExtra xf;
        xf.batch = 2;
        xf.poly_stride = qrn;
        ntt_forward(c, x, x, x, qr_sel(sq, sr, h.aux0), EPI_FWD_CANON, xf, s);
    }
    // tensor product over Q || R: the Q limbs and the R limbs live in different table rows
    const u64 *rhs = square ? x1 : x2;
    launch_tensor(c, x1, rhs, x1, sq, 0, square, s, sqr);
    launch_tensor(c, x1 + qn, rhs + qn, x1 + qn, sr, h.aux0, square, s, sqr);
    NttExtra xi;
    xi.batch = 3;
    xi.poly_stride = qrn;
    ntt_inverse(c, x1, x1, x1, qr_sel(sq, sr, h.aux0), EPI_INV_CANON, xi, s);
    // scale by t/Q and round into base R, then R -> Q (evaluate.cu:800-808): the three polynomials in one launch each
    ScaleRoundArgs ka{tmp, x1, h.frac.p, h.div_mod_r.p, c.d_mod.p, sq, sr, h.aux0, n, rn, qrn};
    launch_scale_round(c, ka, s, 3);
    bconv_hps(c, h.r_to_q, h.d_r_to_q.p, h.r_inv.p, h.alpha_r_mod_q.p, dst, tmp, y, s, 3, qn, rn);
    PHA_API_END
}

// ExpandCRTBasis_Ql_Q (src/rns.cu:1810-1836): the Ql limbs times the product of the dropped primes, the dropped limbs zero.
// blockIdx.y over the |Q| limbs of dst; src has |Ql| limbs with the same limb pitch, so dst == src works in place.
struct ExpandArgs {
    u64 *dst;
    const u64 *src;
    const u64 *c, *c_shoup;
    const DModulus *mod;
    uint32_t size_ql, n;
};
__global__ __launch_bounds__(256) void hps_expand_kernel(const ExpandArgs k) {
    const uint32_t i = blockIdx.y, coeff = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)i * k.n + coeff;
    k.dst[id] = i < k.size_ql ? shoup(k.src[id], u64x2{k.c[i], k.c_shoup[i]}, k.mod[i].value) : 0;
}
static void launch_scale_round_q(Context &c, u64 *dst, const u64 *src, const double *frac, const u64 *tab, uint32_t size_ql,
                                 uint32_t extra, hipStream_t s, uint32_t batch = 1, size_t dst_stride = 0, size_t src_stride = 0) {
    const uint32_t n = (uint32_t)c.n;
    ScaleRoundQArgs ka{dst, src, frac, tab, c.d_mod.p, size_ql, extra, n, dst_stride, src_stride};
    // the carry-free dot product needs the residues (primes size_ql .. size_ql + extra) below 2^62 and the table (Ql primes) below 2^60
    bool split = true;
    for (uint32_t i = 0; i < size_ql; i++) split = split && !(c.primes[i] >> 60);
    for (uint32_t j = 0; j < extra; j++) split = split && !(c.primes[size_ql + j] >> 62);
    const dim3 grid(n / 256, batch), block(256);
#define PHA_SRQ(P)                                                                                     \
    do {                                                                                               \
        if (split) hipLaunchKernelGGL((hps_scale_round_q_kernel<P, true>), grid, block, 0, s, ka);      \
        else hipLaunchKernelGGL((hps_scale_round_q_kernel<P, false>), grid, block, 0, s, ka);           \
    } while (0)
    if (extra <= 8) PHA_SRQ(8);
    else if (extra <= 16) PHA_SRQ(16);
    else if (extra <= 32) PHA_SRQ(32);
    else hipLaunchKernelGGL((hps_scale_round_q_kernel<0, false>), grid, block, 0, s, ka);
#undef PHA_SRQ
    check_launch();
}
static void launch_expand(Context &c, HpsQ &h, u64 *dst, const u64 *src, hipStream_t s) {
    ExpandArgs ka{dst, src, h.drop_mod_q.p, h.drop_mod_q_shoup.p, c.d_mod.p, h.size_q, (uint32_t)c.n};
    hipLaunchKernelGGL(hps_expand_kernel, dim3((unsigned)(c.n / 256), h.size_q_full), dim3(256), 0, s, ka);
    check_launch();
}

// bfv_multiply_hps with mul_tech hps_overq (h of the top level) or hps_overq_leveled with levels dropped (h of a lower level):
// src/evaluate.cu:674-818, the overq branches :709-711, :745-751, :790-795.  Operands and result are over the full base Q.
// ct1 == ct2 (the same pointer) is the reference's squaring shortcut (:720-731), kept as it is.
static void hps_overq_multiply(Context &c, HpsQ &h, const u64 *ct1, const u64 *ct2, u64 *dst, void *stream,
                               bool keep_c2_low = false) {
    hipStream_t s = as_stream(stream);
    const uint32_t n = (uint32_t)c.n, sq = h.size_q, sr = h.size_r, sqr = sq + sr, sqf = h.size_q_full;
    const size_t qn = (size_t)sq * n, rn = (size_t)sr * n, qrn = (size_t)sqr * n, qfn = (size_t)sqf * n;
    const bool square = ct1 == ct2;
    // scratch: x1 [3][Ql+Rl] | x2 [2][Ql+Rl] | y [max(Q, Rl)]
    u64 *base = c.scratch(stream, 5 * qrn + std::max(qfn, rn));
    u64 *x1 = base, *x2 = x1 + 3 * qrn, *y = x2 + 2 * qrn;
    for (uint32_t p = 0; p < 2; p++) {   // first operand: (scaled down to Ql when levels are dropped, :709-710) exact lift to Ql || Rl
        if (h.drop) launch_scale_round_q(c, x1 + p * qrn, ct1 + p * qfn, h.frac_drop.p, h.div_mod_q_drop.p, sq, h.drop, s);
        else PHA_HIP(hipMemcpyAsync(x1 + p * qrn, ct1 + p * qfn, qn * sizeof(u64), hipMemcpyDeviceToDevice, s));
    }
    bconv_hps(c, h.q_to_r, h.d_q_to_r.p, h.q_inv.p, h.alpha_q_mod_r.p, x1 + qn, x1, y, s, 2, qrn, qrn);
    if (!square) {   // second operand: Q -> Rl by bConv_BEHZ_var1, then Rl -> Ql exactly (:745-751); both polynomials per launch
        launch_bconv(c, h.d_q_to_r_var1.p, 0, 2, h.q_to_r_var1.isz, sr, h.q_to_r_var1.split_kind, x2 + qn, qrn, ct2, qfn, nullptr, true, s);
        bconv_hps(c, h.r_to_q, h.d_r_to_q.p, h.r_inv.p, h.alpha_r_mod_q.p, x2, x2 + qn, y, s, 2, qrn, qrn);
    }
    NttExtra xf;
    xf.batch = 2;
    xf.poly_stride = qrn;
    ntt_forward(c, x1, x1, x1, qr_sel(sq, sr, h.aux0), EPI_FWD_CANON, xf, s);
    if (!square) ntt_forward(c, x2, x2, x2, qr_sel(sq, sr, h.aux0), EPI_FWD_CANON, xf, s);
    const u64 *rhs = square ? x1 : x2;
    launch_tensor(c, x1, rhs, x1, sq, 0, square, s, sqr);
    launch_tensor(c, x1 + qn, rhs + qn, x1 + qn, sr, h.aux0, square, s, sqr);
    NttExtra xi;
    xi.batch = 3;
    xi.poly_stride = qrn;
    ntt_inverse(c, x1, x1, x1, qr_sel(sq, sr, h.aux0), EPI_INV_CANON, xi, s);
    // scale by t / Rl and round straight into base Ql (:790-792; the three polynomials in one launch), expand to Q (:794-795)
    launch_scale_round_q(c, dst, x1, h.frac.p, h.div_mod_q.p, sq, sr, s, 3, qfn, qrn);
    for (uint32_t p = 0; p < 3; p++)
        if (h.drop && !(keep_c2_low && p == 2)) launch_expand(c, h, dst + p * qfn, dst + p * qfn, s);   // (:957-958: c2 stays at level l)
}

extern "C" int pha_bfv_multiply_hps_overq(pha_context_t ctx, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst,
                                          void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ct1 || !ct2 || !dst) throw std::invalid_argument("null pointer");
    hps_overq_multiply(ctx->c, ctx->c.hps_overq(), ct1, ct2, dst, stream);
    PHA_API_END
}

extern "C" int pha_bfv_multiply_hps_overq_leveled(pha_context_t ctx, size_t size_Ql, const uint64_t *ct1, const uint64_t *ct2,
                                                  uint64_t *dst, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ct1 || !ct2 || !dst) throw std::invalid_argument("null pointer");
    if (size_Ql < 1 || size_Ql > ctx->c.size_q) throw std::invalid_argument("RNSBase is invalid");
    hps_overq_multiply(ctx->c, ctx->c.hps_overq((uint32_t)size_Ql), ct1, ct2, dst, stream);
    PHA_API_END
}

extern "C" int pha_scaleAndRound_HPS_Q_Ql(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !dst || !src) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql >= c.size_q) throw std::invalid_argument("at least one level has to be dropped");
    HpsQ &h = c.hps_overq((uint32_t)size_Ql);
    launch_scale_round_q(c, dst, src, h.frac_drop.p, h.div_mod_q_drop.p, h.size_q, h.drop, as_stream(stream));
    PHA_API_END
}

extern "C" int pha_ExpandCRTBasis_Ql_Q(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !dst || !src) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql >= c.size_q) throw std::invalid_argument("at least one level has to be dropped");
    launch_expand(c, c.hps_overq((uint32_t)size_Ql), dst, src, as_stream(stream));
    PHA_API_END
}

// dst[i] += src[i] * prod(dropped primes) on the Ql limbs only (ExpandCRTBasis_Ql_Q_add_to_ct, src/rns.cu:1838-1858)
__global__ __launch_bounds__(256) void hps_expand_add_kernel(const ExpandArgs k) {
    const uint32_t i = blockIdx.y, coeff = blockIdx.x * 256 + threadIdx.x;
    const size_t id = (size_t)i * k.n + coeff;
    const u64 q = k.mod[i].value;
    k.dst[id] = add_mod(shoup(k.src[id], u64x2{k.c[i], k.c_shoup[i]}, q), k.dst[id], q);
}

// bfv_mul_relin_hps under hps_overq_leveled with levels dropped (src/evaluate.cu:822-1027): the product's c2 is left at
// level l, key-switched there, and the results go onto the Ql limbs of (c0, c1).  dst [2][Q][N].
extern "C" int pha_bfv_mul_relin_hps_overq_leveled(pha_context_t ctx, size_t size_Ql, const uint64_t *ct1, const uint64_t *ct2,
                                                   const uint64_t *const *rlk, uint64_t *dst, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ct1 || !ct2 || !rlk || !dst) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql >= c.size_q) throw std::invalid_argument("at least one level has to be dropped");
    HpsQ &h = c.hps_overq((uint32_t)size_Ql);
    hipStream_t s = as_stream(stream);
    const size_t n = c.n, qln = size_Ql * n, qfn = (size_t)c.size_q * n;
    u64 *d3 = c.scratch_outer(stream, 3 * qfn + 2 * qln), *res = d3 + 3 * qfn;   // the inner calls use the stream's own arena
    hps_overq_multiply(c, h, ct1, ct2, d3, stream, true);
    PHA_HIP(hipMemsetAsync(res, 0, 2 * qln * sizeof(u64), s));
    const int rc = pha_keyswitch_inplace(ctx, size_Ql, res, d3 + 2 * qfn, rlk, /*scheme bfv*/ 1, stream);
    if (rc != 0) throw std::runtime_error(pha_last_error());
    PHA_HIP(hipMemcpyAsync(dst, d3, 2 * qfn * sizeof(u64), hipMemcpyDeviceToDevice, s));
    for (int p = 0; p < 2; p++) {
        ExpandArgs ka{dst + (size_t)p * qfn, res + (size_t)p * qln, h.drop_mod_q.p, h.drop_mod_q_shoup.p, c.d_mod.p, h.size_q, (uint32_t)n};
        hipLaunchKernelGGL(hps_expand_add_kernel, dim3((unsigned)(n / 256), h.size_q), dim3(256), 0, s, ka);
        check_launch();
    }
    PHA_API_END
}

// keyswitch_inplace for BFV under hps_overq_leveled with levels dropped (src/eval_key_switch.cu:142-147, :170-175): c2 [Q][N] is
// scaled down to Ql, switched with the level's DRNSTool, and both halves are expanded to Q before they are added to ct [2][Q][N]
extern "C" int pha_keyswitch_inplace_bfv_leveled(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *c2,
                                                 const uint64_t *const *rlk, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ct || !c2 || !rlk) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql >= c.size_q) throw std::invalid_argument("at least one level has to be dropped");
    HpsQ &h = c.hps_overq((uint32_t)size_Ql);
    hipStream_t s = as_stream(stream);
    const size_t n = c.n, qln = size_Ql * n, qfn = (size_t)c.size_q * n;
    // buffers of this call (second arena: the inner key switch uses the stream's own): c2 at level l, the level's
    // key-switch result (c0, c1), one expanded polynomial
    u64 *c2l = c.scratch_outer(stream, 3 * qln + qfn), *res = c2l + qln, *full = res + 2 * qln;
    launch_scale_round_q(c, c2l, c2, h.frac_drop.p, h.div_mod_q_drop.p, h.size_q, h.drop, s);
    PHA_HIP(hipMemsetAsync(res, 0, 2 * qln * sizeof(u64), s));
    const int rc = pha_keyswitch_inplace(ctx, size_Ql, res, c2l, rlk, /*scheme bfv*/ 1, stream);
    if (rc != 0) throw std::runtime_error(pha_last_error());
    for (int p = 0; p < 2; p++) {
        launch_expand(c, h, full, res + (size_t)p * qln, s);
        launch_add(c, ct + (size_t)p * qfn, full, ct + (size_t)p * qfn, c.size_q, 0, s);
    }
    PHA_API_END
}

extern "C" int pha_bfv_multiply_behz(pha_context_t ctx, const uint64_t *ct1, const uint64_t *ct2, uint64_t *dst,
                                     void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ct1 || !ct2 || !dst) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    Behz &b = c.behz();
    hipStream_t s = as_stream(stream);
    const uint32_t n = (uint32_t)c.n, sq = b.size_q, sk = b.size_bsk, sb = b.size_b;
    const size_t qn = (size_t)sq * n, bn = (size_t)sk * n;
    const bool square = ct1 == ct2;
    // scratch: q1 [3][Q] | b1 [3][Bsk] | q2 [2][Q] | b2 [2][Bsk] | tmp (y [2][Q] + lift [2][Bsk+1], later conv / floor)
    const size_t tmp_words = std::max(2 * qn + 2 * (bn + n), 6 * bn + 3 * (size_t)n);
    u64 *base = c.scratch(stream, 3 * qn + 3 * bn + 2 * qn + 2 * bn + tmp_words);
    u64 *q1 = base, *b1 = q1 + 3 * qn, *q2 = b1 + 3 * bn, *b2 = q2 + 2 * qn, *tmp = b2 + 2 * bn;
    behz_lift(c, b, ct1, q1, b1, tmp, s);
    if (!square) behz_lift(c, b, ct2, q2, b2, tmp, s);
    // step 4: tensor product in both bases (evaluate.cu:479-498)
    launch_tensor(c, q1, square ? q1 : q2, q1, sq, 0, square, s);
    launch_tensor(c, b1, square ? b1 : b2, b1, sk, b.aux0, square, s);
    // steps 5-6: inverse transforms fused with the multiplication by t (:518-530)
    NttExtra xq;
    xq.batch = 3;
    xq.poly_stride = qn;
    xq.scale = b.t_q.p;
    xq.scale_shoup = b.t_q_shoup.p;
    ntt_inverse(c, q1, q1, q1, plain_sel(0, sq), EPI_INV_SCALE, xq, s);
    NttExtra xb;
    xb.batch = 3;
    xb.poly_stride = bn;
    xb.scale = b.t_bsk.p;
    xb.scale_shoup = b.t_bsk_shoup.p;
    ntt_inverse(c, b1, b1, b1, aux_sel(sk, b.aux0), EPI_INV_SCALE, xb, s);
    // steps 7-8 for the three polynomials in one launch each
    u64 *conv = tmp, *fl = tmp + 3 * bn, *msk = fl + 3 * bn;  // conv [3][Bsk][N], fl [3][Bsk][N], msk [3][N]
    // step 7 fast_floor (rns.cu:1394-1419)
    launch_bconv(c, b.d_q_to_bsk.p, 0, 3, sq, sk, b.q_to_bsk.split_kind, conv, bn, q1, qn, nullptr, true, s);
    FloorArgs fa{fl, b1, conv, c.d_mod.p, b.inv_prod_q_mod_bsk.p, b.aux0, n, bn, bn, bn};
    hipLaunchKernelGGL(fast_floor_kernel, dim3(n / 256, sk, 3), dim3(256), 0, s, fa);
    check_launch();
    // step 8 fastbconv_sk (rns.cu:1470-1510)
    launch_bconv(c, b.d_b_to_msk.p, 0, 3, sb, 1, b.b_to_msk.split_kind, msk, n, fl, bn, nullptr, true, s);   // B -> m_sk
    launch_bconv(c, b.d_b_to_q.p, 0, 3, sb, sq, b.b_to_q.split_kind, dst, qn, fl, bn, nullptr, true, s);
    SkArgs ka{dst, msk, fl + (size_t)sb * n, c.d_mod.p, b.prod_b_mod_q.p, b.inv_prod_b_mod_msk, b.m_sk, n, qn, n, bn};
    hipLaunchKernelGGL(sk_fix_kernel, dim3(n / 256, sq, 3), dim3(256), 0, s, ka);
    check_launch();
    PHA_API_END
}

// ---- DBaseConverter as an object of the ABI (include/rns_bconv.cuh:13-87): a converter between two bases given as
//      rows of the context's prime table; bConv_BEHZ (rns_bconv.cu:212-229) and bConv_HPS (:248-372) on one polynomial ----
struct pha_base_converter {
    pha_context_t ctx;
    pha::BConv conv;
    pha::DevBuf<pha::BConvDev> d_conv;
    pha::DevBuf<double> inv;          // 1 / q_i of the input base (src/host/rns.cu:321-323)
    pha::DevBuf<pha::u64> alpha_mod;  // [isz + 1][osz] alpha * prod(ibase) mod p_j (:459-466)
    bool split_ok = true;
    // bConv_BEHZ_var1 (negPQHatInvModq / QInvModp, src/host/rns.cu:469-496): built on first use
    std::unique_ptr<pha::BConv> var1;
    pha::DevBuf<pha::BConvDev> d_var1;
    std::mutex var1_lock;
    // exact_convert_array (src/rns_bconv.cu:374-431): ONE output modulus, which may be a raw value outside the prime table (the
    // plain modulus t of base_q_to_t_conv_, src/rns.cu:283-284)
    bool raw = false;                 // created by pha_base_converter_create_modulus: exact_convert_array only
    pha::DModulus out_mod{};          // the output modulus when osz == 1
    std::vector<uint32_t> irows;
    pha::DevBuf<pha::u64x2> x_hat_inv;   // [isz] qhat_i^-1 mod q_i (+ Shoup)
    pha::DevBuf<pha::u64> x_mat;         // [isz] qhat_i mod t
    pha::DevBuf<pha::u64> x_big;         // [isz] words of prod(ibase), little-endian
    pha::DevBuf<pha::u64> x_q;           // [isz] q_i
    bool x_ready = false;
};

namespace pha {
struct ExactArgs {
    u64 *dst;
    const u64 *src;
    const u64x2 *hat_inv;
    const u64 *mat, *big, *q;
    DModulus t;
    uint32_t isz, n;
};
// exact_convert_array_kernel src/rns_bconv.cu:374-414, one thread per coefficient.  The mid-loop reduction (:393-396) never
// fires in the reference (`i && reduction_threshold == 0` with reduction_threshold = 15) and is absent here; v is a sum of IEEE
// divisions in limb order; Q mod t by Horner over the words of prod(ibase), most significant first.
__global__ __launch_bounds__(256) void exact_convert_kernel(const ExactArgs k) {
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    u64 lo = 0, hi = 0, q_mod_t = 0;
    double v = 0.0;
    for (uint32_t i = 0; i < k.isz; i++) {
        const u64 qi = k.q[i];
        const u64 yi = shoup(k.src[(size_t)i * k.n + coeff], k.hat_inv[i], qi);
        mac128(yi, k.mat[i], lo, hi);
        q_mod_t = barrett128(k.big[k.isz - i - 1], q_mod_t, k.t);
        v += (double)yi / (double)qi;
    }
    const u64 inner = barrett128(lo, hi, k.t);
    const u64 rounded = (u64)round(v);
    k.dst[coeff] = sub_mod(inner, mul_mod(rounded, q_mod_t, k.t), k.t.value);
}
}  // namespace pha

// constants of exact_convert_array for (input rows, ONE output modulus): qhat_i^-1 mod q_i, qhat_i mod t, prod(ibase)
static void build_exact(pha::Context &c, pha_base_converter &h) {
    using namespace pha;
    using u128 = unsigned __int128;
    const size_t isz = h.irows.size();
    std::vector<u64x2> hat_inv(isz);
    std::vector<u64> mat(isz), q(isz), big(isz + 1, 0);
    const u64 t = h.out_mod.value;
    size_t len = 1;
    big[0] = 1;
    for (size_t i = 0; i < isz; i++) {
        const u64 qi = c.primes[h.irows[i]];
        q[i] = qi;
        u64 hat_q = 1 % qi, hat_t = 1 % t;
        for (size_t kx = 0; kx < isz; kx++)
            if (kx != i) {
                hat_q = (u64)((u128)hat_q * (c.primes[h.irows[kx]] % qi) % qi);
                hat_t = (u64)((u128)hat_t * (c.primes[h.irows[kx]] % t) % t);
            }
        u64 inv = 1, base = hat_q, e = qi - 2;    // q_i is prime
        while (e) {
            if (e & 1) inv = (u64)((u128)inv * base % qi);
            base = (u64)((u128)base * base % qi);
            e >>= 1;
        }
        hat_inv[i] = u64x2{inv, (u64)(((u128)inv << 64) / qi)};
        mat[i] = hat_t;
        u64 carry = 0;
        for (size_t w = 0; w < len; w++) {
            const u128 x = (u128)big[w] * qi + carry;
            big[w] = (u64)x;
            carry = (u64)(x >> 64);
        }
        if (carry) big[len++] = carry;
    }
    big.resize(isz);
    h.x_hat_inv.upload(hat_inv);
    h.x_mat.upload(mat);
    h.x_big.upload(big);
    h.x_q.upload(q);
    h.x_ready = true;
}

extern "C" {

int pha_base_converter_create(pha_context_t ctx, const uint32_t *ibase, size_t ibase_size, const uint32_t *obase,
                              size_t obase_size, pha_base_converter_t *out) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ibase || !obase || !out) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (ibase_size == 0 || ibase_size > 64 || obase_size == 0 || obase_size > 256) throw std::invalid_argument("RNSBase is invalid");
    std::vector<uint32_t> ip(ibase, ibase + ibase_size), op(obase, obase + obase_size);
    for (size_t i = 0; i < ip.size(); i++) {
        if (ip[i] >= c.rows) throw std::invalid_argument("modulus index out of range");
        for (size_t k = 0; k < i; k++)
            if (ip[k] == ip[i]) throw std::invalid_argument("RNSBase is not coprime");
    }
    for (uint32_t row : op)
        if (row >= c.rows) throw std::invalid_argument("modulus index out of range");
    PHA_HIP(hipSetDevice(c.device));
    auto h = std::make_unique<pha_base_converter>();
    h->ctx = ctx;
    build_bconv(c, h->conv, ip, op);
    describe_conv(h->conv, h->d_conv);
    std::vector<double> inv(ip.size());
    for (size_t i = 0; i < ip.size(); i++) {
        inv[i] = 1.0 / (double)c.primes[ip[i]];
        if (c.primes[ip[i]] >> 60) h->split_ok = false;
    }
    std::vector<u64> am((ip.size() + 1) * op.size());
    for (size_t j = 0; j < op.size(); j++) {
        const u64 p = c.primes[op[j]];
        if (p >> 60) h->split_ok = false;
        u64 prod = 1 % p;
        for (uint32_t row : ip) prod = (u64)((unsigned __int128)prod * (c.primes[row] % p) % p);
        for (size_t a = 0; a <= ip.size(); a++) am[a * op.size() + j] = (u64)((unsigned __int128)a * prod % p);
    }
    h->inv.upload(inv);
    h->alpha_mod.upload(am);
    h->irows = ip;
    if (op.size() == 1) h->out_mod = c.mods[op[0]];
    *out = h.release();
    PHA_API_END
}

int pha_base_converter_create_modulus(pha_context_t ctx, const uint32_t *ibase, size_t ibase_size, uint64_t out_modulus,
                                      pha_base_converter_t *out) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx || !ibase || !out) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (ibase_size == 0 || ibase_size > 64) throw std::invalid_argument("RNSBase is invalid");
    if (out_modulus < 2 || (out_modulus >> 61)) throw std::invalid_argument("modulus is invalid");   // Modulus: 2 .. 61 bits
    std::vector<uint32_t> ip(ibase, ibase + ibase_size);
    for (size_t i = 0; i < ip.size(); i++) {
        if (ip[i] >= c.rows) throw std::invalid_argument("modulus index out of range");
        for (size_t k = 0; k < i; k++)
            if (ip[k] == ip[i]) throw std::invalid_argument("RNSBase is not coprime");
    }
    PHA_HIP(hipSetDevice(c.device));
    auto h = std::make_unique<pha_base_converter>();
    h->ctx = ctx;
    h->raw = true;
    h->irows = ip;
    h->out_mod = h_modulus(out_modulus);
    h->conv.isz = (uint32_t)ip.size();
    h->conv.osz = 1;
    *out = h.release();
    PHA_API_END
}

void pha_base_converter_destroy(pha_base_converter_t conv) {
    if (!conv) return;
    (void)hipSetDevice(conv->ctx->c.device);
    (void)hipDeviceSynchronize();
    delete conv;
}

int pha_bConv_BEHZ(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_API_BEGIN
    if (!conv || !dst || !src) throw std::invalid_argument("null pointer");
    if (conv->raw) throw std::invalid_argument("a converter to a raw modulus supports exact_convert_array only");
    Context &c = conv->ctx->c;
    launch_bconv(c, conv->d_conv.p, 0, 1, conv->conv.isz, conv->conv.osz, conv->conv.split_kind, dst, 0, src, 0, nullptr, true,
                 as_stream(stream));
    PHA_API_END
}

int pha_bConv_HPS(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_API_BEGIN
    if (!conv || !dst || !src) throw std::invalid_argument("null pointer");
    if (conv->raw) throw std::invalid_argument("a converter to a raw modulus supports exact_convert_array only");
    Context &c = conv->ctx->c;
    u64 *y = c.scratch(stream, (size_t)conv->conv.isz * c.n);
    bconv_hps(c, conv->conv, conv->d_conv.p, conv->inv.p, conv->alpha_mod.p, dst, src, y, as_stream(stream));
    PHA_API_END
}

// DBaseConverter::bConv_BEHZ_var1 (src/rns_bconv.cu:231-246): phase 1 with negPQHatInvModq, phase 2 with QInvModp
int pha_bConv_BEHZ_var1(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_API_BEGIN
    if (!conv || !dst || !src) throw std::invalid_argument("null pointer");
    if (conv->raw) throw std::invalid_argument("a converter to a raw modulus supports exact_convert_array only");
    Context &c = conv->ctx->c;
    PHA_HIP(hipSetDevice(c.device));
    {
        std::lock_guard<std::mutex> g(conv->var1_lock);
        if (!conv->var1) {
            auto v = std::make_unique<BConv>();
            build_bconv_var1(c, *v, conv->conv.iprime, conv->conv.oprime);
            describe_conv(*v, conv->d_var1);
            conv->var1 = std::move(v);
        }
    }
    launch_bconv(c, conv->d_var1.p, 0, 1, conv->var1->isz, conv->var1->osz, conv->var1->split_kind, dst, 0, src, 0, nullptr, true,
                 as_stream(stream));
    PHA_API_END
}

// DBaseConverter::exact_convert_array (src/rns_bconv.cu:416-431): src [ibase][N] -> dst [N], the output base is ONE modulus
int pha_exact_convert_array(pha_base_converter_t conv, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_API_BEGIN
    if (!conv || !dst || !src) throw std::invalid_argument("null pointer");
    if (conv->conv.osz != 1) throw std::invalid_argument("out base in exact_convert_array must be one.");   // :423-425
    Context &c = conv->ctx->c;
    PHA_HIP(hipSetDevice(c.device));
    {
        std::lock_guard<std::mutex> g(conv->var1_lock);
        if (!conv->x_ready) build_exact(c, *conv);
    }
    ExactArgs k{dst, src, conv->x_hat_inv.p, conv->x_mat.p, conv->x_big.p, conv->x_q.p, conv->out_mod, (uint32_t)conv->irows.size(),
                (uint32_t)c.n};
    hipLaunchKernelGGL(exact_convert_kernel, dim3((unsigned)(c.n / 256)), dim3(256), 0, as_stream(stream), k);
    check_launch();
    PHA_API_END
}

// ---- the DRNSTool steps of the BFV multiplies as entry points of their own (include/rns.cuh:167-200), ONE polynomial per call,
//      so that a caller written like bfv_multiply_behz / bfv_multiply_hps (src/evaluate.cu:447-548, :674-818) links against
//      this library step by step.  The whole multiplies above run the same kernels on two or three polynomials per launch. ----
static void need_top(Context &c, size_t size_Ql, const char *what) {
    if (size_Ql != c.size_q) throw std::invalid_argument(std::string(what) + ": the auxiliary base exists at the top data level only");
}

int pha_tool_aux_sizes(pha_context_t ctx, size_t size_Ql, uint32_t *size_Bsk, uint32_t *size_R, uint32_t *size_Rl) {
    PHA_CTX_BEGIN(ctx)
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql > c.size_q) throw std::invalid_argument("RNSBase is invalid");
    if (size_Bsk) *size_Bsk = size_Ql == c.size_q ? c.behz().size_bsk : 0;
    if (size_R) *size_R = size_Ql == c.size_q ? c.hps().size_r : 0;
    if (size_Rl) *size_Rl = c.hps_overq((uint32_t)size_Ql).size_r;
    PHA_API_END
}

// DRNSTool::fastbconv_m_tilde src/rns.cu:1249-1278: src [Q][N] -> dst [Bsk + 1][N] (last limb modulo m_tilde)
int pha_fastbconv_m_tilde(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!dst || !src) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    need_top(c, size_Ql, "fastbconv_m_tilde");
    Behz &b = c.behz();
    launch_bconv(c, b.d_q_to_bskmt.p, 0, 1, b.size_q, b.size_bsk + 1, b.q_to_bskmt.split_kind, dst, 0, src, 0, nullptr, true,
                 as_stream(stream));
    PHA_API_END
}

// DRNSTool::sm_mrq src/rns.cu:1326-1338 (kernel :1290-1320): src [Bsk + 1][N] -> dst [Bsk][N]
int pha_sm_mrq(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!dst || !src) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    need_top(c, size_Ql, "sm_mrq");
    Behz &b = c.behz();
    MrqArgs ma{dst, src, c.d_mod.p, b.prod_q_mod_bsk.p, b.inv_mt_mod_bsk.p, b.neg_inv_prod_q_mod_mt, b.aux0, b.size_bsk, (uint32_t)c.n};
    hipLaunchKernelGGL(sm_mrq_kernel, dim3((unsigned)(c.n / 256), b.size_bsk, 1), dim3(256), 0, as_stream(stream), ma);
    check_launch();
    PHA_API_END
}

// DRNSTool::fast_floor src/rns.cu:1394-1419: input_base_q [Q][N], input_base_Bsk [Bsk][N] -> out_base_Bsk [Bsk][N]
int pha_fast_floor(pha_context_t ctx, size_t size_Ql, const uint64_t *input_base_q, const uint64_t *input_base_Bsk,
                   uint64_t *out_base_Bsk, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!input_base_q || !input_base_Bsk || !out_base_Bsk) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    need_top(c, size_Ql, "fast_floor");
    Behz &b = c.behz();
    hipStream_t s = as_stream(stream);
    const uint32_t n = (uint32_t)c.n;
    u64 *conv = c.scratch(stream, (size_t)b.size_bsk * n);
    launch_bconv(c, b.d_q_to_bsk.p, 0, 1, b.size_q, b.size_bsk, b.q_to_bsk.split_kind, conv, 0, input_base_q, 0, nullptr, true, s);
    FloorArgs fa{out_base_Bsk, input_base_Bsk, conv, c.d_mod.p, b.inv_prod_q_mod_bsk.p, b.aux0, n, 0, 0, 0};
    hipLaunchKernelGGL(fast_floor_kernel, dim3(n / 256, b.size_bsk, 1), dim3(256), 0, s, fa);
    check_launch();
    PHA_API_END
}

// DRNSTool::fastbconv_sk src/rns.cu:1470-1510: input_base_Bsk [Bsk][N] (B limbs, then the m_sk limb) -> out_base_q [Q][N]
int pha_fastbconv_sk(pha_context_t ctx, size_t size_Ql, const uint64_t *input_base_Bsk, uint64_t *out_base_q, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!input_base_Bsk || !out_base_q) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    need_top(c, size_Ql, "fastbconv_sk");
    Behz &b = c.behz();
    hipStream_t s = as_stream(stream);
    const uint32_t n = (uint32_t)c.n;
    u64 *msk = c.scratch(stream, n);
    launch_bconv(c, b.d_b_to_msk.p, 0, 1, b.size_b, 1, b.b_to_msk.split_kind, msk, 0, input_base_Bsk, 0, nullptr, true, s);
    launch_bconv(c, b.d_b_to_q.p, 0, 1, b.size_b, b.size_q, b.b_to_q.split_kind, out_base_q, 0, input_base_Bsk, 0, nullptr, true, s);
    SkArgs ka{out_base_q, msk, input_base_Bsk + (size_t)b.size_b * n, c.d_mod.p, b.prod_b_mod_q.p, b.inv_prod_b_mod_msk, b.m_sk, n, 0, 0, 0};
    hipLaunchKernelGGL(sk_fix_kernel, dim3(n / 256, b.size_q, 1), dim3(256), 0, s, ka);
    check_launch();
    PHA_API_END
}

// DRNSTool::scaleAndRound_HPS_QR_R src/rns.cu:1739-1746 (kernel :1700-1737): src [Q + R][N] -> dst [R][N]
int pha_scaleAndRound_HPS_QR_R(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!dst || !src) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    need_top(c, size_Ql, "scaleAndRound_HPS_QR_R");
    Hps &h = c.hps();
    ScaleRoundArgs ka{dst, src, h.frac.p, h.div_mod_r.p, c.d_mod.p, h.size_q, h.size_r, h.aux0, (uint32_t)c.n, 0, 0};
    launch_scale_round(c, ka, as_stream(stream), 1);
    PHA_API_END
}

// DRNSTool::scaleAndRound_HPS_QlRl_Ql src/rns.cu:1789-1796 (kernel :1749-1787): src [Ql + Rl][N] -> dst [Ql][N]
int pha_scaleAndRound_HPS_QlRl_Ql(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!dst || !src) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql > c.size_q) throw std::invalid_argument("RNSBase is invalid");
    HpsQ &h = c.hps_overq((uint32_t)size_Ql);
    launch_scale_round_q(c, dst, src, h.frac.p, h.div_mod_q.p, h.size_q, h.size_r, as_stream(stream));
    PHA_API_END
}

// DRNSTool::ExpandCRTBasis_Ql_Q_add_to_ct src/rns.cu:1850-1858 (kernel :1838-1848): dst [Ql limbs] += src [Ql][N] * prod(dropped)
int pha_ExpandCRTBasis_Ql_Q_add_to_ct(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!dst || !src) throw std::invalid_argument("null pointer");
    Context &c = ctx->c;
    if (size_Ql < 1 || size_Ql >= c.size_q) throw std::invalid_argument("at least one level has to be dropped");
    HpsQ &h = c.hps_overq((uint32_t)size_Ql);
    ExpandArgs ka{dst, src, h.drop_mod_q.p, h.drop_mod_q_shoup.p, c.d_mod.p, h.size_q, (uint32_t)c.n};
    hipLaunchKernelGGL(hps_expand_add_kernel, dim3((unsigned)(c.n / 256), h.size_q), dim3(256), 0, as_stream(stream), ka);
    check_launch();
    PHA_API_END
}

}  // extern "C"
