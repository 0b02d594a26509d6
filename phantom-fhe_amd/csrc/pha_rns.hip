// pha_rns.hip -- fast base conversion, hybrid key-switch (mod-up / inner product / mod-down),
// CKKS rescale and Galois permutations on gfx950.
//
// Reference: src/rns_bconv.cu (bconv_mult / bconv_matmul / modup / moddown), src/eval_key_switch.cu,
// src/rns.cu:1082-1184, src/galois.cu.  Differences in HOW (results are bit-identical):
//   * base conversion is one kernel: a thread owns one coefficient, keeps the (scaled) input
//     residues of the digit in registers and walks the output primes, so inputs are read once per
//     output group instead of once per output prime, the q-hat matrix and the output modulus are
//     wave-uniform scalar loads, and stores are coalesced along the coefficient axis (the
//     reference maps adjacent threads to different output limbs);
//   * the q-hat^-1 scaling (bconv phase 1) is fused into the load (no temp round trip);
//   * limb index = blockIdx.y everywhere (no tid / N);
//   * mod-down and rescale reuse the forward NTT's fused epilogue (cx - NTT(delta)) * c.
#include "../../include/phantom_amd.h"
#include "pha_internal.h"
#include "pha_ntt_core.h"

namespace pha {

void launch_add(Context &c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t mod_start, hipStream_t s);

constexpr int kBcThreads = 256;
constexpr int kBcOutPerBlock = 8;

struct BConvArgs {
    u64 *dst;              // output buffer; output j goes to limb j + (j >= pad_start ? pad_len : 0)
    const u64 *src;        // input limbs [isz][n]
    const u64x2 *hat_inv;  // [isz] (only when SCALE_IN)
    const uint32_t *iprime, *oprime;
    const u64 *mat;        // [osz][isz]
    const DModulus *mod;   // QP table
    uint32_t isz, osz, n, pad_start, pad_len;
};

// bconv_mult (+) bconv_matmul: src/rns_bconv.cu:22-60,109-170 and the padded variant :455-485
template <int ISZ_PAD, bool SCALE_IN>
__global__ __launch_bounds__(kBcThreads) void bconv_kernel(const BConvArgs k) {
    const uint32_t coeff = blockIdx.x * kBcThreads + threadIdx.x;
    u64 y[ISZ_PAD];
#pragma unroll
    for (int i = 0; i < ISZ_PAD; i++) {
        y[i] = 0;
        if (i < (int)k.isz) {
            u64 x = k.src[(size_t)i * k.n + coeff];
            if (SCALE_IN) x = shoup(x, k.hat_inv[i], k.mod[k.iprime[i]].value);
            y[i] = x;
        }
    }
    const uint32_t j0 = blockIdx.y * kBcOutPerBlock;
    const uint32_t j1 = min(j0 + kBcOutPerBlock, k.osz);
    for (uint32_t j = j0; j < j1; j++) {
        const DModulus m = k.mod[k.oprime[j]];
        const u64 *row = k.mat + (size_t)j * k.isz;
        u64 lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < ISZ_PAD; i++)
            if (i < (int)k.isz) mac128(y[i], row[i], lo, hi);
        const uint32_t jo = j + (j >= k.pad_start ? k.pad_len : 0);
        k.dst[(size_t)jo * k.n + coeff] = barrett128(lo, hi, m);
    }
}

// generic fallback for very wide input bases (isz > 16): inputs are re-read per output prime
template <bool SCALE_IN>
__global__ __launch_bounds__(kBcThreads) void bconv_wide_kernel(const BConvArgs k) {
    const uint32_t coeff = blockIdx.x * kBcThreads + threadIdx.x;
    const uint32_t j0 = blockIdx.y * kBcOutPerBlock;
    const uint32_t j1 = min(j0 + kBcOutPerBlock, k.osz);
    for (uint32_t j = j0; j < j1; j++) {
        const DModulus m = k.mod[k.oprime[j]];
        const u64 *row = k.mat + (size_t)j * k.isz;
        u64 lo = 0, hi = 0;
        for (uint32_t i = 0; i < k.isz; i++) {
            u64 x = k.src[(size_t)i * k.n + coeff];
            if (SCALE_IN) x = shoup(x, k.hat_inv[i], k.mod[k.iprime[i]].value);
            mac128(x, row[i], lo, hi);
        }
        const uint32_t jo = j + (j >= k.pad_start ? k.pad_len : 0);
        k.dst[(size_t)jo * k.n + coeff] = barrett128(lo, hi, m);
    }
}

static void launch_bconv(Context &c, const BConv &b, u64 *dst, const u64 *src, bool scale_in, uint32_t pad_start,
                         uint32_t pad_len, hipStream_t s) {
    BConvArgs k{};
    k.dst = dst; k.src = src; k.hat_inv = b.hat_inv.p; k.iprime = b.d_iprime.p; k.oprime = b.d_oprime.p;
    k.mat = b.mat.p; k.mod = c.d_mod.p; k.isz = b.isz; k.osz = b.osz; k.n = (uint32_t)c.n;
    k.pad_start = pad_start; k.pad_len = pad_len;
    dim3 grid((unsigned)(c.n / kBcThreads), (b.osz + kBcOutPerBlock - 1) / kBcOutPerBlock);
    dim3 block(kBcThreads);
#define PHA_BC(P)                                                                          \
    do {                                                                                   \
        if (scale_in) hipLaunchKernelGGL((bconv_kernel<P, true>), grid, block, 0, s, k);   \
        else hipLaunchKernelGGL((bconv_kernel<P, false>), grid, block, 0, s, k);           \
    } while (0)
    if (b.isz <= 2) PHA_BC(2);
    else if (b.isz <= 4) PHA_BC(4);
    else if (b.isz <= 8) PHA_BC(8);
    else if (b.isz <= 16) PHA_BC(16);
    else if (scale_in) hipLaunchKernelGGL((bconv_wide_kernel<true>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((bconv_wide_kernel<false>), grid, block, 0, s, k);
#undef PHA_BC
    check_launch();
}

// ---- alpha == 1 fast paths (rns_bconv.cu:432-453, :691-707) -------------------------------------
struct SinglePArgs {
    u64 *dst;
    const u64 *src_raw, *src_normal;
    const DModulus *mod;
    const uint32_t *qlp_prime;
    uint32_t in_limb, n;
};
// out limb = blockIdx.y over [0, count): dst[j] = (j == in_limb) ? src_raw : reduce(src_normal)
__global__ __launch_bounds__(256) void single_p_kernel(const SinglePArgs k) {
    const uint32_t j = blockIdx.y;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    u64 v;
    if (j == k.in_limb) {
        v = k.src_raw[coeff];
    } else {
        const DModulus mo = k.mod[k.qlp_prime[j]];
        const u64 ip = k.mod[k.qlp_prime[k.in_limb]].value;
        v = k.src_normal[coeff];
        if (ip > mo.value) v = barrett64(v, mo.value, mo.ratio1);
    }
    k.dst[(size_t)j * k.n + coeff] = v;
}

// ---- key-switch inner product (eval_key_switch.cu:14-69) ----------------------------------------
struct InnerArgs {
    u64 *cx;
    const u64 *t_mod_up;
    const u64 *const *evks;
    const DModulus *mod;
    const uint32_t *qlp_prime;
    uint32_t n, beta;
    size_t qlp_n, qp_n;
};
__global__ __launch_bounds__(256) void inner_prod_kernel(const InnerArgs k) {
    const uint32_t nid = blockIdx.y;          // limb in [Ql || P]
    const uint32_t twr = k.qlp_prime[nid];    // its row in the key (keys live at full QP width)
    const DModulus m = k.mod[twr];
    const size_t coeff = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t c2_id = (size_t)nid * k.n + coeff;
    const size_t evk_id = (size_t)twr * k.n + coeff;
    u64 a0l = 0, a0h = 0, a1l = 0, a1h = 0, b0l = 0, b0h = 0, b1l = 0, b1h = 0;
    for (uint32_t i = 0; i < k.beta; i++) {
        const u64 *key = k.evks[i];
        const u64x2 v = *reinterpret_cast<const u64x2 *>(k.t_mod_up + (size_t)i * k.qlp_n + c2_id);
        const u64x2 kb = *reinterpret_cast<const u64x2 *>(key + evk_id);
        const u64x2 ka = *reinterpret_cast<const u64x2 *>(key + evk_id + k.qp_n);
        mac128(v.x, kb.x, a0l, a0h);
        mac128(v.y, kb.y, a1l, a1h);
        mac128(v.x, ka.x, b0l, b0h);
        mac128(v.y, ka.y, b1l, b1h);
    }
    u64x2 r0{barrett128(a0l, a0h, m), barrett128(a1l, a1h, m)};
    u64x2 r1{barrett128(b0l, b0h, m), barrett128(b1l, b1h, m)};
    *reinterpret_cast<u64x2 *>(k.cx + c2_id) = r0;
    *reinterpret_cast<u64x2 *>(k.cx + c2_id + k.qlp_n) = r1;
}

// ---- (cx - delta) * c element-wise: moddown_kernel rns_bconv.cu:680-689 and
//      divide_and_round_q_last_kernel rns.cu:1082-1108 (REDUCE_LAST) ---------------------------------
struct SubMulArgs {
    u64 *dst;
    const u64 *cx, *delta;
    const u64x2 *cst;  // per-limb constant (Shoup pair)
    const DModulus *mod;
    uint32_t n;
};
template <bool REDUCE_LAST>
__global__ __launch_bounds__(256) void sub_mul_kernel(const SubMulArgs k) {
    const uint32_t limb = blockIdx.y;
    const DModulus m = k.mod[limb];
    const u64x2 cst = k.cst[limb];
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    u64 d = REDUCE_LAST ? barrett64(k.delta[coeff], m.value, m.ratio1) : k.delta[(size_t)limb * k.n + coeff];
    const u64 t = sub_mod(k.cx[(size_t)limb * k.n + coeff], d, m.value);
    k.dst[(size_t)limb * k.n + coeff] = shoup(t, cst, m.value);
}

// divide_and_round_reduce_q_last_kernel rns.cu:1128-1139: dst[j] = last mod q_j
struct ReduceArgs {
    u64 *dst;
    const u64 *last;
    const DModulus *mod;
    uint32_t n;
};
__global__ __launch_bounds__(256) void reduce_last_kernel(const ReduceArgs k) {
    const uint32_t limb = blockIdx.y;
    const DModulus m = k.mod[limb];
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    k.dst[(size_t)limb * k.n + coeff] = barrett64(k.last[coeff], m.value, m.ratio1);
}

// ---- Galois (src/galois.cu:11-39) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void galois_ntt_kernel(u64 *dst, const u64 *src, const uint32_t *table, uint32_t n) {
    const uint32_t limb = blockIdx.y;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    dst[(size_t)limb * n + coeff] = src[(size_t)limb * n + table[coeff]];
}
__global__ __launch_bounds__(256) void galois_coeff_kernel(u64 *dst, const u64 *src, const DModulus *mod,
                                                           uint32_t mod_start, uint32_t elt, uint32_t n) {
    const uint32_t limb = blockIdx.y;
    const u64 q = mod[mod_start + limb].value;
    const uint32_t coeff = blockIdx.x * 256 + threadIdx.x;
    // index_raw = coeff * galois_elt mod 2n (include/galois.cuh:115-130)
    const uint32_t raw = (uint32_t)(((u64)coeff * elt) & (2 * (u64)n - 1));
    u64 v = src[(size_t)limb * n + coeff];
    if (raw >= n) v = neg_mod(v, q);
    dst[(size_t)limb * n + (raw & (n - 1))] = v;
}

// ------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------
static bool ntt_domain_scheme(int scheme) {
    if (scheme == PHA_SCHEME_CKKS || scheme == PHA_SCHEME_BGV) return true;
    if (scheme == PHA_SCHEME_BFV) return false;
    throw std::invalid_argument("unsupported scheme");
}

// DRNSTool::modup rns_bconv.cu:530-627
static void modup(Context &c, Tool &t, u64 *dst, const u64 *cks, int scheme, u64 *t_cks, hipStream_t s) {
    const size_t n = c.n;
    const uint32_t ql = t.size_ql, qlp = t.size_qlp, alpha = t.alpha;
    const bool ntt_dom = ntt_domain_scheme(scheme);
    if (ntt_dom) {
        if (alpha == 1) {
            ntt_inverse(c, cks, t_cks, t_cks, plain_sel(0, ql), EPI_INV_CANON, NttExtra{}, s);
        } else {
            NttExtra x;  // iNTT fused with x partQlHatInv (bconv phase 1), :558-559
            x.scale = t.part_hat_inv.p;
            x.scale_shoup = t.part_hat_inv_shoup.p;
            ntt_inverse(c, cks, t_cks, t_cks, plain_sel(0, ql), EPI_INV_SCALE, x, s);
        }
    }
    for (uint32_t b = 0; b < t.beta; b++) {
        const uint32_t st = alpha * b;
        const uint32_t len = (b == t.beta - 1) ? ql - alpha * (t.beta - 1) : alpha;
        u64 *out = dst + (size_t)b * qlp * n;
        if (alpha == 1) {
            SinglePArgs k{out, cks + (size_t)st * n, (ntt_dom ? t_cks : cks) + (size_t)st * n, c.d_mod.p,
                          t.d_qlp_prime.p, st, (uint32_t)n};
            hipLaunchKernelGGL(single_p_kernel, dim3((unsigned)(n / 256), qlp), dim3(256), 0, s, k);
            check_launch();
        } else {
            // own limbs are kept verbatim (modup_copy_partQl_kernel :522-528)
            PHA_HIP(hipMemcpyAsync(out + (size_t)st * n, cks + (size_t)st * n, (size_t)len * n * sizeof(u64),
                                   hipMemcpyDeviceToDevice, s));
            // BFV still needs the q-hat^-1 scaling (bconv_mult_kernel :603-607); ckks/bgv got it in the iNTT
            launch_bconv(c, t.digit[b], out, (ntt_dom ? t_cks : cks) + (size_t)st * n, !ntt_dom, st, len, s);
        }
        LimbSel sel = special_sel(0, qlp, c.size_qp, c.size_p);
        if (ntt_dom) { sel.excl_start = st; sel.excl_end = st + len; }  // ntt_modup.cu:422
        ntt_forward(c, out, out, out, sel, EPI_FWD_CANON, NttExtra{}, s);
    }
}

// phantom::key_switch_inner_prod eval_key_switch.cu:71-92
static void inner_prod(Context &c, Tool &t, u64 *cx, const u64 *t_mod_up, const u64 *const *rlk, hipStream_t s) {
    InnerArgs k{};
    k.cx = cx; k.t_mod_up = t_mod_up; k.evks = rlk; k.mod = c.d_mod.p; k.qlp_prime = t.d_qlp_prime.p;
    k.n = (uint32_t)c.n; k.beta = t.beta; k.qlp_n = (size_t)t.size_qlp * c.n; k.qp_n = (size_t)c.size_qp * c.n;
    hipLaunchKernelGGL(inner_prod_kernel, dim3((unsigned)(c.n / 512), t.size_qlp), dim3(256), 0, s, k);
    check_launch();
}

// DRNSTool::moddown_from_NTT rns_bconv.cu:776-828
static void moddown_from_ntt(Context &c, Tool &t, u64 *ct_i, u64 *cx_i, int scheme, u64 *delta, hipStream_t s) {
    const size_t n = c.n;
    const uint32_t ql = t.size_ql, qlp = t.size_qlp;
    if (scheme == PHA_SCHEME_CKKS)
        ntt_inverse(c, cx_i, cx_i, cx_i, special_sel(ql, c.size_p, c.size_qp, c.size_p), EPI_INV_CANON, NttExtra{}, s);
    else if (scheme == PHA_SCHEME_BFV)
        ntt_inverse(c, cx_i, cx_i, cx_i, special_sel(0, qlp, c.size_qp, c.size_p), EPI_INV_CANON, NttExtra{}, s);
    else
        throw std::invalid_argument("unsupported scheme (bgv mod-down is not on the accelerated path yet)");
    if (t.alpha == 1) {
        SinglePArgs k{delta, nullptr, cx_i + (size_t)ql * n, c.d_mod.p, t.d_qlp_prime.p, 0xffffffffu, (uint32_t)n};
        // in_prime is the special prime: pass its limb so the kernel can compare moduli
        k.in_limb = ql;
        hipLaunchKernelGGL(single_p_kernel, dim3((unsigned)(n / 256), ql), dim3(256), 0, s, k);
        check_launch();
    } else {
        launch_bconv(c, t.p_to_ql, delta, cx_i + (size_t)ql * n, true, 0xffffffffu, 0, s);
    }
    if (scheme == PHA_SCHEME_CKKS) {
        NttExtra x;  // NTT(delta) fused with (cx - .) * P^-1 (ntt_moddown.cu:106-261)
        x.scale = t.pinv.p;
        x.scale_shoup = t.pinv_shoup.p;
        x.aux = cx_i;
        ntt_forward(c, delta, delta, ct_i, plain_sel(0, ql), EPI_FWD_MODDOWN, x, s);
    } else {
        SubMulArgs k{ct_i, cx_i, delta, t.pinv2.p, c.d_mod.p, (uint32_t)n};
        hipLaunchKernelGGL(sub_mul_kernel<false>, dim3((unsigned)(n / 256), ql), dim3(256), 0, s, k);
        check_launch();
    }
}

static void check_level(Context &c, size_t size_Ql, bool need_p) {
    if (size_Ql < 1 || size_Ql > c.size_q) throw std::invalid_argument("size_Ql out of range");
    if (need_p && c.size_p == 0) throw std::invalid_argument("context has no special modulus");
}

}  // namespace pha

using namespace pha;

static void need(const void *p) {
    if (!p) throw std::invalid_argument("null device pointer");
}

extern "C" {

int pha_bconv_P_to_Ql(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *src, void *stream) {
    PHA_API_BEGIN
    need(dst); need(src);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    launch_bconv(c, t.p_to_ql, dst, src, true, 0xffffffffu, 0, as_stream(stream));
    PHA_API_END
}

int pha_modup(pha_context_t ctx, size_t size_Ql, uint64_t *dst, const uint64_t *cks, int scheme, void *stream) {
    PHA_API_BEGIN
    need(dst); need(cks);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    u64 *t_cks = c.scratch(stream, size_Ql * c.n);
    modup(c, t, dst, cks, scheme, t_cks, as_stream(stream));
    PHA_API_END
}

int pha_key_switch_inner_prod(pha_context_t ctx, size_t size_Ql, uint64_t *p_cx, const uint64_t *p_t_mod_up,
                              const uint64_t *const *rlk, void *stream) {
    PHA_API_BEGIN
    need(p_cx); need(p_t_mod_up); need(rlk);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    inner_prod(c, c.tool((uint32_t)size_Ql), p_cx, p_t_mod_up, rlk, as_stream(stream));
    PHA_API_END
}

int pha_moddown_from_NTT(pha_context_t ctx, size_t size_Ql, uint64_t *ct_i, uint64_t *cx_i, int scheme,
                         void *stream) {
    PHA_API_BEGIN
    need(ct_i); need(cx_i);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    u64 *delta = c.scratch(stream, size_Ql * c.n);
    moddown_from_ntt(c, t, ct_i, cx_i, scheme, delta, as_stream(stream));
    PHA_API_END
}

int pha_keyswitch_inplace(pha_context_t ctx, size_t size_Ql, uint64_t *ct, const uint64_t *c2,
                          const uint64_t *const *rlk, int scheme, void *stream) {
    PHA_API_BEGIN
    need(ct); need(c2); need(rlk);
    Context &c = ctx->c;
    check_level(c, size_Ql, true);
    Tool &t = c.tool((uint32_t)size_Ql);
    hipStream_t s = as_stream(stream);
    const size_t n = c.n, ql_n = size_Ql * n, qlp_n = (size_t)t.size_qlp * n;
    // scratch: t_cks / delta [Ql][N] | t_mod_up [beta][QlP][N] | cx [2][QlP][N]  (eval_key_switch.cu:151,155)
    u64 *base = c.scratch(stream, ql_n + (size_t)t.beta * qlp_n + 2 * qlp_n);
    u64 *tmp = base, *t_mod_up = base + ql_n, *cx = t_mod_up + (size_t)t.beta * qlp_n;
    modup(c, t, t_mod_up, c2, scheme, tmp, s);
    inner_prod(c, t, cx, t_mod_up, rlk, s);
    for (int i = 0; i < 2; i++) {
        u64 *cx_i = cx + (size_t)i * qlp_n;
        moddown_from_ntt(c, t, cx_i, cx_i, scheme, tmp, s);
        launch_add(c, ct + (size_t)i * ql_n, cx_i, ct + (size_t)i * ql_n, size_Ql, 0, s);  // add_to_ct_kernel
    }
    PHA_API_END
}

int pha_divide_and_round_q_last_ntt(pha_context_t ctx, size_t size_Ql, uint64_t *src, size_t cipher_size,
                                    uint64_t *dst, void *stream) {
    PHA_API_BEGIN
    need(src); need(dst);
    Context &c = ctx->c;
    check_level(c, size_Ql, false);
    if (size_Ql < 2) throw std::invalid_argument("cannot rescale the last remaining modulus");
    Tool &t = c.tool((uint32_t)size_Ql);
    hipStream_t s = as_stream(stream);
    const size_t n = c.n, nl = size_Ql - 1;
    u64 *tmp = c.scratch(stream, nl * n);
    for (size_t p = 0; p < cipher_size; p++) {
        u64 *ci_in = src + p * size_Ql * n, *ci_out = dst + p * nl * n;
        // ci[last] -> coefficient form (rns.cu:1171)
        ntt_inverse(c, ci_in, ci_in, ci_in, plain_sel(nl, 1), EPI_INV_CANON, NttExtra{}, s);
        ReduceArgs k{tmp, ci_in + nl * n, c.d_mod.p, (uint32_t)n};
        hipLaunchKernelGGL(reduce_last_kernel, dim3((unsigned)(n / 256), (unsigned)nl), dim3(256), 0, s, k);
        check_launch();
        // NTT(ci[last] mod qj) fused with (ci[j] - .) * q_last^-1 (rns.cu:1178-1182)
        NttExtra x;
        x.scale = t.inv_q_last.p;
        x.scale_shoup = t.inv_q_last_shoup.p;
        x.aux = ci_in;
        ntt_forward(c, tmp, tmp, ci_out, plain_sel(0, nl), EPI_FWD_MODDOWN, x, s);
    }
    PHA_API_END
}

int pha_divide_and_round_q_last(pha_context_t ctx, size_t size_Ql, const uint64_t *src, size_t cipher_size,
                                uint64_t *dst, void *stream) {
    PHA_API_BEGIN
    need(src); need(dst);
    Context &c = ctx->c;
    check_level(c, size_Ql, false);
    if (size_Ql < 2) throw std::invalid_argument("cannot switch down the last remaining modulus");
    Tool &t = c.tool((uint32_t)size_Ql);
    const size_t n = c.n, nl = size_Ql - 1;
    for (size_t p = 0; p < cipher_size; p++) {
        SubMulArgs k{dst + p * nl * n, src + p * size_Ql * n, src + p * size_Ql * n + nl * n, t.inv_q_last2.p,
                     c.d_mod.p, (uint32_t)n};
        hipLaunchKernelGGL(sub_mul_kernel<true>, dim3((unsigned)(n / 256), (unsigned)nl), dim3(256), 0,
                           as_stream(stream), k);
        check_launch();
    }
    PHA_API_END
}

int pha_apply_galois_ntt(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt, size_t cms,
                         void *stream) {
    PHA_API_BEGIN
    need(src); need(dst);
    if (src == dst) throw std::invalid_argument("apply_galois_ntt cannot run in place");
    Context &c = ctx->c;
    const uint32_t *tab = c.galois_table(galois_elt);
    hipLaunchKernelGGL(galois_ntt_kernel, dim3((unsigned)(c.n / 256), (unsigned)cms), dim3(256), 0,
                       as_stream(stream), dst, src, tab, (uint32_t)c.n);
    check_launch();
    PHA_API_END
}

int pha_apply_galois(pha_context_t ctx, const uint64_t *src, uint64_t *dst, uint32_t galois_elt, size_t cms,
                     size_t mod_start, void *stream) {
    PHA_API_BEGIN
    need(src); need(dst);
    if (src == dst) throw std::invalid_argument("apply_galois cannot run in place");
    Context &c = ctx->c;
    if (!(galois_elt & 1) || galois_elt >= 2 * c.n) throw std::invalid_argument("Galois element is not valid");
    if (mod_start + cms > c.size_qp) throw std::invalid_argument("modulus index out of range");
    hipLaunchKernelGGL(galois_coeff_kernel, dim3((unsigned)(c.n / 256), (unsigned)cms), dim3(256), 0,
                       as_stream(stream), dst, src, c.d_mod.p, (uint32_t)mod_start, galois_elt, (uint32_t)c.n);
    check_launch();
    PHA_API_END
}

}  // extern "C"
